// Does an LDS-DMA load honour EXEC?  One wave, lanes < 20 active, every lane holds a VALID address (active lanes:
// region A, inactive lanes: region B); the LDS kilobyte is dumped afterwards.  Tuning aid, not product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ void glds_v(const void* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void glds_s(const void* gbase, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds) : "memory");
}
__global__ void k(const int* a, const int* b, int* out, int mode, int nact) {
    __shared__ int4 l[128];
    const int lane = threadIdx.x;
    l[lane] = make_int4(-1, -1, -1, -1); l[lane + 64] = make_int4(-2, -2, -2, -2);
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)l;
    if (mode == 0) {
        const int* g = lane < nact ? a + lane * 4 : b + lane * 4;
        if (lane < nact) glds_v(g, __builtin_amdgcn_readfirstlane(base));
    } else if (mode == 1) {
        const unsigned voff = lane < nact ? lane * 16u : 4096u + lane * 16u;   // b = a + 1024 ints
        if (lane < nact) glds_s(a, voff, __builtin_amdgcn_readfirstlane(base));
    } else {
        // odd lanes only
        const int* g = (lane & 1) ? a + lane * 4 : b + lane * 4;
        if (lane & 1) glds_v(g, __builtin_amdgcn_readfirstlane(base));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 128; i += 64) { out[i * 4] = l[i].x; out[i * 4 + 1] = l[i].y; out[i * 4 + 2] = l[i].z; out[i * 4 + 3] = l[i].w; }
}
int main() {
    int *d, *o; hipMalloc(&d, 2048 * 4 + 4096); hipMalloc(&o, 512 * 4);
    int h[2048]; for (int i = 0; i < 1024; ++i) { h[i] = i; h[1024 + i] = 100000 + i; }
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        k<<<1, 64>>>(d, d + 1024, o, mode, 20);
        int r[512]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("mode %d (%s): first dword of each 16-B LDS slot:\n", mode, mode == 0 ? "vaddr form, lanes < 20" : mode == 1 ? "saddr form, lanes < 20" : "vaddr form, odd lanes");
        for (int i = 0; i < 72; ++i) printf("%d%s", r[i * 4], (i % 24 == 23) ? "\n" : " ");
        printf("\n");
    }
    return 0;
}
