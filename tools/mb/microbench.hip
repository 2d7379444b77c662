// Machine-model probes (tuning aid, not product): dependent VALU chain, dependent LDS chain,
// LDS streaming, barrier cost.  Prints ns per op for one wave per SIMD and for full occupancy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_valu(uint32_t* out, int iters) {
    uint32_t x = threadIdx.x, y = blockIdx.x;
    for (int i = 0; i < iters; ++i) { x = x * 3u + y; y = y ^ x; x += 7u; y += x >> 3; }
    if (x == 0x12345678u) out[0] = y;
}
__global__ void k_valu64(uint64_t* out, int iters) {
    uint64_t x = threadIdx.x, y = blockIdx.x + 5;
    for (int i = 0; i < iters; ++i) { x = x > y ? x - y : y - x + 1; y += x >> 3; }
    if (x == 0x12345678u) out[0] = y;
}
__global__ void k_lds_chain(uint32_t* out, int iters) {
    __shared__ uint32_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (i * 97u + 13u) & 4095u;
    __syncthreads();
    uint32_t p = threadIdx.x;
    for (int i = 0; i < iters; ++i) p = s[p];
    if (p == 0xffffffffu) out[0] = p;
}
__global__ void k_lds_rw(uint64_t* out, int iters) {     // bitonic-like: 2 reads, compare, 2 writes per step
    __shared__ uint64_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = i * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    const int t = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        const int j = 1 << (i % 6);
        const int off = t & (j - 1);
        const int lo = (((t - off) << 1) + off) & 4095, hi = (lo + j) & 4095;
        const uint64_t x = s[lo], y = s[hi];
        s[lo] = y < x ? y : x; s[hi] = y < x ? x : y;
        __builtin_amdgcn_wave_barrier();
    }
    if (s[t] == 1) out[0] = 1;
}
__global__ void k_barrier(uint32_t* out, int iters) {
    __shared__ uint32_t s[1024];
    s[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < iters; ++i) { __syncthreads(); s[threadIdx.x] += s[(threadIdx.x + 64) & 1023]; }
    if (s[threadIdx.x] == 0xffffffffu) out[0] = 1;
}

__global__ void k_fma64(double* out, int iters) {      // 8 independent f64 FMA chains per lane: the f64 VALU issue rate
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
        a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = a0;
}

// HBM streams of the PS kernel's size: 400 MB in, 400 MB out
template <int PER, bool NT>
__global__ void __launch_bounds__(1024) k_copy(const int4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    const size_t base = ((size_t)blockIdx.x * PER) * blockDim.x + threadIdx.x;
    int4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) { const size_t i = base + (size_t)q * blockDim.x; v[q] = i < n4 ? in[i] : make_int4(0, 0, 0, 0); }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const size_t i = base + (size_t)q * blockDim.x;
        if (i < n4) {
            const float4 f = make_float4((float)v[q].x, (float)v[q].y, (float)v[q].z, (float)v[q].w);
            if (NT) {
                float* o = reinterpret_cast<float*>(out + i);
                __builtin_nontemporal_store(f.x, o); __builtin_nontemporal_store(f.y, o + 1);
                __builtin_nontemporal_store(f.z, o + 2); __builtin_nontemporal_store(f.w, o + 3);
            } else out[i] = f;
        }
    }
}
__global__ void __launch_bounds__(1024) k_read(const int4* __restrict__ in, int* __restrict__ sink, size_t n4) {
    const size_t base = ((size_t)blockIdx.x * 4) * blockDim.x + threadIdx.x;
    int acc = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const size_t i = base + (size_t)q * blockDim.x; if (i < n4) { const int4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; } }
    if (acc == 0x7fffffff) sink[0] = acc;
}
__global__ void __launch_bounds__(1024) k_write(float4* __restrict__ out, size_t n4) {
    const size_t base = ((size_t)blockIdx.x * 4) * blockDim.x + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const size_t i = base + (size_t)q * blockDim.x; if (i < n4) out[i] = make_float4(1.f, 2.f, 3.f, (float)i); }
}

// lane-xor exchanges by DPP (no LDS crossbar): 1, 2 = quad_perm; 4, 8 = two row shifts with bank masks
__device__ __forceinline__ unsigned lane_xor_dpp(unsigned x, int lm) {
    if (lm == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, false);
    if (lm == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, false);
    if (lm == 4) {
        const int r = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);
        return (unsigned)__builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xF, 0xA, false);
    }
    if (lm == 8) {
        const int r = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x108, 0xF, 0x3, false);
        return (unsigned)__builtin_amdgcn_update_dpp(r, (int)x, 0x118, 0xF, 0xC, false);
    }
    return (unsigned)__shfl_xor((int)x, lm);
}
__global__ void k_dpp_check(unsigned* out) {
    const unsigned x = threadIdx.x * 7u + 3u;
    unsigned bad = 0;
    bad |= lane_xor_dpp(x, 1) != (unsigned)__shfl_xor((int)x, 1) ? 1u : 0u;
    bad |= lane_xor_dpp(x, 2) != (unsigned)__shfl_xor((int)x, 2) ? 2u : 0u;
    bad |= lane_xor_dpp(x, 4) != (unsigned)__shfl_xor((int)x, 4) ? 4u : 0u;
    bad |= lane_xor_dpp(x, 8) != (unsigned)__shfl_xor((int)x, 8) ? 8u : 0u;
    out[threadIdx.x] = bad;
}

template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    void* d; hipMalloc(&d, 1 << 20);
    {
        unsigned h[64]; k_dpp_check<<<1, 64>>>((unsigned*)d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        unsigned any = 0; for (int i = 0; i < 64; ++i) any |= h[i];
        printf("DPP lane-xor self-check: mismatch mask %u (0 = all of xor 1, 2, 4, 8 agree with __shfl_xor)\n", any);
    }
    const int it = 1 << 14;
    for (int rep = 0; rep < 2; ++rep) {
        for (int wgs : {256, 2048}) for (int th : {256, 1024}) {
            float a = timeit([&] { k_valu<<<wgs, th>>>((uint32_t*)d, it); });
            float b = timeit([&] { k_valu64<<<wgs, th>>>((uint64_t*)d, it); });
            float c = timeit([&] { k_lds_chain<<<wgs, th>>>((uint32_t*)d, it); });
            float e = timeit([&] { k_lds_rw<<<wgs, th>>>((uint64_t*)d, it); });
            float g = timeit([&] { k_barrier<<<wgs, 1024>>>((uint32_t*)d, it); });
            printf("wgs %4d threads %4d : valu32 %.2f ns/iter(5 ops)  valu64 %.2f ns/iter  lds chain %.1f ns/hop  lds rw step %.1f ns  barrier+lds %.1f ns\n",
                   wgs, th, a * 1e6 / it, b * 1e6 / it, c * 1e6 / it, e * 1e6 / it, g * 1e6 / it);
        }
    }
    {
        const int it64 = 1 << 14;
        float ms = timeit([&] { k_fma64<<<2048, 1024>>>((double*)d, it64); });
        const double fmas = 2048.0 * 1024 * 8 * it64;
        printf("f64 FMA rate: %.2f T FMA lane-ops/s (= %.1f TFLOP/s), %.3f ms\n", fmas / (ms * 1e-3) / 1e12, 2 * fmas / (ms * 1e-3) / 1e12, ms);
    }
    {
        const size_t bytes = 400u << 20, n4 = bytes / 16;
        int4* din; float4* dout; hipMalloc(&din, bytes); hipMalloc(&dout, bytes); hipMemset(din, 1, bytes);
        auto rep = [&](const char* name, float ms, double gb) { printf("%-34s %.4f ms  %.2f TB/s\n", name, ms, gb / ms); };
        for (int r = 0; r < 2; ++r) {
            rep("copy 400+400 MB, 4 vec/thread", timeit([&] { k_copy<4, false><<<(unsigned)((n4 + 4095) / 4096), 1024>>>(din, dout, n4); }), 0.8388608);
            rep("copy, 4 vec/thread, nontemporal st", timeit([&] { k_copy<4, true><<<(unsigned)((n4 + 4095) / 4096), 1024>>>(din, dout, n4); }), 0.8388608);
            rep("copy, 1 vec/thread", timeit([&] { k_copy<1, false><<<(unsigned)((n4 + 1023) / 1024), 1024>>>(din, dout, n4); }), 0.8388608);
            rep("copy, 8 vec/thread", timeit([&] { k_copy<8, false><<<(unsigned)((n4 + 8191) / 8192), 1024>>>(din, dout, n4); }), 0.8388608);
            rep("read 400 MB", timeit([&] { k_read<<<(unsigned)((n4 + 4095) / 4096), 1024>>>(din, (int*)d, n4); }), 0.4194304);
            rep("write 400 MB", timeit([&] { k_write<<<(unsigned)((n4 + 4095) / 4096), 1024>>>(dout, n4); }), 0.4194304);
        }
    }
    return 0;
}
