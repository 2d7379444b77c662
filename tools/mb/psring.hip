// Structure probes for the PS kernel (tuning aid, not product): how fast can "[n, s] int32 in -> LDS window ->
// K neighbour rows summed per (row, 4-column) item -> float32 quotient out" run under different work structures?
//   tile  : one workgroup per row tile, window = tile + halo, loads -> barrier -> items (the round-2 structure),
//           window loads through registers or by LDS-DMA
//   ring  : persistent workgroups, each streams ONE contiguous row range through a circular LDS buffer:
//           wave 0 issues LDS-DMA pieces (1 KiB) and publishes the landed row count, the other waves take
//           64-item chunks round robin, wait for "their rows + halo" to have landed and publish their progress;
//           no workgroup barrier after the prologue, no halo re-reads, loads and stores in flight all the time.
// Every variant is checked against a host loop on sampled rows.
//   hipcc --offload-arch=gfx950 -O3 -o psring psring.hip && ./psring [n] [s] [K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f div2(v2f a, v2f t) {
    v2f r; r.x = __builtin_amdgcn_rcpf(t.x); r.y = __builtin_amdgcn_rcpf(t.y);
    const v2f one = {1.0f, 1.0f};
    const v2f e = __builtin_elementwise_fma(-t, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    v2f q = a * r;
    v2f rem = __builtin_elementwise_fma(-t, q, a);
    q = __builtin_elementwise_fma(rem, r, q);
    rem = __builtin_elementwise_fma(-t, q, a);
    return __builtin_elementwise_fma(rem, r, q);
}
__device__ __forceinline__ void store_nt(float* p, float a, float b, float c, float d) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ void finish_item(float* out, const int4 own, const unsigned (&acc)[4]) {
    const v2f n0 = {(float)own.x, (float)own.y}, n1 = {(float)own.z, (float)own.w};
    const v2f d0 = {(float)((unsigned)own.x + acc[0]), (float)((unsigned)own.y + acc[1])};
    const v2f d1 = {(float)((unsigned)own.z + acc[2]), (float)((unsigned)own.w + acc[3])};
    const v2f q0 = div2(n0, d0), q1 = div2(n1, d1);
    store_nt(out, q0.x, q0.y, q1.x, q1.y);
}

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst, lds_dst + 1024) (lane l -> lds_dst + 16 l)
template <bool NT> __device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// eight consecutive pieces (8 KiB of the stream -> 8 KiB of LDS) in one statement: the instruction offset moves the
// global AND the LDS address, M0 and the address register are touched twice
template <bool NT> __device__ __forceinline__ void glds16x8(const char* gsrc, unsigned lds_dst) {
    unsigned keep;
    const char* g2 = gsrc + 4096;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off nt\n\tglobal_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, off nt\n\tglobal_load_lds_dwordx4 %2, off offset:1024 nt\n\t"
                     "global_load_lds_dwordx4 %2, off offset:2048 nt\n\tglobal_load_lds_dwordx4 %2, off offset:3072 nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "v"(g2), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, off offset:2048\n\tglobal_load_lds_dwordx4 %2, off offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "v"(g2), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wait_vm(int left) {   // wait until at most `left` vector-memory operations are outstanding
    switch (left) {
#define W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15)
        W(16) W(17) W(18) W(19) W(20) W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31)
        W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40) W(41) W(42) W(43) W(44) W(45) W(46) W(47)
        W(48) W(49) W(50) W(51) W(52) W(53) W(54) W(55) W(56) W(57) W(58) W(59) W(60) W(61) W(62)
#undef W
        default: break;   // 63 or more outstanding allowed: nothing to wait for
    }
}

// ------------------------------------------------------------------ tile structure
struct TileArgs { const int* in; float* out; int n, s, R, H, K; };
template <bool DMA, bool NT>
__global__ void __launch_bounds__(1024) k_tile(TileArgs a) {
    extern __shared__ int4 smem[];
    char* win = reinterpret_cast<char*>(smem);
    const int T = blockDim.x, tid = threadIdx.x;
    const int V = a.s / 4, rowb = a.s * 4;
    const int r0 = blockIdx.x * a.R, nr = min(a.R, a.n - r0);
    const int slo = max(0, r0 - a.H), shi = min(a.n, r0 + nr + a.H), wrows = shi - slo;
    const int zero_off = (a.R + 2 * a.H) * rowb;
    for (int i = tid; i < V; i += T) reinterpret_cast<int4*>(win + zero_off)[i] = make_int4(0, 0, 0, 0);
    const int total = wrows * V;
    const int4* g = reinterpret_cast<const int4*>(a.in + (size_t)slo * a.s);
    if (DMA) {
        const int wave = tid >> 6, lane = tid & 63, nw = T >> 6;
        const unsigned base = (unsigned)(uintptr_t)win;
        for (int p = wave; p * 64 < total; p += nw) {
            const int i = p * 64 + lane;
            if (i < total) glds16<NT>(g + i, __builtin_amdgcn_readfirstlane(base + p * 1024));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        int4* l = reinterpret_cast<int4*>(win);
        int i = tid;
        for (; i + 3 * T < total; i += 4 * T) {
            const int4 v0 = g[i], v1 = g[i + T], v2 = g[i + 2 * T], v3 = g[i + 3 * T];
            l[i] = v0; l[i + T] = v1; l[i + 2 * T] = v2; l[i + 3 * T] = v3;
        }
        for (; i < total; i += T) l[i] = g[i];
    }
    __syncthreads();
    const int items = nr * V, K2 = a.K / 2;
    for (int it = tid; it < items; it += T) {
        const int ri = it / V, c = it - ri * V;
        const int row = r0 + ri;
        unsigned acc[4] = {0, 0, 0, 0};
        for (int d = -K2; d <= K2; ++d) {
            if (d == 0) continue;
            const int q = row + d;
            const int off = (q >= 0 && q < a.n) ? (q - slo) * rowb : zero_off;
            const int4 v = *reinterpret_cast<const int4*>(win + off + c * 16);
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        }
        const int4 own = *reinterpret_cast<const int4*>(win + (row - slo) * rowb + c * 16);
        finish_item(a.out + (size_t)row * a.s + c * 4, own, acc);
    }
}

// ------------------------------------------------------------------ ring structure
struct RingArgs {
    const int* in; float* out;
    int n, s;
    int rows_per_wg;     // contiguous output rows per workgroup
    int ring_rows;       // rows the circular LDS buffer holds
    int H;               // rows a row may reach up / down (halo)
    int K;               // synthetic neighbours per item (rows r-K/2 .. r+K/2)
    int maxfly;          // LDS-DMA pieces in flight (<= 60)
    unsigned* status;    // [0] give-up flag
    unsigned long long* t_end;   // per workgroup: s_memrealtime at exit (optional)
};
struct RingCtl { int landed_rows; int abort; int prog[30]; };

template <bool NT>
__global__ void __launch_bounds__(1024) k_ring(RingArgs a) {
    extern __shared__ int4 smem[];
    const int T = blockDim.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ncw = (T >> 6) - 1;                      // compute waves
    const int V = a.s / 4, rowb = a.s * 4;
    const int ringb = a.ring_rows * rowb;
    char* ring = reinterpret_cast<char*>(smem);
    const int zero_off = ringb;                         // one all-zero row behind the ring
    RingCtl* ctl = reinterpret_cast<RingCtl*>(ring + ringb + rowb);
    const int ra = blockIdx.x * a.rows_per_wg, rb = min(a.n, ra + a.rows_per_wg);
    if (ra >= rb) return;
    const int slo = max(0, ra - a.H), shi = min(a.n, rb + a.H);
    for (int i = tid; i < V; i += T) reinterpret_cast<int4*>(ring + zero_off)[i] = make_int4(0, 0, 0, 0);
    if (tid == 0) { ctl->landed_rows = slo; ctl->abort = 0; }
    if (tid < 30) ctl->prog[tid] = tid < ncw ? ra : 0x7fffffff;
    __syncthreads();
    const int CAP = 1 << 19;
    if (wave == 0) {
        // ---------------- loader: the stream is the bytes of rows [slo, shi), cut into 1 KiB pieces
        __builtin_amdgcn_s_setprio(3);
        const char* gbase = reinterpret_cast<const char*>(a.in + (size_t)slo * a.s);
        const long long total = (long long)(shi - slo) * rowb;
        const int npieces = (int)((total + 1023) >> 10);
        const unsigned ring_base = (unsigned)(uintptr_t)ring;
        int issued = 0, landed = 0, ring_off = 0;
        long long free_hi = ringb;          // stream bytes below this may be issued (ring capacity behind the slowest reader)
        int landed_rows = slo;
        long long landed_row_end = rowb;    // stream byte where row `landed_rows` ends
        int spins = 0;
        while (landed < npieces) {
            bool can = issued < npieces && issued - landed < a.maxfly;
            if (can && ((long long)(issued + 1) << 10) > free_hi) {
                int p = lane < 30 ? __atomic_load_n(&ctl->prog[lane], __ATOMIC_RELAXED) : 0x7fffffff;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) p = min(p, __shfl_xor(p, o));
                const long long lo_row = max(0, min(p, rb) - a.H - slo);
                free_hi = lo_row * rowb + ringb;
                can = ((long long)(issued + 1) << 10) <= free_hi;
            }
            if (can && issued + 8 <= npieces && issued - landed + 8 <= a.maxfly && ring_off + 8192 <= ringb &&
                ((long long)(issued + 8) << 10) <= min(free_hi, total)) {
                const long long pos = (long long)issued << 10;
                glds16x8<NT>(gbase + pos + lane * 16, __builtin_amdgcn_readfirstlane(ring_base + ring_off));
                ring_off += 8192;
                if (ring_off == ringb) ring_off = 0;
                issued += 8;
                spins = 0;
                continue;
            }
            if (can) {
                const long long pos = (long long)issued << 10;
                if (ring_off + 1024 <= ringb) {
                    if (pos + lane * 16 < total) glds16<NT>(gbase + pos + lane * 16, __builtin_amdgcn_readfirstlane(ring_base + ring_off));
                    ring_off += 1024;
                    if (ring_off == ringb) ring_off = 0;
                    ++issued;
                } else {
                    // the piece wraps around the end of the ring: drain, then two partial instructions
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int na = (ringb - ring_off) >> 4;          // lanes that still fit
                    if (lane < na && pos + lane * 16 < total) glds16<NT>(gbase + pos + lane * 16, __builtin_amdgcn_readfirstlane(ring_base + ring_off));
                    if (lane < 64 - na && pos + (na + lane) * 16 < total) glds16<NT>(gbase + pos + (na + lane) * 16, __builtin_amdgcn_readfirstlane(ring_base));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    ring_off = (64 - na) * 16;
                    ++issued;
                    landed = issued - 1;   // falls through to the publish below via the wait branch next trip
                }
                spins = 0;
                continue;
            }
            if (issued > landed) {
                const int step = min(8, issued - landed);
                wait_vm(issued - landed - step);
                landed += step;
                const long long lb = min((long long)landed << 10, total);
                while (landed_row_end <= lb) { ++landed_rows; landed_row_end += rowb; }
                if (lane == 0) __atomic_store_n(&ctl->landed_rows, landed_rows, __ATOMIC_RELAXED);
                spins = 0;
            } else {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > CAP || __atomic_load_n(&ctl->abort, __ATOMIC_RELAXED)) {
                    if (lane == 0) { __atomic_store_n(&ctl->abort, 1, __ATOMIC_RELAXED); atomicOr(a.status, 1u); }
                    break;
                }
            }
        }
    } else {
        // ---------------- compute wave: 64-item chunks cw, cw + ncw, ...
        const int cw = wave - 1;
        const int items = (rb - ra) * V, K2 = a.K / 2;
        const int stride = ncw * 64;
        const int dr = stride / V, dc = stride - dr * V;
        int i0 = cw * 64;                                  // first item of the chunk (uniform)
        int ri = (i0 + lane) / V, c = (i0 + lane) - ri * V;   // this lane's item
        int slot = (ra - slo + ri) % a.ring_rows;
        int rf = i0 / V, cf = i0 - rf * V;                  // first item of the chunk as (row, vector), uniform
        bool dead = false;
        int seen = slo;
        for (; i0 < items; i0 += stride) {
            const int row_first = ra + rf;
            const int i_last = min(items - 1, i0 + 63);
            const int row_last = ra + rf + (cf + (i_last - i0)) / V;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __atomic_store_n(&ctl->prog[cw], row_first, __ATOMIC_RELAXED);
            const int need = min(row_last + a.H + 1, shi);
            int spins = 0;
            while (seen < need) {
                seen = __atomic_load_n(&ctl->landed_rows, __ATOMIC_RELAXED);
                if (seen >= need) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > CAP || __atomic_load_n(&ctl->abort, __ATOMIC_RELAXED)) { dead = true; break; }
            }
            if (dead) break;
            asm volatile("" ::: "memory");
            if (i0 + lane < items) {
                const int row = ra + ri;
                unsigned acc[4] = {0, 0, 0, 0};
                for (int d = -K2; d <= K2; ++d) {
                    if (d == 0) continue;
                    const int q = row + d;
                    int sl = slot + d;
                    sl = sl < 0 ? sl + a.ring_rows : (sl >= a.ring_rows ? sl - a.ring_rows : sl);
                    const int off = (q >= 0 && q < a.n) ? sl * rowb : zero_off;
                    const int4 v = *reinterpret_cast<const int4*>(ring + off + c * 16);
                    acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                }
                const int4 own = *reinterpret_cast<const int4*>(ring + slot * rowb + c * 16);
                finish_item(a.out + (size_t)row * a.s + c * 4, own, acc);
            }
            // next chunk of this wave
            c += dc; ri += dr; slot += dr;
            if (c >= V) { c -= V; ri += 1; slot += 1; }
            while (slot >= a.ring_rows) slot -= a.ring_rows;
            cf += dc; rf += dr;
            if (cf >= V) { cf -= V; rf += 1; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) {
            __atomic_store_n(&ctl->prog[cw], 0x7fffffff, __ATOMIC_RELAXED);
            if (dead) { __atomic_store_n(&ctl->abort, 1, __ATOMIC_RELAXED); atomicOr(a.status, 2u); }
        }
    }
    if (a.t_end && lane == 0 && wave == 1) a.t_end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------ plain copy (no window)
__global__ void __launch_bounds__(1024) k_copy1(const int4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) { const int4 v = in[i]; out[i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
}

template <typename F> float timeit(F f, int warm = 5, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < warm; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

static int check(const std::vector<int>& h_in, const float* d_out, int n, int s, int K, const char* name) {
    std::vector<float> h((size_t)n * s);
    CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
    long long bad = 0;
    const int K2 = K / 2;
    for (int r = 0; r < n; r += (r < 4096 || r > n - 4096) ? 1 : 97) {
        for (int c = 0; c < s; ++c) {
            unsigned long long t = (unsigned)h_in[(size_t)r * s + c];
            for (int d = -K2; d <= K2; ++d) if (d && r + d >= 0 && r + d < n) t += (unsigned)h_in[(size_t)(r + d) * s + c];
            const float want = (float)((double)h_in[(size_t)r * s + c] / (double)t);
            const float got = h[(size_t)r * s + c];
            if (!(want == got || (want != want && got != got))) { if (bad < 5) printf("  %s MISMATCH r %d c %d want %g got %g\n", name, r, c, want, got); ++bad; }
        }
    }
    return bad != 0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000, s = argc > 2 ? atoi(argv[2]) : 100, K = argc > 3 ? atoi(argv[3]) : 8;
    const size_t cells = (size_t)n * s;
    std::vector<int> h_in(cells);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < cells; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h_in[i] = (x & 7) ? (int)((x >> 8) & 1023) : 0; }
    int* d_in; float* d_out; unsigned* d_status; unsigned long long* d_tend;
    CK(hipMalloc(&d_in, cells * 4)); CK(hipMalloc(&d_out, cells * 4)); CK(hipMalloc(&d_status, 64)); CK(hipMalloc(&d_tend, 8 * 4096));
    CK(hipMemcpy(d_in, h_in.data(), cells * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_status, 0, 64));
    const double gb = cells * 8 / 1e9;
    int fails = 0;
    auto report = [&](const char* name, float ms) { printf("%-58s %.4f ms  %.2f TB/s\n", name, ms, gb / ms); fflush(stdout); };
    for (int rep = 0; rep < 2; ++rep) {
        report("plain copy, 1 vec/thread", timeit([&] { k_copy1<<<(unsigned)((cells / 4 + 1023) / 1024), 1024>>>((const int4*)d_in, (float4*)d_out, cells / 4); }));
        // tile structure
        struct TC { int threads, lds, H; bool dma, nt; };
        const TC tcs[] = {{1024, 80 << 10, 16, false, false}, {1024, 80 << 10, 16, true, false}, {1024, 80 << 10, 16, true, true},
                          {1024, 80 << 10, 6, true, true}, {512, 40 << 10, 6, true, true}, {512, 40 << 10, 16, true, true},
                          {256, 20 << 10, 6, true, true}, {1024, 53 << 10, 6, true, true},
                          {1024, 80 << 10, 6, false, false}, {512, 40 << 10, 6, false, false}, {512, 40 << 10, 8, true, true}, {768, 53 << 10, 8, true, true},
                          {1024, 80 << 10, 8, true, true}, {256, 20 << 10, 8, true, true}, {512, 32 << 10, 8, true, true}, {512, 40 << 10, 8, true, false}};
        for (const TC& t : tcs) {
            TileArgs a{d_in, d_out, n, s, 0, t.H, K};
            a.R = (t.lds - 64) / (s * 4) - 2 * t.H - 1;
            if (a.R < 4) continue;
            const unsigned grid = (n + a.R - 1) / a.R;
            char name[128]; snprintf(name, sizeof name, "tile  thr %4d lds %3dK H %2d R %3d %s%s", t.threads, t.lds >> 10, t.H, a.R, t.dma ? "dma" : "reg", t.nt ? " nt" : "");
            CK(hipMemset(d_out, 0xff, cells * 4));
            float ms;
            if (t.dma && t.nt) { CK(hipFuncSetAttribute((const void*)k_tile<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, t.lds)); ms = timeit([&] { k_tile<true, true><<<grid, t.threads, t.lds>>>(a); }); }
            else if (t.dma) { CK(hipFuncSetAttribute((const void*)k_tile<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, t.lds)); ms = timeit([&] { k_tile<true, false><<<grid, t.threads, t.lds>>>(a); }); }
            else { CK(hipFuncSetAttribute((const void*)k_tile<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, t.lds)); ms = timeit([&] { k_tile<false, false><<<grid, t.threads, t.lds>>>(a); }); }
            CK(hipGetLastError());
            report(name, ms);
            if (rep == 0) fails += check(h_in, d_out, n, s, K, name);
        }
        // ring structure
        struct RC { int threads, wg_per_cu, lds, maxfly, split; bool nt; };
        const RC rcs[] = {{1024, 1, 150 << 10, 48, 1, true}, {1024, 1, 150 << 10, 24, 1, true}, {1024, 1, 150 << 10, 48, 1, false},
                          {1024, 2, 78 << 10, 32, 1, true}, {512, 2, 78 << 10, 32, 1, true}, {512, 4, 39 << 10, 16, 1, true},
                          {1024, 2, 78 << 10, 32, 4, true}, {512, 2, 78 << 10, 32, 4, true}, {1024, 1, 150 << 10, 48, 4, true}};
        for (const RC& t : rcs) {
            RingArgs a{d_in, d_out, n, s, 0, 0, 16, K, t.maxfly, d_status, d_tend};
            a.ring_rows = (t.lds - 256 - s * 4) / (s * 4);
            const int wgs = 256 * t.wg_per_cu * t.split;
            a.rows_per_wg = (n + wgs - 1) / wgs;
            if (a.ring_rows < 2 * a.H + 8) continue;
            const unsigned grid = (n + a.rows_per_wg - 1) / a.rows_per_wg;
            const int lds = a.ring_rows * s * 4 + s * 4 + 256;
            char name[160]; snprintf(name, sizeof name, "ring  thr %4d wg/cu %d x%d lds %3dK ring %3d rows fly %2d%s", t.threads, t.wg_per_cu, t.split, lds >> 10, a.ring_rows, t.maxfly, t.nt ? " nt" : "");
            CK(hipMemset(d_out, 0xff, cells * 4));
            float ms;
            if (t.nt) { CK(hipFuncSetAttribute((const void*)k_ring<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); ms = timeit([&] { k_ring<true><<<grid, t.threads, lds>>>(a); }); }
            else { CK(hipFuncSetAttribute((const void*)k_ring<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); ms = timeit([&] { k_ring<false><<<grid, t.threads, lds>>>(a); }); }
            CK(hipGetLastError());
            report(name, ms);
            unsigned st = 0; CK(hipMemcpy(&st, d_status, 4, hipMemcpyDeviceToHost));
            if (st) { printf("  ring gave up: status %u\n", st); CK(hipMemset(d_status, 0, 64)); ++fails; }
            if (rep == 0) {
                fails += check(h_in, d_out, n, s, K, name);
                std::vector<unsigned long long> te(grid);
                CK(hipMemcpy(te.data(), d_tend, grid * 8, hipMemcpyDeviceToHost));
                std::sort(te.begin(), te.end());
                printf("  workgroup end times (100 MHz ticks after the first to end): median %llu  p90 %llu  last %llu\n",
                       te[grid / 2] - te[0], te[grid * 9 / 10] - te[0], te[grid - 1] - te[0]);
            }
        }
    }
    printf(fails ? "FAILED %d\n" : "all variants match the host loop\n", fails);
    return fails ? 1 : 0;
}
