// K3: exclusion-sum + PS over the junction x sample matrix, K4: '.3f' quantise,
// and the --lowCoverageNan scatter.
//
// Replaces SPLICEDICE.calculatePsi (SPLICEDICE.py:297-310): for every junction row the
// reference adds up the count rows of all overlapping junctions (float64 accumulation of
// integer-valued float32 counts -> exact integer sums) and stores
// float32(float64(incl) / float64(incl + excl)).
//
// Design (HBM-bound, 8 algorithmic bytes per PS entry: 4 B count in + 4 B PS out):
//   * one workgroup owns a tile of consecutive output rows x a chunk of columns (all columns
//     up to 256 samples, 128-column chunks above); the chunks of one row tile are dispatched
//     back to back on ONE XCD -- with chunks spread over XCDs and time the cache lines straddling
//     a chunk boundary left L2 as two masked partial writes and the store stream ran at 1.5 TB/s
//     (2M x 500: 2.72 ms -> 1.85 ms with this mapping alone);
//   * it stages the rows [r0 - halo, r1 + halo) of the count matrix into LDS with one
//     coalesced 16 B/lane copy that depends on nothing but the tile index, so every load
//     of the tile is in flight at once (overlap clusters are gene sized: the neighbours of
//     a row lie a few rows away in (chrom,left,right,strand) order);
//   * meanwhile the tile's CSR segment is read and turned into LDS byte offsets of the
//     neighbour rows (or a sentinel for a neighbour outside the staged window);
//   * every (row, 4-column) item then gathers its neighbour rows from LDS with
//     ds_read_b128 and stores a float4 -- each count is read from HBM once per tile
//     (+halo, which is an L2 hit) and each PS value is written once;
//   * a neighbour outside the window (arbitrary user CSR, unusually long junction) is read
//     from global memory instead, so ANY valid CSR gives exact results; the halo only
//     decides how often that slower path runs.
// Arithmetic: sums are exact integers.  When a tile's bound (max count) * (max degree + 1)
// is below 2^24, sums are kept in 32 bits and the quotient is one IEEE float32 division:
// both operands are then exact float32 values and a correctly rounded float32 quotient of
// two such integers equals float32(float64 quotient) (a double rounding could only differ
// if the exact quotient were within 2^-53 of a float32 midpoint, impossible for a
// denominator below 2^24).  Otherwise 64-bit sums and the float64 division are used.
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

struct PsArgs {
    const int32_t* counts;
    const int64_t* row_ptr;
    const int32_t* col;
    int64_t* excl;
    float* ps;
    int64_t n;
    int s;
    int tile_rows;   // output rows per tile
    int halo;        // extra rows staged on each side of the tile
    int chunk_cols;  // columns per chunk == LDS row stride (multiple of VEC)
    int col_cap;     // LDS capacity for staged neighbour offsets (ints)
    int n_tiles;
    int tiles_per_xcd;  // 0 = no remap
    int n_chunks;       // column chunks per row tile
    int prio;           // raise the wave priority while the window loads go out (param ps.prio, default 1)
    int nt;             // non-temporal window loads (param ps.nt_loads, default 1; unchunked tables only)
    // second-generation kernel only:
    const uint32_t* reach;  // reach words of the lists per block of 16 rows (below | beyond << 8, 255 = that far or further), NULL: stage `halo` rows
    int lv_shift;       // column chunks: 6 - log2(vectors per row segment) (chunk_cols / 4 divides 64)
};

__device__ __forceinline__ int4 nt_load(const int4* p) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int nt_load(const int* p) { return __builtin_nontemporal_load(p); }

template <int VEC> struct Vt;
template <> struct Vt<4> { typedef int4 I; };
template <> struct Vt<1> { typedef int I; };

__device__ __forceinline__ float ps_value64(unsigned incl, unsigned long long excl) {
    // float32(float64(incl) / float64(incl + excl)), SPLICEDICE.py:306; 0/0 -> NaN
    const double a = (double)incl;
    const double t = a + (double)excl;
    return (float)(a / t);
}

__device__ __forceinline__ unsigned vmax(const int4& v) {
    return max(max((unsigned)v.x, (unsigned)v.y), max((unsigned)v.z, (unsigned)v.w));
}
__device__ __forceinline__ unsigned vmax(const int& v) { return (unsigned)v; }

template <typename ACC> __device__ __forceinline__ void acc_add(ACC (&acc)[4], const int4& v) {
    acc[0] += (unsigned)v.x; acc[1] += (unsigned)v.y; acc[2] += (unsigned)v.z; acc[3] += (unsigned)v.w;
}
template <typename ACC> __device__ __forceinline__ void acc_add(ACC (&acc)[1], const int& v) { acc[0] += (unsigned)v; }

__device__ __forceinline__ unsigned comp(const int4& v, int q) { return (unsigned)(q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w); }
__device__ __forceinline__ unsigned comp(const int& v, int) { return (unsigned)v; }

template <int VEC> __device__ __forceinline__ void store_f(float* p, const float (&o)[VEC]);
template <> __device__ __forceinline__ void store_f<4>(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
template <> __device__ __forceinline__ void store_f<1>(float* p, const float (&o)[1]) { *p = o[0]; }
__device__ __forceinline__ void store_f4_nt(float* p, const float (&o)[4]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f v = {o[0], o[1], o[2], o[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
}

template <int VEC, typename ACC> __device__ __forceinline__ void store_excl(int64_t* p, const ACC (&acc)[VEC]) {
    if (VEC == 4) {
        longlong2 a, b;
        a.x = (long long)acc[0]; a.y = (long long)acc[1]; b.x = (long long)acc[VEC - 2]; b.y = (long long)acc[VEC - 1];
        reinterpret_cast<longlong2*>(p)[0] = a;
        reinterpret_cast<longlong2*>(p)[1] = b;
    } else {
        *p = (int64_t)acc[0];
    }
}

__device__ __forceinline__ float div_small_ints(float a, float t) {
    // a / t for exact integers 0 <= a <= t < 2^24: the Newton / residual steps of the IEEE
    // float32 division sequence without v_div_scale / v_div_fixup (no scaling is ever needed in
    // this range), hence the same correctly rounded quotient in 8 instead of 13 instructions.
    // t == 0 (then a == 0): rcp = inf, e = NaN -> NaN, as 0/0.
    float r = __builtin_amdgcn_rcpf(t);
    const float e = __builtin_fmaf(-t, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float rem = __builtin_fmaf(-t, q, a);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-t, q, a);
    return __builtin_fmaf(rem, r, q);
}

// The same sequence on two quotients at once: v_pk_fma_f32 / v_pk_mul_f32 carry two float32 lanes
// per instruction (the reciprocal seed has no packed form).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f div_small_ints2(v2f a, v2f t) {
    v2f r;
    r.x = __builtin_amdgcn_rcpf(t.x); r.y = __builtin_amdgcn_rcpf(t.y);
    const v2f one = {1.0f, 1.0f};
    const v2f e = __builtin_elementwise_fma(-t, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    v2f q = a * r;
    v2f rem = __builtin_elementwise_fma(-t, q, a);
    q = __builtin_elementwise_fma(rem, r, q);
    rem = __builtin_elementwise_fma(-t, q, a);
    return __builtin_elementwise_fma(rem, r, q);
}

// k = round-half-even(x * 1000) for a float32 x in [0, 1] (NaN stays NaN), as the decimal formatter of '%.3f' rounds the
// EXACT value of x -- without float64 (v_cvt_f64_f32, v_rndne_f64, v_cvt_f32_f64 are quarter-rate: 13 issue slots per
// value): p = fl(x * 1000) and e = fma(x, 1000, -p) give the exact product p + e (|e| <= ulp(p) / 2 <= 2^-15); k0 =
// rint(p) is the answer unless p sits exactly on a half integer and e is not zero -- then the true value lies on e's
// side of it.  (|p - k0| < 1/2 implies |p - k0| <= 1/2 - ulp(p), so e cannot carry the sum across a half integer.)
__device__ __forceinline__ v2f thousandths2(v2f x) {
    const v2f th = {1000.0f, 1000.0f};
    const v2f p = x * th;
    const v2f e = __builtin_elementwise_fma(x, th, -p);
    v2f k0, k;
    k0.x = __builtin_rintf(p.x); k0.y = __builtin_rintf(p.y);
    const v2f r = p - k0;
    k.x = (__builtin_fabsf(r.x) == 0.5f && e.x != 0.0f) ? p.x + __builtin_copysignf(0.5f, e.x) : k0.x;
    k.y = (__builtin_fabsf(r.y) == 0.5f && e.y != 0.0f) ? p.y + __builtin_copysignf(0.5f, e.y) : k0.y;
    return k;
}
// float32(k / 1000.0) for an integer-valued k in [0, 1000]: k * 0.001f and one residual step give the correctly rounded
// quotient for all 1001 keys (exhaustive test in rational arithmetic: test_ps_of_key_formula_reproduces_the_table)
__device__ __forceinline__ v2f key_over_1000(v2f k) {
    const v2f m = {0.001f, 0.001f}, th = {1000.0f, 1000.0f};
    const v2f q = k * m;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-q, th, k), m, q);
}

// Fast item: every neighbour offset comes from the LDS stage and lies inside the staged window,
// the tile's bound keeps incl + excl < 2^24 -> 32-bit sums, float32 quotient.  Returns false
// (nothing stored) as soon as a neighbour is outside the window; the caller then runs ps_item_slow.
template <int VEC, bool WEXCL, bool WPS, bool CHECK, bool Q3, int ABL>
__device__ __forceinline__ bool ps_item_fast(const PsArgs& a, const char* tileB, const int* colL, int k0, int k1,
                                             int cbytes, int own_off, int64_t out_index, int zero_off) {
    typedef typename Vt<VEC>::I VI;
    unsigned acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0;
    // batches of GB neighbours: all offset reads, then all row reads, then the adds -- two LDS
    // round trips per batch instead of two per neighbour; short batches are padded with the
    // all-zero row kept behind the window
    constexpr int GB = 4;
    for (int k = k0; k < k1; k += GB) {
        // unconditional reads (a short last batch runs into the next row's entries or the words
        // behind the stage -- always inside the LDS allocation) and a select: predicated reads
        // compile to an exec-mask round trip per element
        int raw[GB], off[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) raw[u] = colL[k + u];
#pragma unroll
        for (int u = 0; u < GB; ++u) off[u] = (k + u < k1) ? raw[u] : zero_off;
        if (CHECK) {      // (a tile whose staged neighbours all lie inside the window skips this)
            int mn = off[0];
#pragma unroll
            for (int u = 1; u < GB; ++u) mn = min(mn, off[u]);
            if (mn < 0) return false;
        }
        VI v[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) v[u] = *reinterpret_cast<const VI*>(tileB + off[u] + cbytes);
#pragma unroll
        for (int u = 0; u < GB; ++u) acc_add(acc, v[u]);
    }
    const VI own = *reinterpret_cast<const VI*>(tileB + own_off + cbytes);
    if (WPS) {
        float o[VEC];
        if (VEC == 4) {
            // two packed quotients per instruction
#pragma unroll
            for (int q = 0; q < VEC; q += 2) {
                const unsigned i0 = comp(own, q), i1 = comp(own, q + 1);
                const v2f num = {(float)i0, (float)i1};
                const v2f den = {(float)(i0 + acc[q]), (float)(i1 + acc[q + 1])};
                v2f ps2 = div_small_ints2(num, den);
                if (Q3) ps2 = key_over_1000(thousandths2(ps2));     // fused K4: float32(f'{ps:.3f}')
                o[q] = ps2.x; o[q + 1] = ps2.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const unsigned in = comp(own, q);
                o[q] = div_small_ints((float)in, (float)(in + acc[q]));
                if (Q3) o[q] = div_small_ints((float)rint((double)o[q] * 1000.0), 1000.0f);
            }
        }
        // non-temporal store: the PS matrix is written once and not re-read by this kernel, keeping it
        // out of the L2 leaves the halo rows of the count matrix there (-2.5 % at 1M x 100; ABL 8 = plain store)
        if (VEC == 4 && !(ABL & 8)) { if (!(ABL & 4)) store_f4_nt(a.ps + out_index, reinterpret_cast<const float (&)[4]>(o)); }
        else if (!(ABL & 4)) store_f<VEC>(a.ps + out_index, o);
    }
    if (WEXCL) store_excl<VEC, unsigned>(a.excl + out_index, acc);
    return true;
}

// General item: neighbours from the LDS window or from global memory, indices from the LDS stage
// or from global memory, 64-bit sums, float64 division.  Exact for any valid CSR and any counts.
template <int VEC, bool WEXCL, bool WPS, bool Q3, int ABL>
__device__ __forceinline__ void ps_item_slow(const PsArgs& a, const char* tileB, const int* colL, bool col_in_lds,
                                             int64_t kbase, int k0, int k1, int slo, int wrows, int ldw, int c0, int cvec,
                                             int own_off, int64_t out_index, int wbase) {
    typedef typename Vt<VEC>::I VI;
    unsigned long long acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0;
    const int cbytes = cvec * 4;
    for (int k = k0; k < k1; ++k) {
        int off;
        if (col_in_lds) {
            off = colL[k];
        } else {
            const int j = a.nt ? __builtin_nontemporal_load(a.col + kbase + k) : a.col[kbase + k];
            const unsigned rel = (unsigned)(j - slo);
            off = rel < (unsigned)wrows ? (int)((unsigned)(j - wbase) * (unsigned)ldw * 4u) : -1 - j;
        }
        VI v;
        if (off >= 0) v = *reinterpret_cast<const VI*>(tileB + off + cbytes);
        else v = *reinterpret_cast<const VI*>(a.counts + (int64_t)(-1 - off) * a.s + c0 + cvec);
        acc_add(acc, v);
    }
    const VI own = *reinterpret_cast<const VI*>(tileB + own_off + cbytes);
    if (WPS) {
        float o[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            o[q] = ps_value64(comp(own, q), acc[q]);
            if (Q3) o[q] = (float)(rint((double)o[q] * 1000.0) / 1000.0);
        }
        if (!(ABL & 4)) store_f<VEC>(a.ps + out_index, o);
    }
    if (WEXCL) store_excl<VEC, unsigned long long>(a.excl + out_index, acc);
}

// Q3: store the '.3f' text round trip of PS (K4 fused into the store, param ps.quantize3).
// ABL: timing experiments only (param ps.ablate: 1 skip gather, 2 skip staging loads, 4 skip stores);
//      the production instantiations have ABL = 0 and no trace of it.
template <int VEC, bool WEXCL, bool WPS, bool Q3, int ABL>
__device__ __forceinline__ void ps_tile_work(const PsArgs& a, const int bid, const int tid, int4* smem4) {
    typedef typename Vt<VEC>::I VI;
    const int win_cap = a.tile_rows + 2 * a.halo;
    int* tileL = reinterpret_cast<int*>(smem4);
    int* colL = tileL + (size_t)(win_cap + 1) * a.chunk_cols;   // row win_cap is the all-zero padding row
    int* rpL = colL + a.col_cap;
    unsigned* red = reinterpret_cast<unsigned*>(rpL + a.tile_rows + 1);  // [0] max count, [1] max degree

    const int T = blockDim.x;
    // 1-D grid: the column chunks of one row tile are CONSECUTIVE work on ONE XCD, so the cache
    // lines that two chunks share at a chunk boundary (rows are not 128 B aligned when 4*s is
    // not) meet in that XCD's L2 and leave it as full lines instead of two masked partial writes
    int tile = bid / a.n_chunks;
    int chunk = bid - tile * a.n_chunks;
    if (a.tiles_per_xcd) {
        // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of
        // tiles so that neighbouring tiles (which share halo rows) share one L2.
        const int k = bid >> 3;
        const int t_local = k / a.n_chunks;
        chunk = k - t_local * a.n_chunks;
        tile = (bid & 7) * a.tiles_per_xcd + t_local;
    }
    if (tile >= a.n_tiles) return;
    const int c0 = chunk * a.chunk_cols;
    const int cwc = min(a.chunk_cols, a.s - c0);
    const int V = cwc / VEC;
    const int ldw = a.chunk_cols;
    const int64_t r0 = (int64_t)tile * a.tile_rows;
    const int nr = (int)min((int64_t)a.tile_rows, a.n - r0);
    const int slo = (int)max((int64_t)0, r0 - a.halo);
    const int shi = (int)min(a.n, r0 + nr + a.halo);
    const int wrows = shi - slo;
    // A wave that is sending out its window loads goes ahead of the waves that are gathering: the memory pipe stays fed
    // while the co-resident workgroup computes (measured in one process: 0.1845 -> 0.1757 ms at 1 M x 100, 2.02 -> 1.82 ms
    // at 2 M x 500; priority levels 1..3 are equivalent, holding it until the first barrier is worse).
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    if (tid < 3) red[tid] = 0u;                 // [0] max count, [1] max degree, [2] a staged neighbour is outside the window
    for (int i = tid; i < a.chunk_cols; i += T) tileL[(size_t)win_cap * a.chunk_cols + i] = 0;
    const int zero_off = win_cap * a.chunk_cols * 4;

    // ---- CSR row pointers of the tile (their loads go out first, the data loads right behind)
    int64_t rp_mine[2];
    rp_mine[0] = (tid <= nr) ? a.row_ptr[r0 + tid] : 0;
    rp_mine[1] = (tid + T <= nr) ? a.row_ptr[r0 + tid + T] : 0;
    const int64_t kbase = a.row_ptr[r0];

    // ---- stage rows [slo, shi) x cols [c0, c0+cwc) into LDS, 16 B per lane, 4 loads in flight
    unsigned cmax = 0;
    {
        const int total = wrows * V;
        if (cwc == a.s) {
            const VI* g = reinterpret_cast<const VI*>(a.counts + (int64_t)slo * a.s);
            VI* l = reinterpret_cast<VI*>(tileL);
            int i = tid;
            if (ABL & 2) i = total;
            for (; i + 3 * T < total; i += 4 * T) {
                VI v0, v1, v2, v3;
                if (a.nt) {
                    v0 = nt_load(g + i); v1 = nt_load(g + i + T); v2 = nt_load(g + i + 2 * T); v3 = nt_load(g + i + 3 * T);
                } else { v0 = g[i]; v1 = g[i + T]; v2 = g[i + 2 * T]; v3 = g[i + 3 * T]; }
                l[i] = v0; l[i + T] = v1; l[i + 2 * T] = v2; l[i + 3 * T] = v3;
                cmax = max(max(cmax, vmax(v0)), max(vmax(v1), max(vmax(v2), vmax(v3))));
            }
            for (; i < total; i += T) { const VI v = g[i]; l[i] = v; cmax = max(cmax, vmax(v)); }
        } else {
            // row segments of cwc columns: 4 independent loads in flight per lane
            const int* gbase = a.counts + (int64_t)slo * a.s + c0;
            for (int i = (ABL & 2) ? total : tid; i < total; i += 4 * T) {
                VI v[4];
                int lofs[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ii = min(i + q * T, total - 1);      // the tail re-reads the last vector
                    const int rr = ii / V, cc = ii - rr * V;
                    v[q] = *reinterpret_cast<const VI*>(gbase + (int64_t)rr * a.s + cc * VEC);
                    lofs[q] = rr * ldw + cc * VEC;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (i + q * T < total) {
                        *reinterpret_cast<VI*>(tileL + lofs[q]) = v[q];
                        cmax = max(cmax, vmax(v[q]));
                    }
                }
            }
        }
    }
    if (a.prio) __builtin_amdgcn_s_setprio(0);
    // ---- CSR segment -> LDS: relative row pointers, neighbour LDS offsets, max degree
    const int64_t nk = a.row_ptr[r0 + nr] - kbase;
    const bool col_in_lds = nk <= (int64_t)a.col_cap;
    unsigned dmax = 0;
    if (tid <= nr) rpL[tid] = (int)(rp_mine[0] - kbase);
    if (tid + T <= nr) rpL[tid + T] = (int)(rp_mine[1] - kbase);
    for (int i = tid + 2 * T; i <= nr; i += T) rpL[i] = (int)(a.row_ptr[r0 + i] - kbase);
    if (col_in_lds) {
        bool outside = false;
        for (int k = tid; k < (int)nk; k += T) {
            const int j = a.col[kbase + k];
            const unsigned rel = (unsigned)(j - slo);
            outside = outside || rel >= (unsigned)wrows;
            colL[k] = rel < (unsigned)wrows ? (int)(rel * (unsigned)ldw * 4u) : -1 - j;
        }
        if (__ballot(outside) != 0ull && (tid & 63) == 0) atomicOr(&red[2], 1u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, (unsigned)__shfl_xor((int)cmax, o));
    if ((tid & 63) == 0) atomicMax(&red[0], cmax);
    __syncthreads();
    for (int i = tid; i < nr; i += T) dmax = max(dmax, (unsigned)(rpL[i + 1] - rpL[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, (unsigned)__shfl_xor((int)dmax, o));
    if ((tid & 63) == 0) atomicMax(&red[1], dmax);
    __syncthreads();
    const unsigned tile_cmax = red[0], tile_dmax = red[1];
    const unsigned thr = 0xFFFFFFu / (tile_dmax + 1u);   // per-count bound keeping incl+excl < 2^24
    const bool fast = tile_cmax <= thr && col_in_lds;     // block-uniform
    const bool all_in = red[2] == 0u;                     // block-uniform: no per-batch window check needed

    // ---- per (row, vector) item: gather neighbours from LDS, divide, store
    {
        const char* tileB = reinterpret_cast<const char*>(tileL);
        const int items = nr * V;
        int ri = tid / V, c = tid - ri * V;
        const int dr = T / V, dc = T - dr * V;
        const int own_base = (int)(r0 - slo);
        for (int it = tid; it < items; it += T) {
            const int k0 = rpL[ri], k1 = (ABL & 1) ? k0 : rpL[ri + 1];
            const int64_t o = (r0 + ri) * a.s + c0 + c * VEC;
            const int own_off = (own_base + ri) * ldw * 4;
            bool done = false;
            if (fast)
                done = all_in ? ps_item_fast<VEC, WEXCL, WPS, false, Q3, ABL>(a, tileB, colL, k0, k1, c * VEC * 4, own_off, o, zero_off)
                              : ps_item_fast<VEC, WEXCL, WPS, true, Q3, ABL>(a, tileB, colL, k0, k1, c * VEC * 4, own_off, o, zero_off);
            if (!done)
                ps_item_slow<VEC, WEXCL, WPS, Q3, ABL>(a, tileB, colL, col_in_lds, kbase, k0, k1, slo, wrows, ldw, c0, c * VEC,
                                              own_off, o, slo);
            c += dc; ri += dr;
            if (c >= V) { c -= V; ri += 1; }
        }
    }
}

// One workgroup per (tile, chunk) work item.  (A resident set of workgroups striding over the work items --
// the next tile's loads issued behind the current tile's stores, same LDS -- was measured against this in
// one process: 0.203 ms vs 0.173 ms at 1 M x 100, 2.05 ms vs 1.79 ms at 2 M x 500; the hardware's dynamic
// dispatch balances the tiles better than a static stride and the store drain is not what limits a CU.)
template <int VEC, bool WEXCL, bool WPS, bool Q3, int ABL>
__global__ void __launch_bounds__(1024) ps_tile_kernel(PsArgs a) {
    extern __shared__ int4 smem4[];
    ps_tile_work<VEC, WEXCL, WPS, Q3, ABL>(a, blockIdx.x, threadIdx.x, smem4);
}


// maximum over the 64 lanes by DPP steps (VALU latency; six ds_bpermute round trips on the tile's critical path were
// measurable): the result is valid in lane 63
__device__ __forceinline__ unsigned wave_max_dpp(unsigned x) {
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true));   // row_half_mirror
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true));   // row_mirror
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true));   // row_bcast:15 -> rows 1, 3
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true));   // row_bcast:31 -> rows 2, 3
    return x;
}

// ---------------------------------------------------------------------------------------------------
// Register-staged tile kernel, second generation (VEC = 4 tables).  The window passes through the VGPRs
// (which gives the tile's count bound for free; an LDS-DMA staging, global_load_lds_dwordx4, was built in round 3,
// bit-exact and slower -- DESIGN appendix A.3):
//   * EVERY global load of the tile is issued before the first one is waited for: the tile's own rows (up to four
//     vectors per thread), then -- behind one scalar-load round trip for the first / last row pointer and the reach
//     words -- the halo rows the lists really reach (reach words of the clustering kernel, 16 rows in all on
//     gene-shaped data instead of 2 x 16), the neighbour indices and the row pointers; hipcc's counted vmcnt waits
//     then retire them in issue order while the data moves to LDS;
//   * one workgroup barrier: the count bound goes through one LDS word per wave (DPP reduction, sixteen broadcast
//     reads after the barrier), the degree enters per item: an item is fast iff (degree + 1) * bound < 2^24.
template <bool CHUNKED, bool WEXCL, bool WPS, bool Q3>
__global__ void __launch_bounds__(1024, 8) ps_tile_v3_kernel(PsArgs a) {
    extern __shared__ int4 smem4[];
    constexpr int VEC = 4;
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = T >> 6;
    const int win_cap = a.tile_rows + 2 * a.halo;
    int* tileL = reinterpret_cast<int*>(smem4);
    int* colL = tileL + (win_cap + 1) * a.chunk_cols;            // row win_cap is the all-zero padding row
    int* rpL = colL + a.col_cap;
    unsigned* red = reinterpret_cast<unsigned*>(rpL + ((a.tile_rows + 1 + 3) & ~3));   // one word per wave (16-byte aligned)

    const int bid = blockIdx.x;
    int tile, chunk = 0;
    if (!CHUNKED) {
        tile = a.tiles_per_xcd ? (bid & 7) * a.tiles_per_xcd + (bid >> 3) : bid;
    } else {
        tile = bid / a.n_chunks;
        chunk = bid - tile * a.n_chunks;
        if (a.tiles_per_xcd) {
            const int k = bid >> 3;
            const int t_local = k / a.n_chunks;
            chunk = k - t_local * a.n_chunks;
            tile = (bid & 7) * a.tiles_per_xcd + t_local;
        }
    }
    if (tile >= a.n_tiles) return;
    const int c0 = chunk * a.chunk_cols;
    const int cwc = CHUNKED ? min(a.chunk_cols, a.s - c0) : a.s;
    const int V = cwc / VEC;
    const int ldw = a.chunk_cols, LV = ldw / VEC;
    const int lsh = 6 - a.lv_shift;                        // column chunks: log2(LV)
    const int r0 = tile * a.tile_rows;                     // (n < 2^31)
    const int nr = min(a.tile_rows, (int)a.n - r0);
    const int wbase = r0 - a.halo;                         // row that LDS window row 0 stands for (may be negative)
    const int rowb = ldw * 4;
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    for (int i = tid; i < LV; i += T) reinterpret_cast<int4*>(tileL + win_cap * a.chunk_cols)[i] = make_int4(0, 0, 0, 0);
    const int zero_off = win_cap * a.chunk_cols * 4;
    int4* win4 = reinterpret_cast<int4*>(tileL);

    // ---- (1) the tile's own rows: up to four vectors per thread, all in flight
    const char* gtile = reinterpret_cast<const char*>(a.counts + (int64_t)r0 * a.s + c0);
    const int64_t row_stride = (int64_t)a.s * 4;
    const int ctot = nr * LV;
    int4 cv[4];
    bool cok[4];
    {
        // unconditional loads at clamped positions (a predicated load drags exec-mask code and early waits into the
        // sequence); the predicate acts on the LDS store
        const int4* g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + k * T;
            cok[k] = idx < ctot;
            const int ic = min(idx, ctot - 1);
            if (!CHUNKED) g[k] = reinterpret_cast<const int4*>(gtile) + ic;
            else {
                const int rr = ic >> lsh, cc = ic & (LV - 1);
                cok[k] = cok[k] && cc < V;
                g[k] = reinterpret_cast<const int4*>(gtile + rr * row_stride) + min(cc, V - 1);
            }
        }
        if (a.nt) { cv[0] = nt_load(g[0]); cv[1] = nt_load(g[1]); cv[2] = nt_load(g[2]); cv[3] = nt_load(g[3]); }
        else      { cv[0] = *g[0]; cv[1] = *g[1]; cv[2] = *g[2]; cv[3] = *g[3]; }
    }

    // ---- (2) uniform scalars of the tile (scalar loads: they return on lgkmcnt, the vector queue stays untouched)
    long long kbase, kend;
    int nlo = a.halo, nhi = a.halo;
    {
        const int64_t* pf = a.row_ptr + r0;
        const int64_t* pl = pf + nr;
        if (a.reach) {
            const int b0 = r0 >> 4, bl = (r0 + nr - 1) >> 4;
            const uint32_t* q0 = a.reach + b0;
            const uint32_t* q1 = a.reach + min(b0 + 1, bl);
            const uint32_t* q2 = a.reach + max(bl - 1, b0);
            const uint32_t* q3 = a.reach + bl;
            unsigned w0, w1, w2, w3;
            asm volatile("s_nop 4\n\ts_load_dwordx2 %0, %6, 0x0\n\ts_load_dwordx2 %1, %7, 0x0\n\t"
                         "s_load_dword %2, %8, 0x0\n\ts_load_dword %3, %9, 0x0\n\t"
                         "s_load_dword %4, %10, 0x0\n\ts_load_dword %5, %11, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(kbase), "=&s"(kend), "=&s"(w0), "=&s"(w1), "=&s"(w2), "=&s"(w3)
                         : "s"(pf), "s"(pl), "s"(q0), "s"(q1), "s"(q2), "s"(q3) : "memory");
            // block b reaches (w & 255) rows below row 16 b and (w >> 8 & 255) rows beyond row 16 b + 15
            const int e = r0 + nr - 1;
            int lo_need = (int)(w0 & 255u);
            if (b0 < bl) lo_need = max(lo_need, (int)(w1 & 255u) - 16);
            int hi_need = ((bl << 4) + 15 + (int)((w3 >> 8) & 255u)) - e;
            if (b0 < bl) hi_need = max(hi_need, (bl << 4) - 1 + (int)((w2 >> 8) & 255u) - e);
            nlo = min(max(lo_need, 0), a.halo);
            nhi = min(max(hi_need, 0), a.halo);
        } else {
            asm volatile("s_nop 4\n\ts_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(kbase), "=&s"(kend) : "s"(pf), "s"(pl) : "memory");
        }
    }
    const int nk = (int)(kend - kbase);
    const int slo = max(0, r0 - nlo);
    const int shi = min((int)a.n, r0 + nr + nhi);
    const int wrows = shi - slo;

    // ---- (3) halo rows: the run below the tile, then the run above it, two vectors per thread
    const int lo_rows = r0 - slo, hi_rows = shi - (r0 + nr);
    const int lo_n = lo_rows * LV, hi_n = hi_rows * LV;
    int4 hv[2];
    int hdst[2];                                            // LDS vector index, -1: nothing
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + k * T;
        const bool is_lo = idx < lo_n;
        const int j = is_lo ? idx : idx - lo_n;            // vector inside its run
        bool ok = is_lo || j < hi_n;
        // (clamped: the tile's first vector stands in for a missing one)
        const char* grun = is_lo ? gtile - lo_rows * row_stride : (ok ? gtile + nr * row_stride : gtile);
        const int jc = ok ? j : 0;
        const int ldst = (is_lo ? (slo - wbase) : (a.halo + nr)) * LV + j;
        const int4* g;
        if (!CHUNKED) g = reinterpret_cast<const int4*>(grun) + jc;
        else {
            const int rr = jc >> lsh, cc = jc & (LV - 1);
            ok = ok && cc < V;
            g = reinterpret_cast<const int4*>(grun + rr * row_stride) + min(cc, V - 1);
        }
        hdst[k] = ok ? ldst : -1;
        hv[k] = *g;                                         // (halo rows are another tile's own rows: they may stay in the L2)
    }
    // ---- (4) neighbour indices (three per thread) and (5) row pointers (two per thread)
    const bool col_in_lds = nk <= a.col_cap && nk <= 3 * T;
    int jv[3];
    {
        // (clamped; a tile without list entries reads the row pointers instead: a.col may end exactly at kbase, or be NULL)
        const int* cbase = nk > 0 ? a.col + kbase : reinterpret_cast<const int*>(a.row_ptr + r0);
        const int last = col_in_lds ? max(nk - 1, 0) : 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) jv[k] = cbase[min(tid + k * T, last)];
    }
    int rp_mine[2];                                         // (low words: a tile's lists are < 2^31 entries)
    rp_mine[0] = reinterpret_cast<const int*>(a.row_ptr + r0 + min(tid, nr))[0];
    rp_mine[1] = reinterpret_cast<const int*>(a.row_ptr + r0 + min(tid + T, nr))[0];
    if (a.prio) __builtin_amdgcn_s_setprio(0);

    // ---- retire them in issue order: window vectors to LDS (their maximum on the way), offsets, pointers
    unsigned cmax = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (cok[k]) { win4[a.halo * LV + tid + k * T] = cv[k]; cmax = max(cmax, vmax(cv[k])); }
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (hdst[k] >= 0) { win4[hdst[k]] = hv[k]; cmax = max(cmax, vmax(hv[k])); }
    bool outside = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int kk = tid + k * T;
        if (col_in_lds && kk < nk) {
            const int j = jv[k];
            const bool in = (unsigned)(j - slo) < (unsigned)wrows;
            outside = outside || !in;
            colL[kk] = in ? (int)__umul24((unsigned)(j - wbase), (unsigned)rowb) : -1 - j;      // (window rows < 2^12, row bytes < 2^18: full-rate 24-bit multiply)
        }
    }
    if (tid <= nr) rpL[tid] = rp_mine[0] - (int)kbase;
    if (tid + T <= nr) rpL[tid + T] = rp_mine[1] - (int)kbase;
    {
        cmax = wave_max_dpp(cmax);
        const unsigned long long any_out = __ballot(outside);
        // one word per wave: its largest count (saturating at 2^24: beyond that no item is fast), bit 31 = one of its
        // staged neighbours is outside the window; sixteen words are read back, wave 0 clears the unused ones
        if (lane == 63) red[wave] = min(cmax, 0x1000000u) | (any_out != 0ull ? 0x80000000u : 0u);
        if (wave == 0 && lane >= NW && lane < 16) red[lane] = 0u;
    }
    __syncthreads();
    unsigned tile_cmax, tile_or;
    {
        const uint4* r4 = reinterpret_cast<const uint4*>(red);
        const uint4 x0 = r4[0], x1 = r4[1], x2 = r4[2], x3 = r4[3];
        tile_or = (x0.x | x0.y | x0.z | x0.w) | (x1.x | x1.y | x1.z | x1.w) | (x2.x | x2.y | x2.z | x2.w) | (x3.x | x3.y | x3.z | x3.w);
        const unsigned k = 0x7fffffffu;
        const unsigned m0 = max(max(x0.x & k, x0.y & k), max(x0.z & k, x0.w & k)), m1 = max(max(x1.x & k, x1.y & k), max(x1.z & k, x1.w & k));
        const unsigned m2 = max(max(x2.x & k, x2.y & k), max(x2.z & k, x2.w & k)), m3 = max(max(x3.x & k, x3.y & k), max(x3.z & k, x3.w & k));
        tile_cmax = max(max(m0, m1), max(m2, m3));
    }
    tile_cmax = (unsigned)__builtin_amdgcn_readfirstlane((int)tile_cmax);
    // an item is fast (32-bit sums, float32 quotient) iff (degree + 1) * (largest count of the tile) < 2^24
    const bool fast_tile = tile_cmax <= 0xFFFFFFu && col_in_lds;                         // block-uniform
    const bool all_in = __builtin_amdgcn_readfirstlane((int)tile_or) >= 0;               // block-uniform: no per-batch window check needed

    // ---- per (row, vector) item: gather neighbours from LDS, divide, store
    {
        const char* tileB = reinterpret_cast<const char*>(tileL);
        const int items = nr * V;
        int ri = tid / V, c = tid - ri * V;
        const int dr = T / V, dc = T - dr * V;
        // the outputs are addressed as (tile base: scalar) + (32-bit element offset inside the tile: the host keeps a tile
        // below 2^30 elements), and the offsets of successive items follow by additions: no 64-bit multiply-add, no
        // 32-bit multiplications (quarter-rate) per item -- the kernel runs at 3/4 of the VALU issue rate
        PsArgs b = a;
        {
            const int64_t tile_base = (int64_t)r0 * a.s + c0;
            if (WPS) b.ps = a.ps + tile_base;
            if (WEXCL) b.excl = a.excl + tile_base;
        }
        unsigned o32 = (unsigned)ri * (unsigned)a.s + (unsigned)(c * VEC);
        const unsigned d_o = (unsigned)dr * (unsigned)a.s + (unsigned)(dc * VEC), wrap_o = (unsigned)a.s - (unsigned)(V * VEC);
        int own_off = (a.halo + ri) * rowb;
        const int d_own = dr * rowb;
        // (degree + 1) * (largest count of the tile) < 2^24  <=>  degree + 1 <= deg_lim
        const unsigned deg_lim = tile_cmax ? 0xFFFFFFu / tile_cmax : 0xFFFFFFFFu;
        for (int it = tid; it < items; it += T) {
            const int k0 = rpL[ri], k1 = rpL[ri + 1];
            const int64_t o = (int64_t)(uint64_t)o32;
            bool done = false;
            const unsigned deg = (unsigned)(k1 - k0);
            if (fast_tile && deg < 254u && deg + 1u <= deg_lim)
                done = all_in ? ps_item_fast<VEC, WEXCL, WPS, false, Q3, 0>(b, tileB, colL, k0, k1, c * VEC * 4, own_off, o, zero_off)
                              : ps_item_fast<VEC, WEXCL, WPS, true, Q3, 0>(b, tileB, colL, k0, k1, c * VEC * 4, own_off, o, zero_off);
            if (!done)
                ps_item_slow<VEC, WEXCL, WPS, Q3, 0>(b, tileB, colL, col_in_lds, kbase, k0, k1, slo, wrows, ldw, c0, c * VEC,
                                                     own_off, o, wbase);
            c += dc; ri += dr; o32 += d_o; own_off += d_own;
            if (c >= V) { c -= V; ri += 1; o32 += wrap_o; own_off += rowb; }
        }
    }
}

__global__ void quantize3_kernel(float* __restrict__ x, int64_t n) {
    // f'{x:.3f}' -> float32: x*1000 is exact in float64 for a float32 x (24+10 bits), rint
    // is round-half-even like the decimal formatter, k/1000 -> f32 has no double-rounding
    // hazard (k/1000 is >= 2^-35 away from every f32 rounding boundary).
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4* x4 = reinterpret_cast<float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = x4[i];
        v.x = (float)(rint((double)v.x * 1000.0) / 1000.0);
        v.y = (float)(rint((double)v.y * 1000.0) / 1000.0);
        v.z = (float)(rint((double)v.z * 1000.0) / 1000.0);
        v.w = (float)(rint((double)v.w * 1000.0) / 1000.0);
        x4[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        x[i] = (float)(rint((double)x[i] * 1000.0) / 1000.0);
}

__global__ void mark_low_kernel(float* __restrict__ ps, const int64_t* __restrict__ idx, int64_t n_low,
                                int64_t n_elems) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_low; i += stride) {
        const int64_t k = idx[i];
        if (k >= 0 && k < n_elems) ps[k] = __builtin_nanf("");
    }
}

template <int VEC, bool WE, bool WP, bool Q3, int ABL>
int launch_ps_one(sdice_ctx* ctx, const PsArgs& a, int threads, size_t lds, dim3 grid) {
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ps_tile_kernel<VEC, WE, WP, Q3, ABL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "ps_tile_kernel", (ps_tile_kernel<VEC, WE, WP, Q3, ABL>), grid, dim3(threads), lds, a);
    return SDICE_OK;
}

template <bool CH, bool WE, bool WP, bool Q3>
int launch_ps_v3_one(sdice_ctx* ctx, const PsArgs& a, int threads, size_t lds, dim3 grid) {
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ps_tile_v3_kernel<CH, WE, WP, Q3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "ps_tile_v3_kernel", (ps_tile_v3_kernel<CH, WE, WP, Q3>), grid, dim3(threads), lds, a);
    return SDICE_OK;
}
template <bool CH>
int launch_ps_v3(sdice_ctx* ctx, const PsArgs& a, int threads, size_t lds, dim3 grid, bool wexcl, bool wps, bool q3) {
    if (wexcl && wps) return q3 ? launch_ps_v3_one<CH, true, true, true>(ctx, a, threads, lds, grid)
                                : launch_ps_v3_one<CH, true, true, false>(ctx, a, threads, lds, grid);
    if (wexcl) return launch_ps_v3_one<CH, true, false, false>(ctx, a, threads, lds, grid);
    return q3 ? launch_ps_v3_one<CH, false, true, true>(ctx, a, threads, lds, grid)
              : launch_ps_v3_one<CH, false, true, false>(ctx, a, threads, lds, grid);
}

template <int VEC>
int launch_ps(sdice_ctx* ctx, const PsArgs& a, int threads, size_t lds, dim3 grid, bool wexcl, bool wps, bool q3, int abl) {
    if (abl && VEC == 4 && !wexcl && wps && !q3) {        // timing experiments (one shape only)
        if (abl & 8) return launch_ps_one<4, false, true, false, 8>(ctx, a, threads, lds, grid);
        switch (abl & 7) {
            case 1: return launch_ps_one<4, false, true, false, 1>(ctx, a, threads, lds, grid);
            case 2: return launch_ps_one<4, false, true, false, 2>(ctx, a, threads, lds, grid);
            case 3: return launch_ps_one<4, false, true, false, 3>(ctx, a, threads, lds, grid);
            case 4: return launch_ps_one<4, false, true, false, 4>(ctx, a, threads, lds, grid);
            case 5: return launch_ps_one<4, false, true, false, 5>(ctx, a, threads, lds, grid);
            case 6: return launch_ps_one<4, false, true, false, 6>(ctx, a, threads, lds, grid);
            default: return launch_ps_one<4, false, true, false, 7>(ctx, a, threads, lds, grid);
        }
    }
    if (wexcl && wps) return q3 ? launch_ps_one<VEC, true, true, true, 0>(ctx, a, threads, lds, grid)
                                : launch_ps_one<VEC, true, true, false, 0>(ctx, a, threads, lds, grid);
    if (wexcl) return launch_ps_one<VEC, true, false, false, 0>(ctx, a, threads, lds, grid);
    return q3 ? launch_ps_one<VEC, false, true, true, 0>(ctx, a, threads, lds, grid)
              : launch_ps_one<VEC, false, true, false, 0>(ctx, a, threads, lds, grid);
}

}  // namespace

extern "C" int sdice_ps_dev(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* d_counts,
                            const int64_t* d_row_ptr, const int32_t* d_col, int64_t* d_excl, float* d_ps) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    SD_ARG(n < (int64_t)1 << 31, "n must be < 2^31 (int32 row indices)");
    SD_ARG(d_excl || d_ps, "both outputs are NULL");
    if (n == 0 || s == 0) return SDICE_OK;
    SD_ARG(d_counts && d_row_ptr, "NULL input");
    SD_HIP(hipSetDevice(ctx->device));

    const bool aligned = ((uintptr_t)d_counts % 16 == 0) && (!d_ps || (uintptr_t)d_ps % 16 == 0) &&
                         (!d_excl || (uintptr_t)d_excl % 16 == 0);
    const int vec = (s % 4 == 0 && aligned) ? 4 : 1;
    const int abl = (int)ctx->param("ps.ablate", 0);
    int cw = s;
    const int64_t chunk_param = ctx->param("ps.chunk_cols", 0);
    if (chunk_param > 0) cw = (int)chunk_param;
    else if (s > 256) cw = 128;
    if (cw > s) cw = s;
    if (vec == 4) cw = (cw / 4) * 4;
    if (cw < vec) cw = vec;
    // kernel: 0 = second-generation register-staged kernel (ps_tile_v3_kernel, the default), 2 = first-generation kernel
    // (ps_tile_kernel; also serves tables whose rows are not 16-byte vectors, column chunks whose vectors do not divide
    // 64, tile shapes beyond the register staging of the newer kernel, and the timing experiments)
    const bool pow2chunk = cw == s || (cw / 4 <= 64 && 64 % (cw / 4) == 0);
    int kern = ctx->param("ps.gen1", 0) != 0 ? 2 : 0;
    if (vec != 4 || abl != 0 || !pow2chunk) kern = 2;
    if ((int64_t)s * 2048 >= ((int64_t)1 << 30)) kern = 2;       // (the newer kernel addresses a tile's outputs by 32-bit offsets)
    int64_t lds = ctx->param("ps.lds_bytes", 80 * 1024);   // two workgroups per CU (160 KiB LDS)
    if (lds > 160 * 1024) lds = 160 * 1024;
    if (lds < 8 * 1024) lds = 8 * 1024;
    int threads = (int)ctx->param("ps.threads", 1024);
    threads = (threads / 64) * 64;
    if (threads < 64) threads = 64;
    if (threads > 1024) threads = 1024;
    const int64_t LV = vec == 4 ? cw / 4 : cw;              // 16-byte vectors of a window row

    int64_t R = 0, H = 0;
    bool have_reach = false;
    // tile geometry; a shape that the register staging of the second-generation kernel cannot hold (it stages FOUR
    // vectors of the tile's own rows, TWO of each halo run and the low words of TWO row pointers per thread) falls back
    // to the first-generation kernel, whose loops take any shape
    for (;;) {
        const bool newk = kern == 0;
        // LDS budget (ints): (R + 2H + 1) * cw window + 16 R staged neighbour offsets + (R + 1) row pointers + scratch
        const int64_t L = lds / 4 - (newk ? 64 : 16);      // (the newer kernel keeps 16 per-wave words behind the row pointers)
        have_reach = newk && d_col != nullptr && d_col == ctx->d_col && ctx->reach_n == n && ctx->d_reach != nullptr &&
                     ctx->param("ps.use_reach", 1) != 0;
        H = ctx->param("ps.halo_rows", -1);
        const bool h_auto = H < 0;
        if (h_auto) {
            // rows further than the halo are still summed exactly (global-memory path).  With reach bytes H is the
            // CAPACITY of the window on each side (a tile stages what its lists reach, 8 + 8 rows on average on
            // gene-shaped data, 22 at most); without them H rows are staged on each side: 16 cover >99.9 % of the
            // neighbours of gene-shaped data, less if the clustering saw a smaller reach
            H = 16;
            if (!have_reach && d_col != nullptr && d_col == ctx->d_col && ctx->cluster_reach > 0 && ctx->cluster_reach < H)
                H = ctx->cluster_reach;
        }
        // (a software-pipelined persistent variant -- one workgroup per CU, two LDS buffers, next tile's
        //  loads in flight during the gather -- was built and measured 30 % slower: the kernel is VALU-issue
        //  bound, and halving the resident waves costs more than hiding the load latency gains)
        // caps of the register staging: rows <= 4 T / LV (own rows), rows <= 2 T - 1 (row pointers: nr + 1 of them),
        // halo <= T / LV (the two runs together take two vectors per thread)
        const int64_t r_cap = newk ? std::min<int64_t>(4 * (int64_t)threads / LV, 2 * (int64_t)threads - 1) : 2 * (int64_t)threads;
        const int64_t h_cap = newk ? (int64_t)threads / LV : (int64_t)1 << 20;
        if (H > h_cap) H = h_cap;
        const int64_t per_row = 17;      // 16 staged neighbour offsets + the row pointer
        const int64_t r_param = ctx->param("ps.tile_rows", 0);
        auto fit_rows = [&](int64_t h) {
            int64_t r = r_param > 0 ? r_param : (L - (2 * h + 1) * cw) / (cw + per_row);
            return std::min(r, r_cap);
        };
        R = fit_rows(H);
        while (H > 0 && (R < std::min<int64_t>(8, r_cap) || (R + 2 * H + 1) * cw + per_row * R > L)) {
            // window does not fit: shrink the halo first (misses fall back to global loads), then the tile
            H = H / 2;
            R = fit_rows(H);
        }
        if (R < 1) R = 1;
        while (R > 1 && (R + 2 * H + 1) * cw + per_row * R > L) R -= 1;
        if (newk && (r_cap < 1 || R > r_cap || (R + 2 * H + 1) * cw + per_row * R > L)) { kern = 2; continue; }
        SD_ARG((R + 2 * H + 1) * cw + per_row * R <= L, "row chunk does not fit LDS; lower ps.chunk_cols");
        // (trimming R so that R * V is a multiple of the block size was measured: slower -- the per-tile
        //  fixed cost outweighs the idle lanes of the last pass over the items)
        if (have_reach && R >= 16 && r_param <= 0) R &= ~(int64_t)15;   // tiles are whole 16-row reach blocks
        have_reach = have_reach && R % 16 == 0;
        if (have_reach && h_auto) {
            // the rows lost to the rounding are window capacity: a tile stages only what its reach words ask for, so a
            // larger capacity costs nothing and keeps the rare far-reaching tile off the global-memory path
            // (2 M x 500: 96-row tiles either way, capacity 16 -> 24, 1.686 -> 1.650 ms)
            while (H < 32 && H + 1 <= h_cap && (R + 2 * (H + 1) + 1) * cw + per_row * R <= L) H += 1;
        }
        break;
    }
    const bool use_reach = have_reach;
    if (R > n) { R = n; }

    PsArgs a;
    a.counts = d_counts; a.row_ptr = d_row_ptr; a.col = d_col; a.excl = d_excl; a.ps = d_ps;
    a.n = n; a.s = s;
    a.tile_rows = (int)R;
    a.halo = (int)H;
    a.chunk_cols = cw;
    a.col_cap = (int)(16 * R);
    a.n_tiles = (int)sd_ceil_div(n, R);
    const int n_chunks = (int)sd_ceil_div(s, cw);
    a.n_chunks = n_chunks;
    a.prio = ctx->param("ps.prio", 1) != 0;
    // window loads that do not linger in the L2: -4 % at 1 M x 100 (0.179 -> 0.172 ms in one process); with column chunks
    // the lines shared by two chunks and the halo rows want the L2 (+1.7 % at 2 M x 500): unchunked tables only
    a.nt = ctx->param("ps.nt_loads", 1) != 0 && n_chunks == 1;
    a.reach = use_reach ? ctx->d_reach : nullptr;
    a.lv_shift = 0;                                        // column chunks: log2(row segments per 1 KiB piece)
    for (int b = 0; b <= 6; ++b) if ((cw / 4) == (64 >> b)) a.lv_shift = b;
    int gx = a.n_tiles;
    a.tiles_per_xcd = 0;
    const bool q3 = ctx->param("ps.quantize3", 0) != 0;
    if (ctx->param("ps.xcd_remap", 1) && a.n_tiles >= 64) {
        a.tiles_per_xcd = (int)sd_ceil_div(a.n_tiles, 8);
        gx = a.tiles_per_xcd * 8;
    }
    const size_t lds_bytes = (size_t)lds;
    SD_ARG((int64_t)gx * n_chunks < (int64_t)1 << 31, "grid too large");
    dim3 grid((unsigned)((int64_t)gx * n_chunks));
    if (kern == 0) return n_chunks == 1 && cw == s ? launch_ps_v3<false>(ctx, a, threads, lds_bytes, grid, d_excl != nullptr, d_ps != nullptr, q3)
                                                   : launch_ps_v3<true>(ctx, a, threads, lds_bytes, grid, d_excl != nullptr, d_ps != nullptr, q3);
    if (vec == 4) return launch_ps<4>(ctx, a, threads, lds_bytes, grid, d_excl != nullptr, d_ps != nullptr, q3, abl);
    return launch_ps<1>(ctx, a, threads, lds_bytes, grid, d_excl != nullptr, d_ps != nullptr, q3, abl);
}

extern "C" int sdice_ps(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* counts,
                        const int64_t* row_ptr, const int32_t* col, int64_t* excl, float* ps) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    SD_ARG(excl || ps, "both outputs are NULL");
    if (n == 0 || s == 0) return SDICE_OK;
    SD_ARG(counts && row_ptr, "NULL input");
    const int64_t nnz = row_ptr[n];
    SD_ARG(row_ptr[0] == 0 && nnz >= 0, "row_ptr must start at 0 and be non-decreasing");
    SD_ARG(nnz == 0 || col, "col is NULL");
    for (int64_t i = 0; i < n; ++i) SD_ARG(row_ptr[i + 1] >= row_ptr[i], "row_ptr must be non-decreasing");
    for (int64_t k = 0; k < nnz; ++k) SD_ARG(col[k] >= 0 && col[k] < n, "col index out of range");
    SD_HIP(hipSetDevice(ctx->device));
    const size_t cells = (size_t)n * (size_t)s;
    int32_t *d_counts = nullptr, *d_col = nullptr;
    int64_t *d_rp = nullptr, *d_excl = nullptr;
    float* d_ps = nullptr;
    int rc = SDICE_OK;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        if (d_counts) (void)hipFree(d_counts);
        if (d_col) (void)hipFree(d_col);
        if (d_rp) (void)hipFree(d_rp);
        if (d_excl) (void)hipFree(d_excl);
        if (d_ps) (void)hipFree(d_ps);
    };
#define SD_STEP(expr)                                                    \
    do {                                                                 \
        hipError_t _e = (expr);                                          \
        if (_e != hipSuccess) {                                          \
            sdice_set_error("sdice_ps: %s -> %s", #expr, hipGetErrorString(_e)); \
            cleanup();                                                   \
            return SDICE_ERR_HIP;                                        \
        }                                                                \
    } while (0)
    SD_STEP(hipMalloc((void**)&d_counts, cells * 4));
    SD_STEP(hipMalloc((void**)&d_rp, (size_t)(n + 1) * 8));
    SD_STEP(hipMalloc((void**)&d_col, nnz > 0 ? (size_t)nnz * 4 : 256));
    if (excl) SD_STEP(hipMalloc((void**)&d_excl, cells * 8));
    if (ps) SD_STEP(hipMalloc((void**)&d_ps, cells * 4));
    SD_STEP(hipMemcpyAsync(d_counts, counts, cells * 4, hipMemcpyHostToDevice, ctx->stream));
    SD_STEP(hipMemcpyAsync(d_rp, row_ptr, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nnz > 0) SD_STEP(hipMemcpyAsync(d_col, col, (size_t)nnz * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = sdice_ps_dev(ctx, n, s, d_counts, d_rp, d_col, d_excl, d_ps);
    if (rc == SDICE_OK) {
        if (excl) SD_STEP(hipMemcpyAsync(excl, d_excl, cells * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (ps) SD_STEP(hipMemcpyAsync(ps, d_ps, cells * 4, hipMemcpyDeviceToHost, ctx->stream));
        SD_STEP(hipStreamSynchronize(ctx->stream));
    }
#undef SD_STEP
    cleanup();
    return rc;
}

extern "C" int sdice_quantize3_dev(sdice_ctx* ctx, int64_t n_elems, float* d_ps_inout) {
    SD_ARG(ctx && n_elems >= 0, "bad arguments");
    if (n_elems == 0) return SDICE_OK;
    SD_ARG(d_ps_inout && (uintptr_t)d_ps_inout % 16 == 0, "pointer must be 16-byte aligned");
    int64_t blocks = sd_ceil_div(sd_ceil_div(n_elems, 4), 256);
    if (blocks > 2048) blocks = 2048;
    SD_LAUNCH(ctx, "quantize3_kernel", quantize3_kernel, dim3((unsigned)blocks), dim3(256), 0, d_ps_inout, n_elems);
    return SDICE_OK;
}

extern "C" int sdice_quantize3(sdice_ctx* ctx, int64_t n_elems, float* ps_inout) {
    SD_ARG(ctx && n_elems >= 0, "bad arguments");
    if (n_elems == 0) return SDICE_OK;
    SD_ARG(ps_inout, "NULL pointer");
    float* d = nullptr;
    SD_TRY(sdice_dmalloc(ctx, n_elems * 4, (void**)&d));
    int rc = sdice_h2d(ctx, d, ps_inout, n_elems * 4);
    if (rc == SDICE_OK) rc = sdice_quantize3_dev(ctx, n_elems, d);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, ps_inout, d, n_elems * 4);
    sdice_dfree(ctx, d);
    return rc;
}

extern "C" int sdice_mark_low_dev(sdice_ctx* ctx, int64_t n_elems, float* d_ps, const int64_t* d_low_idx,
                                  int64_t n_low) {
    SD_ARG(ctx && n_elems >= 0 && n_low >= 0, "bad arguments");
    if (n_low == 0) return SDICE_OK;
    SD_ARG(d_ps && d_low_idx, "NULL pointer");
    int64_t blocks = sd_ceil_div(n_low, 256);
    if (blocks > 2048) blocks = 2048;
    SD_LAUNCH(ctx, "mark_low_kernel", mark_low_kernel, dim3((unsigned)blocks), dim3(256), 0, d_ps, d_low_idx, n_low,
              n_elems);
    return SDICE_OK;
}

extern "C" int sdice_mark_low(sdice_ctx* ctx, int64_t n_elems, float* ps, const int64_t* low_idx, int64_t n_low) {
    SD_ARG(ctx && n_elems >= 0 && n_low >= 0, "bad arguments");
    if (n_low == 0) return SDICE_OK;
    SD_ARG(ps && low_idx, "NULL pointer");
    for (int64_t i = 0; i < n_low; ++i) SD_ARG(low_idx[i] >= 0 && low_idx[i] < n_elems, "low index out of range");
    float* d = nullptr;
    int64_t* di = nullptr;
    SD_TRY(sdice_dmalloc(ctx, n_elems * 4, (void**)&d));
    int rc = sdice_dmalloc(ctx, n_low * 8, (void**)&di);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, d, ps, n_elems * 4);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, di, low_idx, n_low * 8);
    if (rc == SDICE_OK) rc = sdice_mark_low_dev(ctx, n_elems, d, di, n_low);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, ps, d, n_elems * 4);
    sdice_dfree(ctx, d);
    sdice_dfree(ctx, di);
    return rc;
}
