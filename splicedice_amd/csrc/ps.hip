// K3: exclusion-sum + PS over the junction x sample matrix, K4: '.3f' quantise,
// and the --lowCoverageNan scatter.
//
// Replaces SPLICEDICE.calculatePsi (SPLICEDICE.py:297-310): for every junction row the
// reference adds up the count rows of all overlapping junctions (float64 accumulation of
// integer-valued float32 counts -> exact integer sums) and stores
// float32(float64(incl) / float64(incl + excl)).
//
// Design (HBM-bound, 8 algorithmic bytes per PS entry: 4 B count in + 4 B PS out):
//   * one workgroup owns a tile of consecutive output rows x a chunk of columns;
//   * it first walks the tile's CSR segment (staging the neighbour indices in LDS) to
//     find the row window [wlo, whi) that holds every neighbour -- overlap clusters are
//     gene sized, so the window is the tile plus a small halo;
//   * the window of the count matrix is copied once, coalesced 16 B per lane, into LDS;
//   * every (row, 4-column) item then gathers its neighbour rows from LDS with
//     ds_read_b128, accumulates in 64-bit integers, divides in float64 and stores a
//     float4 -- each count is read from HBM once per tile (+halo) and each PS value is
//     written once;
//   * neighbours that fall outside the staged window (arbitrary user CSR, or a window
//     larger than LDS) are read from global memory, so any valid CSR gives exact results.
#include "common.h"

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
    int chunk_cols;  // columns per chunk == LDS row stride (multiple of VEC)
    int win_cap;     // LDS window capacity in rows
    int col_cap;     // LDS capacity for staged neighbour indices (ints)
    int n_tiles;
    int tiles_per_xcd;  // 0 = no remap
};

template <int VEC> struct Vt;
template <> struct Vt<4> { typedef int4 I; typedef float4 F; };
template <> struct Vt<1> { typedef int I; typedef float F; };

__device__ __forceinline__ float ps_value(unsigned incl, unsigned long long excl) {
    // float32(float64(incl) / float64(incl + excl)), SPLICEDICE.py:306; 0/0 -> NaN
    const double a = (double)incl;
    const double t = a + (double)excl;
    return (float)(a / t);
}

__device__ __forceinline__ void acc_add(unsigned long long (&acc)[4], const int4& v) {
    acc[0] += (unsigned)v.x; acc[1] += (unsigned)v.y; acc[2] += (unsigned)v.z; acc[3] += (unsigned)v.w;
}
__device__ __forceinline__ void acc_add(unsigned long long (&acc)[1], const int& v) { acc[0] += (unsigned)v; }

__device__ __forceinline__ void store_ps(float* p, const int4& own, const unsigned long long (&acc)[4]) {
    float4 o;
    o.x = ps_value((unsigned)own.x, acc[0]);
    o.y = ps_value((unsigned)own.y, acc[1]);
    o.z = ps_value((unsigned)own.z, acc[2]);
    o.w = ps_value((unsigned)own.w, acc[3]);
    *reinterpret_cast<float4*>(p) = o;
}
__device__ __forceinline__ void store_ps(float* p, const int& own, const unsigned long long (&acc)[1]) {
    *p = ps_value((unsigned)own, acc[0]);
}
__device__ __forceinline__ void store_excl(int64_t* p, const unsigned long long (&acc)[4]) {
    longlong2 a, b;
    a.x = (long long)acc[0]; a.y = (long long)acc[1]; b.x = (long long)acc[2]; b.y = (long long)acc[3];
    reinterpret_cast<longlong2*>(p)[0] = a;
    reinterpret_cast<longlong2*>(p)[1] = b;
}
__device__ __forceinline__ void store_excl(int64_t* p, const unsigned long long (&acc)[1]) { *p = (int64_t)acc[0]; }

template <int VEC, bool WEXCL, bool WPS>
__global__ void __launch_bounds__(1024) ps_tile_kernel(PsArgs a) {
    typedef typename Vt<VEC>::I VI;
    extern __shared__ int4 smem4[];
    int* tileL = reinterpret_cast<int*>(smem4);
    int* colL = tileL + (size_t)a.win_cap * a.chunk_cols;
    int* rpL = colL + a.col_cap;
    int* red = rpL + a.tile_rows + 1;  // [0..15] per-wave min, [16..31] per-wave max

    const int tid = threadIdx.x;
    const int T = blockDim.x;
    int tile = blockIdx.x;
    if (a.tiles_per_xcd) {
        // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of
        // tiles so that neighbouring tiles (which share halo rows) share one L2.
        tile = (blockIdx.x & 7) * a.tiles_per_xcd + (blockIdx.x >> 3);
    }
    if (tile >= a.n_tiles) return;
    const int c0 = blockIdx.y * a.chunk_cols;
    const int cwc = min(a.chunk_cols, a.s - c0);
    const int V = cwc / VEC;
    const int ldw = a.chunk_cols;
    const int64_t r0 = (int64_t)tile * a.tile_rows;
    const int nr = (int)min((int64_t)a.tile_rows, a.n - r0);
    const int64_t kbase = a.row_ptr[r0];
    const int64_t nk = a.row_ptr[r0 + nr] - kbase;
    const bool col_in_lds = nk <= (int64_t)a.col_cap;

    // ---- phase A: CSR segment -> LDS, neighbour window by min/max reduction
    for (int i = tid; i <= nr; i += T) rpL[i] = (int)(a.row_ptr[r0 + i] - kbase);
    int lmin = (int)r0, lmax = (int)r0 + nr - 1;
    for (int64_t k = tid; k < nk; k += T) {
        const int j = a.col[kbase + k];
        if (col_in_lds) colL[k] = j;
        lmin = min(lmin, j);
        lmax = max(lmax, j);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lmin = min(lmin, __shfl_xor(lmin, o));
        lmax = max(lmax, __shfl_xor(lmax, o));
    }
    if ((tid & 63) == 0) { red[tid >> 6] = lmin; red[16 + (tid >> 6)] = lmax; }
    __syncthreads();
    const int nw = (T + 63) >> 6;
    int wlo = red[0], whi = red[16];
    for (int w = 1; w < nw; ++w) { wlo = min(wlo, red[w]); whi = max(whi, red[16 + w]); }
    whi += 1;
    int slo = wlo, shi = whi;
    if (whi - wlo > a.win_cap) {
        const int extra = a.win_cap - nr;
        slo = max(wlo, (int)r0 - extra / 2);
        shi = slo + a.win_cap;
        if (shi > whi) { shi = whi; slo = max(wlo, shi - a.win_cap); }
    }
    const int wrows = shi - slo;

    // ---- phase B: stage rows [slo, shi) x cols [c0, c0+cwc) into LDS (coalesced, 16 B/lane)
    {
        const int total = wrows * V;
        if (cwc == a.s) {
            const VI* g = reinterpret_cast<const VI*>(a.counts + (int64_t)slo * a.s);
            VI* l = reinterpret_cast<VI*>(tileL);
            for (int i = tid; i < total; i += T) l[i] = g[i];
        } else {
            int rr = tid / V, cc = tid - rr * V;
            const int dr = T / V, dc = T - dr * V;
            for (int i = tid; i < total; i += T) {
                *reinterpret_cast<VI*>(tileL + rr * ldw + cc * VEC) =
                    *reinterpret_cast<const VI*>(a.counts + (int64_t)(slo + rr) * a.s + c0 + cc * VEC);
                cc += dc; rr += dr;
                if (cc >= V) { cc -= V; rr += 1; }
            }
        }
    }
    __syncthreads();

    // ---- phase C: per (row, vector) item: gather neighbours, divide, store
    {
        const int items = nr * V;
        int ri = tid / V, c = tid - ri * V;
        const int dr = T / V, dc = T - dr * V;
        const int own_base = (int)(r0 - slo);
        for (int it = tid; it < items; it += T) {
            unsigned long long acc[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = 0ull;
            const int k0 = rpL[ri], k1 = rpL[ri + 1];
            for (int k = k0; k < k1; ++k) {
                const int j = col_in_lds ? colL[k] : a.col[kbase + k];
                const unsigned rel = (unsigned)(j - slo);
                VI v;
                if (rel < (unsigned)wrows)
                    v = *reinterpret_cast<const VI*>(tileL + rel * ldw + c * VEC);
                else
                    v = *reinterpret_cast<const VI*>(a.counts + (int64_t)j * a.s + c0 + c * VEC);
                acc_add(acc, v);
            }
            const VI own = *reinterpret_cast<const VI*>(tileL + (own_base + ri) * ldw + c * VEC);
            const int64_t o = (r0 + ri) * a.s + c0 + c * VEC;
            if (WPS) store_ps(a.ps + o, own, acc);
            if (WEXCL) store_excl(a.excl + o, acc);
            c += dc; ri += dr;
            if (c >= V) { c -= V; ri += 1; }
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

template <int VEC>
int launch_ps(sdice_ctx* ctx, const PsArgs& a, int threads, size_t lds, dim3 grid, bool wexcl, bool wps) {
#define SD_PS_CASE(WE, WP)                                                                         \
    do {                                                                                           \
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ps_tile_kernel<VEC, WE, WP>),      \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
        SD_LAUNCH(ctx, "ps_tile_kernel", (ps_tile_kernel<VEC, WE, WP>), grid, dim3(threads), lds, a); \
    } while (0)
    if (wexcl && wps) SD_PS_CASE(true, true);
    else if (wexcl) SD_PS_CASE(true, false);
    else SD_PS_CASE(false, true);
#undef SD_PS_CASE
    return SDICE_OK;
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
    int64_t lds = ctx->param("ps.lds_bytes", 64 * 1024);
    if (lds > 160 * 1024) lds = 160 * 1024;
    if (lds < 8 * 1024) lds = 8 * 1024;
    int threads = (int)ctx->param("ps.threads", 512);
    threads = (threads / 64) * 64;
    if (threads < 64) threads = 64;
    if (threads > 1024) threads = 1024;

    int cw = s;
    const int64_t chunk_param = ctx->param("ps.chunk_cols", 0);
    if (chunk_param > 0) cw = (int)chunk_param;
    else if (s > 256) cw = 128;
    if (cw > s) cw = s;
    if (vec == 4) cw = (cw / 4) * 4;
    if (cw < vec) cw = vec;

    const int64_t L = lds / 4 - 64;  // ints available (64 reserved for the reduction scratch)
    int64_t R = ctx->param("ps.tile_rows", 0);
    if (R <= 0) R = (int64_t)(L / (1.5 * cw + 17.0));
    if (R < 1) R = 1;
    // shrink the tile until a window of at least R rows fits
    while (R > 1 && (L - 17 * R - 1) / cw < R) R = R * 3 / 4;
    int64_t win = (L - 17 * R - 1) / cw;
    SD_ARG(win >= R && win >= 1, "row chunk does not fit LDS; lower ps.chunk_cols");
    if (R > n) { R = n; }

    PsArgs a;
    a.counts = d_counts; a.row_ptr = d_row_ptr; a.col = d_col; a.excl = d_excl; a.ps = d_ps;
    a.n = n; a.s = s;
    a.tile_rows = (int)R;
    a.chunk_cols = cw;
    a.win_cap = (int)win;
    a.col_cap = (int)(16 * R);
    a.n_tiles = (int)sd_ceil_div(n, R);
    const int n_chunks = (int)sd_ceil_div(s, cw);
    SD_ARG(n_chunks <= 65535, "too many column chunks");
    int gx = a.n_tiles;
    a.tiles_per_xcd = 0;
    if (ctx->param("ps.xcd_remap", 1) && a.n_tiles >= 64) {
        a.tiles_per_xcd = (int)sd_ceil_div(a.n_tiles, 8);
        gx = a.tiles_per_xcd * 8;
    }
    const size_t lds_bytes = (size_t)lds;
    dim3 grid(gx, n_chunks);
    if (vec == 4) return launch_ps<4>(ctx, a, threads, lds_bytes, grid, d_excl != nullptr, d_ps != nullptr);
    return launch_ps<1>(ctx, a, threads, lds_bytes, grid, d_excl != nullptr, d_ps != nullptr);
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
