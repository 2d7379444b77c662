// PS of a float64 count table: the arithmetic of counts_to_ps.writePsValues (counts_to_ps.py:58-70).
//
//   exclusion = counts[junction].copy()
//   for overlap in clusters[junction]: exclusion += counts[overlap]      (list order, float64)
//   ps = counts[junction] / exclusion                                    (0/0 -> nan, x/0 -> inf)
//
// The reference parses the table with dtype=float (:50), so fractional / normalised counts are legal
// input and the additions round: they are done here in the same order, one IEEE add at a time, and the
// quotient is one IEEE division (the library is built with -fno-fast-math -ffp-contract=off).  For
// integer-valued tables this is the same number as ps_tile_kernel's integer sums followed by a float64
// division; the tile kernel stays the path of quant (float32 result) and of pairwise (int64 sums).
//
// One wave per output row, lanes across the columns: every load of a neighbour row is a coalesced run
// of 64 doubles, the list bounds and entries are wave-uniform (scalar loads).  HBM/L2-bound gather:
// 8 B in + 8 B out per entry plus the neighbour rows, which are L2 hits for sorted tables.
#include "common.h"

namespace {

constexpr int PF_WAVES = 4;

// EXCL: store the sum of the listed rows alone (additions in list order, starting from the first row) instead of
// own / (own + sum): `pairwise` on a fractional count table (np.sum(counts[rows], axis=0), pairwise_fisher.py:158-160,
// adds the rows in table order; fisher_exact then truncates the float sums to int64)
template <bool EXCL>
__global__ void __launch_bounds__(PF_WAVES * 64) ps_f64_kernel(int64_t n_out, int s, const double* __restrict__ counts,
                                                                const int64_t* __restrict__ row_ptr,
                                                                const int32_t* __restrict__ col, double* __restrict__ ps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * PF_WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t stride = (int64_t)gridDim.x * PF_WAVES;
    for (int64_t row = wave0; row < n_out; row += stride) {
        const int64_t k0 = row_ptr[row], k1 = row_ptr[row + 1];
        const double* own_row = counts + row * s;
        for (int j = lane; j < s; j += 64) {
            const double own = EXCL ? 0.0 : own_row[j];
            double acc = own;
            int64_t k = k0;
            if (EXCL && k < k1) { acc = counts[(int64_t)col[k] * s + j]; ++k; }
            for (; k + 4 <= k1; k += 4) {          // four neighbour rows in flight, added in list order
                const double a0 = counts[(int64_t)col[k] * s + j];
                const double a1 = counts[(int64_t)col[k + 1] * s + j];
                const double a2 = counts[(int64_t)col[k + 2] * s + j];
                const double a3 = counts[(int64_t)col[k + 3] * s + j];
                acc += a0; acc += a1; acc += a2; acc += a3;
            }
            for (; k < k1; ++k) acc += counts[(int64_t)col[k] * s + j];
            ps[row * s + j] = EXCL ? acc : own / acc;
        }
    }
}

}  // namespace

static int ps_f64_dev(sdice_ctx* ctx, bool excl, int64_t n_out, int64_t n_rows, int32_t s, const double* d_counts,
                      const int64_t* d_row_ptr, const int32_t* d_col, double* d_ps) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n_out >= 0 && n_rows >= n_out && s >= 0, "need 0 <= n_out <= n_rows, s >= 0");
    if (n_out == 0 || s == 0) return SDICE_OK;
    SD_ARG(d_counts && d_row_ptr && d_ps, "NULL pointer");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t blocks = sd_ceil_div(n_out, PF_WAVES);
    const unsigned grid = (unsigned)(blocks < 256 * 64 ? blocks : 256 * 64);
    if (excl) SD_LAUNCH(ctx, "ps_f64_kernel", (ps_f64_kernel<true>), dim3(grid), dim3(PF_WAVES * 64), 0, n_out, (int)s, d_counts,
                        d_row_ptr, d_col, d_ps);
    else SD_LAUNCH(ctx, "ps_f64_kernel", (ps_f64_kernel<false>), dim3(grid), dim3(PF_WAVES * 64), 0, n_out, (int)s, d_counts,
                   d_row_ptr, d_col, d_ps);
    return SDICE_OK;
}

extern "C" int sdice_ps_f64_dev(sdice_ctx* ctx, int64_t n_out, int64_t n_rows, int32_t s, const double* d_counts,
                                const int64_t* d_row_ptr, const int32_t* d_col, double* d_ps) {
    return ps_f64_dev(ctx, false, n_out, n_rows, s, d_counts, d_row_ptr, d_col, d_ps);
}

extern "C" int sdice_excl_f64_dev(sdice_ctx* ctx, int64_t n_out, int64_t n_rows, int32_t s, const double* d_counts,
                                  const int64_t* d_row_ptr, const int32_t* d_col, double* d_excl) {
    return ps_f64_dev(ctx, true, n_out, n_rows, s, d_counts, d_row_ptr, d_col, d_excl);
}

static int ps_f64_host(sdice_ctx* ctx, bool excl, int64_t n_out, int64_t n_rows, int32_t s, const double* counts,
                       const int64_t* row_ptr, const int32_t* col, double* ps) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n_out >= 0 && n_rows >= n_out && s >= 0, "need 0 <= n_out <= n_rows, s >= 0");
    if (n_out == 0 || s == 0) return SDICE_OK;
    SD_ARG(counts && row_ptr && ps, "NULL pointer");
    const int64_t nnz = row_ptr[n_out];
    SD_ARG(row_ptr[0] == 0 && nnz >= 0, "row_ptr must start at 0 and be non-decreasing");
    SD_ARG(nnz == 0 || col, "col is NULL");
    for (int64_t i = 0; i < n_out; ++i) SD_ARG(row_ptr[i + 1] >= row_ptr[i], "row_ptr must be non-decreasing");
    for (int64_t k = 0; k < nnz; ++k) SD_ARG(col[k] >= 0 && col[k] < n_rows, "col index out of range");
    SD_HIP(hipSetDevice(ctx->device));
    const size_t in_bytes = (size_t)n_rows * (size_t)s * 8, out_bytes = (size_t)n_out * (size_t)s * 8;
    double *d_counts = nullptr, *d_ps = nullptr;
    int64_t* d_rp = nullptr;
    int32_t* d_col = nullptr;
    int rc = SDICE_OK;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        if (d_counts) (void)hipFree(d_counts);
        if (d_ps) (void)hipFree(d_ps);
        if (d_rp) (void)hipFree(d_rp);
        if (d_col) (void)hipFree(d_col);
    };
#define SD_STEP(expr)                                                        \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) {                                              \
            sdice_set_error("sdice_ps_f64: %s -> %s", #expr, hipGetErrorString(_e)); \
            cleanup();                                                       \
            return SDICE_ERR_HIP;                                            \
        }                                                                    \
    } while (0)
    SD_STEP(hipMalloc((void**)&d_counts, in_bytes));
    SD_STEP(hipMalloc((void**)&d_ps, out_bytes));
    SD_STEP(hipMalloc((void**)&d_rp, (size_t)(n_out + 1) * 8));
    SD_STEP(hipMalloc((void**)&d_col, nnz > 0 ? (size_t)nnz * 4 : 256));
    SD_STEP(hipMemcpyAsync(d_counts, counts, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    SD_STEP(hipMemcpyAsync(d_rp, row_ptr, (size_t)(n_out + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nnz > 0) SD_STEP(hipMemcpyAsync(d_col, col, (size_t)nnz * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = ps_f64_dev(ctx, excl, n_out, n_rows, s, d_counts, d_rp, d_col, d_ps);
    if (rc == SDICE_OK) {
        SD_STEP(hipMemcpyAsync(ps, d_ps, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        SD_STEP(hipStreamSynchronize(ctx->stream));
    }
#undef SD_STEP
    cleanup();
    return rc;
}

extern "C" int sdice_ps_f64(sdice_ctx* ctx, int64_t n_out, int64_t n_rows, int32_t s, const double* counts,
                            const int64_t* row_ptr, const int32_t* col, double* ps) {
    return ps_f64_host(ctx, false, n_out, n_rows, s, counts, row_ptr, col, ps);
}

extern "C" int sdice_excl_f64(sdice_ctx* ctx, int64_t n_out, int64_t n_rows, int32_t s, const double* counts,
                              const int64_t* row_ptr, const int32_t* col, double* excl) {
    return ps_f64_host(ctx, true, n_out, n_rows, s, counts, row_ptr, col, excl);
}
