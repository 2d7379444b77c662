// K8: `similarity` scoring (reference similarity.py:25-47).
//
// For every event row that the comparison table marks significant (p <= 0.05, delta != 0) and
// every sample column with a non-NaN PS:  count += 1;  score += (ps < midpoint) when delta < 0,
// (ps > midpoint) when delta > 0, midpoint = median1 - delta/2.  The reference compares Python
// floats parsed from the TEXT of both tables, so PS and midpoints are float64 here.
//
// HBM-bound: 8 B per PS entry read once, 16 B per column written.  One thread owns one column of
// a block's row range (lanes = consecutive columns: coalesced row segments), accumulates in
// registers and issues two atomics per column at the end.
#include "common.h"

namespace {

constexpr int SIM_THREADS = 256;

__global__ void __launch_bounds__(SIM_THREADS) similarity_kernel(const double* __restrict__ ps, int64_t n, int s,
                                                                 const double* __restrict__ mid,
                                                                 const int8_t* __restrict__ sign, int rows_per_block,
                                                                 unsigned long long* __restrict__ scores,
                                                                 unsigned long long* __restrict__ counts) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(n, r0 + rows_per_block);
    for (int c = threadIdx.x; c < s; c += SIM_THREADS) {
        unsigned long long sc = 0, cnt = 0;
        for (int64_t r = r0; r < r1; ++r) {
            const int sg = sign[r];            // block-uniform
            if (sg == 0) continue;
            const double m = mid[r];
            const double v = ps[r * s + c];
            if (v == v) {
                ++cnt;
                sc += (sg < 0) ? (v < m) : (v > m);
            }
        }
        if (cnt) {
            atomicAdd(&counts[c], cnt);
            if (sc) atomicAdd(&scores[c], sc);
        }
    }
}

}  // namespace

extern "C" int sdice_similarity_dev(sdice_ctx* ctx, int64_t n, int32_t s, const double* d_ps, const double* d_mid,
                                    const int8_t* d_sign, int64_t* d_scores, int64_t* d_counts) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    if (s == 0) return SDICE_OK;
    SD_ARG(d_scores && d_counts, "NULL output");
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipMemsetAsync(d_scores, 0, (size_t)s * 8, ctx->stream));
    SD_HIP(hipMemsetAsync(d_counts, 0, (size_t)s * 8, ctx->stream));
    if (n == 0) return SDICE_OK;
    SD_ARG(d_ps && d_mid && d_sign, "NULL input");
    int64_t blocks = (int64_t)ctx->n_cu * 8;
    int64_t rows = sd_ceil_div(n, blocks);
    if (rows < 16) rows = 16;
    blocks = sd_ceil_div(n, rows);
    SD_LAUNCH(ctx, "similarity_kernel", similarity_kernel, dim3((unsigned)blocks), dim3(SIM_THREADS), 0, d_ps, n, (int)s,
              d_mid, d_sign, (int)rows, reinterpret_cast<unsigned long long*>(d_scores),
              reinterpret_cast<unsigned long long*>(d_counts));
    return SDICE_OK;
}

extern "C" int sdice_similarity(sdice_ctx* ctx, int64_t n, int32_t s, const double* ps, const double* mid,
                                const int8_t* sign, int64_t* scores, int64_t* counts) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    if (s == 0) return SDICE_OK;
    SD_ARG(scores && counts, "NULL output");
    SD_ARG(n == 0 || (ps && mid && sign), "NULL input");
    double *d_ps = nullptr, *d_mid = nullptr;
    int8_t* d_sign = nullptr;
    int64_t* d_out = nullptr;
    int rc = sdice_dmalloc(ctx, (int64_t)s * 16, (void**)&d_out);
    if (rc == SDICE_OK && n) rc = sdice_dmalloc(ctx, n * s * 8, (void**)&d_ps);
    if (rc == SDICE_OK && n) rc = sdice_dmalloc(ctx, n * 8, (void**)&d_mid);
    if (rc == SDICE_OK && n) rc = sdice_dmalloc(ctx, n, (void**)&d_sign);
    if (rc == SDICE_OK && n) rc = sdice_h2d(ctx, d_ps, ps, n * s * 8);
    if (rc == SDICE_OK && n) rc = sdice_h2d(ctx, d_mid, mid, n * 8);
    if (rc == SDICE_OK && n) rc = sdice_h2d(ctx, d_sign, sign, n);
    if (rc == SDICE_OK) rc = sdice_similarity_dev(ctx, n, s, d_ps, d_mid, d_sign, d_out, d_out + s);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, scores, d_out, (int64_t)s * 8);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, counts, d_out + s, (int64_t)s * 8);
    sdice_dfree(ctx, d_ps); sdice_dfree(ctx, d_mid); sdice_dfree(ctx, d_sign); sdice_dfree(ctx, d_out);
    return rc;
}
