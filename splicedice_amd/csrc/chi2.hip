// pairwise --chi2: Yates-corrected chi-square test of independence for every sample pair.
//
// Replaces pairwise_fisher.py:133-136,179 with test_method = scipy.stats.chi2_contingency
// (scipy/stats/contingency.py, 1.15.3): for table [[a, b], [c, d]]
//   expected[i][j] = row_i * col_j / total                      (float64)
//   any expected == 0  -> scipy raises ValueError and the reference run aborts; here the table is
//                         counted in *n_bad (p = NaN) and the host raises
//   dof = 1, Yates: observed += sign(expected - observed) * min(0.5, |expected - observed|)
//   chi2 = sum (observed - expected)^2 / expected  (row-major order), p = chdtrc(1, chi2)
//        = erfc(sqrt(chi2 / 2))
// HBM-bound in principle (8 B per p-value, ~40 flops); same pair enumeration as the Fisher kernel.
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ double chi2_yates_p(double a, double b, double c, double d, bool& bad) {
    const double r0 = a + b, r1 = c + d, c0 = a + c, c1 = b + d, tot = r0 + r1;
    const double e[4] = {r0 * c0 / tot, r0 * c1 / tot, r1 * c0 / tot, r1 * c1 / tot};
    const double o[4] = {a, b, c, d};
    if (!(tot > 0.0) || e[0] == 0.0 || e[1] == 0.0 || e[2] == 0.0 || e[3] == 0.0) {
        bad = true;
        return __builtin_nan("");
    }
    double stat = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double diff = e[i] - o[i];
        const double mag = fmin(0.5, fabs(diff));
        const double dir = diff > 0.0 ? 1.0 : (diff < 0.0 ? -1.0 : 0.0);
        const double oc = o[i] + mag * dir;
        const double t = oc - e[i];
        stat += t * t / e[i];
    }
    return erfc(sqrt(0.5 * stat));
}

__global__ void __launch_bounds__(256) chi2_pairs_kernel(const int32_t* __restrict__ incl,
                                                         const int64_t* __restrict__ excl, int64_t n, int s,
                                                         double* __restrict__ p, unsigned long long* __restrict__ n_bad) {
    extern __shared__ double smd[];
    double* inc = smd;
    double* exc = smd + s;
    const int64_t n_pairs = (int64_t)s * (s - 1) / 2;
    unsigned long long bad_count = 0;
    for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
        __syncthreads();
        for (int k = threadIdx.x; k < s; k += blockDim.x) {
            inc[k] = (double)incl[row * s + k];
            exc[k] = (double)excl[row * s + k];
        }
        __syncthreads();
        double* out = p + row * n_pairs;
        for (int64_t q = threadIdx.x; q < n_pairs; q += blockDim.x) {
            // invert q = i*s - i(i+1)/2 + (j-i-1)   (pairwise_fisher.py:142-147)
            const double bb = 2.0 * s - 1.0;
            int i = (int)((bb - sqrt(bb * bb - 8.0 * (double)q)) * 0.5);
            if (i < 0) i = 0;
            if (i > s - 2) i = s - 2;
            while (i > 0 && (int64_t)i * s - (int64_t)i * (i + 1) / 2 > q) --i;
            while ((int64_t)(i + 1) * s - (int64_t)(i + 1) * (i + 2) / 2 <= q) ++i;
            const int j = (int)(q - ((int64_t)i * s - (int64_t)i * (i + 1) / 2)) + i + 1;
            bool bad = false;
            out[q] = chi2_yates_p(inc[i], inc[j], exc[i], exc[j], bad);
            bad_count += bad ? 1 : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad_count += __shfl_xor(bad_count, o);
    if ((threadIdx.x & 63) == 0 && bad_count) atomicAdd(n_bad, bad_count);
}

}  // namespace

extern "C" int sdice_chi2_pairs_dev(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* d_incl, const int64_t* d_excl,
                                    double* d_p, int64_t* d_n_bad) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    SD_ARG(d_n_bad, "d_n_bad is NULL");
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipMemsetAsync(d_n_bad, 0, 8, ctx->stream));
    if (n == 0 || s < 2) return SDICE_OK;
    SD_ARG(d_incl && d_excl && d_p, "NULL pointer");
    SD_ARG(s <= 8192, "more than 8192 samples per junction is not supported");
    int64_t blocks = n;
    const int64_t cap = (int64_t)ctx->n_cu * 32;
    if (blocks > cap) blocks = cap;
    const size_t lds = (size_t)s * 16;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chi2_pairs_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "chi2_pairs_kernel", chi2_pairs_kernel, dim3((unsigned)blocks), dim3(256), lds, d_incl, d_excl, n, (int)s,
              d_p, reinterpret_cast<unsigned long long*>(d_n_bad));
    return SDICE_OK;
}

extern "C" int sdice_chi2_pairs(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* incl, const int64_t* excl, double* p,
                                int64_t* n_bad) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    SD_ARG(n_bad, "n_bad is NULL");
    *n_bad = 0;
    if (n == 0 || s < 2) return SDICE_OK;
    SD_ARG(incl && excl && p, "NULL pointer");
    for (int64_t i = 0; i < n * s; ++i) SD_ARG(incl[i] >= 0 && excl[i] >= 0, "counts must be non-negative");
    const int64_t n_pairs = (int64_t)s * (s - 1) / 2;
    int32_t* di = nullptr;
    int64_t *de = nullptr, *db = nullptr;
    double* dp = nullptr;
    int rc = sdice_dmalloc(ctx, n * s * 4, (void**)&di);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * s * 8, (void**)&de);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * n_pairs * 8, (void**)&dp);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, 8, (void**)&db);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, di, incl, n * s * 4);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, de, excl, n * s * 8);
    if (rc == SDICE_OK) rc = sdice_chi2_pairs_dev(ctx, n, s, di, de, dp, db);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, p, dp, n * n_pairs * 8);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, n_bad, db, 8);
    sdice_dfree(ctx, di); sdice_dfree(ctx, de); sdice_dfree(ctx, dp); sdice_dfree(ctx, db);
    return rc;
}
