// K6: two-sided Fisher exact test for every sample pair of every junction.
//
// Replaces the inner loop of pairwise (pairwise_fisher.py:164-179):
//   table = [[incl_a, incl_b], [excl_a, excl_b]];  p = scipy.stats.fisher_exact(table)[1]
// scipy (1.15.3, _stats_py.py fisher_exact): any zero margin -> 1.0; otherwise, with
// n1 = a+b, n2 = c+d, n = a+c and the hypergeometric pmf over k = table[0][0], the
// two-sided p is the sum of pmf(k) over the support with pmf(k) <= pmf(a) * (1 + 1e-14)
// (its cdf/sf + binary-search branches compute exactly that set), clipped at 1.
//
// Here: pmf(a) from a log-factorial table in HBM/L2 (device lgamma beyond the table), and
// the RATIOS r_k = pmf(k)/pmf(a) by the exact one-step recurrence outward from k = a in both
// directions, so the tie test r_k <= 1 + 1e-12 is decided to ~1e-14, far tighter than a
// log-gamma difference could.  p = pmf(a) * (1 + sum of the accepted r_k).  Monotone tails are
// cut once the remaining mass is below 1e-18 of the sum (the pmf is log-concave).
// The kernel is f64-VALU bound (O(support) steps per p-value), not HBM bound.
#include "common.h"
#include <math.h>

namespace {

struct LfTable {
    const double* lf;   // lf[k] = lgamma(k + 1), k in [0, n)
    long long n;
};

__device__ __forceinline__ double logfact(const LfTable& t, long long k) {
    return k < t.n ? t.lf[k] : lgamma((double)k + 1.0);
}

// 1/y for y an integer-valued double in [1, 2^63): hardware seed + two Newton steps (no scaling or
// fix-up is ever needed in this range).  Accurate to ~1 ulp; the step ratios it feeds carry a
// relative error of a few 1e-16, against a tie slack of 1e-12 and a p-value tolerance of 1e-9.
// The IEEE division it replaces was ~2/3 of the instructions of a walk step.
__device__ __forceinline__ double rcp_pos(double y) {
    double x = __builtin_amdgcn_rcp(y);
    double e = fma(-y, x, 1.0);
    x = fma(x, e, x);
    e = fma(-y, x, 1.0);
    return fma(x, e, x);
}

// one-directional walk of the ratios; dir = +1 (k increasing) or -1
template <int DIR>
__device__ __forceinline__ double walk_side(double a, double n1, double n2, double n, double bound_steps) {
    // up:   rho = (n1-k)(n-k) / ((k+1)(n2-n+k+1))
    // down: rho = k (n2-n+k) / ((n1-k+1)(n-k+1))
    double u1, u2, v1, v2;   // numerator factors u (decreasing by 1 per step), denominator factors v (increasing)
    if (DIR > 0) { u1 = n1 - a; u2 = n - a; v1 = a + 1.0; v2 = n2 - n + a + 1.0; }
    else { u1 = a; u2 = n2 - n + a; v1 = n1 - a + 1.0; v2 = n - a + 1.0; }
    const double slack = 1.0 + 1e-12;
    double r = 1.0, total = 0.0;
    int e = 0;   // r is scaled by 2^(500 e) while it is astronomically above 1
    for (double t = 0.0; t < bound_steps; t += 1.0) {
        const double rho = (u1 * u2) * rcp_pos(v1 * v2);
        r *= rho;
        u1 -= 1.0; u2 -= 1.0; v1 += 1.0; v2 += 1.0;
        if (r > 0x1p500) { r *= 0x1p-500; ++e; }
        else if (e > 0 && r < 0x1p-100) { r *= 0x1p500; --e; }
        if (e == 0 && r <= slack) {
            total += r;
            if (rho < 0.5 && r < 1e-18 * (1.0 + total)) break;
        }
    }
    return total;
}

__device__ double fisher_two_sided(long long a, long long b, long long c, long long d, const LfTable& t) {
    const long long n1 = a + b, n2 = c + d, n = a + c, m = b + d;
    if (n1 == 0 || n2 == 0 || n == 0 || m == 0) return 1.0;   // a zero margin (scipy: p = 1)
    const long long M = n1 + n2;
    const long long lo = n - n2 > 0 ? n - n2 : 0;
    const long long hi = n1 < n ? n1 : n;
    const double logp = logfact(t, n1) + logfact(t, n2) + logfact(t, n) + logfact(t, M - n) - logfact(t, M) -
                        logfact(t, a) - logfact(t, n1 - a) - logfact(t, n - a) - logfact(t, n2 - n + a);
    const double pexact = exp(logp);
    const double da = (double)a, dn1 = (double)n1, dn2 = (double)n2, dn = (double)n;
    double total = 1.0;
    total += walk_side<-1>(da, dn1, dn2, dn, (double)(a - lo));
    total += walk_side<+1>(da, dn1, dn2, dn, (double)(hi - a));
    const double p = pexact * total;
    return p < 1.0 ? p : 1.0;
}

__global__ void __launch_bounds__(256) lf_table_kernel(double* __restrict__ lf, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lf[i] = lgamma((double)i + 1.0);
}

__global__ void __launch_bounds__(256) fisher_tables_kernel(const int64_t* __restrict__ abcd, int64_t m,
                                                            double* __restrict__ p, LfTable t) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    p[i] = fisher_two_sided(abcd[4 * i], abcd[4 * i + 1], abcd[4 * i + 2], abcd[4 * i + 3], t);
}

// one workgroup per junction; pairs q = (i, j), i < j, row-major (pairwise_fisher.py:142-147)
__global__ void __launch_bounds__(256) fisher_pairs_kernel(const int32_t* __restrict__ incl,
                                                           const int64_t* __restrict__ excl, int64_t n, int s,
                                                           double* __restrict__ p, LfTable t) {
    extern __shared__ long long sm[];
    long long* inc = sm;
    long long* exc = sm + s;
    const int64_t n_pairs = (int64_t)s * (s - 1) / 2;
    for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
        __syncthreads();
        for (int k = threadIdx.x; k < s; k += blockDim.x) {
            inc[k] = incl[row * s + k];
            exc[k] = excl[row * s + k];
        }
        __syncthreads();
        double* out = p + row * n_pairs;
        for (int64_t q = threadIdx.x; q < n_pairs; q += blockDim.x) {
            // invert q = i*s - i(i+1)/2 + (j-i-1)
            const double bb = 2.0 * s - 1.0;
            int i = (int)((bb - sqrt(bb * bb - 8.0 * (double)q)) * 0.5);
            if (i < 0) i = 0;
            if (i > s - 2) i = s - 2;
            while (i > 0 && (int64_t)i * s - (int64_t)i * (i + 1) / 2 > q) --i;
            while ((int64_t)(i + 1) * s - (int64_t)(i + 1) * (i + 2) / 2 <= q) ++i;
            const int j = (int)(q - ((int64_t)i * s - (int64_t)i * (i + 1) / 2)) + i + 1;
            out[q] = fisher_two_sided(inc[i], inc[j], exc[i], exc[j], t);
        }
    }
}

}  // namespace

// log-factorial table owned by the context (built on first use)
static int get_lf_table(sdice_ctx* ctx, LfTable* out) {
    long long want = ctx->param("fisher.table_max", 1 << 20);
    if (want < 2) want = 2;
    if (ctx->d_lf && ctx->lf_n == want) {
        out->lf = ctx->d_lf;
        out->n = want;
        return SDICE_OK;
    }
    if (ctx->d_lf) {
        SD_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_lf);
        ctx->d_lf = nullptr;
        ctx->lf_n = 0;
    }
    double* lf = nullptr;
    hipError_t e = hipMalloc((void**)&lf, (size_t)want * 8);
    if (e != hipSuccess) {
        sdice_set_error("fisher: hipMalloc of log-factorial table failed: %s", hipGetErrorString(e));
        return SDICE_ERR_NOMEM;
    }
    SD_LAUNCH(ctx, "lf_table_kernel", lf_table_kernel, dim3((unsigned)sd_ceil_div(want, 256)), dim3(256), 0, lf, want);
    ctx->d_lf = lf;
    ctx->lf_n = want;
    out->lf = lf;
    out->n = want;
    return SDICE_OK;
}

extern "C" int sdice_fisher_pairs_dev(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* d_incl,
                                      const int64_t* d_excl, double* d_p) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    if (n == 0 || s < 2) return SDICE_OK;
    SD_ARG(d_incl && d_excl && d_p, "NULL pointer");
    SD_ARG(s <= 8192, "more than 8192 samples per junction is not supported");
    SD_HIP(hipSetDevice(ctx->device));
    LfTable t;
    SD_TRY(get_lf_table(ctx, &t));
    int threads = (int)ctx->param("fisher.threads", 256);
    threads = (threads / 64) * 64;
    if (threads < 64) threads = 64;
    if (threads > 256) threads = 256;
    int64_t blocks = n;
    const int64_t cap = (int64_t)ctx->n_cu * 32;
    if (blocks > cap) blocks = cap;
    const size_t lds = (size_t)s * 16;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fisher_pairs_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "fisher_pairs_kernel", fisher_pairs_kernel, dim3((unsigned)blocks), dim3(threads), lds, d_incl, d_excl,
              n, (int)s, d_p, t);
    return SDICE_OK;
}

extern "C" int sdice_fisher_pairs(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* incl, const int64_t* excl,
                                  double* p) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    if (n == 0 || s < 2) return SDICE_OK;
    SD_ARG(incl && excl && p, "NULL pointer");
    for (int64_t i = 0; i < n * s; ++i) SD_ARG(incl[i] >= 0 && excl[i] >= 0, "counts must be non-negative");
    const int64_t n_pairs = (int64_t)s * (s - 1) / 2;
    int32_t* di = nullptr;
    int64_t* de = nullptr;
    double* dp = nullptr;
    int rc = sdice_dmalloc(ctx, n * s * 4, (void**)&di);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * s * 8, (void**)&de);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * n_pairs * 8, (void**)&dp);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, di, incl, n * s * 4);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, de, excl, n * s * 8);
    if (rc == SDICE_OK) rc = sdice_fisher_pairs_dev(ctx, n, s, di, de, dp);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, p, dp, n * n_pairs * 8);
    sdice_dfree(ctx, di); sdice_dfree(ctx, de); sdice_dfree(ctx, dp);
    return rc;
}

extern "C" int sdice_fisher_tables(sdice_ctx* ctx, int64_t m, const int64_t* abcd, double* p) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(m >= 0, "negative size");
    if (m == 0) return SDICE_OK;
    SD_ARG(abcd && p, "NULL pointer");
    for (int64_t i = 0; i < 4 * m; ++i) SD_ARG(abcd[i] >= 0, "table entries must be non-negative");
    SD_HIP(hipSetDevice(ctx->device));
    LfTable t;
    SD_TRY(get_lf_table(ctx, &t));
    int64_t* dt = nullptr;
    double* dp = nullptr;
    int rc = sdice_dmalloc(ctx, m * 32, (void**)&dt);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, m * 8, (void**)&dp);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, dt, abcd, m * 32);
    if (rc == SDICE_OK) {
        hipLaunchKernelGGL(fisher_tables_kernel, dim3((unsigned)sd_ceil_div(m, 256)), dim3(256), 0, ctx->stream, dt, m, dp,
                           t);
        if (hipGetLastError() != hipSuccess) { sdice_set_error("fisher_tables_kernel launch failed"); rc = SDICE_ERR_HIP; }
    }
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, p, dp, m * 8);
    sdice_dfree(ctx, dt); sdice_dfree(ctx, dp);
    return rc;
}
