// K7: Benjamini-Hochberg FDR.
//
// Replaces statsmodels.stats.multitest.multipletests(p, method="fdr_bh")[1]
// (compareSampleSets.py:235; pairwise_fisher.py:185,190): sort ascending,
// p_(i) / (i/m), running minimum from the largest rank down, clip at 1, unsort.
// (statsmodels is not installable in the build image: "parity unpinned" for this call,
// cross-checked against scipy.stats.false_discovery_control.)
//
// Device: radix sort of the IEEE-754 bit patterns (non-negative doubles order like
// unsigned integers) with the original index as payload, an elementwise kernel that
// writes p_(i)/(i/m) in REVERSED order, an inclusive min-scan (on bit patterns, again
// order preserving), and a scatter back to the original positions.
#include "common.h"

int sd_inclusive_min_scan_u64(sdice_ctx* ctx, int64_t n, const uint64_t* d_in, uint64_t* d_out);

namespace {

__global__ void __launch_bounds__(256) bh_keys_kernel(const double* __restrict__ p, int64_t m,
                                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double v = p[i];
    if (v == 0.0) v = 0.0;   // -0.0 -> +0.0
    keys[i] = (uint64_t)__double_as_longlong(v);
    idx[i] = (uint32_t)i;
}

// raw_rev[m-1-i] = p_(i) / ((i+1)/m)
__global__ void __launch_bounds__(256) bh_raw_kernel(const uint64_t* __restrict__ sorted, int64_t m,
                                                     uint64_t* __restrict__ raw_rev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double ps = __longlong_as_double((long long)sorted[i]);
    const double ecdf = (double)(i + 1) / (double)m;
    const double raw = ps / ecdf;
    raw_rev[m - 1 - i] = (uint64_t)__double_as_longlong(raw);
}

__global__ void __launch_bounds__(256) bh_scatter_kernel(const uint64_t* __restrict__ cummin_rev,
                                                         const uint32_t* __restrict__ idx, int64_t m,
                                                         double* __restrict__ q) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double v = __longlong_as_double((long long)cummin_rev[m - 1 - i]);
    if (v > 1.0) v = 1.0;
    q[idx[i]] = v;
}

__global__ void __launch_bounds__(256) transpose_f64_kernel(const double* __restrict__ in, int64_t rows, int64_t cols,
                                                            double* __restrict__ out) {
    // out[c, r] = in[r, c]; 32x32 tiles through LDS
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) tile[k][tx] = in[r * cols + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][k];
    }
}

}  // namespace

static int bh_core(sdice_ctx* ctx, int64_t m, const double* d_p, double* d_q) {
    Arena& A = ctx->arena;
    const size_t M = (size_t)m;
    uint64_t* kA = (uint64_t*)A.alloc(M * 8);
    uint64_t* kB = (uint64_t*)A.alloc(M * 8);
    uint64_t* kC = (uint64_t*)A.alloc(M * 8);
    uint32_t* vA = (uint32_t*)A.alloc(M * 4);
    uint32_t* vB = (uint32_t*)A.alloc(M * 4);
    uint32_t* vC = (uint32_t*)A.alloc(M * 4);
    if (!kA || !kB || !kC || !vA || !vB || !vC) return SDICE_ERR_NOMEM;
    const unsigned g = (unsigned)sd_ceil_div(m, 256);
    SD_LAUNCH(ctx, "bh_keys_kernel", bh_keys_kernel, dim3(g), dim3(256), 0, d_p, m, kA, vA);
    // p-values live in [0, 1] (or NaN): sign bit and bit 62 never vary, every other digit may
    SD_TRY(sd_radix_sort_pairs(ctx, m, kA, vA, kB, vB, kC, vC, 0x7fffffffffffffffull));
    SD_LAUNCH(ctx, "bh_raw_kernel", bh_raw_kernel, dim3(g), dim3(256), 0, kB, m, kA);
    SD_TRY(sd_inclusive_min_scan_u64(ctx, m, kA, kC));
    SD_LAUNCH(ctx, "bh_scatter_kernel", bh_scatter_kernel, dim3(g), dim3(256), 0, kC, vB, m, d_q);
    return SDICE_OK;
}

extern "C" int sdice_bh_dev(sdice_ctx* ctx, int64_t m, const double* d_p, double* d_q) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(m >= 0 && m < ((int64_t)1 << 32), "m out of range");
    if (m == 0) return SDICE_OK;
    SD_ARG(d_p && d_q, "NULL pointer");
    SD_HIP(hipSetDevice(ctx->device));
    SD_TRY(ctx->arena.reset(ctx->stream));
    return bh_core(ctx, m, d_p, d_q);
}

extern "C" int sdice_bh(sdice_ctx* ctx, int64_t m, const double* p, double* q) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(m >= 0, "negative size");
    if (m == 0) return SDICE_OK;
    SD_ARG(p && q, "NULL pointer");
    double *dp = nullptr, *dq = nullptr;
    int rc = sdice_dmalloc(ctx, m * 8, (void**)&dp);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, m * 8, (void**)&dq);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, dp, p, m * 8);
    if (rc == SDICE_OK) rc = sdice_bh_dev(ctx, m, dp, dq);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, q, dq, m * 8);
    sdice_dfree(ctx, dp); sdice_dfree(ctx, dq);
    return rc;
}

extern "C" int sdice_bh_columns(sdice_ctx* ctx, int64_t n, int64_t cols, double* p_inout) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && cols >= 0, "negative size");
    if (n == 0 || cols == 0) return SDICE_OK;
    SD_ARG(p_inout, "NULL pointer");
    SD_HIP(hipSetDevice(ctx->device));
    double *d_rm = nullptr, *d_cm = nullptr, *d_q = nullptr;
    int rc = sdice_dmalloc(ctx, n * cols * 8, (void**)&d_rm);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * cols * 8, (void**)&d_cm);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * cols * 8, (void**)&d_q);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, d_rm, p_inout, n * cols * 8);
    if (rc == SDICE_OK) {
        dim3 g((unsigned)sd_ceil_div(cols, 32), (unsigned)sd_ceil_div(n, 32));
        hipLaunchKernelGGL(transpose_f64_kernel, g, dim3(256), 0, ctx->stream, d_rm, n, cols, d_cm);
        for (int64_t c = 0; c < cols && rc == SDICE_OK; ++c) {
            rc = ctx->arena.reset(ctx->stream);
            if (rc == SDICE_OK) rc = bh_core(ctx, n, d_cm + c * n, d_q + c * n);
        }
        if (rc == SDICE_OK) {
            dim3 g2((unsigned)sd_ceil_div(n, 32), (unsigned)sd_ceil_div(cols, 32));
            hipLaunchKernelGGL(transpose_f64_kernel, g2, dim3(256), 0, ctx->stream, d_q, cols, n, d_rm);
            if (hipGetLastError() != hipSuccess) { sdice_set_error("transpose launch failed"); rc = SDICE_ERR_HIP; }
        }
    }
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, p_inout, d_rm, n * cols * 8);
    sdice_dfree(ctx, d_rm); sdice_dfree(ctx, d_cm); sdice_dfree(ctx, d_q);
    return rc;
}
