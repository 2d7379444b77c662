// K9: per-row np.nanmean / np.nanstd over a selected set of columns (findOutliers.py:125-135; SURVEY
// 8(f) rank 4), bit-identical to numpy in the matrix's own dtype (float32 or float64).
//
// numpy (lib/_nanfunctions_impl.py): NaNs are REPLACED BY ZERO IN PLACE (the summation order is that
// of all K selected values, not of the valid ones), cnt = number of valid values,
//   mean = T(f64(pairwise_sum(x0)) / f64(cnt)),
//   std  = sqrt(T(f64(pairwise_sum(d*d)) / f64(cnt))),  d = (x0 - mean) with the NaN slots back at zero,
// where pairwise_sum is numpy's 8-accumulator / halving summation in T and the two divisions run in
// float64 (a float32 value divided by an integer count promotes) before rounding to T.
//
// One wave per row, K <= 1024 selected columns; the K values sit in LDS, the sum uses the same
// lane = leaf * 8 + accumulator layout as the rank-sum kernels (ranksum.hip).
#include "common.h"
#include <math.h>

namespace {

#define RS_WAVE_SYNC()                                         \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

constexpr int RS_DEPTH = 4;        // 1024 values: up to four halvings, sixteen leaves

template <int DEPTH>
__device__ __forceinline__ void rs_leaves(int off, int len, int* leaf_off, int& nl, bool writer) {
    if (DEPTH == 0 || len <= 128) {
        if (writer) leaf_off[nl] = off;
        ++nl;
    } else {
        int n2 = len / 2;
        n2 -= n2 % 8;
        rs_leaves<(DEPTH > 0 ? DEPTH - 1 : 0)>(off, n2, leaf_off, nl, writer);
        rs_leaves<(DEPTH > 0 ? DEPTH - 1 : 0)>(off + n2, len - n2, leaf_off, nl, writer);
    }
}

template <int DEPTH, typename T>
__device__ __forceinline__ T rs_combine(int len, const T* leaf_sum, int& next) {
    if (DEPTH == 0 || len <= 128) return leaf_sum[next++];
    int n2 = len / 2;
    n2 -= n2 % 8;
    const T l = rs_combine<(DEPTH > 0 ? DEPTH - 1 : 0), T>(n2, leaf_sum, next);
    const T r = rs_combine<(DEPTH > 0 ? DEPTH - 1 : 0), T>(len - n2, leaf_sum, next);
    return l + r;
}

__device__ __forceinline__ float rs_sqrt(float v) { return sqrtf(v); }     // correctly rounded (default for HIP)
__device__ __forceinline__ double rs_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float rs_shfl_xor(float v, int m) { return __shfl_xor(v, m); }
__device__ __forceinline__ double rs_shfl_xor(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

// numpy pairwise_sum of C[0..n) by one wave (n <= 1024); leaf boundaries come from leaf_off[0..nl]
template <typename T>
__device__ __forceinline__ T rs_wave_sum(const T* C, int n, int lane, const int* leaf_off, int nl, T* leaf_sum) {
    const int j = lane & 7;
    for (int base = 0; base < nl; base += 8) {
        const int L = base + (lane >> 3);
        int off = 0, len = 0;
        if (L < nl) { off = leaf_off[L]; len = leaf_off[L + 1] - off; }
        const int main_n = len - (len & 7);
        T r = 0;
        if (len >= 8) {
            r = C[off + j];
            for (int i = 8; i < main_n; i += 8) r += C[off + i + j];
        }
        r = r + rs_shfl_xor(r, 1);
        r = r + rs_shfl_xor(r, 2);
        r = r + rs_shfl_xor(r, 4);
        for (int i = (len >= 8 ? main_n : 0); i < len; ++i) r += C[off + i];
        if (j == 0 && L < nl) leaf_sum[L] = r;
    }
    RS_WAVE_SYNC();
    int next = 0;
    const T out = rs_combine<RS_DEPTH, T>(n, leaf_sum, next);
    RS_WAVE_SYNC();
    return out;
}

template <typename T>
__global__ void __launch_bounds__(256) rowstats_kernel(const T* __restrict__ data, int64_t n, int s,
                                                       const int32_t* __restrict__ idx, int K, T* __restrict__ mean_out,
                                                       T* __restrict__ std_out, int32_t* __restrict__ nan_out) {
    extern __shared__ __align__(16) unsigned char smem_rs[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
    const size_t per_wave = ((size_t)K * sizeof(T) + 15) / 16 * 16 + 24 * sizeof(T) + 24 * sizeof(int);
    unsigned char* base = smem_rs + (size_t)wave * per_wave;
    T* C = reinterpret_cast<T*>(base);
    T* leaf_sum = reinterpret_cast<T*>(base + ((size_t)K * sizeof(T) + 15) / 16 * 16);
    int* leaf_off = reinterpret_cast<int*>(leaf_sum + 24);
    // the leaf layout depends on K only
    int nl = 0;
    rs_leaves<RS_DEPTH>(0, K, leaf_off, nl, lane == 0);
    if (lane == 0) leaf_off[nl] = K;
    RS_WAVE_SYNC();
    for (int64_t row = (int64_t)blockIdx.x * wpb + wave; row < n; row += (int64_t)gridDim.x * wpb) {
        const T* prow = data + row * s;
        int n_nan = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            T v = 0;
            bool isn = false;
            if (k < K) { v = prow[idx[k]]; isn = v != v; }
            n_nan += __popcll(__ballot(isn));
            if (k < K) C[k] = isn ? (T)0 : v;
        }
        RS_WAVE_SYNC();
        const double cnt = (double)(K - n_nan);
        const T sum1 = rs_wave_sum<T>(C, K, lane, leaf_off, nl, leaf_sum);
        const T avg = (T)((double)sum1 / cnt);
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            if (k < K) {
                const T v = prow[idx[k]];              // L1/L2 hit; tells the NaN slots apart from real zeros
                const T d = (v != v) ? (T)0 : (C[k] - avg);
                C[k] = d * d;
            }
        }
        RS_WAVE_SYNC();
        const T sum2 = rs_wave_sum<T>(C, K, lane, leaf_off, nl, leaf_sum);
        const T var = (T)((double)sum2 / cnt);
        if (lane == 0) {
            mean_out[row] = avg;
            std_out[row] = rs_sqrt(var);
            nan_out[row] = n_nan;
        }
        RS_WAVE_SYNC();
    }
}

template <typename T>
int launch_rowstats(sdice_ctx* ctx, const T* d_data, int64_t n, int s, const int32_t* d_idx, int K, T* d_mean, T* d_std,
                    int32_t* d_nan) {
    const int waves = 4;
    const size_t per_wave = ((size_t)K * sizeof(T) + 15) / 16 * 16 + 24 * sizeof(T) + 24 * sizeof(int);
    const size_t lds = per_wave * waves;
    int64_t blocks = sd_ceil_div(n, waves);
    const int64_t cap = (int64_t)ctx->n_cu * 8;
    if (blocks > cap) blocks = cap;
    SD_LAUNCH(ctx, "rowstats_kernel", (rowstats_kernel<T>), dim3((unsigned)blocks), dim3(waves * 64), lds, d_data, n, s, d_idx, K,
              d_mean, d_std, d_nan);
    return SDICE_OK;
}

}  // namespace

extern "C" int sdice_rowstats_dev(sdice_ctx* ctx, int64_t n, int32_t s, const void* d_data, int dtype,
                                  const int32_t* d_idx, int32_t k, void* d_mean, void* d_std, int32_t* d_nan) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0 && k >= 0, "negative size");
    SD_ARG(dtype == 0 || dtype == 1, "dtype must be 0 (float32) or 1 (float64)");
    SD_ARG(k >= 1 && k <= 1024, "between 1 and 1024 selected columns are supported");
    if (n == 0) return SDICE_OK;
    SD_ARG(d_data && d_idx && d_mean && d_std && d_nan, "NULL pointer");
    SD_HIP(hipSetDevice(ctx->device));
    if (dtype == 0)
        return launch_rowstats<float>(ctx, (const float*)d_data, n, s, d_idx, k, (float*)d_mean, (float*)d_std, d_nan);
    return launch_rowstats<double>(ctx, (const double*)d_data, n, s, d_idx, k, (double*)d_mean, (double*)d_std, d_nan);
}

extern "C" int sdice_rowstats(sdice_ctx* ctx, int64_t n, int32_t s, const void* data, int dtype, const int32_t* idx,
                              int32_t k, void* mean, void* std_, int32_t* n_nan) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0 && k >= 0, "negative size");
    SD_ARG(dtype == 0 || dtype == 1, "dtype must be 0 (float32) or 1 (float64)");
    if (n == 0) return SDICE_OK;
    SD_ARG(data && idx && mean && std_ && n_nan, "NULL pointer");
    for (int i = 0; i < k; ++i) SD_ARG(idx[i] >= 0 && idx[i] < s, "column index out of range");
    const int64_t w = dtype == 0 ? 4 : 8;
    void *d_data = nullptr, *d_mean = nullptr, *d_std = nullptr;
    int32_t *d_idx = nullptr, *d_nan = nullptr;
    int rc = sdice_dmalloc(ctx, n * s * w, &d_data);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, (int64_t)(k > 0 ? k : 1) * 4, (void**)&d_idx);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * w, &d_mean);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * w, &d_std);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * 4, (void**)&d_nan);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, d_data, data, n * s * w);
    if (rc == SDICE_OK && k) rc = sdice_h2d(ctx, d_idx, idx, (int64_t)k * 4);
    if (rc == SDICE_OK) rc = sdice_rowstats_dev(ctx, n, s, d_data, dtype, d_idx, k, d_mean, d_std, d_nan);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, mean, d_mean, n * w);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, std_, d_std, n * w);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, n_nan, d_nan, n * 4);
    sdice_dfree(ctx, d_data); sdice_dfree(ctx, d_idx); sdice_dfree(ctx, d_mean); sdice_dfree(ctx, d_std);
    sdice_dfree(ctx, d_nan);
    return rc;
}
