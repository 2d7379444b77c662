// Device scans: per-block reduce, then per-block apply in which every block folds the aggregates of
// its predecessors itself (2 launches; beyond 4096 blocks a single-block scan of the aggregates sits
// in between, 3 launches).  Used for row_ptr (sum), the segmented prefix maximum of the clustering
// (max on composite keys) and the BH running minimum (min on IEEE bit patterns).
#include "common.h"

namespace {

// ---- scans ---------------------------------------------------------------------------
struct OpSumI64 {
    typedef int64_t T;
    __device__ static T identity() { return 0; }
    __device__ static T apply(T a, T b) { return a + b; }
};
struct OpMaxU64 {
    typedef uint64_t T;
    __device__ static T identity() { return 0; }
    __device__ static T apply(T a, T b) { return a > b ? a : b; }
};
struct OpMinU64 {
    typedef uint64_t T;
    __device__ static T identity() { return ~0ull; }
    __device__ static T apply(T a, T b) { return a < b ? a : b; }
};

template <typename T>
__device__ __forceinline__ T shfl_up64(T v, int o) {
    unsigned lo = (unsigned)((uint64_t)v & 0xffffffffu), hi = (unsigned)((uint64_t)v >> 32);
    lo = __shfl_up(lo, o);
    hi = __shfl_up(hi, o);
    return (T)(((uint64_t)hi << 32) | lo);
}

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;  // per thread -> 2048 per block

// inclusive scan inside a block of SCAN_THREADS*SCAN_ITEMS consecutive elements;
// returns each thread's items (inclusive, block-local) and the block aggregate
template <typename Op>
__device__ __forceinline__ void block_scan_items(typename Op::T (&v)[SCAN_ITEMS], typename Op::T& block_total,
                                                 typename Op::T* wsum /* [4] shared */) {
    typedef typename Op::T T;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int q = 1; q < SCAN_ITEMS; ++q) v[q] = Op::apply(v[q - 1], v[q]);
    T x = v[SCAN_ITEMS - 1];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T y = shfl_up64<T>(x, o);
        if (lane >= o) x = Op::apply(y, x);
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    T pre = Op::identity();
    for (int k = 0; k < w; ++k) pre = Op::apply(pre, wsum[k]);
    // exclusive prefix of this thread = pre (+) (x without own)
    T excl_lane = shfl_up64<T>(x, 1);
    if (lane == 0) excl_lane = Op::identity();
    const T tpre = Op::apply(pre, excl_lane);
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) v[q] = Op::apply(tpre, v[q]);
    T tot = Op::identity();
    for (int k = 0; k < SCAN_THREADS / 64; ++k) tot = Op::apply(tot, wsum[k]);
    block_total = tot;
    __syncthreads();
}

template <typename Op>
__global__ void __launch_bounds__(SCAN_THREADS) scan_reduce_kernel(const typename Op::T* __restrict__ in, int64_t n,
                                                                   typename Op::T* __restrict__ block_sums) {
    typedef typename Op::T T;
    __shared__ T wsum[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_THREADS * SCAN_ITEMS + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) v[q] = (base + q < n) ? in[base + q] : Op::identity();
    T tot;
    block_scan_items<Op>(v, tot, wsum);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single-block exclusive scan of the block sums (in place); total -> *total_out
template <typename Op>
__global__ void __launch_bounds__(SCAN_THREADS) scan_sums_kernel(typename Op::T* __restrict__ sums, int64_t m,
                                                                 typename Op::T* __restrict__ total_out) {
    typedef typename Op::T T;
    __shared__ T wsum[SCAN_THREADS / 64];
    __shared__ T last_incl[SCAN_THREADS];
    T carry = Op::identity();
    for (int64_t base0 = 0; base0 < m; base0 += SCAN_THREADS * SCAN_ITEMS) {
        const int64_t base = base0 + (int64_t)threadIdx.x * SCAN_ITEMS;
        T v[SCAN_ITEMS];
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS; ++q) v[q] = (base + q < m) ? sums[base + q] : Op::identity();
        T tot;
        block_scan_items<Op>(v, tot, wsum);
        // exclusive result = carry (+) inclusive value of the previous element
        last_incl[threadIdx.x] = v[SCAN_ITEMS - 1];
        __syncthreads();
        const T tprev = threadIdx.x == 0 ? Op::identity() : last_incl[threadIdx.x - 1];
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS; ++q) {
            const T e = q == 0 ? tprev : v[q - 1];
            if (base + q < m) sums[base + q] = Op::apply(carry, e);
        }
        carry = Op::apply(carry, tot);
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// final pass: out[i] = block_prefix (+) local scan; EXCLUSIVE selects exclusive/inclusive output
template <typename Op, bool EXCLUSIVE>
__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const typename Op::T* __restrict__ in, int64_t n,
                                                                  const typename Op::T* __restrict__ block_prefix,
                                                                  typename Op::T* __restrict__ out) {
    typedef typename Op::T T;
    __shared__ T wsum[SCAN_THREADS / 64];
    __shared__ T last_incl[SCAN_THREADS];
    const int64_t base = (int64_t)blockIdx.x * SCAN_THREADS * SCAN_ITEMS + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) v[q] = (base + q < n) ? in[base + q] : Op::identity();
    T tot;
    block_scan_items<Op>(v, tot, wsum);
    const T bp = block_prefix[blockIdx.x];
    if (EXCLUSIVE) {
        last_incl[threadIdx.x] = v[SCAN_ITEMS - 1];
        __syncthreads();
        const T tprev = threadIdx.x == 0 ? Op::identity() : last_incl[threadIdx.x - 1];
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS; ++q) {
            const T e = q == 0 ? tprev : v[q - 1];
            if (base + q < n) out[base + q] = Op::apply(bp, e);
        }
    } else {
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS; ++q)
            if (base + q < n) out[base + q] = Op::apply(bp, v[q]);
    }
}

// Two-launch variant for up to SCAN_SELF_MAX blocks: every block folds the aggregates of the blocks
// before it by itself (a few KB from L2) instead of waiting for a single-block scan of the
// aggregates -- one launch less on pipelines whose kernels sit at the ~5 us launch/drain floor.
// All three operators are associative and commutative, so the fold order is free.
constexpr int64_t SCAN_SELF_MAX = 4096;

template <typename Op, bool EXCLUSIVE>
__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_self_kernel(const typename Op::T* __restrict__ in, int64_t n,
                                                                       const typename Op::T* __restrict__ block_sums,
                                                                       typename Op::T* __restrict__ out,
                                                                       typename Op::T* __restrict__ total_out) {
    typedef typename Op::T T;
    __shared__ T wsum[SCAN_THREADS / 64];
    __shared__ T last_incl[SCAN_THREADS];
    __shared__ T fold[SCAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T acc = Op::identity();
    for (int64_t i = tid; i < (int64_t)blockIdx.x; i += SCAN_THREADS) acc = Op::apply(acc, block_sums[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned lo = (unsigned)((uint64_t)acc & 0xffffffffu), hi = (unsigned)((uint64_t)acc >> 32);
        lo = __shfl_xor(lo, o);
        hi = __shfl_xor(hi, o);
        acc = Op::apply(acc, (T)(((uint64_t)hi << 32) | lo));
    }
    if (lane == 0) fold[w] = acc;
    const int64_t base = (int64_t)blockIdx.x * SCAN_THREADS * SCAN_ITEMS + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) v[q] = (base + q < n) ? in[base + q] : Op::identity();
    T tot;
    block_scan_items<Op>(v, tot, wsum);          // (contains the barriers that publish fold[])
    T bp = Op::identity();
    for (int k = 0; k < SCAN_THREADS / 64; ++k) bp = Op::apply(bp, fold[k]);
    if (total_out && blockIdx.x == gridDim.x - 1 && tid == 0) *total_out = Op::apply(bp, tot);
    if (EXCLUSIVE) {
        last_incl[threadIdx.x] = v[SCAN_ITEMS - 1];
        __syncthreads();
        const T tprev = threadIdx.x == 0 ? Op::identity() : last_incl[threadIdx.x - 1];
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS; ++q) {
            const T e = q == 0 ? tprev : v[q - 1];
            if (base + q < n) out[base + q] = Op::apply(bp, e);
        }
    } else {
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS; ++q)
            if (base + q < n) out[base + q] = Op::apply(bp, v[q]);
    }
}

template <typename Op, bool EXCLUSIVE>
int scan_impl(sdice_ctx* ctx, int64_t n, const typename Op::T* d_in, typename Op::T* d_out,
              typename Op::T* d_total, const char* tag) {
    typedef typename Op::T T;
    if (n <= 0) {
        if (d_total) SD_HIP(hipMemsetAsync(d_total, 0, sizeof(T), ctx->stream));
        return SDICE_OK;
    }
    const int64_t per_block = SCAN_THREADS * SCAN_ITEMS;
    const int64_t nb = sd_ceil_div(n, per_block);
    T* sums = (T*)ctx->arena.alloc((size_t)nb * sizeof(T));
    if (!sums) return SDICE_ERR_NOMEM;
    (void)tag;
    SD_LAUNCH(ctx, "scan_reduce_kernel", (scan_reduce_kernel<Op>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, d_in, n, sums);
    if (nb <= SCAN_SELF_MAX) {
        SD_LAUNCH(ctx, "scan_apply_self_kernel", (scan_apply_self_kernel<Op, EXCLUSIVE>), dim3((unsigned)nb),
                  dim3(SCAN_THREADS), 0, d_in, n, sums, d_out, d_total);
        return SDICE_OK;
    }
    SD_LAUNCH(ctx, "scan_sums_kernel", (scan_sums_kernel<Op>), dim3(1), dim3(SCAN_THREADS), 0, sums, nb, d_total);
    SD_LAUNCH(ctx, "scan_apply_kernel", (scan_apply_kernel<Op, EXCLUSIVE>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0,
              d_in, n, sums, d_out);
    return SDICE_OK;
}

}  // namespace

int sd_exclusive_scan_i64(sdice_ctx* ctx, int64_t n, const int64_t* d_in, int64_t* d_out, int64_t* d_total) {
    return scan_impl<OpSumI64, true>(ctx, n, d_in, d_out, d_total, "sum");
}

int sd_inclusive_max_scan_u64(sdice_ctx* ctx, int64_t n, const uint64_t* d_in, uint64_t* d_out) {
    return scan_impl<OpMaxU64, false>(ctx, n, d_in, d_out, nullptr, "max");
}

int sd_inclusive_min_scan_u64(sdice_ctx* ctx, int64_t n, const uint64_t* d_in, uint64_t* d_out) {
    return scan_impl<OpMinU64, false>(ctx, n, d_in, d_out, nullptr, "min");
}
