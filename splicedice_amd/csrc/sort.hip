// Device primitives: stable LSD radix sort of (u64 key, u32 value) pairs and scans.
//
// Radix sort (8-bit digits, 3 launches per digit):
//   1. radix_hist_kernel   : one 64-lane wave per chunk of keys -> 256-bin histogram in LDS
//                            -> hist[bin][chunk] (bin-major)
//   2. radix_binscan_kernel: one workgroup per bin: exclusive scan of hist[bin][*] in place,
//                            bin total -> bin_total[bin]
//   3. radix_scatter_kernel: one wave per chunk again; scans bin_total (256 values) for the
//                            digit bases, then walks its chunk 64 keys at a time; the stable
//                            rank of a key among equal digits of the wave comes from eight
//                            64-bit ballots (wave-wide match), running per-digit offsets
//                            live in LDS.
// Digits whose bits are constant over all keys (bit_mask) are skipped entirely.
#include "common.h"

namespace {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SCATTER_ROUNDS = 8;   // keys per lane held in registers by the scatter kernel

__global__ void __launch_bounds__(64) radix_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                        int keys_per_chunk, int n_chunks,
                                                        uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[RADIX];
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    for (int b = lane; b < RADIX; b += 64) h[b] = 0;
    __syncthreads();
    const int64_t beg = (int64_t)chunk * keys_per_chunk;
    const int64_t end = min(n, beg + keys_per_chunk);
    for (int64_t i = beg + lane; i < end; i += 64) {
        const uint32_t d = (uint32_t)(keys[i] >> shift) & (RADIX - 1);
        atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    for (int b = lane; b < RADIX; b += 64) hist[(int64_t)b * n_chunks + chunk] = h[b];
}

__global__ void __launch_bounds__(256) radix_binscan_kernel(uint32_t* __restrict__ hist, int n_chunks,
                                                            uint32_t* __restrict__ bin_total) {
    // exclusive scan of hist[bin][0..n_chunks) in place
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    const int bin = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t* row = hist + (int64_t)bin * n_chunks;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_chunks; base += 256) {
        const int i = base + tid;
        const uint32_t v = i < n_chunks ? row[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        uint32_t wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        const uint32_t carry = carry_s;
        if (i < n_chunks) row[i] = carry + wbase + x - v;
        __syncthreads();
        if (tid == 255) carry_s = carry + wbase + x;
        __syncthreads();
    }
    if (tid == 0) bin_total[bin] = carry_s;
}

__global__ void __launch_bounds__(64) radix_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in,
                                                           uint64_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                           int keys_per_chunk, int n_chunks,
                                                           const uint32_t* __restrict__ hist,
                                                           const uint32_t* __restrict__ bin_total) {
    __shared__ uint32_t off[RADIX];
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    // digit bases: exclusive scan of the 256 bin totals, 4 bins per lane
    {
        uint32_t t[4];
        uint32_t s = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { t[q] = bin_total[lane * 4 + q]; s += t[q]; }
        uint32_t x = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        uint32_t run = x - s;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = lane * 4 + q;
            off[b] = run + hist[(int64_t)b * n_chunks + chunk];
            run += t[q];
        }
    }
    __syncthreads();
    const int64_t beg = (int64_t)chunk * keys_per_chunk;
    const int64_t end = min(n, beg + keys_per_chunk);
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // the whole chunk goes to registers first: one memory round trip per chunk instead of one
    // per 64 keys (keys_per_chunk <= 64 * SCATTER_ROUNDS)
    uint64_t rk[SCATTER_ROUNDS];
    uint32_t rv[SCATTER_ROUNDS];
#pragma unroll
    for (int r = 0; r < SCATTER_ROUNDS; ++r) {
        const int64_t i = beg + r * 64 + lane;
        rk[r] = 0; rv[r] = 0;
        if (i < end) { rk[r] = keys_in[i]; rv[r] = vals_in[i]; }
    }
#pragma unroll
    for (int r = 0; r < SCATTER_ROUNDS; ++r) {
        const int64_t i = beg + r * 64 + lane;
        if (beg + r * 64 >= end) break;
        const bool active = i < end;
        const uint64_t key = rk[r];
        const uint32_t val = rv[r];
        const uint32_t d = (uint32_t)(key >> shift) & (RADIX - 1);
        uint64_t peers = __ballot(active);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        uint32_t pos = 0;
        if (active) pos = off[d] + (uint32_t)__popcll(peers & lt_mask);
        __syncthreads();  // all lanes have read off[] before the leaders update it
        if (active && (peers & lt_mask) == 0) off[d] += (uint32_t)__popcll(peers);
        __syncthreads();
        if (active) { keys_out[pos] = key; vals_out[pos] = val; }
    }
}

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

int sd_radix_sort_pairs(sdice_ctx* ctx, int64_t n, const uint64_t* d_keys_in, const uint32_t* d_vals_in,
                        uint64_t* d_keys_out, uint32_t* d_vals_out, uint64_t* d_keys_tmp, uint32_t* d_vals_tmp,
                        uint64_t bit_mask) {
    if (n <= 0) return SDICE_OK;
    // digits that actually vary
    int shifts[8], np = 0;
    for (int d = 0; d < 8; ++d)
        if ((bit_mask >> (8 * d)) & 0xffull) shifts[np++] = 8 * d;
    if (np == 0) {
        if (d_keys_in != d_keys_out) {
            SD_HIP(hipMemcpyAsync(d_keys_out, d_keys_in, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
            SD_HIP(hipMemcpyAsync(d_vals_out, d_vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return SDICE_OK;
    }
    // one wave per chunk; a chunk is at most 64 * SCATTER_ROUNDS keys (held in registers)
    int64_t kpc = sd_ceil_div(n, 4096);
    kpc = sd_ceil_div(kpc, 64) * 64;
    if (kpc < 128) kpc = 128;
    if (kpc > 64 * SCATTER_ROUNDS) kpc = 64 * SCATTER_ROUNDS;
    const int64_t n_chunks = sd_ceil_div(n, kpc);
    uint32_t* hist = (uint32_t*)ctx->arena.alloc((size_t)RADIX * n_chunks * 4);
    uint32_t* bin_total = (uint32_t*)ctx->arena.alloc(RADIX * 4);
    if (!hist || !bin_total) return SDICE_ERR_NOMEM;
    // ping-pong so that the last pass lands in *_out
    const uint64_t* kin = d_keys_in;
    const uint32_t* vin = d_vals_in;
    for (int p = 0; p < np; ++p) {
        const bool to_out = ((np - 1 - p) % 2) == 0;
        uint64_t* kout = to_out ? d_keys_out : d_keys_tmp;
        uint32_t* vout = to_out ? d_vals_out : d_vals_tmp;
        SD_LAUNCH(ctx, "radix_hist_kernel", radix_hist_kernel, dim3((unsigned)n_chunks), dim3(64), 0, kin, n, shifts[p],
                  (int)kpc, (int)n_chunks, hist);
        SD_LAUNCH(ctx, "radix_binscan_kernel", radix_binscan_kernel, dim3(RADIX), dim3(256), 0, hist, (int)n_chunks,
                  bin_total);
        SD_LAUNCH(ctx, "radix_scatter_kernel", radix_scatter_kernel, dim3((unsigned)n_chunks), dim3(64), 0, kin, vin, kout,
                  vout, n, shifts[p], (int)kpc, (int)n_chunks, hist, bin_total);
        kin = kout;
        vin = vout;
    }
    return SDICE_OK;
}
