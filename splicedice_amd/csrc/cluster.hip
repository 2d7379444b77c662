// K1 + K2: interval sort and overlap "clusters" (per-junction adjacency lists).
//
// Replaces SPLICEDICE.getClusters (SPLICEDICE.py:230-255; twin counts_to_ps.py:16-41) and
// the junctionIndex sort (SPLICEDICE.py:96).  The reference sorts junctions by
// (chrom, strand, left, right) and sweeps, keeping a most-recent-first list of still-open
// prior junctions; two junctions on one chromosome+strand are neighbours iff their closed
// intervals overlap (prior.right >= cur.left, :250).  Output rows use a DIFFERENT order,
// (chrom, left, right, strand) (:96).  Per junction the neighbour list order is: earlier
// junctions of the sweep, most recent first, then later ones in sweep order.
//
// Device formulation (no sweep):
//   1. ONE radix sort into sweep order on a packed key
//        chrom | strand | left - min_left | right - left        (field widths from a reduction)
//      (falls back to two sorts on wider keys when the fields do not fit 64 bits).
//   2. Everything else is decoded from the sorted keys, no gathers: composite keys
//        ckL[p] = seg<<32 | left,  ckR[p] = seg<<32 | right,   seg = chrom<<1 | strand.
//   3. The output row of sweep position p: both strands of a chromosome occupy the same index
//      range in row order and in sweep order, and within one strand the two orders agree, so
//        row(p) = chrom_start + (p - own_strand_start) + #{other strand: (left,right) < or <= mine}
//      -- one binary search in the other strand's sorted segment.
//   4. Later neighbours of p are the contiguous range (p, ub_p),
//        ub_p = upper_bound(ckL, seg_p<<32 | right_p)           (galloping binary search);
//      earlier neighbours are the q < p with ckR[q] >= seg_p<<32 | left_p; the backward walk
//      stops at the first q whose running prefix maximum of ckR drops below the target and
//      skips 64-aligned blocks whose block maximum is below it.
//   5. degree -> exclusive scan (row order) -> fill.
#include "common.h"
#include <unistd.h>

int sd_inclusive_max_scan_u64(sdice_ctx* ctx, int64_t n, const uint64_t* d_in, uint64_t* d_out);

namespace {

// red[0..1] OR/AND of (left<<32 | right<<1 | strand); red[2..3] OR/AND of chrom; red[4] error flag;
// red[5] max row reach (filled later); red[6] min left; red[7] max left; red[8] max (right-left)
constexpr int RED_WORDS = 16;   // one 128-byte line per slot
constexpr int RED_SLOTS = 32;

__global__ void __launch_bounds__(256) reduce_fields_kernel(const int32_t* __restrict__ chrom,
                                                            const int32_t* __restrict__ left,
                                                            const int32_t* __restrict__ right,
                                                            const int8_t* __restrict__ strand, int64_t n,
                                                            unsigned long long* __restrict__ red) {
    __shared__ unsigned long long sh[RED_WORDS][4];
    unsigned long long o1 = 0, a1 = ~0ull, o2 = 0, a2 = ~0ull, err = 0, mnl = ~0ull, mxl = 0, mxn = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t l = left[i], r = right[i], c = chrom[i];
        const int st = strand[i];
        if (l < 0 || r < l || c < 0 || (st != 0 && st != 1)) { err = 1; continue; }
        const uint64_t k = ((uint64_t)(uint32_t)l << 32) | ((uint64_t)(uint32_t)r << 1) | (uint64_t)(st & 1);
        o1 |= k; a1 &= k;
        o2 |= (uint64_t)(uint32_t)c; a2 &= (uint64_t)(uint32_t)c;
        mnl = min(mnl, (unsigned long long)l); mxl = max(mxl, (unsigned long long)l);
        mxn = max(mxn, (unsigned long long)(r - l));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        o1 |= __shfl_xor(o1, o); a1 &= __shfl_xor(a1, o);
        o2 |= __shfl_xor(o2, o); a2 &= __shfl_xor(a2, o);
        err |= __shfl_xor(err, o);
        mnl = min(mnl, (unsigned long long)__shfl_xor(mnl, o));
        mxl = max(mxl, (unsigned long long)__shfl_xor(mxl, o));
        mxn = max(mxn, (unsigned long long)__shfl_xor(mxn, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[0][w] = o1; sh[1][w] = a1; sh[2][w] = o2; sh[3][w] = a2; sh[4][w] = err;
        sh[6][w] = mnl; sh[7][w] = mxl; sh[8][w] = mxn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            sh[0][0] |= sh[0][k]; sh[1][0] &= sh[1][k]; sh[2][0] |= sh[2][k]; sh[3][0] &= sh[3][k]; sh[4][0] |= sh[4][k];
            sh[6][0] = min(sh[6][0], sh[6][k]); sh[7][0] = max(sh[7][0], sh[7][k]); sh[8][0] = max(sh[8][0], sh[8][k]);
        }
        red += (blockIdx.x % RED_SLOTS) * RED_WORDS;
        atomicOr(&red[0], sh[0][0]); atomicAnd(&red[1], sh[1][0]);
        atomicOr(&red[2], sh[2][0]); atomicAnd(&red[3], sh[3][0]);
        if (sh[4][0]) atomicOr(&red[4], 1ull);
        atomicMin(&red[6], sh[6][0]); atomicMax(&red[7], sh[7][0]); atomicMax(&red[8], sh[8][0]);
    }
}

// ------------------------------------------------------------------ packed (single sort) path
struct Pack {
    int bN;            // bits of (right - left)
    int bL;            // bits of (left - min_left)
    uint32_t min_left;
    __host__ __device__ int seg_shift() const { return bN + bL; }
};

__global__ void __launch_bounds__(256) pack_keys_kernel(const int32_t* __restrict__ chrom,
                                                        const int32_t* __restrict__ left,
                                                        const int32_t* __restrict__ right,
                                                        const int8_t* __restrict__ strand, int64_t n, Pack pk,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t seg = ((uint64_t)(uint32_t)chrom[i] << 1) | (uint64_t)(strand[i] & 1);
    const uint64_t l = (uint32_t)left[i] - pk.min_left;
    const uint64_t len = (uint32_t)(right[i] - left[i]);
    keys[i] = (seg << pk.seg_shift()) | (l << pk.bN) | len;
    idx[i] = (uint32_t)i;
}

__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t lo, int64_t hi, uint64_t t) {
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < t) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// seg_start[t] = first sweep position whose seg >= t, t in [0, n_seg]
__global__ void __launch_bounds__(256) seg_table_kernel(const uint64_t* __restrict__ keys, int64_t n, Pack pk,
                                                        int64_t n_seg, int64_t* __restrict__ seg_start) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_seg) return;
    seg_start[t] = lower_bound_u64(keys, 0, n, (uint64_t)t << pk.seg_shift());
}

// row of every sweep position (see header, step 3).  The same pass unpacks the composite sweep keys
// (ckL = seg|left, ckR = seg|right), takes the 64-wide maxima of ckR and clears the small accumulators
// of the neighbour kernels -- three launches and two memsets less on a latency-bound pipeline.
__global__ void __launch_bounds__(256) rank_rows_kernel(const uint64_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ idx, int64_t n, Pack pk,
                                                        const int64_t* __restrict__ seg_start /* may be null */,
                                                        uint32_t* __restrict__ srow, int32_t* __restrict__ row_of,
                                                        uint64_t* __restrict__ ckL, uint64_t* __restrict__ ckR,
                                                        uint64_t* __restrict__ bmax,
                                                        unsigned long long* __restrict__ reach_slots,
                                                        int64_t* __restrict__ deg_tail) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < 1024) reach_slots[p] = 0ull;
    if (p == 0) *deg_tail = 0;
    {
        uint64_t r = 0ull;
        if (p < n) {
            const uint64_t kk = keys[p];
            const uint64_t sg = kk >> pk.seg_shift();
            const uint64_t l = ((kk >> pk.bN) & ((1ull << pk.bL) - 1ull)) + pk.min_left;
            r = (sg << 32) | (l + (kk & ((1ull << pk.bN) - 1ull)));
            ckL[p] = (sg << 32) | l;
            ckR[p] = r;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t y = __shfl_xor(r, o);
            r = y > r ? y : r;
        }
        if ((threadIdx.x & 63) == 0 && p < n) bmax[p >> 6] = r;
    }
    if (p >= n) return;
    const uint64_t k = keys[p];
    const int sh = pk.seg_shift();
    const uint64_t seg = k >> sh;
    const uint64_t pos_key = k & ((1ull << sh) - 1ull);
    const uint64_t other = seg ^ 1ull;
    int64_t own_start, oth_start, oth_end;
    if (seg_start) {
        own_start = seg_start[seg]; oth_start = seg_start[other]; oth_end = seg_start[other + 1];
    } else {
        own_start = lower_bound_u64(keys, 0, n, seg << sh);
        oth_start = lower_bound_u64(keys, 0, n, other << sh);
        oth_end = lower_bound_u64(keys, oth_start, n, (other + 1) << sh);
    }
    // '+' (strand 0) precedes '-' on equal (left, right): '+' counts strictly smaller, '-' counts <=
    const uint64_t target = (other << sh) | pos_key;
    const int64_t lb = lower_bound_u64(keys, oth_start, oth_end, (seg & 1ull) ? target + 1 : target);
    const int64_t chrom_start = own_start < oth_start ? own_start : oth_start;
    const uint32_t row = (uint32_t)(chrom_start + (p - own_start) + (lb - oth_start));
    srow[p] = row;
    row_of[idx[p]] = (int32_t)row;
}

// ------------------------------------------------------------------ generic (two sorts) path
__global__ void __launch_bounds__(256) build_keys_kernel(const int32_t* __restrict__ left,
                                                         const int32_t* __restrict__ right,
                                                         const int8_t* __restrict__ strand, int64_t n,
                                                         uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((uint64_t)(uint32_t)left[i] << 32) | ((uint64_t)(uint32_t)right[i] << 1) | (uint64_t)(strand[i] & 1);
    idx[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) gather_chrom_kernel(const int32_t* __restrict__ chrom,
                                                           const uint32_t* __restrict__ idx, int64_t n,
                                                           uint64_t* __restrict__ key2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key2[i] = (uint64_t)(uint32_t)chrom[idx[i]];
}

// rows are in (chrom,left,right,strand) order: perm[r] = input index of row r
__global__ void __launch_bounds__(256) rows_kernel(const uint32_t* __restrict__ perm,
                                                   const int32_t* __restrict__ chrom,
                                                   const int32_t* __restrict__ left,
                                                   const int32_t* __restrict__ right,
                                                   const int8_t* __restrict__ strand, int64_t n,
                                                   int32_t* __restrict__ row_of, uint64_t* __restrict__ key3,
                                                   uint32_t* __restrict__ rid, int32_t* __restrict__ rowL,
                                                   int32_t* __restrict__ rowR) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t i = perm[r];
    row_of[i] = (int32_t)r;
    key3[r] = ((uint64_t)(uint32_t)chrom[i] << 1) | (uint64_t)(strand[i] & 1);
    rid[r] = (uint32_t)r;
    rowL[r] = left[i];
    rowR[r] = right[i];
}

// sweep order: srow[p] = row at sweep position p, seg[p] = its (chrom<<1|strand)
__global__ void __launch_bounds__(256) sweep_keys_kernel(const uint64_t* __restrict__ seg,
                                                         const uint32_t* __restrict__ srow,
                                                         const int32_t* __restrict__ rowL,
                                                         const int32_t* __restrict__ rowR, int64_t n,
                                                         uint64_t* __restrict__ ckL, uint64_t* __restrict__ ckR) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t r = srow[p];
    const uint64_t s = seg[p] << 32;
    ckL[p] = s | (uint32_t)rowL[r];
    ckR[p] = s | (uint32_t)rowR[r];
}

// ------------------------------------------------------------------ neighbour lists
__global__ void __launch_bounds__(64) blockmax_kernel(const uint64_t* __restrict__ ckR, int64_t n,
                                                      uint64_t* __restrict__ bmax) {
    const int64_t p = (int64_t)blockIdx.x * 64 + threadIdx.x;
    uint64_t v = p < n ? ckR[p] : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t y = __shfl_xor(v, o);
        v = y > v ? y : v;
    }
    if (threadIdx.x == 0) bmax[blockIdx.x] = v;
}

__device__ __forceinline__ int64_t upper_bound_from(const uint64_t* __restrict__ ck, int64_t p, int64_t n,
                                                    uint64_t target) {
    // first q > p with ck[q] > target (ck sorted non-decreasing); gallop then bisect
    int64_t lo = p + 1, step = 1;
    int64_t hi = lo;
    while (hi < n && ck[hi] <= target) { lo = hi + 1; hi += step; step <<= 1; }
    if (hi > n) hi = n;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (ck[mid] <= target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <bool FILL>
__global__ void __launch_bounds__(256) neighbours_kernel(const uint64_t* __restrict__ ckL,
                                                         const uint64_t* __restrict__ ckR,
                                                         const uint64_t* __restrict__ pmax,
                                                         const uint64_t* __restrict__ bmax,
                                                         const uint32_t* __restrict__ srow, int64_t n,
                                                         int64_t* __restrict__ deg /* by row (count pass) */,
                                                         const int64_t* __restrict__ row_ptr,
                                                         int32_t* __restrict__ col,
                                                         unsigned long long* __restrict__ reach_out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned reach = 0;
    if (p < n) {
        const uint64_t seg_hi = ckL[p] & 0xffffffff00000000ull;
        const uint64_t tgt_r = seg_hi | (ckR[p] & 0xffffffffull);  // later: ckL[q] <= seg|right_p
        const uint64_t tgt_l = ckL[p];                             // earlier: ckR[q] >= seg|left_p
        const uint32_t row = srow[p];
        const int64_t ub = upper_bound_from(ckL, p, n, tgt_r);
        int64_t cnt = 0;
        int32_t* out = FILL ? col + row_ptr[row] : nullptr;
        // earlier neighbours, most recent first
        int64_t q = p - 1;
        int64_t far = p;   // farthest earlier neighbour
        while (q >= 0 && pmax[q] >= tgt_l) {
            if ((q & 63) == 63 && bmax[q >> 6] < tgt_l) { q -= 64; continue; }
            if (ckR[q] >= tgt_l) {
                if (FILL) out[cnt] = (int32_t)srow[q];
                far = q;
                ++cnt;
            }
            --q;
        }
        if (FILL) {
            for (int64_t t = p + 1; t < ub; ++t) out[cnt++] = (int32_t)srow[t];
        } else {
            deg[row] = cnt + (ub - p - 1);
            // row order restricted to one (chrom,strand) segment equals sweep order, so the
            // extreme neighbours in sweep order are the extreme ones in row order
            if (far < p) reach = row - srow[far];
            if (ub - 1 > p) reach = max(reach, srow[ub - 1] - row);
        }
    }
    if (!FILL) {
        // one write per block into a per-slot maximum (1024 slots): no same-address atomic storm
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) reach = max(reach, (unsigned)__shfl_xor((int)reach, o));
        __shared__ unsigned wreach[4];
        if ((threadIdx.x & 63) == 0) wreach[threadIdx.x >> 6] = reach;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned r = max(max(wreach[0], wreach[1]), max(wreach[2], wreach[3]));
            if (r) atomicMax(&reach_out[blockIdx.x & 1023], (unsigned long long)r);
        }
    }
}

inline unsigned grid_for(int64_t n, int threads) { return (unsigned)sd_ceil_div(n, threads); }

inline int bits_for(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

}  // namespace

// The generic path (any n < 2^31, any field widths); sdice_cluster_dev (cluster_fast.hip) takes it
// for more than 8 M junctions, when asked to by parameter and as its fallback.
int sd_cluster_legacy(sdice_ctx* ctx, int64_t n, const int32_t* d_chrom, const int32_t* d_left, const int32_t* d_right,
                      const int8_t* d_strand, int32_t* d_row_of, int64_t* d_row_ptr, int64_t* nnz_out) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && n < ((int64_t)1 << 31), "n out of range");
    SD_HIP(hipSetDevice(ctx->device));
    ctx->nnz = 0;
    ctx->cluster_reach = 0;
    if (nnz_out) *nnz_out = 0;
    if (n == 0) {
        if (d_row_ptr) SD_HIP(hipMemsetAsync(d_row_ptr, 0, 8, ctx->stream));
        return SDICE_OK;
    }
    SD_ARG(d_chrom && d_left && d_right && d_strand && d_row_of && d_row_ptr, "NULL pointer");
    // scratch: 3 x u64 + 3 x u32 keys/values, degrees, generic-path row copies, block maxima, scan
    // sums, segment table, radix histograms
    SD_TRY(ctx->arena.reserve((size_t)n * 60 + ((size_t)1 << 24) + (1 << 20), ctx->stream));
    Arena& A = ctx->arena;
    const size_t N = (size_t)n;
    uint64_t* kA = (uint64_t*)A.alloc(N * 8);
    uint64_t* kB = (uint64_t*)A.alloc(N * 8);
    uint64_t* kC = (uint64_t*)A.alloc(N * 8);
    uint32_t* vA = (uint32_t*)A.alloc(N * 4);
    uint32_t* vB = (uint32_t*)A.alloc(N * 4);
    uint32_t* vC = (uint32_t*)A.alloc(N * 4);
    int64_t* deg = (int64_t*)A.alloc((N + 1) * 8);
    unsigned long long* red = (unsigned long long*)A.alloc(RED_SLOTS * RED_WORDS * 8);
    const int64_t nb64 = sd_ceil_div(n, 64);
    uint64_t* bmax = (uint64_t*)A.alloc((size_t)nb64 * 8);
    unsigned long long* reach_slots = (unsigned long long*)A.alloc(1024 * 8);
    if (!kA || !kB || !kC || !vA || !vB || !vC || !deg || !red || !bmax || !reach_slots) return SDICE_ERR_NOMEM;

    // ---- field ranges (and validation) in one pass, one read-back
    // every block ends with 8 atomics; RED_SLOTS copies of the accumulators (one 128-B line each,
    // block b uses slot b % RED_SLOTS) keep them from queueing on one address; the host folds the slots
    int64_t* hp = ctx->h_pinned;
    for (int s = 0; s < RED_SLOTS; ++s) {
        int64_t* q = hp + s * RED_WORDS;
        for (int i = 0; i < RED_WORDS; ++i) q[i] = 0;
        q[1] = -1; q[3] = -1; q[6] = -1;   // AND / min identities
    }
    SD_HIP(hipMemcpyAsync(red, hp, RED_SLOTS * RED_WORDS * 8, hipMemcpyHostToDevice, ctx->stream));
    {
        int64_t blocks = sd_ceil_div(n, 256 * 4);
        if (blocks > 1024) blocks = 1024;
        SD_LAUNCH(ctx, "reduce_fields_kernel", reduce_fields_kernel, dim3((unsigned)blocks), dim3(256), 0, d_chrom, d_left,
                  d_right, d_strand, n, red);
    }
    int64_t* hr = hp + RED_SLOTS * RED_WORDS;
    SD_HIP(hipMemcpyAsync(hr, red, RED_SLOTS * RED_WORDS * 8, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t f[RED_WORDS];
    for (int i = 0; i < RED_WORDS; ++i) f[i] = (uint64_t)hr[i];
    for (int s = 1; s < RED_SLOTS; ++s) {
        const uint64_t* q = (const uint64_t*)(hr + s * RED_WORDS);
        f[0] |= q[0]; f[1] &= q[1]; f[2] |= q[2]; f[3] &= q[3]; f[4] |= q[4];
        f[6] = q[6] < f[6] ? q[6] : f[6]; f[7] = q[7] > f[7] ? q[7] : f[7]; f[8] = q[8] > f[8] ? q[8] : f[8];
    }
    if (f[4]) {
        sdice_set_error("sdice_cluster: invalid junction (need 0 <= left <= right, chrom_rank >= 0, strand in {0,1})");
        return SDICE_ERR_ARG;
    }
    const uint64_t maskQ = f[0] ^ f[1];
    const uint64_t maskC = (f[2] ^ f[3]) & 0xffffffffull;
    const uint64_t max_chrom = f[2];   // the OR of all chrom ranks bounds the maximum from above
    const uint64_t min_left = f[6], max_left = f[7], max_len = f[8];

    Pack pk;
    pk.bN = bits_for(max_len);
    pk.bL = bits_for(max_left - min_left);
    pk.min_left = (uint32_t)min_left;
    const int bS = bits_for((max_chrom << 1) | 1ull);
    // 62: (n_seg + 1) << seg_shift must not overflow in the segment table
    const bool packed = (pk.bN + pk.bL + bS <= 62) && !ctx->param("cluster.generic", 0);

    uint32_t* srow;
    uint64_t *ckL, *ckR, *pmax;
    if (packed) {
        SD_LAUNCH(ctx, "pack_keys_kernel", pack_keys_kernel, dim3(grid_for(n, 256)), dim3(256), 0, d_chrom, d_left, d_right,
                  d_strand, n, pk, kA, vA);
        const int total_bits = pk.bN + pk.bL + bS;
        const uint64_t mask = total_bits >= 64 ? ~0ull : ((1ull << total_bits) - 1ull);
        SD_TRY(sd_radix_sort_pairs(ctx, n, kA, vA, kB, vB, kC, vC, mask));   // -> kB (keys), vB (input index)
        ckL = kA; ckR = kC;
        const int64_t n_seg = (int64_t)(((max_chrom << 1) | 1ull) + 1ull);   // seg values are < n_seg
        int64_t* seg_start = nullptr;
        if (n_seg <= (1 << 20)) {
            seg_start = (int64_t*)A.alloc((size_t)(n_seg + 2) * 8);
            if (!seg_start) return SDICE_ERR_NOMEM;
            SD_LAUNCH(ctx, "seg_table_kernel", seg_table_kernel, dim3(grid_for(n_seg + 2, 256)), dim3(256), 0, kB, n, pk,
                      n_seg + 1, seg_start);
        }
        srow = vA;
        SD_LAUNCH(ctx, "rank_rows_kernel", rank_rows_kernel, dim3(grid_for(n > 1024 ? n : 1024, 256)), dim3(256), 0, kB, vB,
                  n, pk, (const int64_t*)seg_start, srow, d_row_of, ckL, ckR, bmax, reach_slots, deg + n);
        pmax = kB;   // the sorted keys are dead after rank_rows (stream order)
    } else {
        int32_t* rowL = (int32_t*)A.alloc(N * 4);
        int32_t* rowR = (int32_t*)A.alloc(N * 4);
        if (!rowL || !rowR) return SDICE_ERR_NOMEM;
        SD_LAUNCH(ctx, "build_keys_kernel", build_keys_kernel, dim3(grid_for(n, 256)), dim3(256), 0, d_left, d_right,
                  d_strand, n, kA, vA);
        // row order (chrom, left, right, strand): sort by key, then stably by chrom
        SD_TRY(sd_radix_sort_pairs(ctx, n, kA, vA, kB, vB, kC, vC, maskQ));   // -> kB, vB
        const uint32_t* perm = vB;
        if (maskC) {
            SD_LAUNCH(ctx, "gather_chrom_kernel", gather_chrom_kernel, dim3(grid_for(n, 256)), dim3(256), 0, d_chrom, vB, n,
                      kA);
            SD_TRY(sd_radix_sort_pairs(ctx, n, kA, vB, kC, vA, kB, vC, maskC));   // -> kC, vA
            perm = vA;
        }
        // per-row data, then sweep order (chrom, strand, left, right) by one more stable sort
        uint64_t* key3 = kA;
        uint32_t* rid = (perm == vA) ? vB : vA;
        SD_LAUNCH(ctx, "rows_kernel", rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, perm, d_chrom, d_left, d_right,
                  d_strand, n, d_row_of, key3, rid, rowL, rowR);
        const uint64_t mask3 = (maskC << 1) | (maskQ & 1ull);
        srow = (rid == vA) ? vB : vA;   // perm no longer needed after rows_kernel
        SD_TRY(sd_radix_sort_pairs(ctx, n, key3, rid, kB, srow, kC, vC, mask3));   // -> kB (seg), srow
        ckL = kA; ckR = kC;
        SD_LAUNCH(ctx, "sweep_keys_kernel", sweep_keys_kernel, dim3(grid_for(n, 256)), dim3(256), 0, kB, srow, rowL, rowR, n,
                  ckL, ckR);
        pmax = kB;  // seg dead after sweep_keys
    }
    SD_TRY(sd_inclusive_max_scan_u64(ctx, n, ckR, pmax));

    // ---- degrees -> row_ptr
    if (!packed) {      // (the packed path did these three inside rank_rows_kernel)
        SD_LAUNCH(ctx, "blockmax_kernel", blockmax_kernel, dim3((unsigned)nb64), dim3(64), 0, ckR, n, bmax);
        SD_HIP(hipMemsetAsync(reach_slots, 0, 1024 * 8, ctx->stream));
        SD_HIP(hipMemsetAsync(deg + n, 0, 8, ctx->stream));
    }
    SD_LAUNCH(ctx, "neighbours_count_kernel", (neighbours_kernel<false>), dim3(grid_for(n, 256)), dim3(256), 0, ckL, ckR,
              pmax, bmax, srow, n, deg, (const int64_t*)nullptr, (int32_t*)nullptr, reach_slots);
    SD_TRY(sd_exclusive_scan_i64(ctx, n + 1, deg, d_row_ptr, nullptr));
    SD_HIP(hipMemcpyAsync(hp + 32, d_row_ptr + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipMemcpyAsync(hp + 64, reach_slots, 1024 * 8, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t nnz = hp[32];
    int64_t reach = 0;
    for (int i = 0; i < 1024; ++i) reach = hp[64 + i] > reach ? hp[64 + i] : reach;
    ctx->cluster_reach = (int)reach;
    SD_TRY(sd_cluster_check_nnz(ctx, nnz, false));
    if (nnz > ctx->col_cap) {
        if (ctx->d_col) (void)hipFree(ctx->d_col);
        ctx->d_col = nullptr;
        ctx->col_cap = 0;
        const int64_t cap = nnz + nnz / 8 + 1024;
        hipError_t e = hipMalloc((void**)&ctx->d_col, (size_t)cap * 4);
        if (e != hipSuccess) {
            sdice_set_error("sdice_cluster: hipMalloc of %lld neighbour indices failed: %s", (long long)cap,
                            hipGetErrorString(e));
            return SDICE_ERR_NOMEM;
        }
        ctx->col_cap = cap;
    }
    if (nnz > 0)
        SD_LAUNCH(ctx, "neighbours_fill_kernel", (neighbours_kernel<true>), dim3(grid_for(n, 256)), dim3(256), 0, ckL, ckR,
                  pmax, bmax, srow, n, (int64_t*)nullptr, (const int64_t*)d_row_ptr, ctx->d_col,
                  (unsigned long long*)nullptr);
    ctx->nnz = nnz;
    if (nnz_out) *nnz_out = nnz;
    return SDICE_OK;
}

int sd_cluster_check_nnz(sdice_ctx* ctx, int64_t nnz, bool host) {
    int64_t cap = ctx->param("cluster.max_nnz", 0);
    const char* what = "param cluster.max_nnz";
    if (cap <= 0) {
        size_t free_b = 0, total_b = 0;
        SD_HIP(hipMemGetInfo(&free_b, &total_b));
        const int64_t held = ctx->col_cap * 4;                       // (the buffer it would replace)
        cap = (int64_t)((double)(free_b + (size_t)held) * 0.9) / 4;
        what = "free device memory";
        if (host) {
            const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
            const int64_t hcap = pages > 0 && psz > 0 ? (int64_t)pages * psz / 2 / 4 : cap;
            if (hcap < cap) { cap = hcap; what = "half of the host memory"; }
        }
    }
    if (nnz > cap) {
        sdice_set_error("sdice_cluster: the neighbour list has %lld entries (%.2f GB), more than %s allows (%lld entries); "
                        "the junction set is too dense to materialise its overlap lists",
                        (long long)nnz, (double)nnz * 4e-9, what, (long long)cap);
        return SDICE_ERR_NOMEM;
    }
    return SDICE_OK;
}

extern "C" int sdice_cluster_col_dev(sdice_ctx* ctx, const int32_t** d_col, int64_t* nnz) {
    SD_ARG(ctx && d_col, "bad arguments");
    *d_col = ctx->d_col;
    if (nnz) SD_TRY(sd_cluster_resolve(ctx));      // (an asynchronous clustering learns its nnz here)
    if (nnz) *nnz = ctx->nnz;
    return SDICE_OK;
}

extern "C" int sdice_cluster(sdice_ctx* ctx, int64_t n, const int32_t* chrom_rank, const int32_t* left,
                             const int32_t* right, const int8_t* strand, int32_t* row_of, int64_t* row_ptr,
                             int64_t* nnz) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && n < ((int64_t)1 << 31), "n out of range");
    SD_ARG(row_ptr, "row_ptr is NULL");
    if (n == 0) { row_ptr[0] = 0; if (nnz) *nnz = 0; ctx->nnz = 0; return SDICE_OK; }
    SD_ARG(chrom_rank && left && right && strand && row_of, "NULL pointer");
    int32_t *dc = nullptr, *dl = nullptr, *dr = nullptr, *drow = nullptr;
    int8_t* ds = nullptr;
    int64_t* drp = nullptr;
    int rc = sdice_dmalloc(ctx, n * 4, (void**)&dc);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * 4, (void**)&dl);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * 4, (void**)&dr);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n, (void**)&ds);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * 4, (void**)&drow);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, (n + 1) * 8, (void**)&drp);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, dc, chrom_rank, n * 4);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, dl, left, n * 4);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, dr, right, n * 4);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, ds, strand, n);
    int64_t z = 0;
    if (rc == SDICE_OK) rc = sdice_cluster_dev(ctx, n, dc, dl, dr, ds, drow, drp, &z);
    if (rc == SDICE_OK) rc = sd_cluster_check_nnz(ctx, z, true);        // (the caller is about to hold the list on the host)
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, row_of, drow, n * 4);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, row_ptr, drp, (n + 1) * 8);
    if (rc == SDICE_OK && nnz) *nnz = z;
    sdice_dfree(ctx, dc); sdice_dfree(ctx, dl); sdice_dfree(ctx, dr); sdice_dfree(ctx, ds);
    sdice_dfree(ctx, drow); sdice_dfree(ctx, drp);
    return rc;
}

extern "C" int sdice_cluster_col(sdice_ctx* ctx, int32_t* col, int64_t capacity) {
    SD_ARG(ctx, "ctx is NULL");
    SD_TRY(sd_cluster_resolve(ctx));
    SD_ARG(capacity >= ctx->nnz, "capacity smaller than nnz");
    if (ctx->nnz == 0) return SDICE_OK;
    SD_ARG(col, "col is NULL");
    return sdice_d2h(ctx, col, ctx->d_col, ctx->nnz * 4);
}
