// K7 for MANY columns: Benjamini-Hochberg down each of `segs` columns of m p-values of a row-major table
// (the per-pair-column mode of `pairwise`, pairwise_fisher.py:187-191), in place.
//
// The generic path (bh.hip) sorts (p, index) pairs with eight 8-bit LSD radix passes: ~150 B of HBM
// traffic per p-value.  BH does not need a stable full sort, though -- only, per value, its rank and the
// running minimum of p*m/rank from the top rank down -- and it needs no order inside a tie group (all
// members of a tie group end with the group's last, smallest p*m/rank).  So here, per column:
//   1. sample:   8 jittered regular samples per bucket, read straight from the row-major table, sorted in
//                LDS as (p bits, index) pairs -- the index breaks ties, so a column of ONE repeated value
//                (Fisher p-values are discrete, p = 1 is common) still splits evenly; every 8th is a splitter;
//   2. transpose + count: a strip of 16 columns goes row-major -> column-major through LDS (full 128-byte
//                lines in, 512-byte runs out) and every value is classified on the way (binary search over
//                its column's splitters in LDS, LDS histogram, one global atomic per (block, column, bucket));
//   3. scan:     bucket starts per column (buckets are contiguous rank ranges);
//   4. scatter:  tile of 4096 values of one column, the search again, one global atomic per (tile, bucket)
//                reserves a run; the 8-byte KEYS land in their bucket, and every value's (bucket, slot) goes to
//                pos[column][row] -- a coalesced 4-byte store.  No index travels with the keys;
//   5. buckets:  ONE WAVE per bucket (mean ~200 values): key-only bitonic network on registers (DPP and lane
//                swaps, no LDS crossbar), ranks = bucket start + position, p*m/rank exactly as the generic path
//                computes it, suffix minimum inside the bucket; sorted keys and minima parked in the wave's LDS,
//                every value finds its rank by binary search there and its result is written back IN PLACE,
//                to the slot its key came from (a coalesced 8-byte store); bucket minimum to a small table;
//   6. suffix minima of the bucket minima per column;
//   7. finish:   the transpose back to the row-major table GATHERS: value (row, column) reads pos, takes its
//                result from the slot, applies min(own, later buckets' minimum, 1).
// Nothing is scattered at 8-byte granularity (rounds 1-3 scattered the results to [column][row] from the bucket
// kernel: 27 GB written for 4 GB of results at 25 000 x 19 900); the one random access left is step 7's 8-byte
// read inside a column's 200 KB of results, which the block -> XCD mapping keeps in one L2.  ~36 B of
// algorithmic traffic per value; bit-identical to the generic path.
#include "common.h"
#include <algorithm>
#include <stdio.h>

namespace {

constexpr int TILE_T = 512;         // threads of a scatter tile
constexpr int TILE_E = 8;           // values per thread
constexpr int TILE = TILE_T * TILE_E;
constexpr int MAX_B = 1024;
constexpr int POS_SHIFT = 18;       // pos word: bucket << 18 | slot inside the column (m <= 2^18, buckets <= 2^10)
constexpr unsigned POS_MASK = (1u << POS_SHIFT) - 1u;

struct BhsArgs {
    const double* p_rm;     // row-major input: value (r, c) at p_rm[r * pitch + c]
    int64_t pitch;
    double* p_cm;           // [segs][m] transposed input
    int64_t m;
    int segs;
    int B;                  // buckets per segment
    int spb;                // samples per bucket
    int S, S2;              // samples per segment, next power of two
    uint64_t* spl_k;        // [segs][B] (B-1 used)
    uint32_t* spl_i;
    unsigned* gcount;       // [segs][B]
    unsigned* start;        // [segs][B+1]
    unsigned* cursor;       // [segs][B]
    uint64_t* keyS;         // [segs][m] bucketed keys; the bucket kernel overwrites every key with its result
    uint32_t* pos_cm;       // [segs][m] bucket << 18 | slot of each value (NULL: one bucket, slot = row)
    uint64_t* bmin;         // [segs][B]
    uint64_t* sfx;          // [segs][B] minimum over the LATER buckets
    unsigned* big_count;    // buckets beyond the main kernel's network: work list of the second bucket kernel
    int64_t* big_list;
    uint64_t* spill;        // [segs * m] scratch of the in-HBM bucket path (adversarial input only), bump-allocated
    unsigned long long* spill_n;
    int reg_cap;            // buckets beyond this many values take the in-HBM path (2048; lower in tests)
};

__device__ __forceinline__ uint64_t key_of(double v) {
    if (v == 0.0) v = 0.0;   // -0.0 -> +0.0
    return (uint64_t)__double_as_longlong(v);
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// c += (ka, ia) < (kb, ib) in composite order: the borrow of the 96-bit subtraction, added with carry -- four VALU
// instructions (the compiler turns "ka < kb || (ka == kb && ia < ib)", and __builtin_subc chains too, into five compares
// and their mask arithmetic)
__device__ __forceinline__ void count_less96(unsigned& c, uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
    unsigned t;
    asm("v_sub_co_u32 %0, vcc, %2, %3\n\t"
        "v_subb_co_u32 %0, vcc, %4, %5, vcc\n\t"
        "v_subb_co_u32 %0, vcc, %6, %7, vcc\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(t), "+v"(c)
        : "v"(ia), "v"(ib), "v"((unsigned)ka), "v"((unsigned)kb), "v"((unsigned)(ka >> 32)), "v"((unsigned)(kb >> 32))
        : "vcc");
}

// number of splitters <= (k, i) in composite order
__device__ __forceinline__ int find_bucket(const uint64_t* sk, const uint32_t* si, int ns, uint64_t k, uint32_t i) {
    int lo = 0, hi = ns;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const uint64_t s = sk[mid];
        const bool le = s < k || (s == k && si[mid] <= i);
        lo = le ? mid + 1 : lo;
        hi = le ? hi : mid;
    }
    return lo;
}
// ---------------------------------------------------------------- 1. splitters
__global__ void __launch_bounds__(256) bhs_sample_kernel(BhsArgs a) {
    extern __shared__ uint64_t smem_s[];
    uint64_t* sk = smem_s;
    uint32_t* si = reinterpret_cast<uint32_t*>(sk + a.S2);
    const int seg = blockIdx.x, tid = threadIdx.x;
    const double* p = a.p_rm + seg;                      // column `seg` of the row-major table
    for (int j = tid; j < a.S2; j += 256) {
        uint64_t k = ~0ull;
        uint32_t i = ~0u;
        if (j < a.S) {
            const int64_t lo = (int64_t)j * a.m / a.S, hi = (int64_t)(j + 1) * a.m / a.S;
            const int64_t pos = lo + (int64_t)(hash32((unsigned)seg * 40503u + (unsigned)j) % (unsigned)(hi - lo));
            k = key_of(p[pos * a.pitch]);
            i = (uint32_t)pos;
        }
        sk[j] = k;
        si[j] = i;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= a.S2; k2 <<= 1) {
        for (int j = k2 >> 1; j >= 1; j >>= 1) {
            for (int t = tid; t < (a.S2 >> 1); t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = (i & k2) == 0;
                const uint64_t ka = sk[i], kb = sk[l];
                const uint32_t ia = si[i], ib = si[l];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == up) { sk[i] = kb; sk[l] = ka; si[i] = ib; si[l] = ia; }
            }
            __syncthreads();
        }
    }
    for (int b = tid + 1; b < a.B; b += 256) {
        a.spl_k[(int64_t)seg * a.B + b - 1] = sk[a.spb * b];
        a.spl_i[(int64_t)seg * a.B + b - 1] = si[a.spb * b];
    }
}

// ---------------------------------------------------------------- 2. transpose in (+ count)
// A workgroup owns a strip of TC_COLS columns x rows_per_block rows: row segments of 128 bytes in, runs of TC_ROWS
// doubles per column out.  COUNT: every value is classified against its column's splitters (LDS) on the way; the
// histogram stays in LDS until the workgroup is through its rows.
constexpr int TC_COLS = 16, TC_ROWS = 128, TC_T = 256;
// 1-D grid, workgroups dealt round-robin over the 8 XCDs: XCD x takes the strips [x spx, (x + 1) spx), neighbouring
// strips of one row block back to back -- a 128-byte row piece starts wherever (row x pitch) mod 128 puts it, so every
// piece shares its first and last line with the neighbouring strips, and those lines should meet in ONE L2.
template <bool COUNT>
__global__ void __launch_bounds__(TC_T) bhs_transpose_kernel(BhsArgs a, int rows_per_block, int strips_per_xcd) {
    extern __shared__ uint64_t smem_tc[];
    __shared__ double tile[TC_COLS][TC_ROWS + 1];
    const int ns = a.B - 1;
    uint64_t* sk = smem_tc;                                               // [TC_COLS][ns]
    uint32_t* si = reinterpret_cast<uint32_t*>(sk + (COUNT ? TC_COLS * ns : 0));    // [TC_COLS][ns]
    unsigned* hist = reinterpret_cast<unsigned*>(si + (COUNT ? TC_COLS * ns : 0));  // [TC_COLS][B]

    const int tid = threadIdx.x;
    const int64_t kb = (int64_t)(blockIdx.x >> 3);
    const int64_t c0 = ((int64_t)(blockIdx.x & 7) * strips_per_xcd + kb % strips_per_xcd) * TC_COLS;
    if (c0 >= a.segs) return;
    const int ncol = (int)min((int64_t)TC_COLS, (int64_t)a.segs - c0);
    const int64_t rb0 = (kb / strips_per_xcd) * rows_per_block, rb1 = min(a.m, rb0 + rows_per_block);
    if (COUNT) {
        for (int i = tid; i < ncol * ns; i += TC_T) {
            const int c = i / ns, b = i - c * ns;
            sk[c * ns + b] = a.spl_k[(c0 + c) * a.B + b];
            si[c * ns + b] = a.spl_i[(c0 + c) * a.B + b];
        }
        for (int i = tid; i < TC_COLS * a.B; i += TC_T) hist[i] = 0u;
        __syncthreads();
    }
    const int col = tid & (TC_COLS - 1), rsub = tid >> 4;                 // 16 rows x 16 columns per pass
    constexpr int PASSES = TC_ROWS / (TC_T / TC_COLS);                    // 8
    for (int64_t r0 = rb0; r0 < rb1; r0 += TC_ROWS) {
        double v[PASSES];
#pragma unroll
        for (int j = 0; j < PASSES; ++j) {
            const int64_t row = r0 + rsub + 16 * j;
            v[j] = (row < rb1 && col < ncol) ? __builtin_nontemporal_load(a.p_rm + row * a.pitch + c0 + col) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < PASSES; ++j) {
            const int64_t row = r0 + rsub + 16 * j;
            tile[col][rsub + 16 * j] = v[j];
            if (COUNT && row < rb1 && col < ncol) {
                const int bkt = find_bucket(sk + col * ns, si + col * ns, ns, key_of(v[j]), (uint32_t)row);
                atomicAdd(&hist[col * a.B + bkt], 1u);
            }
        }
        __syncthreads();
        const int nrow = (int)min((int64_t)TC_ROWS, rb1 - r0);
        for (int i = tid; i < TC_COLS * TC_ROWS; i += TC_T) {
            const int c = i / TC_ROWS, rr = i - c * TC_ROWS;
            if (c < ncol && rr < nrow) a.p_cm[(c0 + c) * a.m + r0 + rr] = tile[c][rr];
        }
        __syncthreads();
    }
    if (COUNT) {
        for (int i = tid; i < ncol * a.B; i += TC_T)
            if (hist[i]) atomicAdd(&a.gcount[c0 * a.B + i], hist[i]);         // (hist is [column][B], as gcount)
    }
}

// ---------------------------------------------------------------- 2b. / 4. count (many buckets: the splitters of a
// strip do not fit LDS beside the transpose) and scatter
template <bool SCATTER>
__global__ void __launch_bounds__(TILE_T) bhs_tile_kernel(BhsArgs a, int tiles) {
    extern __shared__ uint64_t smem_t[];
    uint64_t* sk = smem_t;                                            // [B]
    uint32_t* si = reinterpret_cast<uint32_t*>(sk + a.B);             // [B]
    unsigned* hist = reinterpret_cast<unsigned*>(si + a.B);           // [B]
    unsigned* base = hist + a.B;                                      // [B]
    // 1-D grid, workgroups dealt round-robin over the 8 XCDs: all tiles of a segment go to ONE XCD (segment mod 8), so
    // that the runs a bucket receives from the segment's tiles -- their ends share cache lines -- meet in one L2
    const int64_t kk = (int64_t)(blockIdx.x >> 3);
    const int64_t seg64 = (kk / tiles) * 8 + (blockIdx.x & 7);
    if (seg64 >= a.segs) return;
    const int seg = (int)seg64, tile_x = (int)(kk % tiles), tid = threadIdx.x;
    const int ns = a.B - 1;
    for (int b = tid; b < a.B; b += TILE_T) {
        hist[b] = 0;
        if (b < ns) { sk[b] = a.spl_k[(int64_t)seg * a.B + b]; si[b] = a.spl_i[(int64_t)seg * a.B + b]; }
    }
    __syncthreads();
    const int64_t e0 = (int64_t)tile_x * TILE;
    const double* p = a.p_cm + (int64_t)seg * a.m;
    uint64_t key[TILE_E];
    int bkt[TILE_E];
    unsigned off[TILE_E];
#pragma unroll
    for (int q = 0; q < TILE_E; ++q) {
        const int64_t e = e0 + q * TILE_T + tid;
        key[q] = e < a.m ? key_of(__builtin_nontemporal_load(p + e)) : 0;
    }
#pragma unroll
    for (int q = 0; q < TILE_E; ++q) {
        const int64_t e = e0 + q * TILE_T + tid;
        bkt[q] = -1;
        if (e < a.m) {
            bkt[q] = find_bucket(sk, si, ns, key[q], (uint32_t)e);
            off[q] = atomicAdd(&hist[bkt[q]], 1u);
        }
    }
    __syncthreads();
    if (!SCATTER) {
        for (int b = tid; b < a.B; b += TILE_T)
            if (hist[b]) atomicAdd(&a.gcount[(int64_t)seg * a.B + b], hist[b]);
        return;
    }
    for (int b = tid; b < a.B; b += TILE_T)
        base[b] = hist[b] ? atomicAdd(&a.cursor[(int64_t)seg * a.B + b], hist[b]) : 0u;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TILE_E; ++q) {
        if (bkt[q] >= 0) {
            const unsigned slot = base[bkt[q]] + off[q];
            a.keyS[(int64_t)seg * a.m + slot] = key[q];
            a.pos_cm[(int64_t)seg * a.m + e0 + q * TILE_T + tid] = ((unsigned)bkt[q] << POS_SHIFT) | slot;      // (coalesced)
        }
    }
}

// ---------------------------------------------------------------- 3. bucket starts
__global__ void __launch_bounds__(256) bhs_scan_kernel(BhsArgs a) {
    __shared__ unsigned wsum[4];
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned* c = a.gcount + (int64_t)seg * a.B;
    unsigned v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = tid * 4 + q;
        v[q] = b < a.B ? c[b] : 0u;
        s += v[q];
    }
    unsigned x = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    unsigned pre = x - s;
    for (int k = 0; k < w; ++k) pre += wsum[k];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = tid * 4 + q;
        if (b < a.B) {
            a.start[(int64_t)seg * (a.B + 1) + b] = pre;
            a.cursor[(int64_t)seg * a.B + b] = pre;
        }
        pre += v[q];
    }
    if (tid == 255) a.start[(int64_t)seg * (a.B + 1) + a.B] = pre;
}

// ---------------------------------------------------------------- 5. one wave per bucket
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(v & 0xffffffffu), mask);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), mask);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_down_u64(uint64_t v, int d) {
    const unsigned lo = (unsigned)__shfl_down((int)(unsigned)(v & 0xffffffffu), d);
    const unsigned hi = (unsigned)__shfl_down((int)(unsigned)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t bcast_lane0_u64(uint64_t v) {
    return ((uint64_t)(unsigned)__shfl((int)(unsigned)(v >> 32), 0) << 32) | (unsigned)__shfl((int)(unsigned)(v & 0xffffffffu), 0);
}

__device__ __forceinline__ uint64_t raw_bits(uint64_t key, int64_t rank1, int64_t m) {
    // p_(i) / (i / m), the arithmetic of the generic path (bh.hip bh_raw_kernel)
    const double ps = __longlong_as_double((long long)key);
    const double ecdf = (double)rank1 / (double)m;
    return (uint64_t)__double_as_longlong(ps / ecdf);
}
// the same value with the first division by arithmetic: rank1 and m are integers below 2^53 with rank1 <= m <= 2^18, so
// the quotient lies at least 2^-18 ulp off every rounding boundary, and q0 = rank1 * (1 / m) corrected by its exact
// residual is the correctly rounded quotient (exhaustive over all ranks of 3000 random and 21 chosen column lengths:
// tests/test_abi_and_host.py::test_rank_over_m_by_reciprocal); inv_m = 1.0 / m by the IEEE division, once per wave
__device__ __forceinline__ uint64_t raw_bits_inv(uint64_t key, int rank1, double m, double inv_m) {
    const double ps = __longlong_as_double((long long)key);
    const double a = (double)rank1;
    const double q0 = a * inv_m;
    const double ecdf = __builtin_fma(__builtin_fma(-q0, m, a), inv_m, q0);
    return (uint64_t)__double_as_longlong(ps / ecdf);
}

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint64_t y = shfl_xor_u64(v, o); v = y < v ? y : v; }
    return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint64_t y = shfl_xor_u64(v, o); v = y > v ? y : v; }
    return v;
}

// ---- ranking a bucket by a COUNTING SORT in LDS instead of a comparison network.  The bucket's keys lie between two
// splitters, so (key - lo) >> sh -- sh the shift that brings the bucket's key range below the number of bins; integer
// arithmetic on the bit patterns: monotone for ANY input, linear in the value inside a binade -- spreads them over the
// bins at less than one key per bin.  An LDS atomic per value counts its bin and hands it a slot in it, a scan over the
// bins gives their starts, the keys go to their bin's list, and every value counts the members of its OWN bin that
// come before it in (key, slot) order: rank = bin start + that count (a bin of ONE repeated value -- Fisher's p = 1
// fills whole buckets -- is recognised and ranked by slot alone).  A bin of many DIFFERENT values (values a few ulps
// apart in a wide bucket) degrades to the quadratic count, at worst about the cost of a sorting network.  Then
// p * m / rank as the generic path computes it, results to L[rank], suffix minima over the ranks in place, and every
// value reads its own back: members of a tie group hold distinct ranks in arbitrary order, and since p * m / rank falls
// inside a group, the suffix minimum from any of them is the group's last, i.e. the group's value.
// (Rounds 1-3 sorted (key, index) pairs per bucket with a bitonic network on registers, one wave per bucket; a
// key-only network with DPP / lane-swap exchanges and a wave-per-bucket form of this counting sort were measured on the
// way: DESIGN.md appendix A.4.)

// rare (adversarial input): more values in one bucket than the widest network holds -- its wave sorts a copy in HBM
// (all-ascending bitonic network: out-of-range partners count as +inf and never move), ranks the bucket's values by
// binary search in the copy, turns the copy into suffix minima and puts every value's result into its slot
__device__ void bucket_in_hbm(const BhsArgs& a, uint64_t* ks, int n_b, unsigned start, int lane, uint64_t* bmin_out) {
    unsigned long long so = 0;
    if (lane == 0) so = atomicAdd(a.spill_n, (unsigned long long)n_b);
    uint64_t* t = a.spill + bcast_lane0_u64(so);
    for (int p = lane; p < n_b; p += 64) t[p] = ks[p];
    __threadfence_block();
    int P = 1;
    while (P < n_b) P <<= 1;
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j = k2 >> 1; j >= 1; j >>= 1) {
            for (int q = lane; q < (P >> 1); q += 64) {
                int i, l;
                if (j == (k2 >> 1)) {              // first step of a merge: mirror partner
                    const int blk = q / j, r = q - blk * j;
                    i = blk * k2 + r;
                    l = blk * k2 + k2 - 1 - r;
                } else {
                    i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                    l = i | j;
                }
                if (l < n_b) {
                    const uint64_t ka = t[i], kb = t[l];
                    if (kb < ka) { t[i] = kb; t[l] = ka; }
                }
            }
            __threadfence_block();
        }
    }
    for (int p = lane; p < n_b; p += 64) {         // rank of value p: lower bound in the sorted copy
        const uint64_t k = ks[p];
        int lo = 0, hi = n_b;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (t[mid] < k) lo = mid + 1; else hi = mid; }
        ks[p] = (uint64_t)lo;
    }
    __threadfence_block();
    uint64_t carry = ~0ull;
    for (int c = (n_b - 1) / 64; c >= 0; --c) {
        const int p = c * 64 + lane;
        uint64_t x = p < n_b ? raw_bits(t[p], (int64_t)start + p + 1, a.m) : ~0ull;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = shfl_down_u64(x, o);
            if (lane + o < 64) x = y < x ? y : x;
        }
        x = carry < x ? carry : x;
        if (p < n_b) t[p] = x;
        carry = bcast_lane0_u64(x);
    }
    __threadfence_block();
    for (int p = lane; p < n_b; p += 64) ks[p] = t[ks[p]];
    if (lane == 0) *bmin_out = carry;
}

// what a wave needs to know about bucket g
struct BucketRef {
    int seg, b, n_b;
    unsigned start;
    const double* pd;
    uint64_t* ks;
    uint64_t* bm;
    uint64_t lo, hi;        // bounds of the bucket's keys (the splitters on either side)
    bool end;               // first / last bucket of a column, or the column's only one: bounds from the keys
};
__device__ __forceinline__ BucketRef bucket_ref(const BhsArgs& a, int64_t g) {
    BucketRef r;
    r.seg = (int)(g / a.B);
    r.b = (int)(g - (int64_t)r.seg * a.B);
    const int64_t seg_off = (int64_t)r.seg * a.m;
    r.start = 0;
    r.n_b = (int)a.m;
    r.pd = nullptr;
    r.ks = a.keyS + seg_off;
    r.lo = 0; r.hi = 0;
    r.end = r.b == 0 || r.b == a.B - 1;
    if (a.B == 1) {
        r.pd = a.p_cm + seg_off;                   // short columns: no partition, straight from the p-values
    } else {
        r.start = a.start[(int64_t)r.seg * (a.B + 1) + r.b];
        r.n_b = (int)(a.start[(int64_t)r.seg * (a.B + 1) + r.b + 1] - r.start);
        r.ks += r.start;
        if (!r.end) { r.lo = a.spl_k[(int64_t)r.seg * a.B + r.b - 1]; r.hi = a.spl_k[(int64_t)r.seg * a.B + r.b]; }
    }
    r.bm = a.bmin + (int64_t)r.seg * a.B + r.b;
    return r;
}

// ---- one WORKGROUP per bucket: T threads, K values per thread, NS = T K slots, 2 NS bins.  Large buckets mean few of
// them per column -- long runs in the scatter, few search steps everywhere, a small sample to sort -- and a workgroup
// ranks a large bucket at K = 4 values per thread (48 VGPRs) where one wave per bucket would need 16 and more per lane.
// Workgroup barriers between the phases; the resolving loop's trip count stays per WAVE (the largest mixed bin among ITS
// values).  LDS: L[NS] (8 B) | CNT[2 NS + 4] (4 B) | wred[32] (8 B) | wtot[16] (4 B) | hv[4] (4 B) = 16 NS + 352 bytes.
// ZOOM: with the second-level pass for crowded bins (below).  Without it -- the main kernel, which keeps to 64 VGPRs and
// eight waves per SIMD: with the pass inlined it needs 74, runs six, and takes 4.38 instead of 3.75 ms on a table that
// never needs the pass -- a bucket that needs the pass is given back (false) and the caller lists it for the zoom kernel.
template <int T, int K, bool ZOOM>
__device__ __forceinline__ bool bucket_wg_rank(const BhsArgs& a, const BucketRef& r, uint64_t* smem) {
    constexpr int NS = T * K, BINS = 2 * NS, NW = T / 64;
    constexpr int LOGB = NS == 1024 ? 11 : NS == 2048 ? 12 : 13;
    static_assert(NS == 1024 || NS == 2048 || NS == 4096, "slots");
    uint64_t* L = smem;                                               // [NS]
    uint64_t* wred = L + NS;                                          // [2 NW]
    unsigned* CNT = reinterpret_cast<unsigned*>(wred + 32);          // [BINS + 4]
    unsigned* wtot = CNT + BINS + 4;                                  // [NW]
    unsigned* hv = wtot + 16;                                         // [0] a bin beyond HEAVY exists, [1] mixed ones among them
    constexpr int HEAVY = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_b = r.n_b;
    uint64_t key[K];
    bool ok[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = k * T + tid;                // coalesced
        ok[k] = p < n_b;
        key[k] = 0;
        if (ok[k]) key[k] = r.pd ? key_of(r.pd[p]) : __builtin_nontemporal_load(r.ks + p);
    }
    for (int i = tid; i < BINS + 1; i += T) CNT[i] = 0u;
    if (tid == 0) { hv[0] = 0u; hv[1] = 0u; }
    uint64_t lo = r.lo, hi = r.hi;
    if (r.end) {                                   // (block-uniform) bounds from the keys
        uint64_t mn = ~0ull, mx = 0ull;
#pragma unroll
        for (int k = 0; k < K; ++k) { if (ok[k]) { mn = key[k] < mn ? key[k] : mn; mx = key[k] > mx ? key[k] : mx; } }
        mn = wave_min_u64(mn);
        mx = wave_max_u64(mx);
        if (lane == 0) { wred[wave] = mn; wred[NW + wave] = mx; }
        __syncthreads();
        lo = wred[0]; hi = wred[NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) { lo = wred[w] < lo ? wred[w] : lo; hi = wred[NW + w] > hi ? wred[NW + w] : hi; }
    }
    const uint64_t range = hi - lo;
    if (range == 0ull) {
        // (block-uniform) a bucket of ONE repeated value -- Fisher's p = 1 and the other discrete levels fill whole buckets:
        // nothing to rank (any order is a valid one), p * m / rank falls with the rank, so every member's value inside the
        // bucket is that of the LAST rank.  No LDS, no atomics (1 024 atomic increments of ONE counter are what such a
        // bucket would cost in the counting sort).
        const uint64_t last = n_b > 0 ? raw_bits(lo, (int64_t)r.start + n_b, a.m) : ~0ull;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (ok[k]) r.ks[k * T + tid] = last;
        if (tid == 0) *r.bm = last;
        return true;
    }
    const int bits = 64 - __builtin_clzll(range);
    const int sh = __builtin_amdgcn_readfirstlane(bits > LOGB ? bits - LOGB : 0);
    // (block-uniform) A bucket whose bounds lie in different binades -- a column's FIRST bucket reaches from its smallest
    // p-value, many binades down, to the first splitter -- is badly served by bins that are linear in the key BITS alone:
    // uniformly distributed values put half the bucket into the top binade, i.e. into 2 of the 2 048 bins, and the
    // resolving loop below runs once per member of the fullest bin.  Such a bucket takes the MEAN of two monotone maps,
    // linear in the bits (spreads a tail over many decades) and linear in the value (spreads the uniform part): the mean
    // of two non-decreasing maps is non-decreasing, which is all the ranking needs.  Keys between two finite
    // non-negative bounds are finite non-negative doubles, so the value map is monotone over the whole bucket.
    const bool blend = (hi >> 52) != (lo >> 52) && hi < 0x7ff0000000000000ull;
    const double plo = __longlong_as_double((long long)lo);
    const double scale = blend ? (double)(BINS - 1) / (__longlong_as_double((long long)hi) - plo) : 0.0;
    __syncthreads();
    int bin[K];
    unsigned slot[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint64_t d = (key[k] - lo) >> sh;
        unsigned bn = (unsigned)(d < (uint64_t)(BINS - 1) ? d : (uint64_t)(BINS - 1));
        if (blend) {
            const unsigned bv = (unsigned)((__longlong_as_double((long long)key[k]) - plo) * scale);
            bn = (bn + (bv < (unsigned)(BINS - 1) ? bv : (unsigned)(BINS - 1))) >> 1;
        }
        bin[k] = (int)bn;
        slot[k] = 0u;
        if (ok[k]) slot[k] = atomicAdd(&CNT[bin[k]], 1u);
    }
    __syncthreads();
    {   // exclusive scan over the bins: thread t owns bins [2 K t, 2 K t + 2 K)
        unsigned c[2 * K], tot = 0, cm = 0;
#pragma unroll
        for (int j = 0; j < 2 * K; ++j) { c[j] = CNT[tid * 2 * K + j]; tot += c[j]; cm = c[j] > cm ? c[j] : cm; }
        if (cm > (unsigned)HEAVY) hv[0] = 1u;
        unsigned x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        unsigned pre = x - tot;
        for (int w = 0; w < wave; ++w) pre += wtot[w];
#pragma unroll
        for (int j = 0; j < 2 * K; ++j) { CNT[tid * 2 * K + j] = pre; pre += c[j]; }
        if (tid == T - 1) CNT[BINS] = pre;
    }
    __syncthreads();
    const bool heavy_path = hv[0] != 0u;            // (block-uniform)
    int sb[K], cb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        sb[k] = (int)CNT[bin[k]];
        cb[k] = ok[k] ? (int)CNT[bin[k] + 1] - sb[k] : 0;
        if (!ok[k]) sb[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (ok[k]) L[sb[k] + (int)slot[k]] = key[k];
    __syncthreads();
    // a bin of ONE repeated value needs no order (rank = start + slot); a value that differs from its bin's first member
    // marks the bin (top bit of its start word) and only marked bins are resolved by counting
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (ok[k] && L[sb[k]] != key[k]) atomicOr(&CNT[bin[k]], 0x80000000u);
    __syncthreads();
    unsigned before[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool mixed = ok[k] && (CNT[bin[k]] & 0x80000000u) != 0u;
        before[k] = mixed ? 0u : slot[k];
        if (!mixed) cb[k] = 0;
    }
    // The resolving loop below runs, for a whole wave, once per member of the FULLEST mixed bin among the wave's values.
    // Where distinct values crowd inside a bucket far closer than a bin is wide (the bench's Fisher columns hold ~1 500
    // sums within 10^-13 of 1, next to the exact ones) one bin of hundreds of members would hold all four waves for
    // hundreds of trips.  The members of mixed bins beyond HEAVY entries leave the loop and are ranked by a SECOND counting
    // sort inside their bin, with a window the bin's own members choose: centred on the bin's first member, as wide as 16 x
    // the median distance of the next seven members from it (copies of it aside; a median: a stray ordinary value in the crowd's bin does not
    // widen it), cut into SB sub-bins.  All heavy bins of the bucket share one array of NS sub-bin counters -- the bin
    // counters are dead by now -- and one scan; the listed keys are rewritten into their bin's stretch of L in sub-bin
    // order, a sub-bin of one repeated value is again ranked by position alone, and what is left (mixed sub-bins: a few
    // members each) is counted by the owner.  (block-uniform decisions throughout; six more barriers, for ~2 % of the
    // buckets of the bench's table)
    if (heavy_path && !ZOOM) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < K; ++k) any = any || cb[k] > HEAVY;
        if (any) hv[1] = 1u;
        __syncthreads();
        if (hv[1] != 0u) return false;
    }
    if (heavy_path && ZOOM) {
        unsigned* SC = CNT;                         // [NS + 1] sub-bin counters, then their exclusive scan (top bit: mixed)
        unsigned* HT = CNT + NS + 1;                // [NS] by bin start: heavy-bin index | median distance exponent << 16
        bool lst[K];
        __syncthreads();                            // every read of the bin counters is done
        for (int i = tid; i < NS + 1; i += T) SC[i] = 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            lst[k] = cb[k] > HEAVY;
            if (lst[k] && slot[k] == 0u) {          // the bin's first member (L[sb] is its own key) sizes the window
                unsigned e[7], nz = 0u;                 // (copies of the first member itself say nothing about the spread)
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const uint64_t kj = L[sb[k] + 1 + j];
                    const uint64_t d = kj > key[k] ? kj - key[k] : key[k] - kj;
                    e[j] = d != 0ull ? 63u - (unsigned)__builtin_clzll(d) : 64u;
                    nz += d != 0ull ? 1u : 0u;
                }
#define BH_CE(x, y) { const unsigned lo_ = e[x] < e[y] ? e[x] : e[y]; e[y] = e[x] < e[y] ? e[y] : e[x]; e[x] = lo_; }
                BH_CE(0, 6) BH_CE(2, 3) BH_CE(4, 5) BH_CE(0, 2) BH_CE(1, 4) BH_CE(3, 6) BH_CE(0, 1) BH_CE(2, 5)
                BH_CE(3, 4) BH_CE(1, 2) BH_CE(4, 6) BH_CE(2, 3) BH_CE(4, 5) BH_CE(1, 2) BH_CE(3, 4) BH_CE(5, 6)
#undef BH_CE
                const unsigned mi = nz >> 1;            // the median of the nz distances that are not zero (sorted first)
                unsigned med = e[0];
#pragma unroll
                for (int j = 1; j < 7; ++j) med = mi == (unsigned)j ? e[j] : med;
                if (nz == 0u) med = 0u;
                HT[sb[k]] = atomicAdd(&hv[1], 1u) | (med << 16);
            }
        }
        __syncthreads();
        const int n_heavy = (int)hv[1];
        if (n_heavy > 0) {                          // (a heavy bin of ONE value is not mixed: nothing listed)
            // every heavy bin holds more than HEAVY values: n_heavy <= NS / 9, SB >= 8, n_heavy x SB <= NS
            const int sbl = 31 - __builtin_clz((unsigned)(NS / n_heavy));
            const unsigned half = 1u << (sbl - 1);
            unsigned sub[K], ss[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                sub[k] = 0u; ss[k] = 0u;
                if (lst[k]) {
                    const unsigned ht = HT[sb[k]];
                    const uint64_t ref = L[sb[k]];
                    const int s2 = (int)(ht >> 16) + 5 - sbl;          // window 2^(exponent + 5) wide, SB sub-bins
                    const int sh2 = s2 > 0 ? s2 : 0;
                    unsigned in_bin;                                    // monotone in the key, saturating at both ends
                    if (key[k] >= ref) {
                        const uint64_t d = (key[k] - ref) >> sh2;
                        in_bin = half + (unsigned)(d < (uint64_t)(half - 1u) ? d : (uint64_t)(half - 1u));
                    } else {
                        const uint64_t d = (ref - key[k] - 1ull) >> sh2;
                        in_bin = half - 1u - (unsigned)(d < (uint64_t)(half - 1u) ? d : (uint64_t)(half - 1u));
                    }
                    sub[k] = ((ht & 0xffffu) << sbl) + in_bin;
                    ss[k] = atomicAdd(&SC[sub[k]], 1u);
                }
            }
            __syncthreads();
            {   // exclusive scan over the NS sub-bin counters: thread t owns [K t, K t + K)
                unsigned c[K], tot = 0;
#pragma unroll
                for (int j = 0; j < K; ++j) { c[j] = SC[tid * K + j]; tot += c[j]; }
                unsigned x = tot;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o); if (lane >= o) x += y; }
                if (lane == 63) wtot[wave] = x;
                __syncthreads();
                unsigned pre = x - tot;
                for (int w = 0; w < wave; ++w) pre += wtot[w];
#pragma unroll
                for (int j = 0; j < K; ++j) { SC[tid * K + j] = pre; pre += c[j]; }
                if (tid == T - 1) SC[NS] = pre;
            }
            __syncthreads();
            unsigned g0[K], cs[K], pos[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                g0[k] = 0u; cs[k] = 0u; pos[k] = 0u;
                if (lst[k]) {
                    const unsigned g = SC[sub[k]];
                    cs[k] = SC[sub[k] + 1] - g;
                    g0[k] = g - SC[(sub[k] >> sbl) << sbl];            // the sub-bin's start inside its bin
                    pos[k] = g0[k] + ss[k];
                    L[sb[k] + (int)pos[k]] = key[k];                    // (the stretch of a heavy bin is read by its members only,
                }                                                       //  and their reads of it lie before the last barrier)
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (lst[k] && L[sb[k] + (int)g0[k]] != key[k]) atomicOr(&SC[sub[k]], 0x80000000u);
            __syncthreads();
            unsigned smax = 0u, cnt[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                cnt[k] = 0u;
                if (!lst[k] || (SC[sub[k]] & 0x80000000u) == 0u) { cnt[k] = ss[k]; cs[k] = 0u; }
                smax = cs[k] > smax ? cs[k] : smax;
            }
            for (unsigned i = 0; __ballot(i < smax) != 0ull; ++i) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool in = i < cs[k];
                    const uint64_t kj = L[sb[k] + (int)g0[k] + (int)(in ? i : 0u)];
                    count_less96(cnt[k], in ? kj : ~0ull, in ? g0[k] + i : ~0u, key[k], pos[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (lst[k]) { before[k] = g0[k] + cnt[k]; cb[k] = 0; }
        }
    }
    int cmax = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) cmax = cb[k] > cmax ? cb[k] : cmax;
    for (int i = 0; __ballot(i < cmax) != 0ull; ++i) {
        uint64_t kj[K];
#pragma unroll
        for (int k = 0; k < K; ++k) kj[k] = L[sb[k] + (i < cb[k] ? i : 0)];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool in = i < cb[k];
            count_less96(before[k], in ? kj[k] : ~0ull, in ? (unsigned)i : ~0u, key[k], slot[k]);
        }
    }
    int rank[K];
    uint64_t raw[K];
    const double md = (double)a.m, inv_m = 1.0 / md;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        rank[k] = sb[k] + (int)before[k];
        raw[k] = raw_bits_inv(key[k], (int)r.start + rank[k] + 1, md, inv_m);
    }
    __syncthreads();                               // every read of the lists is done: L becomes the rank-ordered array
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (ok[k]) L[rank[k]] = raw[k];
    __syncthreads();
    {   // suffix minima over the ranks in place: thread t owns ranks [t K, t K + K)
        uint64_t sfx[K];
        uint64_t run = ~0ull;
#pragma unroll
        for (int j = K - 1; j >= 0; --j) {
            const int p = tid * K + j;
            const uint64_t v = p < n_b ? L[p] : ~0ull;
            run = v < run ? v : run;
            sfx[j] = run;
        }
        uint64_t x = run;                          // inclusive suffix minimum over the lanes >= this one
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = shfl_down_u64(x, o);
            if (lane + o < 64) x = y < x ? y : x;
        }
        uint64_t ex = shfl_down_u64(x, 1);
        if (lane == 63) ex = ~0ull;
        if (lane == 0) wred[wave] = x;
        __syncthreads();
        for (int w = wave + 1; w < NW; ++w) ex = wred[w] < ex ? wred[w] : ex;     // (minima of the waves behind this one)
        if (tid == 0) {
            uint64_t all = wred[0];
            for (int w = 1; w < NW; ++w) all = wred[w] < all ? wred[w] : all;
            *r.bm = all;
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (tid * K + j < n_b) L[tid * K + j] = sfx[j] < ex ? sfx[j] : ex;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (ok[k]) r.ks[k * T + tid] = L[rank[k]];        // in place: the slot the key came from
    return true;
}
template <int T, int K> constexpr size_t bucket_wg_lds() { return (size_t)T * K * 16 + 16 + 32 * 8 + 16 * 4 + 16; }

// one workgroup of T threads per bucket of up to 4 T values; larger ones, and the few with a crowded bin (see ZOOM), go to
// the work list of the second kernel.
// Workgroups are dealt round-robin over the 8 XCDs: all buckets of one column go to workgroups of ONE XCD (column mod
// 8), next to the tiles that filled them.
template <int T>
__global__ void __launch_bounds__(T) bhs_bucket_kernel(BhsArgs a) {
    extern __shared__ uint64_t smem_bk[];
    const int64_t k = (int64_t)(blockIdx.x >> 3);
    const int64_t seg_x = (k / a.B) * 8 + (blockIdx.x & 7);
    if (seg_x >= a.segs) return;
    const BucketRef r = bucket_ref(a, seg_x * a.B + k % a.B);
    const bool done = r.n_b <= 4 * T && r.n_b <= a.reg_cap && bucket_wg_rank<T, 4, false>(a, r, smem_bk);      // (block-uniform)
    if (!done && threadIdx.x == 0) a.big_list[atomicAdd(a.big_count, 1u)] = seg_x * a.B + k % a.B;
}

// the listed buckets in turn: up to 2048 values by the workgroup (T x K slots, with the second-level pass for crowded
// bins), beyond that (or beyond bh.reg_cap) by its first wave in HBM.  (512 / 1024 threads for ALL buckets were measured:
// the barriers between the phases cost more than the larger bucket saves -- EXPERIMENTS.md A.4; for the listed ones alone
// see the launch.)
template <int T, int K>
__global__ void __launch_bounds__(T) bhs_bucket_big_kernel(BhsArgs a) {
    extern __shared__ uint64_t smem_bk[];
    const unsigned n_big = *a.big_count;
#pragma nounroll
    for (unsigned w = blockIdx.x; w < n_big; w += gridDim.x) {
        const BucketRef r = bucket_ref(a, a.big_list[w]);
        if (r.n_b <= 1024 && r.n_b <= a.reg_cap) continue;            // (block-uniform) bhs_bucket_zoom_kernel's
        if ((r.n_b > a.reg_cap && !r.pd) || r.n_b > 2048) {
            if (threadIdx.x < 64) bucket_in_hbm(a, r.ks, r.n_b, r.start, (int)threadIdx.x, r.bm);
        } else {
            bucket_wg_rank<T, K, true>(a, r, smem_bk);
        }
        __syncthreads();
    }
}

// the listed buckets of up to 1024 values -- the main kernel gave them back for their crowded bins -- at 4 values per thread
// with the second-level pass: 74 VGPRs, six waves per SIMD, where the 8-per-thread kernel above runs two
__global__ void __launch_bounds__(256) bhs_bucket_zoom_kernel(BhsArgs a) {
    extern __shared__ uint64_t smem_bk[];
    const unsigned n_big = *a.big_count;
#pragma nounroll
    for (unsigned w = blockIdx.x; w < n_big; w += gridDim.x) {
        const BucketRef r = bucket_ref(a, a.big_list[w]);
        if (r.n_b <= 1024 && r.n_b <= a.reg_cap) bucket_wg_rank<256, 4, true>(a, r, smem_bk);
        __syncthreads();
    }
}

// ---------------------------------------------------------------- 6. minima of the later buckets
__global__ void __launch_bounds__(256) bhs_suffix_kernel(BhsArgs a) {
    __shared__ uint64_t bm[MAX_B];
    __shared__ uint64_t part[256];
    const int seg = blockIdx.x, tid = threadIdx.x;
    for (int b = tid; b < a.B; b += 256) bm[b] = a.bmin[(int64_t)seg * a.B + b];
    __syncthreads();
    // thread t owns buckets [4t, 4t+4): minimum of its chunk, then of the chunks behind it
    uint64_t mine = ~0ull;
    for (int q = 0; q < 4; ++q) {
        const int b = tid * 4 + q;
        if (b < a.B) mine = bm[b] < mine ? bm[b] : mine;
    }
    part[tid] = mine;
    __syncthreads();
    uint64_t later = ~0ull;
    for (int t = tid + 1; t < 256 && t * 4 < a.B; ++t) later = part[t] < later ? part[t] : later;
    for (int q = 3; q >= 0; --q) {
        const int b = tid * 4 + q;
        if (b < a.B) {
            a.sfx[(int64_t)seg * a.B + b] = later;
            later = bm[b] < later ? bm[b] : later;
        }
    }
}

// ---------------------------------------------------------------- 7. transpose back + final minimum, gathering
// out[r * pitch + c] = min(result of value (r, c) in its slot, sfx[c][its bucket], 1).  A workgroup owns COLS columns x
// 2048 / COLS rows: pos is read in runs per column, the results are gathered (8-byte reads inside the column's m x 8
// bytes: L2 hits while the strip's workgroups -- consecutive on one XCD -- are at work), rows leave as 8 COLS-byte pieces.
template <int COLS, bool NT>
__global__ void __launch_bounds__(TC_T) bhs_finish_kernel(BhsArgs a, double* __restrict__ out, int row_tiles) {
    constexpr int ROWS = 2048 / COLS;
    __shared__ double tile[COLS][ROWS + 1];
    const int tid = threadIdx.x;
    // blocks are dealt round-robin over the XCDs: strip = (k / row_tiles) * 8 + xcd, row tile = k mod row_tiles
    const int64_t k = (int64_t)(blockIdx.x >> 3);
    const int64_t strip = (k / row_tiles) * 8 + (blockIdx.x & 7);
    const int64_t c0 = strip * COLS;
    if (c0 >= a.segs) return;
    const int ncol = (int)min((int64_t)COLS, (int64_t)a.segs - c0);
    const int64_t r0 = (k % row_tiles) * ROWS;
    const int nrow = (int)min((int64_t)ROWS, a.m - r0);
    constexpr int PER = 2048 / TC_T;                                  // 8 values per thread
    unsigned pw[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + j * TC_T, c = i / ROWS, rr = i - c * ROWS;
        const bool ok = c < ncol && rr < nrow;
        pw[j] = ok ? (a.pos_cm ? __builtin_nontemporal_load(a.pos_cm + (c0 + c) * a.m + r0 + rr) : (unsigned)(r0 + rr)) : 0xFFFFFFFFu;
    }
    uint64_t own[PER], later[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = (tid + j * TC_T) / ROWS;
        const bool ok = pw[j] != 0xFFFFFFFFu;
        const uint64_t* src = a.keyS + (c0 + c) * a.m + (pw[j] & POS_MASK);
        own[j] = ok ? (NT ? __builtin_nontemporal_load(src) : *src) : 0ull;
        later[j] = ok ? a.sfx[(c0 + c) * a.B + (pw[j] >> POS_SHIFT)] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + j * TC_T, c = i / ROWS, rr = i - c * ROWS;
        double v = __longlong_as_double((long long)(own[j] < later[j] ? own[j] : later[j]));
        if (v > 1.0) v = 1.0;
        tile[c][rr] = v;
    }
    __syncthreads();
    const int col = tid & (COLS - 1), rsub = tid / COLS;
#pragma unroll
    for (int j = 0; j < ROWS / (TC_T / COLS); ++j) {
        const int rr = rsub + (TC_T / COLS) * j;
        if (col < ncol && rr < nrow) out[(r0 + rr) * a.pitch + c0 + col] = tile[col][rr];
    }
}

// =====================================================================================================
// ONE long vector (compare_sample_sets: 1 M p-values; `--multiple_test_correction all`): sample sort in four
// launches (+ one memset) instead of the 29 of the radix path (8 passes x 3 kernels + 5), which were pure launch latency
// (0.26 ms for 16 MB of algorithmic traffic):
//   bhv_rank     ranks of a jittered regular sample (16 per bucket) by brute force over the grid;
//   bhv_scatter  every 16th ranked sample is a splitter; tile of 4096 values, binary search in LDS, one global atomic
//                per (tile, bucket) reserves a run in the bucket's SLOT (2.75 x the mean bucket: no counting pass;
//                an element that finds its slot full goes to an overflow list), 12-byte (key, index) elements;
//                counts the present entries (masked variant);
//   bhv_bucket   one WORKGROUP per bucket (~2048 values, one copy in LDS, two workgroups per CU): a local sample
//                into sub-buckets of ~40, every thread ranks its values inside their sub-bucket by counting,
//                rank = bucket start + position, p * m / rank with the generic path's arithmetic, suffix minima over
//                the bucket, partial results scattered to the original positions with the bucket id;
//   bhv_finish   suffix minima over the bucket minima (every workgroup for itself), min(own, later buckets, 1).
// Ties are broken by index in every comparison (a vector of ONE repeated value still splits evenly); absent entries
// (masked variant) carry the key ~0, sort behind every p-value and are counted out of m.  Bit-identical to the radix path.
struct BhvArgs {
    const double* p; const uint8_t* tested; int masked;
    int64_t n;
    int B, spb, S;
    uint32_t* rank;        // [S]
    unsigned* tile_done;   // [ceil(S / 256)] j tiles finished per column of the ranking grid
    uint64_t* spl_k; uint32_t* spl_i;   // [B] splitters
    unsigned* cursor;      // [B] fill of the slots = bucket sizes
    unsigned long long* m_eff;
    uint64_t* keyS; uint32_t* idxS;     // [B][cap] the buckets' slots
    unsigned long long* ovf_n;          // elements that found their slot full (adversarial input only) ...
    uint64_t* ovfK; uint32_t* ovfI; uint16_t* ovfB;     // [n] ... with their bucket
    unsigned long long* spill_n;        // [n] where the slow path puts such a bucket together
    uint64_t* spillK; uint32_t* spillI;
    uint64_t* qpart;       // [n]
    uint16_t* bid;         // [n]
    uint64_t* bmin;        // [B]
    int cap;               // values a bucket workgroup holds in LDS
    double* q;
};
constexpr int BHV_T = 1024;           // threads of every bhv kernel but the ranking
constexpr int BHV_E = 4;              // values per thread in a count / scatter / finish tile

// absent entries sort behind every p-value (NaN patterns included) and in front of the padding (~0) of the register sorts
constexpr uint64_t BHV_ABSENT = ~0ull - 1;
__device__ __forceinline__ uint64_t bhv_key(const BhvArgs& a, int64_t i) {
    const double v = a.p[i];
    const bool present = !a.masked || (a.tested ? a.tested[i] != 0 : !(v < 0.0));
    return present ? key_of(v) : BHV_ABSENT;
}
__device__ __forceinline__ int64_t bhv_sample_pos(const BhvArgs& a, int j) {
    // jittered regular sample: one position in every stride of n / S values (32-bit arithmetic: n <= 2 Mi)
    const unsigned stride = (unsigned)(a.n / a.S);
    return (int64_t)((unsigned)j * stride + hash32((unsigned)j * 2654435761u + 12345u) % stride);
}

__global__ void __launch_bounds__(256) bhv_rank_kernel(BhvArgs a) {
    __shared__ __align__(16) uint64_t jk[256];
    __shared__ __align__(16) uint32_t ji[256];
    const int t = threadIdx.x;
    const int i = blockIdx.x * 256 + t, j = blockIdx.y * 256 + t;
    uint64_t ik = 0; uint32_t ii = 0;
    if (i < a.S) { const int64_t pos = bhv_sample_pos(a, i); ik = bhv_key(a, pos); ii = (uint32_t)pos; }
    jk[t] = ~0ull; ji[t] = ~0u;                  // (beyond the sample: a key above every key, never counted)
    if (j < a.S) { const int64_t pos = bhv_sample_pos(a, j); jk[t] = bhv_key(a, pos); ji[t] = (uint32_t)pos; }
    __syncthreads();
    // four samples per trip: two 16-byte reads of keys and one of indices (broadcasts) for four compares
    unsigned cnt = 0;
    if (i < a.S) {
#pragma unroll 4
        for (int q = 0; q < 256; q += 4) {
            const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(&jk[q]);
            const ulonglong2 k23 = *reinterpret_cast<const ulonglong2*>(&jk[q + 2]);
            const uint4 i4 = *reinterpret_cast<const uint4*>(&ji[q]);
            count_less96(cnt, k01.x, i4.x, ik, ii);
            count_less96(cnt, k01.y, i4.y, ik, ii);
            count_less96(cnt, k23.x, i4.z, ik, ii);
            count_less96(cnt, k23.y, i4.w, ik, ii);
        }
        if (cnt) atomicAdd(&a.rank[i], cnt);
    }
    // the workgroup that completes a column of the grid (all j tiles of these 256 samples) reads the final ranks and
    // writes the splitters: sample of rank spb * (b + 1) is splitter b -- the tile kernel then loads B - 1 finished
    // splitters instead of walking the S ranks (eight dependent global round trips per workgroup)
    // (no fence: the adds are device-scope atomics, performed at the memory side once vmcnt reaches zero, and the final
    //  ranks are read back with an atomic as well -- a release fence here writes the L2 back and cost 50 us)
    __shared__ unsigned last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) last = atomicAdd(&a.tile_done[blockIdx.x], 1u) == gridDim.y - 1 ? 1u : 0u;
    __syncthreads();
    if (last && i < a.S) {
        const unsigned r = atomicAdd(&a.rank[i], 0u);                     // (at the L2, where the other workgroups' adds landed)
        if (r != 0u && (r & (unsigned)(a.spb - 1)) == 0u) {               // (spb is a power of two)
            const int b = (int)(r >> (31 - __builtin_clz((unsigned)a.spb))) - 1;
            if (b < a.B - 1) { a.spl_k[b] = ik; a.spl_i[b] = ii; }
        }
    }
}

// splitters into LDS (written by the ranking kernel)
__device__ __forceinline__ void bhv_load_splitters(const BhvArgs& a, uint64_t* sk, uint32_t* si, int tid) {
    for (int b = tid; b < a.B - 1; b += BHV_T) { sk[b] = a.spl_k[b]; si[b] = a.spl_i[b]; }
}
// exclusive scan of up to 1024 counters held one per thread (BHV_T threads); returns this thread's prefix
__device__ __forceinline__ unsigned bhv_block_excl_scan(unsigned v, unsigned* wsum /*[16]*/, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    unsigned x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    unsigned pre = x - v;
    for (int k = 0; k < w; ++k) pre += wsum[k];
    return pre;
}

// classify a tile against the splitters and move it straight into the buckets' SLOTS of `cap` elements each -- no
// counting pass in front: a slot is 2.75 x the mean bucket, and the (practically impossible) elements beyond it go to
// an overflow list that the bucket's slow path collects.  cursor[b] ends as the size of bucket b.
__global__ void __launch_bounds__(BHV_T) bhv_scatter_kernel(BhvArgs a) {
    extern __shared__ uint64_t smem_v[];
    uint64_t* sk = smem_v;                                            // [B]
    uint32_t* si = reinterpret_cast<uint32_t*>(sk + a.B);             // [B]
    unsigned* hist = reinterpret_cast<unsigned*>(si + a.B);           // [B]
    unsigned* base = hist + a.B;                                      // [B]
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * (BHV_T * BHV_E);
    uint64_t key[BHV_E];
    int bkt[BHV_E];
    unsigned off[BHV_E];
    unsigned present = 0;
#pragma unroll
    for (int q = 0; q < BHV_E; ++q) {                 // (the tile's values travel while the splitters are put together)
        const int64_t e = e0 + q * BHV_T + tid;
        key[q] = e < a.n ? bhv_key(a, e) : 0;
        present += (e < a.n && key[q] != BHV_ABSENT) ? 1u : 0u;
    }
    for (int b = tid; b < a.B; b += BHV_T) hist[b] = 0;
    bhv_load_splitters(a, sk, si, tid);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BHV_E; ++q) {
        const int64_t e = e0 + q * BHV_T + tid;
        bkt[q] = -1;
        if (e < a.n) {
            bkt[q] = find_bucket(sk, si, a.B - 1, key[q], (uint32_t)e);
            off[q] = atomicAdd(&hist[bkt[q]], 1u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) present += (unsigned)__shfl_xor((int)present, o);
    if ((tid & 63) == 0) wsum[tid >> 6] = present;
    __syncthreads();
    if (tid == 0) {
        unsigned tot = 0;
        for (int w = 0; w < BHV_T / 64; ++w) tot += wsum[w];
        if (tot) atomicAdd(a.m_eff, (unsigned long long)tot);
    }
    if (tid < a.B) base[tid] = hist[tid] ? atomicAdd(&a.cursor[tid], hist[tid]) : 0u;      // (B <= 1024: one bucket per thread)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BHV_E; ++q) {
        if (bkt[q] >= 0) {
            const unsigned d = base[bkt[q]] + off[q];
            const uint32_t idx = (uint32_t)(e0 + q * BHV_T + tid);
            if (d < (unsigned)a.cap) {
                const int64_t dst = (int64_t)bkt[q] * a.cap + d;
                a.keyS[dst] = key[q];
                a.idxS[dst] = idx;
            } else {
                const unsigned long long o = atomicAdd(a.ovf_n, 1ull);
                a.ovfK[o] = key[q]; a.ovfI[o] = idx; a.ovfB[o] = (uint16_t)bkt[q];
            }
        }
    }
}

// a bucket beyond the LDS capacity (practically never): in-place network in HBM by one wave
__device__ uint64_t bhv_sub_in_lds(uint64_t* ks, uint32_t* is, int n_s, int64_t rank0, int64_t m_eff, int lane) {
    int P = 1;
    while (P < n_s) P <<= 1;
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j = k2 >> 1; j >= 1; j >>= 1) {
            for (int t = lane; t < (P >> 1); t += 64) {
                int i, l;
                if (j == (k2 >> 1)) { const int blk = t / j, r = t - blk * j; i = blk * k2 + r; l = blk * k2 + k2 - 1 - r; }
                else { i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); l = i | j; }
                if (l < n_s) {
                    const uint64_t ka = ks[i], kb = ks[l];
                    if (kb < ka) { const uint32_t ia = is[i], ib = is[l]; ks[i] = kb; ks[l] = ka; is[i] = ib; is[l] = ia; }
                }
            }
            __threadfence_block();
        }
    }
    uint64_t carry = ~0ull;
    for (int c = (n_s - 1) / 64; c >= 0; --c) {
        const int p = c * 64 + lane;
        uint64_t x = ~0ull;
        if (p < n_s) {
            const int64_t rank1 = rank0 + p + 1;
            x = rank1 <= m_eff ? raw_bits(ks[p], rank1, m_eff) : 0x7ff0000000000000ull;
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = shfl_down_u64(x, o);
            if (lane + o < 64) x = y < x ? y : x;
        }
        x = carry < x ? carry : x;
        if (p < n_s) ks[p] = x;
        carry = ((uint64_t)(unsigned)__shfl((int)(unsigned)(x >> 32), 0) << 32) | (unsigned)__shfl((int)(unsigned)(x & 0xffffffffu), 0);
    }
    return carry;
}

// LDS per value: key (8) + index (4) + sub-bucket (1) -- ONE copy of the bucket: a thread reads its values from HBM into
// registers, scatters them into sub-bucket order, and after the ranking holds the finished (p m / rank, index, position)
// in registers across a barrier before it overwrites the same arrays in sorted order.  5632 values = 73 KB, so that TWO
// workgroups share a CU and all ~490 buckets of 1 M values are resident at once (with two copies, 150 KB, the
// workgroups ran in two rounds: 99 us, half of it waiting).
constexpr int BHV_EPT = 6;            // values per thread of a bucket workgroup (cap <= 6144)
// sub-buckets: up to 128 of ~16 values, 2 samples each (sizes ~ Gamma(2): a value meets ~24 candidates on average; with
// 64 sub-buckets of ~32 and 4 samples each -- Gamma(4) -- it met ~40, and the counting loop is the kernel's VALU load)
constexpr int BHV_SPS_LOG = 1, BHV_SPS = 1 << BHV_SPS_LOG, BHV_NSB_MAX = 256 / BHV_SPS, BHV_SUB_MEAN = 20;
__global__ void __launch_bounds__(BHV_T, 8) bhv_bucket_kernel(BhvArgs a) {
    extern __shared__ uint64_t smem_b[];
    uint64_t* K2 = smem_b;
    uint32_t* I2 = reinterpret_cast<uint32_t*>(K2 + a.cap);
    uint8_t* SB = reinterpret_cast<uint8_t*>(I2 + a.cap);
    __shared__ uint64_t samK[256], ssk[BHV_NSB_MAX], smin[16];
    __shared__ uint32_t samI[256], ssi[BHV_NSB_MAX];
    __shared__ unsigned scount[BHV_NSB_MAX], sstart[BHV_NSB_MAX + 1], srank[256];
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int64_t m_eff = (int64_t)*a.m_eff;
    const uint64_t* slotK = a.keyS + (int64_t)b * a.cap;
    const uint32_t* slotI = a.idxS + (int64_t)b * a.cap;
    // bucket start = sum of the sizes before it
    const unsigned gc = tid < a.B ? a.cursor[tid] : 0u;
    unsigned before = tid < b ? gc : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += (unsigned)__shfl_xor((int)before, o);
    if (lane == 0) wsum[wave] = before;
    if (tid < BHV_NSB_MAX) scount[tid] = 0;
    if (tid < 256) srank[tid] = 0;
    __syncthreads();
    unsigned start = 0;
    for (int w = 0; w < BHV_T / 64; ++w) start += wsum[w];
    const int n_b = (int)a.cursor[b];
    if (n_b == 0) { if (tid == 0) a.bmin[b] = ~0ull; return; }
    if (n_b > a.cap) {
        // (adversarial input only) the bucket does not fit its slot, let alone LDS: its first wave puts it together --
        // the slot and the bucket's entries of the overflow list -- and sorts it in place in HBM
        if (wave == 0) {
            unsigned long long so = 0;
            if (lane == 0) so = atomicAdd(a.spill_n, (unsigned long long)n_b);
            so = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(so >> 32), 0) << 32) | (unsigned)__shfl((int)(unsigned)(so & 0xffffffffu), 0);
            uint64_t* ks = a.spillK + so; uint32_t* is = a.spillI + so;
            for (int p = lane; p < a.cap; p += 64) { ks[p] = slotK[p]; is[p] = slotI[p]; }
            const long long n_o = (long long)*a.ovf_n;
            int fill = a.cap;
            for (long long t0 = 0; t0 < n_o; t0 += 64) {
                const long long t = t0 + lane;
                const bool mine = t < n_o && a.ovfB[t] == (uint16_t)b;
                const unsigned long long mm = __ballot(mine);
                if (mine) { const int d = fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, 0u)); ks[d] = a.ovfK[t]; is[d] = a.ovfI[t]; }
                fill += __popcll(mm);
            }
            __threadfence_block();
            const uint64_t mn = bhv_sub_in_lds(ks, is, n_b, (int64_t)start, m_eff, lane);
            __threadfence();
            for (int p = lane; p < n_b; p += 64) { a.qpart[is[p]] = ks[p]; a.bid[is[p]] = (uint16_t)b; }
            if (lane == 0) a.bmin[b] = mn;
        }
        return;
    }
    uint64_t key[BHV_EPT];
    uint32_t idx[BHV_EPT];
#pragma unroll
    for (int q = 0; q < BHV_EPT; ++q) {
        const int i = tid + q * BHV_T;
        key[q] = 0; idx[q] = 0;
        if (i < n_b) { key[q] = slotK[i]; idx[q] = slotI[i]; }
    }
    // sub-buckets: a power of two
    int nsb = 1;
    while (nsb < BHV_NSB_MAX && nsb * BHV_SUB_MEAN < n_b) nsb <<= 1;
    if (nsb > 1) {
        // SPS * nsb regular samples (<= 256), ranked by counting: thread t counts, for sample t mod ns, the smaller samples
        // among slice t / ns of the sample; the sample of rank SPS (b + 1) is splitter b
        const int ns = BHV_SPS * nsb;
        const int nsh = 31 - __builtin_clz((unsigned)ns);            // ns is a power of two; j * n_b < 2^21
        if (tid < ns) {
            const int pos = (tid * n_b) >> nsh;
            samK[tid] = slotK[pos]; samI[tid] = slotI[pos];
        }
        __syncthreads();
        const int slices = BHV_T / ns;                                // >= 4
        const int per = ns / slices;                                  // samples per slice (a power of two >= 1)
        {
            const int sm = tid & (ns - 1), sl = tid >> nsh;
            const uint64_t km = samK[sm];
            const uint32_t im = samI[sm];
            unsigned c = 0;
            for (int j0 = sl * per; j0 < (sl + 1) * per; j0 += 8) {
                uint64_t kj[8];
                uint32_t ij[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int pj = min(j0 + u, ns - 1); kj[u] = samK[pj]; ij[u] = samI[pj]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = j0 + u < (sl + 1) * per;
                    count_less96(c, in ? kj[u] : ~0ull, in ? ij[u] : ~0u, km, im);
                }
            }
            if (c) atomicAdd(&srank[sm], c);
        }
        __syncthreads();
        if (tid < ns) {
            const unsigned c = srank[tid];
            if (c != 0u && (c & (unsigned)(BHV_SPS - 1)) == 0u) { ssk[(c >> BHV_SPS_LOG) - 1] = samK[tid]; ssi[(c >> BHV_SPS_LOG) - 1] = samI[tid]; }
        }
        __syncthreads();
    }
    // classify, count, scatter into sub-bucket order
    {
        int sub[BHV_EPT];
        unsigned off[BHV_EPT];
#pragma unroll
        for (int q = 0; q < BHV_EPT; ++q) {
            const int i = tid + q * BHV_T;
            sub[q] = -1;
            if (i < n_b) {
                sub[q] = nsb > 1 ? find_bucket(ssk, ssi, nsb - 1, key[q], idx[q]) : 0;
                off[q] = atomicAdd(&scount[sub[q]], 1u);
            }
        }
        __syncthreads();
        if (wave == 0) {                       // lane l: sub-buckets [l * PER, l * PER + PER)
            constexpr int PER = BHV_NSB_MAX / 64 > 0 ? BHV_NSB_MAX / 64 : 1;
            unsigned cq[PER], tot = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int sb = lane * PER + q; cq[q] = sb < nsb ? scount[sb] : 0u; tot += cq[q]; }
            unsigned x = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o); if (lane >= o) x += y; }
            unsigned pre = x - tot;
#pragma unroll
            for (int q = 0; q < PER; ++q) { const int sb = lane * PER + q; if (sb < BHV_NSB_MAX) sstart[sb] = pre; pre += cq[q]; }
            if (lane == 63) sstart[BHV_NSB_MAX] = x;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < BHV_EPT; ++q) {
            if (sub[q] >= 0) {
                const unsigned d = sstart[sub[q]] + off[q];
                K2[d] = key[q]; I2[d] = idx[q]; SB[d] = (uint8_t)sub[q];
            }
        }
    }
    __syncthreads();
    // every thread ranks its own values inside their sub-bucket by COUNTING the smaller ones (composite (key, index)
    // order; eight LDS reads in flight per trip; neighbouring threads share a sub-bucket, so the reads are near-broadcasts).
    // (A register sorting network per sub-bucket cost ~1.4 ds_bpermute per value and stage on the LDS crossbar; one wave
    //  per sub-bucket left most of the workgroup idle behind chains of LDS round trips.)
    {
        uint64_t ok[BHV_EPT];
        uint32_t oi[BHV_EPT];
        int op[BHV_EPT];
#pragma unroll
        for (int q = 0; q < BHV_EPT; ++q) {
            const int d = tid + q * BHV_T;
            op[q] = -1; ok[q] = 0; oi[q] = 0;
            if (d < n_b) {
                const int sb = SB[d];
                const int s0 = (int)sstart[sb], n_s = (int)scount[sb];
                const uint64_t km = K2[d];
                const uint32_t im = I2[d];
                unsigned c = 0;
                int j0 = 0;
                for (; j0 + 8 <= n_s; j0 += 8) {              // whole trips: eight reads in flight, no bound checks
                    uint64_t kj[8];
                    uint32_t ij[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { kj[u] = K2[s0 + j0 + u]; ij[u] = I2[s0 + j0 + u]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) count_less96(c, kj[u], ij[u], km, im);
                }
                if (j0 < n_s) {
                    uint64_t kj[8];
                    uint32_t ij[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {             // (beyond the sub-bucket: a key above every key)
                        const bool in = j0 + u < n_s;
                        const int j = s0 + min(j0 + u, n_s - 1);
                        kj[u] = in ? K2[j] : ~0ull; ij[u] = in ? I2[j] : ~0u;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) count_less96(c, kj[u], ij[u], km, im);
                }
                const int pos = s0 + (int)c;
                const int64_t rank1 = (int64_t)start + pos + 1;
                ok[q] = rank1 <= m_eff ? raw_bits(km, rank1, m_eff) : 0x7ff0000000000000ull;     // absent entries: +inf
                oi[q] = im;
                op[q] = pos;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < BHV_EPT; ++q)
            if (op[q] >= 0) { K2[op[q]] = ok[q]; I2[op[q]] = oi[q]; }
    }
    __syncthreads();
    // suffix minimum over the bucket's sorted positions: thread t owns positions [t * EPT, t * EPT + EPT)
    {
        uint64_t v[BHV_EPT];
        uint64_t run = ~0ull;
#pragma unroll
        for (int k = BHV_EPT - 1; k >= 0; --k) {
            const int pos = tid * BHV_EPT + k;
            const uint64_t r = pos < n_b ? K2[pos] : ~0ull;
            run = r < run ? r : run;
            v[k] = run;
        }
        uint64_t x = run;                       // inclusive suffix minimum over the lanes >= this one
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = shfl_down_u64(x, o);
            if (lane + o < 64) x = y < x ? y : x;
        }
        if (lane == 0) smin[wave] = x;          // (smin: one word per wave here)
        __syncthreads();
        uint64_t later = ~0ull;                 // minimum over the waves behind this one
        for (int w = wave + 1; w < BHV_T / 64; ++w) later = smin[w] < later ? smin[w] : later;
        uint64_t ex = shfl_down_u64(x, 1);      // ... and over the lanes behind this one
        if (lane == 63) ex = ~0ull;
        ex = ex < later ? ex : later;
        if (tid == 0) a.bmin[b] = x < later ? x : later;
#pragma unroll
        for (int k = 0; k < BHV_EPT; ++k) {
            const int pos = tid * BHV_EPT + k;
            if (pos < n_b) {
                const uint32_t dst = I2[pos];
                a.qpart[dst] = v[k] < ex ? v[k] : ex;
                a.bid[dst] = (uint16_t)b;
            }
        }
    }
}

__global__ void __launch_bounds__(BHV_T) bhv_finish_kernel(BhvArgs a) {
    __shared__ uint64_t sfx[1024];
    __shared__ uint64_t wmin[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // sfx[b] = minimum of the bucket minima behind b
    uint64_t x = tid < a.B ? a.bmin[tid] : ~0ull;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = shfl_down_u64(x, o);
        if (lane + o < 64) x = y < x ? y : x;
    }
    if (lane == 0) wmin[wave] = x;          // inclusive suffix minimum of the wave's 64 buckets
    __syncthreads();
    uint64_t later_waves = ~0ull;
    for (int w = wave + 1; w < BHV_T / 64; ++w) later_waves = wmin[w] < later_waves ? wmin[w] : later_waves;
    uint64_t ex = shfl_down_u64(x, 1);
    if (lane == 63) ex = ~0ull;
    sfx[tid] = ex < later_waves ? ex : later_waves;
    __syncthreads();
    const int64_t m_eff = (int64_t)*a.m_eff;
    (void)m_eff;
    const int64_t e0 = (int64_t)blockIdx.x * (BHV_T * BHV_E);
#pragma unroll
    for (int q = 0; q < BHV_E; ++q) {
        const int64_t e = e0 + q * BHV_T + tid;
        if (e < a.n) {
            const uint64_t own = a.qpart[e], later = sfx[a.bid[e]];
            double v = __longlong_as_double((long long)(own < later ? own : later));
            if (v > 1.0) v = 1.0;
            if (a.masked && bhv_key(a, e) == BHV_ABSENT) v = 0.0;      // absent entries get 0 (as the radix path leaves them)
            a.q[e] = v;
        }
    }
}

}  // namespace

// scratch bytes the sample-sort path needs for `segs` columns of m values: transposed input (8 B per value), bucketed
// keys / results (8), position words (4), the in-HBM path's copies (8), per-bucket tables
size_t sd_bh_cols_scratch(int64_t m, int64_t segs) {
    const size_t vals = (size_t)m * (size_t)segs;
    return vals * 28 + (size_t)segs * (size_t)(MAX_B + 1) * 50 + (1 << 16);
}

bool sd_bh_cols_supported(int64_t m, int64_t segs) {
    return m >= 1 && m <= ((int64_t)1 << POS_SHIFT) && segs >= 1 && segs <= 0x7fffffff;
}

// BH down each of the `segs` columns of the row-major table d_rm (m rows, row pitch `pitch` elements), in place.
// Scratch from the context arena (caller reserved sd_bh_cols_scratch bytes).
int sd_bh_cols_samplesort(sdice_ctx* ctx, int64_t m, int64_t segs, double* d_rm, int64_t pitch) {
    Arena& A = ctx->arena;
    BhsArgs a;
    a.p_rm = d_rm;
    a.pitch = pitch;
    a.m = m;
    a.segs = (int)segs;
    // buckets: bh.wg threads per bucket workgroup (256 / 512 / 1024: buckets of up to 1024 / 2048 / 4096 values), mean bucket
    // (bh.mean, default: half the workgroup's capacity)
    int wg_threads = (int)ctx->param("bh.wg", 256);
    wg_threads = wg_threads >= 1024 ? 1024 : wg_threads >= 512 ? 512 : 256;
    int B = 1;
    if (m > 4 * (int64_t)wg_threads) {
        int64_t mean = ctx->param("bh.mean", 0);
        // (25 000 x 19 900, 256 threads: mean 400 / 450 / 512 / 600 -> 14.7 / 14.3 / 14.1 / 13.9 ms: fewer, fuller workgroups
        //  against more buckets beyond the capacity, ~1 % at 0.55 of it)
        if (mean <= 0) mean = wg_threads * 4 * 55 / 100;
        mean = std::max<int64_t>(mean, sd_ceil_div(m, (int64_t)MAX_B));
        B = (int)sd_ceil_div(m, mean);
        if (B > MAX_B) B = MAX_B;
    }
    a.B = B;
    a.reg_cap = (int)std::min<int64_t>(2048, std::max<int64_t>(0, ctx->param("bh.reg_cap", 2048)));
    a.spb = (int)ctx->param("bh.spb", 8);
    if (a.spb < 1) a.spb = 1;
    while (a.spb > 1 && (int64_t)a.spb * B > 8192) a.spb >>= 1;      // the sample is sorted in 96 KB of LDS
    while (a.spb > 1 && (int64_t)a.spb * B * 2 > m) a.spb >>= 1;
    a.S = a.spb * B;
    a.S2 = 1;
    while (a.S2 < a.S) a.S2 <<= 1;
    const size_t vals = (size_t)m * (size_t)segs;
    const size_t sb = (size_t)segs * (size_t)B;
    a.p_cm = (double*)A.alloc(vals * 8);
    a.keyS = (uint64_t*)A.alloc(vals * 8);
    a.spl_k = (uint64_t*)A.alloc(sb * 8);
    a.bmin = (uint64_t*)A.alloc(sb * 8);
    a.sfx = (uint64_t*)A.alloc(sb * 8);
    a.big_list = (int64_t*)A.alloc(sb * 8);
    a.spl_i = (uint32_t*)A.alloc(sb * 4);
    a.gcount = (unsigned*)A.alloc(sb * 4);
    a.cursor = (unsigned*)A.alloc(sb * 4);
    a.start = (unsigned*)A.alloc((size_t)segs * (size_t)(B + 1) * 4);
    unsigned long long* zeroed = (unsigned long long*)A.alloc(16);    // spill_n | big_count
    a.pos_cm = B > 1 ? (uint32_t*)A.alloc(vals * 4) : nullptr;
    a.spill = B > 1 ? (uint64_t*)A.alloc(vals * 8) : nullptr;
    if (!a.p_cm || !a.keyS || !a.spl_k || !a.bmin || !a.sfx || !a.big_list || !a.spl_i || !a.gcount || !a.cursor || !a.start || !zeroed ||
        (B > 1 && (!a.pos_cm || !a.spill)))
        return SDICE_ERR_NOMEM;
    a.spill_n = zeroed;
    a.big_count = reinterpret_cast<unsigned*>(zeroed + 1);
    SD_HIP(hipMemsetAsync(zeroed, 0, 16, ctx->stream));
    const int64_t strips = sd_ceil_div(segs, (int64_t)TC_COLS);
    SD_ARG(strips < ((int64_t)1 << 31), "bh: too many columns");
    const int rows_per_block = (int)std::max<int64_t>(TC_ROWS, std::min<int64_t>(ctx->param("bh.rows_per_block", 2048), m));
    const int strips_per_xcd = (int)sd_ceil_div(strips, (int64_t)8);
    const int64_t tblocks = (int64_t)strips_per_xcd * 8 * sd_ceil_div(m, (int64_t)rows_per_block);
    SD_ARG(tblocks < ((int64_t)1 << 31), "bh: too many tiles");
    const dim3 tgrid((unsigned)tblocks);
    if (B > 1) {
        SD_HIP(hipMemsetAsync(a.gcount, 0, sb * 4, ctx->stream));
        const size_t lds_s = (size_t)a.S2 * 12;
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bhs_sample_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
        SD_LAUNCH(ctx, "bhs_sample_kernel", bhs_sample_kernel, dim3((unsigned)segs), dim3(256), lds_s, a);
        const size_t lds_t = (size_t)B * 20;
        const int64_t tiles = sd_ceil_div(m, (int64_t)TILE);
        const int64_t tile_blocks = sd_ceil_div(segs, (int64_t)8) * 8 * tiles;
        SD_ARG(tile_blocks < ((int64_t)1 << 31), "bh: too many tiles");
        // the splitters of a 16-column strip beside the transpose tile: up to ~190 buckets per column; beyond, the
        // transpose runs alone and the tiles of the scatter kernel count first
        const size_t lds_tc = (size_t)TC_COLS * ((size_t)(B - 1) * 12 + (size_t)B * 4);
        if (lds_tc <= 48 * 1024 && ctx->param("bh.fused_count", 1)) {
            SD_LAUNCH(ctx, "bhs_transpose_count_kernel", (bhs_transpose_kernel<true>), tgrid, dim3(TC_T), lds_tc, a, rows_per_block, strips_per_xcd);
        } else {
            SD_LAUNCH(ctx, "bhs_transpose_kernel", (bhs_transpose_kernel<false>), tgrid, dim3(TC_T), 0, a, rows_per_block, strips_per_xcd);
            SD_LAUNCH(ctx, "bhs_count_kernel", (bhs_tile_kernel<false>), dim3((unsigned)tile_blocks), dim3(TILE_T), lds_t, a, (int)tiles);
        }
        SD_LAUNCH(ctx, "bhs_scan_kernel", bhs_scan_kernel, dim3((unsigned)segs), dim3(256), 0, a);
        SD_LAUNCH(ctx, "bhs_scatter_kernel", (bhs_tile_kernel<true>), dim3((unsigned)tile_blocks), dim3(TILE_T), lds_t, a, (int)tiles);
    } else {
        SD_LAUNCH(ctx, "bhs_transpose_kernel", (bhs_transpose_kernel<false>), tgrid, dim3(TC_T), 0, a, rows_per_block, strips_per_xcd);
    }
    const int64_t n_buckets = segs * B;
    const int64_t bucket_blocks = sd_ceil_div(segs, (int64_t)8) * 8 * B;
    SD_ARG(bucket_blocks < ((int64_t)1 << 31), "bh: too many buckets");
    // threads of a bucket workgroup (4 values each): twice the mean bucket, so that ~1 % of the buckets (sizes ~ Gamma(8))
    // overflow to the second kernel
    if (wg_threads == 256) {
        SD_LAUNCH(ctx, "bhs_bucket_kernel", (bhs_bucket_kernel<256>), dim3((unsigned)bucket_blocks), dim3(256), (bucket_wg_lds<256, 4>()), a);
    } else if (wg_threads == 512) {
        SD_LAUNCH(ctx, "bhs_bucket_kernel", (bhs_bucket_kernel<512>), dim3((unsigned)bucket_blocks), dim3(512), (bucket_wg_lds<512, 4>()), a);
    } else {
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bhs_bucket_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)bucket_wg_lds<1024, 4>()));
        SD_LAUNCH(ctx, "bhs_bucket_kernel", (bhs_bucket_kernel<1024>), dim3((unsigned)bucket_blocks), dim3(1024), (bucket_wg_lds<1024, 4>()), a);
    }
    const int64_t big_blocks = std::max<int64_t>(1, std::min<int64_t>(n_buckets, (int64_t)ctx->n_cu * 4));
    // the listed buckets of 1 025 ... 2 048 values: 512 threads x 4 values (109 VGPRs) take 0.61 ms on the bench's table where
    // 256 x 8 (173 VGPRs, two waves per SIMD) take 0.77
    if (ctx->param("bh.big_wg", 512) >= 512) {
        SD_LAUNCH(ctx, "bhs_bucket_big_kernel", (bhs_bucket_big_kernel<512, 4>), dim3((unsigned)big_blocks), dim3(512), (bucket_wg_lds<512, 4>()), a);
    } else {
        SD_LAUNCH(ctx, "bhs_bucket_big_kernel", (bhs_bucket_big_kernel<256, 8>), dim3((unsigned)big_blocks), dim3(256), (bucket_wg_lds<256, 8>()), a);
    }
    SD_LAUNCH(ctx, "bhs_bucket_zoom_kernel", bhs_bucket_zoom_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_buckets, (int64_t)ctx->n_cu * 6))), dim3(256), (bucket_wg_lds<256, 4>()), a);
    SD_LAUNCH(ctx, "bhs_suffix_kernel", bhs_suffix_kernel, dim3((unsigned)segs), dim3(256), 0, a);
    // bh.finish_cols: columns per workgroup of the last kernel (16: full 128-byte lines out, 3.2 MB of results per strip
    // at 25 000 rows; 8: half lines, half the L2 footprint of the gather); bh.finish_nt: non-temporal gather loads
    const int fcols = ctx->param("bh.finish_cols", 16) >= 16 ? 16 : 8;
    const bool fnt = ctx->param("bh.finish_nt", 0) != 0;
    const int64_t fstrips = sd_ceil_div(segs, (int64_t)fcols);
    const int64_t row_tiles = sd_ceil_div(m, (int64_t)(2048 / fcols));
    const int64_t fin_blocks = sd_ceil_div(fstrips, (int64_t)8) * 8 * row_tiles;
    SD_ARG(fin_blocks < ((int64_t)1 << 31), "bh: too many tiles");
    if (fcols == 16) {
        if (fnt) SD_LAUNCH(ctx, "bhs_finish_kernel", (bhs_finish_kernel<16, true>), dim3((unsigned)fin_blocks), dim3(TC_T), 0, a, d_rm, (int)row_tiles);
        else SD_LAUNCH(ctx, "bhs_finish_kernel", (bhs_finish_kernel<16, false>), dim3((unsigned)fin_blocks), dim3(TC_T), 0, a, d_rm, (int)row_tiles);
    } else {
        if (fnt) SD_LAUNCH(ctx, "bhs_finish_kernel", (bhs_finish_kernel<8, true>), dim3((unsigned)fin_blocks), dim3(TC_T), 0, a, d_rm, (int)row_tiles);
        else SD_LAUNCH(ctx, "bhs_finish_kernel", (bhs_finish_kernel<8, false>), dim3((unsigned)fin_blocks), dim3(TC_T), 0, a, d_rm, (int)row_tiles);
    }
    return SDICE_OK;
}

// One vector of n p-values (masked: entries with tested == 0 -- or p < 0 without a mask -- are absent).  Supported sizes:
// the sample fits the brute-force ranking and a bucket fits LDS.
bool sd_bh_vector_supported(int64_t n) { return n >= 16384 && n <= ((int64_t)2 << 20); }      // (up to 1024 buckets of 2048 on average)
// geometry of the one-vector path: buckets of ~2048 values, 16 samples per bucket (sizes ~ Gamma(16): sigma = mean / 4);
// a bucket beyond its slot / the LDS capacity (5632 = mean + 7 sigma) practically never occurs (slow path: one wave, HBM)
static void bhv_geometry(sdice_ctx* ctx, int64_t n, int* B_out, int* cap_out) {
    int64_t mean = ctx->param("bhv.mean", 2048);
    if (mean < 512) mean = 512;
    if (mean < sd_ceil_div(n, (int64_t)1024)) mean = sd_ceil_div(n, (int64_t)1024);
    int B = (int)sd_ceil_div(n, mean);
    if (B < 2) B = 2;
    if (B > 1024) B = 1024;
    int cap = (int)ctx->param("bhv.cap", 5632);     // 5632 x 13 B = 73 KB of LDS: two bucket workgroups per CU
    if (cap < 64) cap = 64;
    if (cap > BHV_T * BHV_EPT) cap = BHV_T * BHV_EPT;
    *B_out = B; *cap_out = cap;
}
// slots (12 B x cap per bucket) + partial results (10 B per value) + the overflow list and the slow path's assembly area
// (14 + 12 B per value) + the sample ranks
size_t sd_bh_vector_scratch(sdice_ctx* ctx, int64_t n) {
    int B, cap;
    bhv_geometry(ctx, n, &B, &cap);
    return (size_t)n * 36 + (size_t)B * (size_t)cap * 12 + (size_t)B * 80 + (1 << 16);
}

int sd_bh_vector_samplesort(sdice_ctx* ctx, int64_t n, const double* d_p, const uint8_t* d_tested, bool masked, double* d_q) {
    Arena& A = ctx->arena;
    BhvArgs a;
    a.p = d_p; a.tested = d_tested; a.masked = masked ? 1 : 0; a.n = n; a.q = d_q;
    int B;
    bhv_geometry(ctx, n, &B, &a.cap);
    a.B = B;
    a.spb = 16;
    a.S = a.spb * B;
    const size_t N = (size_t)n;
    // one zeroed block: rank[S] | cursor[B] | tile_done[S / 256] | m_eff | ovf_n | spill_n
    const unsigned gs = (unsigned)sd_ceil_div(a.S, 256);
    const size_t zwords = (size_t)a.S + (size_t)B + gs + 8;
    unsigned* z = (unsigned*)A.alloc(zwords * 4);
    a.keyS = (uint64_t*)A.alloc((size_t)B * a.cap * 8);
    a.qpart = (uint64_t*)A.alloc(N * 8);
    a.ovfK = (uint64_t*)A.alloc(N * 8);
    a.spillK = (uint64_t*)A.alloc(N * 8);
    a.bmin = (uint64_t*)A.alloc((size_t)B * 8);
    a.idxS = (uint32_t*)A.alloc((size_t)B * a.cap * 4);
    a.ovfI = (uint32_t*)A.alloc(N * 4);
    a.spillI = (uint32_t*)A.alloc(N * 4);
    a.bid = (uint16_t*)A.alloc(N * 2);
    a.ovfB = (uint16_t*)A.alloc(N * 2);
    if (!z || !a.keyS || !a.qpart || !a.ovfK || !a.spillK || !a.bmin || !a.idxS || !a.ovfI || !a.spillI || !a.bid || !a.ovfB)
        return SDICE_ERR_NOMEM;
    a.rank = z;
    a.cursor = z + a.S;
    a.tile_done = a.cursor + B;
    a.m_eff = (unsigned long long*)(a.tile_done + gs + ((a.S + B + gs) & 1));      // 8-byte aligned
    a.spl_k = (uint64_t*)A.alloc((size_t)B * 8);
    a.spl_i = (uint32_t*)A.alloc((size_t)B * 4);
    if (!a.spl_k || !a.spl_i) return SDICE_ERR_NOMEM;
    a.ovf_n = a.m_eff + 1;
    a.spill_n = a.m_eff + 2;
    SD_HIP(hipMemsetAsync(z, 0, zwords * 4, ctx->stream));
    SD_LAUNCH(ctx, "bhv_rank_kernel", bhv_rank_kernel, dim3(gs, gs), dim3(256), 0, a);
    const unsigned tiles = (unsigned)sd_ceil_div(n, (int64_t)(BHV_T * BHV_E));
    const size_t lds_t = (size_t)B * 20;
    SD_LAUNCH(ctx, "bhv_scatter_kernel", bhv_scatter_kernel, dim3(tiles), dim3(BHV_T), lds_t, a);
    const size_t lds_b = (((size_t)a.cap * 13 + 15) / 16) * 16;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bhv_bucket_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    SD_LAUNCH(ctx, "bhv_bucket_kernel", bhv_bucket_kernel, dim3((unsigned)B), dim3(BHV_T), lds_b, a);
    SD_LAUNCH(ctx, "bhv_finish_kernel", bhv_finish_kernel, dim3(tiles), dim3(BHV_T), 0, a);
    return SDICE_OK;
}
