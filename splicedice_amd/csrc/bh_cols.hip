// K7 for MANY columns: Benjamini-Hochberg down each of `segs` equally long, contiguous segments of m
// p-values (the per-pair-column mode of `pairwise`, pairwise_fisher.py:187-191, after the transpose).
//
// The generic path (bh.hip) sorts (p, index) pairs with eight 8-bit LSD radix passes: ~150 B of HBM
// traffic per p-value and 58 of the 78 ms that BH took at config 4's per-GPU shard.  BH does not need
// a stable full sort, though -- only, per value, its rank and the running minimum of p*m/rank from
// the top rank down.  So here:
//   1. sample:   per segment, 8 jittered regular samples per bucket, sorted in LDS as (p bits, index)
//                pairs -- the index breaks ties, so a segment of ONE repeated value (Fisher p-values
//                are discrete, p = 1 is common) still splits evenly; every 8th is a splitter;
//   2. count:    tile of 4096 values, binary search over the splitters in LDS, one global atomic
//                per (tile, bucket);
//   3. scan:     bucket starts per segment (buckets are contiguous rank ranges);
//   4. scatter:  same search again, one global atomic per (tile, bucket) reserves a run, 12-byte
//                (key, index) elements land in their bucket;
//   5. buckets:  ONE WAVE per bucket (mean ~160 values, up to 1024): bitonic network on registers
//                (4/8/16 keys per lane, cross-lane steps by ds_bpermute, no LDS storage, no
//                barriers), ranks = bucket start + position, p*m/rank exactly as the generic path
//                computes it, suffix minimum inside the bucket, scattered to [segment][index] with
//                the bucket id beside it; bucket minimum to a small table.  A bucket beyond 1024
//                values (probability ~1e-10 per bucket) is sorted in place in HBM by its wave;
//   6. suffix minima of the bucket minima per segment;
//   7. the transpose back to the row-major table applies min(own, later buckets' minimum, 1).
// ~70 B of traffic per value and no multi-pass sort; ties need no order (all members of a tie group
// end with the group's last, smallest p*m/rank), so the result is bit-identical to the generic path.
#include "common.h"
#include <algorithm>

namespace {

constexpr int TILE_T = 512;         // threads of a count / scatter tile
constexpr int TILE_E = 8;           // values per thread
constexpr int TILE = TILE_T * TILE_E;
constexpr int MAX_B = 1024;

struct BhsArgs {
    const double* p_cm;     // [segs][m]
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
    uint64_t* keyS;         // [segs][m] bucketed keys
    uint32_t* idxS;
    uint64_t* q_cm;         // [segs][m] p*m/rank, suffix minimum inside the bucket (f64 bits)
    uint16_t* bid_cm;       // [segs][m] bucket of each value
    uint64_t* bmin;         // [segs][B]
    uint64_t* sfx;          // [segs][B] minimum over the LATER buckets
    unsigned* big_count;    // buckets of more than 256 values: work list of the second bucket kernel
    int64_t* big_list;      // [segs * B]
    int reg_cap;            // buckets beyond this many values take the in-HBM path (1024; lower in tests)
};

__device__ __forceinline__ uint64_t key_of(double v) {
    if (v == 0.0) v = 0.0;   // -0.0 -> +0.0
    return (uint64_t)__double_as_longlong(v);
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
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
    const double* p = a.p_cm + (int64_t)seg * a.m;
    for (int j = tid; j < a.S2; j += 256) {
        uint64_t k = ~0ull;
        uint32_t i = ~0u;
        if (j < a.S) {
            const int64_t lo = (int64_t)j * a.m / a.S, hi = (int64_t)(j + 1) * a.m / a.S;
            const int64_t pos = lo + (int64_t)(hash32((unsigned)seg * 40503u + (unsigned)j) % (unsigned)(hi - lo));
            k = key_of(p[pos]);
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

// ---------------------------------------------------------------- 2. / 4. count and scatter
template <bool SCATTER>
__global__ void __launch_bounds__(TILE_T) bhs_tile_kernel(BhsArgs a) {
    extern __shared__ uint64_t smem_t[];
    uint64_t* sk = smem_t;                                            // [B]
    uint32_t* si = reinterpret_cast<uint32_t*>(sk + a.B);             // [B]
    unsigned* hist = reinterpret_cast<unsigned*>(si + a.B);           // [B]
    unsigned* base = hist + a.B;                                      // [B]
    const int seg = blockIdx.y, tid = threadIdx.x;
    const int ns = a.B - 1;
    for (int b = tid; b < a.B; b += TILE_T) {
        hist[b] = 0;
        if (b < ns) { sk[b] = a.spl_k[(int64_t)seg * a.B + b]; si[b] = a.spl_i[(int64_t)seg * a.B + b]; }
    }
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * TILE;
    const double* p = a.p_cm + (int64_t)seg * a.m;
    uint64_t key[TILE_E];
    int bkt[TILE_E];
    unsigned off[TILE_E];
#pragma unroll
    for (int q = 0; q < TILE_E; ++q) {
        const int64_t e = e0 + q * TILE_T + tid;
        key[q] = e < a.m ? key_of(p[e]) : 0;
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
            const int64_t dst = (int64_t)seg * a.m + base[bkt[q]] + off[q];
            a.keyS[dst] = key[q];
            a.idxS[dst] = (uint32_t)(e0 + q * TILE_T + tid);
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

// ascending bitonic network over 64*K (key, val) pairs; pair at position lane*K + k
template <int K>
__device__ __forceinline__ void wave_bitonic(uint64_t (&key)[K], uint32_t (&val)[K], const int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64 * K; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j >= 1; j >>= 1) {
            if (j >= K) {
                const int lm = j / K;
                const bool lower = (lane & lm) == 0;
                const bool up = k2 >= 64 * K ? true : (lane & (k2 / K)) == 0;
                const bool keep_min = lower == up;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint64_t o = shfl_xor_u64(key[k], lm);
                    const uint32_t ov = (uint32_t)__shfl_xor((int)val[k], lm);
                    const bool take = keep_min ? o < key[k] : o > key[k];
                    key[k] = take ? o : key[k];
                    val[k] = take ? ov : val[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int kp = k ^ j;
                    if (kp > k) {
                        bool up;
                        if (k2 < K) up = (k & k2) == 0;
                        else if (k2 >= 64 * K) up = true;
                        else up = (lane & (k2 / K)) == 0;
                        const bool sw = up ? key[kp] < key[k] : key[kp] > key[k];
                        const uint64_t tk = key[k];
                        const uint32_t tv = val[k];
                        key[k] = sw ? key[kp] : tk;
                        val[k] = sw ? val[kp] : tv;
                        key[kp] = sw ? tk : key[kp];
                        val[kp] = sw ? tv : val[kp];
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ uint64_t raw_bits(uint64_t key, int64_t rank1, int64_t m) {
    // p_(i) / (i / m), the arithmetic of the generic path (bh.hip bh_raw_kernel)
    const double ps = __longlong_as_double((long long)key);
    const double ecdf = (double)rank1 / (double)m;
    return (uint64_t)__double_as_longlong(ps / ecdf);
}

template <int K>
__device__ __forceinline__ void bucket_in_regs(const BhsArgs& a, const uint64_t* ks, const uint32_t* is, const double* pd,
                                               int n_b, int64_t seg_off, unsigned start, int bucket, int lane,
                                               uint64_t* bmin_out) {
    uint64_t key[K];
    uint32_t val[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = k * 64 + lane;              // coalesced; the starting arrangement is arbitrary anyway
        key[k] = ~0ull;
        val[k] = 0u;
        if (p < n_b) {
            if (pd) { key[k] = key_of(pd[p]); val[k] = (uint32_t)p; }
            else { key[k] = ks[p]; val[k] = is[p]; }
        }
    }
    wave_bitonic<K>(key, val, lane);
    uint64_t s[K];
    uint64_t run = ~0ull;
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
        const int p = lane * K + k;
        const uint64_t r = p < n_b ? raw_bits(key[k], (int64_t)start + p + 1, a.m) : ~0ull;
        run = r < run ? r : run;
        s[k] = run;
    }
    uint64_t x = run;                              // inclusive suffix minimum over the lanes >= this one
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = shfl_down_u64(x, o);
        if (lane + o < 64) x = y < x ? y : x;
    }
    uint64_t ex = shfl_down_u64(x, 1);
    if (lane == 63) ex = ~0ull;
    if (lane == 0) *bmin_out = x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = lane * K + k;
        if (p < n_b) {
            const uint64_t v = s[k] < ex ? s[k] : ex;
            a.q_cm[seg_off + val[k]] = v;
            a.bid_cm[seg_off + val[k]] = (uint16_t)bucket;
        }
    }
}

// rare: more than 1024 values in one bucket -- its wave sorts it in place in HBM (all-ascending
// bitonic network: out-of-range partners count as +inf and never move), then walks it from the end
__device__ void bucket_in_hbm(const BhsArgs& a, uint64_t* ks, uint32_t* is, int n_b, int64_t seg_off, unsigned start,
                              int bucket, int lane, uint64_t* bmin_out) {
    int P = 1;
    while (P < n_b) P <<= 1;
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j = k2 >> 1; j >= 1; j >>= 1) {
            for (int t = lane; t < (P >> 1); t += 64) {
                int i, l;
                if (j == (k2 >> 1)) {              // first step of a merge: mirror partner
                    const int blk = t / j, r = t - blk * j;
                    i = blk * k2 + r;
                    l = blk * k2 + k2 - 1 - r;
                } else {
                    i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    l = i | j;
                }
                if (l < n_b) {
                    const uint64_t ka = ks[i], kb = ks[l];
                    if (kb < ka) {
                        const uint32_t ia = is[i], ib = is[l];
                        ks[i] = kb; ks[l] = ka; is[i] = ib; is[l] = ia;
                    }
                }
            }
            __threadfence_block();
        }
    }
    uint64_t carry = ~0ull;
    for (int c = (n_b - 1) / 64; c >= 0; --c) {
        const int p = c * 64 + lane;
        uint64_t x = p < n_b ? raw_bits(ks[p], (int64_t)start + p + 1, a.m) : ~0ull;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = shfl_down_u64(x, o);
            if (lane + o < 64) x = y < x ? y : x;
        }
        x = carry < x ? carry : x;
        if (p < n_b) {
            a.q_cm[seg_off + is[p]] = x;
            a.bid_cm[seg_off + is[p]] = (uint16_t)bucket;
        }
        carry = ((uint64_t)(unsigned)__shfl((int)(unsigned)(x >> 32), 0) << 32) |
                (unsigned)__shfl((int)(unsigned)(x & 0xffffffffu), 0);     // lane 0's value for everyone
    }
    if (lane == 0) *bmin_out = carry;
}

// what a wave needs to know about bucket g
struct BucketRef {
    int seg, b, n_b;
    unsigned start;
    int64_t seg_off;
    const double* pd;
    uint64_t* ks;
    uint32_t* is;
    uint64_t* bm;
};
__device__ __forceinline__ BucketRef bucket_ref(const BhsArgs& a, int64_t g) {
    BucketRef r;
    r.seg = (int)(g / a.B);
    r.b = (int)(g - (int64_t)r.seg * a.B);
    r.seg_off = (int64_t)r.seg * a.m;
    r.start = 0;
    r.n_b = (int)a.m;
    r.pd = nullptr; r.ks = nullptr; r.is = nullptr;
    if (a.B == 1) {
        r.pd = a.p_cm + r.seg_off;                 // short segments: no partition, straight from the p-values
    } else {
        r.start = a.start[(int64_t)r.seg * (a.B + 1) + r.b];
        r.n_b = (int)(a.start[(int64_t)r.seg * (a.B + 1) + r.b + 1] - r.start);
        r.ks = a.keyS + r.seg_off + r.start;
        r.is = a.idxS + r.seg_off + r.start;
    }
    r.bm = a.bmin + (int64_t)r.seg * a.B + r.b;
    return r;
}

// buckets of up to 256 values (nearly all of them): 4 keys per lane, ~40 VGPRs, 8 waves per SIMD.
// Larger buckets go to a work list for bhs_bucket_big_kernel (whose 16-keys-per-lane network needs
// 150 VGPRs: in one kernel it held every wave to 3 per SIMD and the loads' latency was not hidden).
// Workgroups are dealt round-robin over the 8 XCDs and each XCD has its own L2: all buckets of one segment
// go to workgroups of ONE XCD (segment mod 8), so that the scattered 8-byte results of a segment -- every
// cache line of its 200 KB gets its 16 values from 16 different buckets -- meet in one L2 and leave it as
// full lines.  (With buckets dealt over all XCDs every L2 wrote its own partial lines: 26.7 GB of write
// traffic for 5 GB of results.)
__global__ void __launch_bounds__(256) bhs_bucket_kernel(BhsArgs a, int blocks_per_seg) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)(blockIdx.x >> 3);
    const int64_t seg_x = (k / blocks_per_seg) * 8 + (blockIdx.x & 7);
    const int b_x = (int)(k % blocks_per_seg) * 4 + (int)(threadIdx.x >> 6);
    if (seg_x >= a.segs || b_x >= a.B) return;
    const int64_t g = seg_x * a.B + b_x;
    const BucketRef r = bucket_ref(a, g);
    if (r.n_b <= 256 && r.n_b <= a.reg_cap) {
        bucket_in_regs<4>(a, r.ks, r.is, r.pd, r.n_b, r.seg_off, r.start, r.b, lane, r.bm);
    } else if (lane == 0) {
        const unsigned slot = atomicAdd(a.big_count, 1u);
        a.big_list[slot] = g;
    }
}

__global__ void __launch_bounds__(256) bhs_bucket_big_kernel(BhsArgs a) {
    const int lane = threadIdx.x & 63;
    const unsigned n_big = *a.big_count;
    const unsigned waves = gridDim.x * 4;
#pragma nounroll
    for (unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6); w < n_big; w += waves) {
        const BucketRef r = bucket_ref(a, a.big_list[w]);
        if (r.n_b > a.reg_cap && !r.pd) bucket_in_hbm(a, r.ks, r.is, r.n_b, r.seg_off, r.start, r.b, lane, r.bm);
        else if (r.n_b <= 256) bucket_in_regs<4>(a, r.ks, r.is, r.pd, r.n_b, r.seg_off, r.start, r.b, lane, r.bm);
        else if (r.n_b <= 512) bucket_in_regs<8>(a, r.ks, r.is, r.pd, r.n_b, r.seg_off, r.start, r.b, lane, r.bm);
        else if (r.n_b <= 1024) bucket_in_regs<16>(a, r.ks, r.is, r.pd, r.n_b, r.seg_off, r.start, r.b, lane, r.bm);
        else bucket_in_hbm(a, r.ks, r.is, r.n_b, r.seg_off, r.start, r.b, lane, r.bm);
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

// ---------------------------------------------------------------- 7. transpose back + final minimum
// out[r * out_pitch + c] = min(q_cm[c][r], sfx[c][bid_cm[c][r]], 1)
__global__ void __launch_bounds__(256) bhs_finish_kernel(BhsArgs a, double* __restrict__ out, int64_t out_pitch) {
    __shared__ double tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = c0 + k, r = r0 + tx;
        if (c < a.segs && r < a.m) {
            const uint64_t own = a.q_cm[c * a.m + r];
            const uint64_t later = a.sfx[c * a.B + a.bid_cm[c * a.m + r]];
            double v = __longlong_as_double((long long)(own < later ? own : later));
            if (v > 1.0) v = 1.0;
            tile[k][tx] = v;
        }
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + k, c = c0 + tx;
        if (c < a.segs && r < a.m) out[r * out_pitch + c] = tile[tx][k];
    }
}

}  // namespace

// scratch bytes the sample-sort path needs for `segs` segments of m values (without the transposed input)
size_t sd_bh_cols_scratch(int64_t m, int64_t segs) {
    const size_t vals = (size_t)m * (size_t)segs;
    return vals * 22 + (size_t)segs * (size_t)(MAX_B + 1) * 48 + (1 << 16);
}

bool sd_bh_cols_supported(int64_t m, int64_t segs) {
    return m >= 1 && m <= ((int64_t)1 << 18) && segs >= 1 && segs <= 0x7fffffff;
}

// BH inside each of `segs` contiguous segments of d_cm ([segs][m]); the result goes, transposed back,
// to d_out[r * out_pitch + c] (r < m, c < segs).  Scratch from the context arena (caller reserved it).
int sd_bh_cols_samplesort(sdice_ctx* ctx, int64_t m, int64_t segs, const double* d_cm, double* d_out, int64_t out_pitch) {
    Arena& A = ctx->arena;
    BhsArgs a;
    a.p_cm = d_cm;
    a.m = m;
    a.segs = (int)segs;
    int B = 1;
    if (m > 1024) {
        int64_t mean = std::max<int64_t>(ctx->param("bh.mean", 160), sd_ceil_div(m, (int64_t)MAX_B));
        B = (int)sd_ceil_div(m, mean);
        if (B > MAX_B) B = MAX_B;
    }
    a.B = B;
    a.reg_cap = (int)std::min<int64_t>(1024, std::max<int64_t>(0, ctx->param("bh.reg_cap", 1024)));
    a.spb = (int)ctx->param("bh.spb", 8);
    if (a.spb < 1) a.spb = 1;
    while (a.spb > 1 && (int64_t)a.spb * B > 8192) a.spb >>= 1;      // the sample is sorted in 96 KB of LDS
    while (a.spb > 1 && (int64_t)a.spb * B * 2 > m) a.spb >>= 1;
    a.S = a.spb * B;
    a.S2 = 1;
    while (a.S2 < a.S) a.S2 <<= 1;
    const size_t vals = (size_t)m * (size_t)segs;
    const size_t sb = (size_t)segs * (size_t)B;
    a.spl_k = (uint64_t*)A.alloc(sb * 8);
    a.bmin = (uint64_t*)A.alloc(sb * 8);
    a.sfx = (uint64_t*)A.alloc(sb * 8);
    a.spl_i = (uint32_t*)A.alloc(sb * 4);
    a.gcount = (unsigned*)A.alloc(sb * 4);
    a.cursor = (unsigned*)A.alloc(sb * 4);
    a.start = (unsigned*)A.alloc((size_t)segs * (size_t)(B + 1) * 4);
    a.q_cm = (uint64_t*)A.alloc(vals * 8);
    a.bid_cm = (uint16_t*)A.alloc(vals * 2);
    a.big_list = (int64_t*)A.alloc(sb * 8);
    a.big_count = (unsigned*)A.alloc(8);
    a.keyS = B > 1 ? (uint64_t*)A.alloc(vals * 8) : nullptr;
    a.idxS = B > 1 ? (uint32_t*)A.alloc(vals * 4) : nullptr;
    if (!a.spl_k || !a.bmin || !a.sfx || !a.spl_i || !a.gcount || !a.cursor || !a.start || !a.q_cm || !a.bid_cm || !a.big_list || !a.big_count ||
        (B > 1 && (!a.keyS || !a.idxS)))
        return SDICE_ERR_NOMEM;
    if (B > 1) {
        SD_HIP(hipMemsetAsync(a.gcount, 0, sb * 4, ctx->stream));
        const size_t lds_s = (size_t)a.S2 * 12;
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bhs_sample_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
        SD_LAUNCH(ctx, "bhs_sample_kernel", bhs_sample_kernel, dim3((unsigned)segs), dim3(256), lds_s, a);
        const size_t lds_t = (size_t)B * 20;
        const int64_t tiles = sd_ceil_div(m, (int64_t)TILE);
        // blockIdx.y is limited to 65535: segments go in slices
        for (int64_t s0 = 0; s0 < segs; s0 += 65535) {
            BhsArgs b = a;
            const int64_t sc = std::min<int64_t>(65535, segs - s0);
            b.p_cm += s0 * m; b.spl_k += s0 * B; b.spl_i += s0 * B; b.gcount += s0 * B;
            SD_LAUNCH(ctx, "bhs_count_kernel", (bhs_tile_kernel<false>), dim3((unsigned)tiles, (unsigned)sc), dim3(TILE_T), lds_t, b);
        }
        SD_LAUNCH(ctx, "bhs_scan_kernel", bhs_scan_kernel, dim3((unsigned)segs), dim3(256), 0, a);
        for (int64_t s0 = 0; s0 < segs; s0 += 65535) {
            BhsArgs b = a;
            const int64_t sc = std::min<int64_t>(65535, segs - s0);
            b.p_cm += s0 * m; b.spl_k += s0 * B; b.spl_i += s0 * B; b.cursor += s0 * B;
            b.keyS += s0 * m; b.idxS += s0 * m;
            SD_LAUNCH(ctx, "bhs_scatter_kernel", (bhs_tile_kernel<true>), dim3((unsigned)tiles, (unsigned)sc), dim3(TILE_T), lds_t, b);
        }
    }
    const int64_t n_buckets = segs * B;
    SD_ARG(sd_ceil_div(n_buckets, (int64_t)4) < ((int64_t)1 << 31), "bh: too many buckets");
    SD_HIP(hipMemsetAsync(a.big_count, 0, 8, ctx->stream));
    const int blocks_per_seg = (int)sd_ceil_div((int64_t)B, (int64_t)4);
    const int64_t bucket_blocks = sd_ceil_div(segs, (int64_t)8) * 8 * blocks_per_seg;
    SD_ARG(bucket_blocks < ((int64_t)1 << 31), "bh: too many buckets");
    SD_LAUNCH(ctx, "bhs_bucket_kernel", bhs_bucket_kernel, dim3((unsigned)bucket_blocks), dim3(256), 0, a, blocks_per_seg);
    SD_LAUNCH(ctx, "bhs_bucket_big_kernel", bhs_bucket_big_kernel,
              dim3((unsigned)std::min<int64_t>(sd_ceil_div(n_buckets, (int64_t)4), (int64_t)ctx->n_cu * 4)), dim3(256), 0, a);
    SD_LAUNCH(ctx, "bhs_suffix_kernel", bhs_suffix_kernel, dim3((unsigned)segs), dim3(256), 0, a);
    const int64_t gx = sd_ceil_div(m, (int64_t)32);
    for (int64_t c0 = 0; c0 < segs; c0 += (int64_t)65535 * 32) {
        BhsArgs b = a;
        const int64_t cc = std::min<int64_t>((int64_t)65535 * 32, segs - c0);
        b.segs = (int)cc;
        b.q_cm += c0 * m; b.bid_cm += c0 * m; b.sfx += c0 * B;
        SD_LAUNCH(ctx, "bhs_finish_kernel", bhs_finish_kernel, dim3((unsigned)gx, (unsigned)sd_ceil_div(cc, (int64_t)32)),
                  dim3(256), 0, b, d_out + c0, out_pitch);
    }
    return SDICE_OK;
}
