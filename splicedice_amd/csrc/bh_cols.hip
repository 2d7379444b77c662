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
//   5. buckets:  ONE WAVE per bucket (mean ~200 values, up to 1024): bitonic network on registers
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
#include <stdio.h>

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
    unsigned* big_count;    // [8] buckets of more than 256 values: work lists of the second bucket kernel, one per XCD
    int64_t* big_list;      // [8][big_region]: list x holds buckets of the segments with seg mod 8 == x
    int64_t big_region;
    int reg_cap;            // buckets beyond this many values take the in-HBM path (1024; lower in tests)
    int ablate;             // timing experiments (param bh.ablate; results are wrong when set): 1 no q store, 2 no bucket-id store
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

// ---- key-only network (the default bucket path) --------------------------------------------------------------
// BH needs no order inside a tie group (all its members end with the group's last, smallest p * m / rank), so the
// network moves KEYS ONLY -- half the registers and half the exchange traffic of (key, index) pairs -- and every value
// then finds its rank by a binary search over the sorted keys, which the wave parks in LDS together with the suffix
// minima (lower bound: the first member of its tie group; the suffix minimum from there is the group's value).
// Lane exchanges by DPP (xor 1, 2, 4, 8: VALU rate, no LDS crossbar) and the gfx950 lane swaps (xor 16, 32); a swap
// hands over BOTH operands of the compare-exchange, which is symmetric in them.
template <int M> __device__ __forceinline__ unsigned dpp_xor(unsigned x) {
    if (M == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
    if (M == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);          // quad_perm [2,3,0,1]
    if (M == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, true);         // row_ror:8
    // xor 4: banks 1, 3 take lane - 4 (row_ror:4), banks 0, 2 take lane + 4 (row_ror:12)
    const int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x124, 0xF, 0xA, false);
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x12C, 0xF, 0x5, false);
}
// compare-exchange of `key` with lane ^ M: the lane keeps the smaller key iff keep_min
template <int M> __device__ __forceinline__ uint64_t cx_lane(uint64_t key, bool keep_min) {
    uint64_t a, b;
    if (M >= 16) {
        const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
        if (M == 32) {
            const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            a = ((uint64_t)rh[0] << 32) | rl[0]; b = ((uint64_t)rh[1] << 32) | rl[1];
        } else {
            const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            a = ((uint64_t)rh[0] << 32) | rl[0]; b = ((uint64_t)rh[1] << 32) | rl[1];
        }
    } else {
        a = key;
        b = ((uint64_t)dpp_xor<M>((unsigned)(key >> 32)) << 32) | dpp_xor<M>((unsigned)key);
    }
    return ((a < b) == keep_min) ? a : b;
}
// ascending bitonic network over 64 * K keys; key at position lane * K + k
template <int K>
__device__ __forceinline__ void wave_bitonic_keys(uint64_t (&key)[K], const int lane) {
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
                    if (lm == 1) key[k] = cx_lane<1>(key[k], keep_min);
                    else if (lm == 2) key[k] = cx_lane<2>(key[k], keep_min);
                    else if (lm == 4) key[k] = cx_lane<4>(key[k], keep_min);
                    else if (lm == 8) key[k] = cx_lane<8>(key[k], keep_min);
                    else if (lm == 16) key[k] = cx_lane<16>(key[k], keep_min);
                    else key[k] = cx_lane<32>(key[k], keep_min);
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
                        const uint64_t x = key[k], y = key[kp];
                        const bool sw = (y < x) == up;
                        key[k] = sw ? y : x;
                        key[kp] = sw ? x : y;
                    }
                }
            }
        }
    }
}

// one bucket of up to 64 * K values by one wave; SK / SQ: this wave's 64 * K words of LDS each
template <int K>
__device__ __forceinline__ void bucket_keys_only(const BhsArgs& a, const uint64_t* ks, const uint32_t* is, const double* pd,
                                                 int n_b, int64_t seg_off, unsigned start, int bucket, int lane,
                                                 uint64_t* bmin_out, uint64_t* SK, uint64_t* SQ) {
    uint64_t key[K], sk[K];
    uint32_t val[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = k * 64 + lane;              // coalesced; the starting arrangement is arbitrary anyway
        key[k] = ~0ull;
        val[k] = 0u;
        if (p < n_b) {
            if (pd) { key[k] = key_of(pd[p]); val[k] = (uint32_t)p; }
            else { key[k] = __builtin_nontemporal_load(ks + p); val[k] = __builtin_nontemporal_load(is + p); }
        }
        sk[k] = key[k];
    }
    wave_bitonic_keys<K>(sk, lane);
    uint64_t s[K];
    uint64_t run = ~0ull;
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
        const int p = lane * K + k;
        const uint64_t r = p < n_b ? raw_bits(sk[k], (int64_t)start + p + 1, a.m) : ~0ull;
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
        SK[lane * K + k] = sk[k];
        SQ[lane * K + k] = s[k] < ex ? s[k] : ex;
    }
    // (LDS operations of one wave execute in order: the reads below see the writes above)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lower bound of each own key among the sorted keys: branch-free, K independent chains
    int pos[K];
#pragma unroll
    for (int k = 0; k < K; ++k) pos[k] = 0;
#pragma unroll
    for (int st = 32 * K; st >= 1; st >>= 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) pos[k] += SK[pos[k] + st - 1] < key[k] ? st : 0;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k * 64 + lane < n_b) {
            if (!(a.ablate & 1)) a.q_cm[seg_off + val[k]] = SQ[pos[k]];
            if (!(a.ablate & 2)) a.bid_cm[seg_off + val[k]] = (uint16_t)bucket;
            if (a.ablate & 4) a.q_cm[seg_off + start + k * 64 + lane] = SQ[pos[k]];      // (coalesced instead)
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
template <int K>
__global__ void __launch_bounds__(256) bhs_bucket_kernel(BhsArgs a, int blocks_per_seg) {
    __shared__ uint64_t SK[4][64 * K], SQ[4][64 * K];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (scalar: the bucket's addresses stay in SGPRs)
    const int64_t k = (int64_t)(blockIdx.x >> 3);
    const int64_t seg_x = (k / blocks_per_seg) * 8 + (blockIdx.x & 7);
    const int b_x = (int)(k % blocks_per_seg) * 4 + wave;
    if (seg_x >= a.segs || b_x >= a.B) return;
    const int64_t g = seg_x * a.B + b_x;
    const BucketRef r = bucket_ref(a, g);
    if (r.n_b <= 64 * K && r.n_b <= a.reg_cap) {
        bucket_keys_only<K>(a, r.ks, r.is, r.pd, r.n_b, r.seg_off, r.start, r.b, lane, r.bm, SK[wave], SQ[wave]);
    } else if (lane == 0) {
        // (one list per XCD: the big kernel's workgroups keep to the list of "their" segments, so that the scattered
        //  results of a segment still meet in ONE L2 -- a single list in arrival order spread them over all eight)
        const int x = (int)(seg_x & 7);
        const unsigned slot = atomicAdd(&a.big_count[x], 1u);
        a.big_list[(int64_t)x * a.big_region + slot] = g;
    }
}

__global__ void __launch_bounds__(256) bhs_bucket_big_kernel(BhsArgs a) {
    const int lane = threadIdx.x & 63;
    const int x = (int)(blockIdx.x & 7);
    const unsigned n_big = a.big_count[x];
    const unsigned waves = (gridDim.x >> 3) * 4;
    const int64_t* list = a.big_list + (int64_t)x * a.big_region;
#pragma nounroll
    for (unsigned w = (blockIdx.x >> 3) * 4 + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); w < n_big; w += waves) {
        const BucketRef r = bucket_ref(a, list[w]);
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
        // (mean bucket size, same process, 25 000 x 19 900: 96 -> 29.1 ms, 128 -> 23.6, 160 -> 23.3, 200 -> 22.0, 224 -> 28.2,
        //  256 -> 35.4: per-bucket overhead below, the 8 / 16-keys-per-lane kernel for buckets beyond 256 values above)
        int64_t mean = std::max<int64_t>(ctx->param("bh.mean", 200), sd_ceil_div(m, (int64_t)MAX_B));
        B = (int)sd_ceil_div(m, mean);
        if (B > MAX_B) B = MAX_B;
    }
    a.B = B;
    a.ablate = (int)ctx->param("bh.ablate", 0);
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
    a.big_region = sd_ceil_div(segs, (int64_t)8) * B;
    a.big_list = (int64_t*)A.alloc((size_t)a.big_region * 8 * 8);
    a.big_count = (unsigned*)A.alloc(32);
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
        const int64_t tile_blocks = sd_ceil_div(segs, (int64_t)8) * 8 * tiles;
        SD_ARG(tile_blocks < ((int64_t)1 << 31), "bh: too many tiles");
        SD_LAUNCH(ctx, "bhs_count_kernel", (bhs_tile_kernel<false>), dim3((unsigned)tile_blocks), dim3(TILE_T), lds_t, a, (int)tiles);
        SD_LAUNCH(ctx, "bhs_scan_kernel", bhs_scan_kernel, dim3((unsigned)segs), dim3(256), 0, a);
        SD_LAUNCH(ctx, "bhs_scatter_kernel", (bhs_tile_kernel<true>), dim3((unsigned)tile_blocks), dim3(TILE_T), lds_t, a, (int)tiles);
    }
    const int64_t n_buckets = segs * B;
    SD_ARG(sd_ceil_div(n_buckets, (int64_t)4) < ((int64_t)1 << 31), "bh: too many buckets");
    SD_HIP(hipMemsetAsync(a.big_count, 0, 32, ctx->stream));
    const int blocks_per_seg = (int)sd_ceil_div((int64_t)B, (int64_t)4);
    const int64_t bucket_blocks = sd_ceil_div(segs, (int64_t)8) * 8 * blocks_per_seg;
    SD_ARG(bucket_blocks < ((int64_t)1 << 31), "bh: too many buckets");
    // bh.keys: keys per lane of the main bucket kernel (4: buckets of up to 256 values, 8: up to 512); larger buckets go
    // to the (key, index) networks of the second kernel
    if (ctx->param("bh.keys", 4) >= 8)
        SD_LAUNCH(ctx, "bhs_bucket_kernel", (bhs_bucket_kernel<8>), dim3((unsigned)bucket_blocks), dim3(256), 0, a, blocks_per_seg);
    else
        SD_LAUNCH(ctx, "bhs_bucket_kernel", (bhs_bucket_kernel<4>), dim3((unsigned)bucket_blocks), dim3(256), 0, a, blocks_per_seg);
    // (a multiple of 8 workgroups, at least 8: workgroup w serves the list of XCD w mod 8)
    const int64_t big_blocks = std::max<int64_t>(8, sd_ceil_div(std::min<int64_t>(sd_ceil_div(n_buckets, (int64_t)4), (int64_t)ctx->n_cu * 4), (int64_t)8) * 8);
    SD_LAUNCH(ctx, "bhs_bucket_big_kernel", bhs_bucket_big_kernel, dim3((unsigned)big_blocks), dim3(256), 0, a);
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
