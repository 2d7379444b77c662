// K1 + K2, fast path: junction sort and overlap lists in FOUR kernels, no host round trip.
//
// Replaces SPLICEDICE.getClusters (SPLICEDICE.py:230-255; twin counts_to_ps.py:16-41) and the
// junctionIndex sort (SPLICEDICE.py:96), like cluster.hip (which stays as the generic path for more
// than 8 M junctions and as the fallback).  Reference semantics kept: two junctions of one
// chromosome+strand are neighbours iff their closed intervals overlap (prior.right >= cur.left, :250);
// output rows are in (chrom, left, right, strand) order (:96); the list of a junction holds the
// earlier junctions of the (chrom, strand, left, right) sweep most recent first, then the later ones
// in sweep order.
//
// Formulation.  Within one chromosome+strand the sweep order and the row order agree (both are
// (left, right)), so ONE sort into ROW order is enough: the earlier neighbours of row r are the rows
// q < r of its chromosome with the same strand and right_q >= left_r, visited downwards; the later
// ones are the rows q > r with the same strand and left_q <= right_r, visited upwards; rows of the
// other strand that lie in between are stepped over.
//
//   sample_rank_kernel     ranks of a regular (jittered) sample of 8 keys per bucket by brute force
//                          (every sample key against every other, 256 x 256 per workgroup)
//   bucket_scatter_kernel  every 8th ranked sample key is a splitter; a tile of keys is classified by
//                          binary search over the splitters in LDS, counted per bucket in LDS, and
//                          placed into fixed-capacity bucket slots in HBM (one global atomic per
//                          (tile, bucket)); validates the input, takes the maximum junction length
//   bucket_sort_kernel     one workgroup per bucket (mean 2048 keys): bitonic sort of 16-byte
//                          {chrom, left, right<<1|strand, input index} elements in LDS (buckets
//                          beyond the LDS capacity are sorted in place in HBM by the same code);
//                          writes the row-order arrays, row_of, and 64-row maxima of `right`
//   neighbours_kernel      one tile of rows per workgroup: window of the row arrays in LDS, count
//                          walk, workgroup scan, decoupled look-back over the tiles for the global
//                          offset (8-byte {flag, value} granules, agent-scope relaxed atomics),
//                          fill walk into an LDS stage, contiguous stores of row_ptr and col
//
// The backward walk ends at the first row whose left + (maximum junction length) is below the
// target and skips 64-row blocks whose maximum `right` is below it.
//
// Nothing is read back by the host inside the chain: validation results, the total list length and
// the neighbour reach live in a device status block that is fetched when the caller asks for nnz (or
// at the next sdice_sync / sdice_cluster_status); a failed chain leaves row_ptr all zero (safe for a
// dependent PS launch) and the failure is reported then.
#include "common.h"

// Timing experiments (tools/ablate_cluster.py, param cluster.ablate) are compiled in only with
// `make EXTRA=-DSDICE_CLUSTER_ABLATE=1`; the shipped kernels carry no trace of them.
#ifndef SDICE_CLUSTER_ABLATE
#define SDICE_CLUSTER_ABLATE 0
#endif
#define SD_ABL(a, bit) (SDICE_CLUSTER_ABLATE && ((a).ablate & (bit)))

int sd_cluster_legacy(sdice_ctx* ctx, int64_t n, const int32_t* d_chrom, const int32_t* d_left, const int32_t* d_right,
                      const int8_t* d_strand, int32_t* d_row_of, int64_t* d_row_ptr, int64_t* nnz_out);

namespace {

constexpr int SPB = 8;                 // sample keys per bucket
constexpr int BUCKET_MEAN = 2048;      // keys per bucket (mean)
constexpr int MAX_BUCKETS = 4096;      // splitters + per-bucket counters must fit LDS of the scatter kernel
constexpr int SLOT_FACTOR = 8;         // bucket slot capacity in HBM = 8 x mean
constexpr int64_t FAST_MAX_N = (int64_t)MAX_BUCKETS * BUCKET_MEAN;

enum : unsigned long long { ST_INVALID = 1, ST_SLOT_OVERFLOW = 2, ST_DUP = 4, ST_COL_OVERFLOW = 8, ST_LOOKBACK = 16 };
// status block (uint64 words)
constexpr int SB_FLAGS = 0, SB_NNZ = 2, SB_MAXLEN = 32, SB_REACH = 64, SB_WORDS = 96;
// word SB_STICKY lies BEHIND the words a chain clears: every flag is also OR-ed into it, and only the call that reports
// the status clears it -- the failure of an asynchronous chain survives the chains enqueued behind it
constexpr int SB_STICKY = SB_WORDS, SB_ALLOC_WORDS = SB_WORDS + 8;
constexpr int SB_SLOTS = 32;

struct FastArgs {
    const int32_t* chrom; const int32_t* left; const int32_t* right; const int8_t* strand;
    int64_t n;
    int B;                 // buckets
    int S;                 // sample keys (SPB * B, 0 when B == 1)
    int64_t slot_cap;      // elements per bucket slot
    int lds_cap;           // elements the sort kernel may hold in LDS
    unsigned long long* sb;   // status block
    uint32_t* rank;        // [S]
    uint32_t* col_done;    // [ceil(S / 256)] j tiles finished per column of the ranking grid
    uint64_t* spl;         // [B] splitters (B - 1 used)
    uint32_t* cursor;      // [B]
    uint4* slots;          // [B * slot_cap]
    uint32_t* rowC; uint32_t* rowL; uint32_t* rowR2;   // row-order arrays [n]
    uint32_t* bmax64;      // [ceil(n/64)] max right per 64 rows
    unsigned long long* tile_state;   // look-back granules of the neighbour kernel
    int32_t* row_of; int64_t* row_ptr; int32_t* col; int64_t col_cap;
    uint32_t* blkneed;     // [ceil(n/16)] per block of 16 rows: rows its lists reach below the block's first row | beyond its
                           // last row << 8 (each saturating at 255): the PS kernel sizes a tile's halo from it
    int spb;               // sample keys per bucket
    int sample_sort;       // buckets of 512..4096 keys: LDS-local sample sort (param cluster.sample_sort, default 1)
    int ablate;            // timing experiments only (param cluster.ablate): results are wrong when set
};

__device__ __forceinline__ bool key_less(uint32_t ac, uint32_t al, uint32_t ar, uint32_t bc, uint32_t bl, uint32_t br) {
    return ac < bc || (ac == bc && (al < bl || (al == bl && ar < br)));
}
__device__ __forceinline__ bool elem_less(const uint4& a, const uint4& b) { return key_less(a.x, a.y, a.z, b.x, b.y, b.z); }

__device__ __forceinline__ int64_t sample_pos(int64_t i, int64_t n, int64_t S) {
    const int64_t stride = n / S;                              // >= 1: S <= n / 128 by construction
    const uint64_t h = ((uint64_t)i * 0x9E3779B97F4A7C15ull) >> 33;
    return i * stride + (int64_t)(h % (uint64_t)stride);
}

// ------------------------------------------------------------------ 1. ranks of the sample keys
// Sample keys and splitters are the 64-bit prefix (chrom << 32 | left) of the sort key: junctions that
// share chromosome and left end always land in one bucket, and a key comparison is one instruction.
__device__ __forceinline__ uint64_t sample_key(const FastArgs& a, int64_t i) {
    const int64_t p = sample_pos(i, a.n, a.S);
    return ((uint64_t)(uint32_t)a.chrom[p] << 32) | (uint64_t)(uint32_t)a.left[p];
}

// c += (ka, ia) < (kb, ib) in composite order: the borrow of the 96-bit subtraction, added with carry (four VALU
// instructions; the compiler makes five compares and their mask arithmetic of the || / && form)
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

__global__ void __launch_bounds__(256) sample_rank_kernel(FastArgs a) {
    __shared__ __align__(16) uint64_t jk[256];
    __shared__ unsigned last;
    const int t = threadIdx.x;
    // (the status block is first written by the next kernel: cleared here, no launch of its own)
    if (blockIdx.x == 0 && blockIdx.y == 0 && t < SB_WORDS) a.sb[t] = 0ull;
    const int64_t i = (int64_t)blockIdx.x * 256 + t, j = (int64_t)blockIdx.y * 256 + t;
    jk[t] = j < a.S ? sample_key(a, j) : ~0ull;
    const uint64_t ik = i < a.S ? sample_key(a, i) : 0ull;
    __syncthreads();
    // rank of key i among the sample = #{j : key_j < key_i, or equal and j < i}; j runs over this block's 256 keys,
    // two per 16-byte broadcast read; (key_j, q) < (key_i, split) with q = j - j0, split = i - j0 clamped to [0, 256]
    const int64_t j0 = (int64_t)blockIdx.y * 256;
    const int64_t split64 = i - j0;                     // j0 + q < i  <=>  q < split
    const uint32_t split = split64 < 0 ? 0u : split64 > 256 ? 256u : (uint32_t)split64;
    uint32_t cnt = 0;
#pragma unroll 8
    for (int q = 0; q < (SD_ABL(a, 8) ? 0 : 256); q += 2) {
        const ulonglong2 k2 = *reinterpret_cast<const ulonglong2*>(&jk[q]);
        count_less96(cnt, k2.x, (uint32_t)q, ik, split);
        count_less96(cnt, k2.y, (uint32_t)q + 1u, ik, split);
    }
    if (i < a.S && cnt) atomicAdd(&a.rank[i], cnt);
    // the workgroup that completes a column of the grid (all j tiles of these 256 sample keys) reads their final ranks and
    // writes the splitters -- every spb-th sample key in rank order -- so that the scatter kernel's workgroups load B - 1
    // finished keys instead of walking the S ranks (up to a dozen dependent global round trips each).  No fence: the adds
    // are device-scope atomics, performed at the memory side once vmcnt is zero, and the ranks are read back by an atomic
    // (a release fence writes the L2 back: 50 us in the BH kernel of the same shape).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) last = atomicAdd(&a.col_done[blockIdx.x], 1u) == gridDim.y - 1 ? 1u : 0u;
    __syncthreads();
    if (last && i < a.S) {
        const uint32_t rk = atomicAdd(&a.rank[i], 0u);
        if (rk != 0 && rk % (uint32_t)a.spb == 0) a.spl[rk / (uint32_t)a.spb - 1] = ik;                  // slots 0 .. B-2
    }
}

// ------------------------------------------------------------------ 2. classify + scatter into slots
constexpr int SC_KPT = 4;
template <int T>
__global__ void __launch_bounds__(T) bucket_scatter_kernel(FastArgs a) {
    extern __shared__ uint64_t sm64[];
    const int B = a.B;
    uint64_t* spl = sm64;                                        // [B] (B-1 used)
    uint32_t* cnt = reinterpret_cast<uint32_t*>(sm64 + B);       // [B]
    uint32_t* base = cnt + B;                                    // [B]
    __shared__ uint32_t wmax[T / 64];
    __shared__ uint32_t s_bad, s_over;
    const int t = threadIdx.x;
    if (t == 0) { s_bad = 0; s_over = 0; }
    // every spb-th sample key in rank order is a splitter (written by the ranking kernel)
    for (int b = t; b < B; b += T) { spl[b] = b < B - 1 ? a.spl[b] : ~0ull; cnt[b] = 0; }
    __syncthreads();
    const int64_t tile0 = (int64_t)blockIdx.x * T * SC_KPT;
    uint32_t kc[SC_KPT], kl[SC_KPT], kr[SC_KPT], bk[SC_KPT], lr[SC_KPT];
    uint32_t maxlen = 0, bad = 0;
#pragma unroll
    for (int q = 0; q < SC_KPT; ++q) {
        const int64_t i = tile0 + q * T + t;
        bk[q] = 0xffffffffu;
        kc[q] = kl[q] = kr[q] = lr[q] = 0;
        if (i < a.n) {
            const int32_t c = a.chrom[i], l = a.left[i], r = a.right[i];
            const int st = a.strand[i];
            if (l < 0 || r < l || c < 0 || (st != 0 && st != 1)) bad = 1;
            else maxlen = max(maxlen, (uint32_t)(r - l));
            kc[q] = (uint32_t)c; kl[q] = (uint32_t)l; kr[q] = ((uint32_t)r << 1) | (uint32_t)(st & 1);
        }
    }
#pragma unroll
    for (int q = 0; q < SC_KPT; ++q) {
        if (tile0 + q * T + t < a.n) {
            const uint64_t k = ((uint64_t)kc[q] << 32) | kl[q];
            int lo = 0, hi = B - 1;                            // bucket = number of splitters <= key
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (spl[mid] <= k) lo = mid + 1; else hi = mid;
            }
            bk[q] = (uint32_t)lo;
            lr[q] = atomicAdd(&cnt[lo], 1u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o));
    if ((t & 63) == 0) wmax[t >> 6] = maxlen;
    if (bad) s_bad = 1;
    __syncthreads();
    for (int b = t; b < B; b += T) {
        const uint32_t c = cnt[b];
        base[b] = (c && !SD_ABL(a, 32)) ? atomicAdd(&a.cursor[b], c) : 0u;
    }
    if (t == 0) {
        uint32_t m = 0;
        for (int w = 0; w < T / 64; ++w) m = max(m, wmax[w]);
        if (m) atomicMax(&a.sb[SB_MAXLEN + (blockIdx.x % SB_SLOTS)], (unsigned long long)m);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SC_KPT; ++q) {
        if (bk[q] == 0xffffffffu) continue;
        const int64_t i = tile0 + q * T + t;
        const uint64_t pos = (uint64_t)base[bk[q]] + lr[q];
        if (SD_ABL(a, 16)) continue;
        if (pos < (uint64_t)a.slot_cap) a.slots[(int64_t)bk[q] * a.slot_cap + (int64_t)pos] = make_uint4(kc[q], kl[q], kr[q], (uint32_t)i);
        else s_over = 1;
    }
    __syncthreads();
    if (t == 0 && (s_bad | s_over)) {
        const unsigned long long f = (s_bad ? (unsigned long long)ST_INVALID : 0ull) | (s_over ? (unsigned long long)ST_SLOT_OVERFLOW : 0ull);
        atomicOr(&a.sb[SB_FLAGS], f);
        atomicOr(&a.sb[SB_STICKY], f);
    }
}

// ------------------------------------------------------------------ 3. per-bucket sort, row arrays
__device__ __forceinline__ void cmpswap(uint4* buf, int lo, int hi) {
    const uint4 x = buf[lo], y = buf[hi];
    if (elem_less(y, x)) { buf[lo] = y; buf[hi] = x; }
}
__device__ __forceinline__ void cmpswap(uint64_t* buf, int lo, int hi) {
    const uint64_t x = buf[lo], y = buf[hi];
    buf[lo] = y < x ? y : x;                 // unconditional: no exec-mask juggling around two LDS stores
    buf[hi] = y < x ? x : y;
}

// Bitonic network in its "flip + disperse" form: every comparator puts the smaller element at the
// lower index, so positions >= count (virtual +inf) never take part and P need not equal count.
// With comparator c handled by thread c % T, a wave's 64 comparators of a flip with k <= 128 or of a
// disperse with j <= 64 all lie inside one 128-element segment that belongs to that wave alone;
// consecutive such steps need no workgroup barrier (LDS operations of one wave complete in order).
template <bool WAVE_LOCAL_OK, typename E>
__device__ __forceinline__ void bitonic_sort(E* buf, int count, int P, int t, int T) {
    const int half = P >> 1;
    bool prev_confined = false;
    for (int k = 2; k <= P; k <<= 1) {
        {
            const bool confined = WAVE_LOCAL_OK && k <= 128;
            if (confined && prev_confined) __builtin_amdgcn_wave_barrier(); else __syncthreads();
            prev_confined = confined;
            const int hk = k >> 1, lg = __ffs(hk) - 1;
            for (int c = t; c < half; c += T) {
                const int off = c & (hk - 1);
                const int b0 = (c >> lg) << (lg + 1);
                const int lo = b0 + off, hi = b0 + k - 1 - off;
                if (hi < count) cmpswap(buf, lo, hi);
            }
        }
        for (int j = k >> 2; j >= 1; j >>= 1) {
            const bool confined = WAVE_LOCAL_OK && j <= 64;
            if (confined && prev_confined) __builtin_amdgcn_wave_barrier(); else __syncthreads();
            prev_confined = confined;
            for (int c = t; c < half; c += T) {
                const int off = c & (j - 1);
                const int lo = ((c - off) << 1) + off, hi = lo + j;
                if (hi < count) cmpswap(buf, lo, hi);
            }
        }
    }
    __syncthreads();
}

// The same network on 64-bit keys with FOUR elements per thread and TWO levels per pass: a group of
// four elements that is closed under two consecutive levels is loaded once, exchanged in registers
// and stored once -- half the LDS instructions of the one-level form (the sort is bound by the
// LDS instruction rate: 2 reads + 2 writes per comparator there).
//   pass A(k)  = flip(k) + disperse(k/4):  e0 = b+off, e1 = e0+k/4, e3 = b+k-1-off, e2 = e3-k/4
//   pass B(j)  = disperse(j) + disperse(j/2): e0 = b+off, e1 = e0+j/2, e2 = e0+j, e3 = e0+3j/2
//   pass C     = disperse(1) alone (also flip(2)): four consecutive elements
// A wave's 64 groups lie in one 256-element segment of its own whenever the block size (k resp. 2j)
// is <= 256: consecutive such passes need no workgroup barrier.
__device__ __forceinline__ void ce64(uint64_t& a, uint64_t& b) {
    const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}

template <int MODE>
__device__ __forceinline__ void bitonic_pass4(uint64_t* buf, int count, int P, int t, int T, int param) {
    const int lg = MODE == 0 ? __ffs(param >> 2) - 1 : MODE == 1 ? __ffs(param >> 1) - 1 : 0;
    for (int g = t; g < (P >> 2); g += T) {
        int e0, e1, e2, e3;
        if (MODE == 0) {
            const int q = param >> 2, off = g & (q - 1), b0 = (g >> lg) << (lg + 2);
            e0 = b0 + off; e1 = e0 + q; e3 = b0 + param - 1 - off; e2 = e3 - q;
        } else if (MODE == 1) {
            const int h = param >> 1, off = g & (h - 1), b0 = (g >> lg) << (lg + 2);
            e0 = b0 + off; e1 = e0 + h; e2 = e1 + h; e3 = e2 + h;
        } else {
            e0 = g << 2; e1 = e0 + 1; e2 = e0 + 2; e3 = e0 + 3;
        }
        if (e0 >= count) continue;                             // (e0 is the lowest index of the group)
        uint64_t v0 = buf[e0];
        uint64_t v1 = e1 < count ? buf[e1] : ~0ull, v2 = e2 < count ? buf[e2] : ~0ull, v3 = e3 < count ? buf[e3] : ~0ull;
        if (MODE == 0) { ce64(v0, v3); ce64(v1, v2); ce64(v0, v1); ce64(v2, v3); }
        else if (MODE == 1) { ce64(v0, v2); ce64(v1, v3); ce64(v0, v1); ce64(v2, v3); }
        else { ce64(v0, v1); ce64(v2, v3); }
        buf[e0] = v0;
        if (e1 < count) buf[e1] = v1;
        if (e2 < count) buf[e2] = v2;
        if (e3 < count) buf[e3] = v3;
    }
}

__device__ __forceinline__ void bitonic_sort_u64(uint64_t* buf, int count, int P, int t, int T) {
    if (P < 4) P = 4;
    bool prev_confined = false;
    auto sync = [&](bool confined) {
        if (confined && prev_confined) __builtin_amdgcn_wave_barrier(); else __syncthreads();
        prev_confined = confined;
    };
    sync(true);
    bitonic_pass4<2>(buf, count, P, t, T, 0);                  // k = 2
    for (int k = 4; k <= P; k <<= 1) {
        sync(k <= 256);
        bitonic_pass4<0>(buf, count, P, t, T, k);              // flip(k), disperse(k/4)
        int j = k >> 3;
        while (j >= 2) {
            sync(j <= 128);
            bitonic_pass4<1>(buf, count, P, t, T, j);          // disperse(j), disperse(j/2)
            j >>= 2;
        }
        if (j == 1) {
            sync(true);
            bitonic_pass4<2>(buf, count, P, t, T, 0);          // disperse(1)
        }
    }
    __syncthreads();
}

constexpr int SORT_T_ = 1024;
// ---- sorting a bucket's packed keys WITHOUT the workgroup-wide network: an LDS-local sample sort.
// The network above costs ~66 dependent LDS round trips for a 2 048-key bucket, most of them behind a
// workgroup barrier -- ~50 us per workgroup, and the whole grid is one round of workgroups, so that IS
// the kernel's time.  Here: 256 regular samples ranked against each other by all threads (keys are
// distinct: the slot index rides in the low bits), every 16th a splitter; each key finds its sub-bucket
// (4 probes), one LDS atomic per (wave, sub-bucket) hands out offsets, a list of 16-bit positions per
// sub-bucket goes to the unused upper half of the key buffer; then ONE WAVE per sub-bucket gathers its
// keys and sorts them in registers (2 / 4 / 8 keys per lane, cross-lane steps by ds_bpermute, no LDS
// storage, no barrier) and writes them back in order.  Four workgroup barriers instead of ~25.
template <int K>
__device__ __forceinline__ void wave_sort_u64(uint64_t (&key)[K], const int lane) {
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
                    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(key[k] & 0xffffffffu), lm);
                    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(key[k] >> 32), lm);
                    const uint64_t o = ((uint64_t)hi << 32) | lo;
                    const bool take = keep_min ? o < key[k] : o > key[k];
                    key[k] = take ? o : key[k];
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
                        key[k] = sw ? key[kp] : tk;
                        key[kp] = sw ? tk : key[kp];
                    }
                }
            }
        }
    }
}

constexpr int SS_NB = 16;            // sub-buckets = waves of the workgroup
constexpr int SS_MIN = 512, SS_MAX = 6400, SS_SUB_CAP = 1024;
constexpr int SS_KPT = (SS_MAX + SORT_T_ - 1) / SORT_T_;      // keys per thread
static_assert((8192 - SS_MAX) * 8 >= SS_MAX * 2 && (8192 - SS_MAX) * 8 >= 256 * 12, "scratch behind the keys must hold the lists / the sample");

template <int K>
__device__ __forceinline__ void ss_sort_sub(const uint64_t* kb, const unsigned short* list, int n, int lane, uint64_t (&key)[16]) {
    uint64_t k_[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = k * 64 + lane;
        k_[k] = p < n ? kb[list[p]] : ~0ull;
    }
    wave_sort_u64<K>(k_, lane);
#pragma unroll
    for (int k = 0; k < K; ++k) key[k] = k_[k];
}

// true: kb[0..count) is sorted.  false (a sub-bucket above SS_SUB_CAP keys): kb is untouched.
// The 8192-key buffer holds `count` keys; what lies behind them is scratch: first the sample and its ranks,
// then the per-sub-bucket lists of 16-bit key positions.
__device__ __forceinline__ bool lds_sample_sort(uint64_t* kb, int count, int t) {
    __shared__ unsigned s_cnt[SS_NB], s_base[SS_NB + 1], s_max;
    __shared__ uint64_t s_spl[SS_NB];
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    uint64_t* smp = kb + count;
    const int ns = count <= 2048 ? 128 : 256;                           // 8 or 16 samples per sub-bucket
    const int per = ns / SS_NB;
    unsigned* rnk = reinterpret_cast<unsigned*>(smp + ns);
    unsigned short* list = reinterpret_cast<unsigned short*>(kb + count);    // (after the sample is done with)
    __syncthreads();
    if (t < ns) { smp[t] = kb[(int)((int64_t)t * count / ns)]; rnk[t] = 0u; }
    if (t < SS_NB) s_cnt[t] = 0u;
    __syncthreads();
    {
        // every thread ranks one sample against one chunk of the sample (keys are distinct)
        const int i = t & (ns - 1), chunk = SORT_T_ / ns, c = t / ns, len = ns / chunk;
        const uint64_t me = smp[i];
        unsigned below = 0;
        for (int j = c * len; j < c * len + len; ++j) below += smp[j] < me ? 1u : 0u;
        atomicAdd(&rnk[i], below);
    }
    __syncthreads();
    if (t < ns) {
        const unsigned r = rnk[t];
        if (r > 0 && r % (unsigned)per == 0) s_spl[r / (unsigned)per - 1] = smp[t];
    }
    __syncthreads();
    // sub-bucket and offset of this thread's keys, packed (offset << 4 | sub-bucket), -1 = none
    int where[SS_KPT];
#pragma unroll
    for (int q = 0; q < SS_KPT; ++q) {
        const int i = q * SORT_T_ + t;
        const bool act = i < count;
        int b = 0;
        if (act) {
            const uint64_t key = kb[i];
            int lo = 0, hi = SS_NB - 1;                                 // number of splitters <= key
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const bool le = s_spl[mid] <= key;
                lo = le ? mid + 1 : lo;
                hi = le ? hi : mid;
            }
            b = lo;
        }
        unsigned long long m = __ballot(act);
#pragma unroll
        for (int bit = 0; bit < 4; ++bit) {
            const unsigned long long bb = __ballot((b >> bit) & 1);
            m &= ((b >> bit) & 1) ? bb : ~bb;
        }
        unsigned basew = 0;
        if (act) {
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) basew = atomicAdd(&s_cnt[b], (unsigned)__popcll(m));
            basew = (unsigned)__shfl((int)basew, leader);
        }
        where[q] = act ? (int)(((basew + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) << 4) | (unsigned)b) : -1;
    }
    __syncthreads();
    if (w == 0) {
        const unsigned c = lane < SS_NB ? s_cnt[lane] : 0u;
        unsigned x = c, mx = c;
#pragma unroll
        for (int o = 1; o < SS_NB; o <<= 1) {
            const unsigned y = (unsigned)__shfl_up((int)x, o);
            if (lane >= o) x += y;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        if (lane < SS_NB) s_base[lane + 1] = x;
        if (lane == 0) { s_base[0] = 0u; s_max = mx; }
    }
    __syncthreads();
    if (s_max > (unsigned)SS_SUB_CAP) return false;
#pragma unroll
    for (int q = 0; q < SS_KPT; ++q)
        if (where[q] >= 0) list[s_base[where[q] & 15] + (unsigned)(where[q] >> 4)] = (unsigned short)(q * SORT_T_ + t);
    __syncthreads();
    const int n = (int)s_cnt[w];
    const unsigned base = s_base[w];
    uint64_t key[16];
    const int K = n <= 128 ? 2 : n <= 256 ? 4 : n <= 512 ? 8 : 16;
    if (K == 2) ss_sort_sub<2>(kb, list + base, n, lane, key);
    else if (K == 4) ss_sort_sub<4>(kb, list + base, n, lane, key);
    else if (K == 8) ss_sort_sub<8>(kb, list + base, n, lane, key);
    else ss_sort_sub<16>(kb, list + base, n, lane, key);
    __syncthreads();                                                    // every wave holds its keys: kb may be overwritten
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int p = lane * K + k;
        if (k < K && p < n) kb[base + p] = key[k];
    }
    __syncthreads();
    return true;
}

constexpr int SORT_T = SORT_T_;
constexpr int SORT_WAVES = SORT_T / 64;
constexpr int RDX_CAP = 8192;                    // packed 64-bit keys a bucket may hold in LDS
constexpr int RDX_IDX_BITS = 13;                 // slot index of an element rides in the low bits of its key
constexpr int SORT_LDS_ELEMS = 4096;             // 16-byte elements of the unpacked fallback (same LDS)
constexpr size_t SORT_LDS_BYTES = (size_t)RDX_CAP * 8;     // 64 KB: two workgroups per CU
static_assert(SORT_LDS_BYTES >= (size_t)SORT_LDS_ELEMS * 16, "unpacked fallback must fit");

__device__ __forceinline__ int bits_u32(uint32_t v) { return v ? 32 - __clz(v) : 0; }

__global__ void __launch_bounds__(SORT_T, 8) bucket_sort_kernel(FastArgs a) {      // 64 VGPRs: two workgroups per CU
    extern __shared__ uint4 lds_elems[];
    __shared__ unsigned long long wsum[SORT_WAVES];
    __shared__ uint32_t red[5][SORT_WAVES];
    __shared__ uint32_t s_dup;
    const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    if (a.sb[SB_FLAGS] & (ST_INVALID | ST_SLOT_OVERFLOW)) return;      // uniform: written by the previous kernel
    const int count = (int)a.cursor[b];
    // first row of this bucket = number of keys in the buckets before it
    unsigned long long part = 0;
    for (int i = t; i < b; i += SORT_T) part += a.cursor[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) wsum[w] = part;
    if (t == 0) s_dup = 0;
    __syncthreads();
    int64_t start = 0;
    for (int k = 0; k < SORT_WAVES; ++k) start += (int64_t)wsum[k];
    if (count == 0) return;
    uint4* src = a.slots + (int64_t)b * a.slot_cap;

    // ---- choose the path: packed local keys + LDS radix, LDS bitonic, or in-place bitonic in HBM
    int total_bits = 99, bl = 0, bn = 0;
    uint32_t cmin = 0, lmin = 0;
    const bool small = count <= RDX_CAP && count <= a.lds_cap;
    if (small) {
        uint32_t c0 = 0xffffffffu, c1 = 0, l0 = 0xffffffffu, l1 = 0, n1 = 0;
        for (int i = t; i < count; i += SORT_T) {
            const uint4 e = src[i];
            c0 = min(c0, e.x); c1 = max(c1, e.x); l0 = min(l0, e.y); l1 = max(l1, e.y); n1 = max(n1, (e.z >> 1) - e.y);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c0 = min(c0, (uint32_t)__shfl_xor((int)c0, o)); c1 = max(c1, (uint32_t)__shfl_xor((int)c1, o));
            l0 = min(l0, (uint32_t)__shfl_xor((int)l0, o)); l1 = max(l1, (uint32_t)__shfl_xor((int)l1, o));
            n1 = max(n1, (uint32_t)__shfl_xor((int)n1, o));
        }
        if (lane == 0) { red[0][w] = c0; red[1][w] = c1; red[2][w] = l0; red[3][w] = l1; red[4][w] = n1; }
        __syncthreads();
        for (int k = 0; k < SORT_WAVES; ++k) {
            c0 = min(c0, red[0][k]); c1 = max(c1, red[1][k]); l0 = min(l0, red[2][k]); l1 = max(l1, red[3][k]);
            n1 = max(n1, red[4][k]);
        }
        cmin = c0; lmin = l0;
        bl = bits_u32(l1 - l0); bn = bits_u32(n1);
        total_bits = bits_u32(c1 - c0) + bl + bn + 1;
    }
    uint32_t dup = 0;
    // rows, row_of, 64-row maxima for the sorted element i (fetched by `get`); `same_as_next` flags a duplicate key
    auto emit_all = [&](auto get) {
        for (int i0 = 0; i0 < count; i0 += SORT_T) {
            const int i = i0 + t;
            const bool act = i < count;
            uint32_t rr = 0, blk = 0xffffffffu;
            if (act) {
                const uint4 e = get(i);
                const int64_t r = start + i;
                a.rowC[r] = e.x; a.rowL[r] = e.y; a.rowR2[r] = e.z;
                a.row_of[e.w] = (int32_t)r;
                rr = e.z >> 1;
                blk = (uint32_t)(r >> 6);
            }
            // 64 consecutive rows touch at most two 64-row blocks: one maximum for each
            const uint32_t blk0 = (uint32_t)__shfl((int)blk, 0);
            uint32_t m0 = (act && blk == blk0) ? rr : 0u, m1 = (act && blk != blk0) ? rr : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                m0 = max(m0, (uint32_t)__shfl_xor((int)m0, o));
                m1 = max(m1, (uint32_t)__shfl_xor((int)m1, o));
            }
            if (lane == 0 && blk0 != 0xffffffffu) {
                if (m0) atomicMax(&a.bmax64[blk0], m0);
                if (m1) atomicMax(&a.bmax64[blk0 + 1], m1);
            }
        }
    };
    if (small && total_bits + RDX_IDX_BITS <= 64) {
        uint64_t* kb0 = reinterpret_cast<uint64_t*>(lds_elems);
        for (int i = t; i < count; i += SORT_T) {
            const uint4 e = src[i];
            uint64_t k = (((uint64_t)(e.x - cmin) << bl) | (uint64_t)(e.y - lmin)) << bn;
            k = ((k | (uint64_t)((e.z >> 1) - e.y)) << 1) | (uint64_t)(e.z & 1u);
            kb0[i] = (k << RDX_IDX_BITS) | (uint64_t)i;
        }
        // bitonic network on the packed keys: one LDS round trip per step, most steps wave-local.
        // (An LSD radix with per-wave counters was measured at ~5 us per 9-bit pass -- four barriers
        //  and a 8192-counter scan each -- against ~1 us per ten steps of this network.)
        int P = 1;
        while (P < count) P <<= 1;
        bool sorted_already = false;
        if (a.sample_sort && count >= SS_MIN && count <= SS_MAX && !SD_ABL(a, 1)) sorted_already = lds_sample_sort(kb0, count, t);
        if (sorted_already) {}
        else if (P > 1 && !SD_ABL(a, 1)) bitonic_sort_u64(kb0, count, P, t, SORT_T); else __syncthreads();
        const uint64_t* sorted = kb0;
        for (int i = t; i + 1 < count; i += SORT_T)
            if ((sorted[i] >> RDX_IDX_BITS) == (sorted[i + 1] >> RDX_IDX_BITS)) dup = 1;
        if (!SD_ABL(a, 2)) emit_all([&](int i) { return src[sorted[i] & ((1u << RDX_IDX_BITS) - 1u)]; });
    } else {
        int P = 1;
        while (P < count) P <<= 1;
        uint4* buf;
        if (count <= SORT_LDS_ELEMS && count <= a.lds_cap) {
            for (int i = t; i < count; i += SORT_T) lds_elems[i] = src[i];
            buf = lds_elems;
            if (P > 1) bitonic_sort<true, uint4>(buf, count, P, t, SORT_T); else __syncthreads();
        } else {
            buf = src;                                         // rare: sorted in place in HBM
            bitonic_sort<false, uint4>(buf, count, P, t, SORT_T);
        }
        for (int i = t; i + 1 < count; i += SORT_T) {
            const uint4 e = buf[i], f = buf[i + 1];
            if (f.x == e.x && f.y == e.y && f.z == e.z) dup = 1;
        }
        emit_all([&](int i) { return buf[i]; });
    }
    if (dup) s_dup = 1;
    __syncthreads();
    if (t == 0 && s_dup) { atomicOr(&a.sb[SB_FLAGS], (unsigned long long)ST_DUP); atomicOr(&a.sb[SB_STICKY], (unsigned long long)ST_DUP); }
}

// ------------------------------------------------------------------ 4. neighbour lists
constexpr unsigned long long LB_A = 1ull << 62, LB_P = 2ull << 62, LB_MASK = (1ull << 62) - 1;

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exclusive prefix of `agg` over the tiles before `tile` (called by the 64 lanes of one wave).
// Forward progress: EVERY tile is drawn from a device-side ticket counter by a workgroup that is already running, in
// increasing order, and a workgroup publishes a tile's aggregate before it waits for anything.  The tiles before a given
// one therefore all belong to workgroups that hold a CU and never wait on a later tile: every wait ends, whatever the
// dispatcher does and whoever else is on the device.  (Rounds 1-3 launched one workgroup per tile and relied on in-order
// dispatch.  The first version of this loop gave a workgroup its FIRST tile by block index: with two processes on one GPU
// -- two neighbour kernels of capped grids, each holding CUs the other's not-yet-dispatched workgroups needed -- both
// chains waited on tiles nobody had started and gave up; `tools/gpu_rehearse2.sh`.)  The grid is capped at what the device
// holds at once (launch_neighbours) only to save dispatches.  The spin stays bounded all the same: a wait that outlasts
// ~1 s gives up, flags ST_LOOKBACK (the chain's result is then discarded: the synchronous caller falls back to the generic
// path, the asynchronous one reports it at the next sync) and the grid drains.
// fail: cluster.ablate & 256 (tests; the one ablation bit that is compiled into the shipped kernel): tile 1 gives up at once.
__device__ __forceinline__ unsigned long long lookback_exclusive(unsigned long long* state, int tile, unsigned long long agg,
                                                                 int lane, unsigned long long* sb, bool fail) {
    if (lane == 0) st_agent(&state[tile], (tile == 0 ? LB_P : LB_A) | agg);
    unsigned long long excl = 0;
    int idx = tile - 1;
    unsigned spins = fail && tile == 1 ? (1u << 21) : 0u;
    while (idx >= 0) {
        const int j = idx - lane;
        const unsigned long long v = j >= 0 ? ld_agent(&state[j]) : LB_P;
        const unsigned flag = (unsigned)(v >> 62);
        const unsigned long long empty = __ballot(flag == 0u);
        const unsigned long long pm = __ballot(flag == 2u);
        const int first_p = pm ? __ffsll((long long)pm) - 1 : 64;
        const unsigned long long need = first_p >= 63 ? ~0ull : ((1ull << (first_p + 1)) - 1ull);
        if ((empty & need) || (fail && tile == 1)) {
            if (++spins > (1u << 21)) {
                if (lane == 0) { atomicOr(&sb[SB_FLAGS], (unsigned long long)ST_LOOKBACK); atomicOr(&sb[SB_STICKY], (unsigned long long)ST_LOOKBACK); }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        unsigned long long x = lane <= first_p ? (v & LB_MASK) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        excl += x;
        if (first_p < 64) break;
        idx -= 64;
    }
    if (lane == 0 && tile != 0) st_agent(&state[tile], LB_P | (excl + agg));
    return excl;
}

// Tile of T rows per workgroup, tile = workgroup id (see lookback_exclusive).
template <int T>
struct NbCfg {
    static constexpr int HB = 256;             // rows staged before the tile
    static constexpr int HF = 128;             // rows staged after it
    static constexpr int W = T + HB + HF;
    static constexpr int STAGE = 11 * T;       // list entries staged in LDS per tile (11, not 12: 40 120 B per workgroup -> FOUR per CU)
    static constexpr int KMAX = 16;            // entries a row keeps from its (single) walk, as 16-bit row distances
    static constexpr int WIN_WORDS = 6 * W;    // cL, running maximum (64 bit each), right, window index + position (16 bit each)
    static constexpr int UNION_WORDS = WIN_WORDS > STAGE ? WIN_WORDS : STAGE;
};

// walks outside the staged window (long junctions, dense loci, rows that redo their list): row arrays
// from global memory, 64-row blocks without a reaching `right` skipped, end at left + maxlen < target
template <class Emit>
__device__ __forceinline__ void walk_back_global(const FastArgs& a, int q, uint32_t c, uint32_t l, uint32_t st, uint32_t maxlen,
                                                 Emit emit) {
    while (q >= 0) {
        const uint32_t qc = a.rowC[q], ql = a.rowL[q], qr = a.rowR2[q];
        if (qc != c || ql + maxlen < l) break;
        if ((q & 63) == 63 && a.bmax64[q >> 6] < l) { q -= 64; continue; }
        if ((qr & 1u) == st && (qr >> 1) >= l) emit(q);
        --q;
    }
}
template <class Emit>
__device__ __forceinline__ void walk_fwd_global(const FastArgs& a, int q, int n, uint32_t c, uint32_t rgt, uint32_t st, Emit emit) {
    while (q < n) {
        const uint32_t qc = a.rowC[q], ql = a.rowL[q], qr = a.rowR2[q];
        if (qc != c || ql > rgt) break;
        if ((qr & 1u) == st) emit(q);
        ++q;
    }
}

template <int T>
__global__ void __launch_bounds__(T, 8) neighbours_kernel(FastArgs a) {      // four workgroups per CU (three: +3 % on the quant step)
    typedef NbCfg<T> Cfg;
    constexpr int W = Cfg::W, KMAX = Cfg::KMAX;
    constexpr int EPT = (W + T - 1) / T;       // window rows per thread while the window is built
    // The window [wlo, whi) is held COMPACTED BY STRAND, so that a walk only ever visits rows of its own
    // strand (half the visits of a walk over the interleaved rows): the '+' rows fill the arrays from the
    // front in row order, the '-' rows from the back in reverse order, position k holds
    //   cL[k]  = (chrom - chrom[window start]) << 32 | left      forward walks end at cL > (c << 32 | right)
    //   pm[k]  = running maximum over the window rows of this strand up to this one of (chrom - ..) << 32 | right:
    //            a backward walk for (c, left) ends at the first row with pm < (c << 32 | left) -- nothing at
    //            or before it (of this chromosome and strand) reaches `left`
    //   wR[k]  = right,  rid[k] = window index of the row,  pos[i] = position of window row i.
    // The same LDS is the list stage afterwards.
    __shared__ __align__(16) uint32_t u_mem[Cfg::UNION_WORDS];
    __shared__ int16_t tmp[(KMAX + 1) * T];     // (row KMAX is a dummy target for non-hits)
    __shared__ uint32_t wsum[T / 64];
    __shared__ unsigned long long wmax0[T / 64], wmax1[T / 64];
    __shared__ uint32_t s_misc[4];             // [1] maxlen, [2] reach
    __shared__ unsigned long long s_base;
    unsigned long long* cL = reinterpret_cast<unsigned long long*>(u_mem);
    unsigned long long* pm = cL + W;
    uint32_t* wR = reinterpret_cast<uint32_t*>(pm + W);
    uint16_t* rid = reinterpret_cast<uint16_t*>(wR + W);
    uint16_t* pos = rid + W;
    int32_t* stage = reinterpret_cast<int32_t*>(u_mem);
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = (int)a.n;
    const int n_tiles = (n + T - 1) / T;
    const bool failed_chain = (a.sb[SB_FLAGS] & (ST_INVALID | ST_SLOT_OVERFLOW | ST_DUP)) != 0;
    // a workgroup draws its tiles from a counter (see lookback_exclusive; a workgroup that drew cheap tiles takes more of
    // them, as the dispatcher would have arranged: a fixed stride cost 30 % of the kernel at 2 M junctions).  A failed
    // chain (no look-back) walks by stride.
    __shared__ int s_tile;
    unsigned long long* ticket = a.tile_state + n_tiles;
    if (t == 0) s_tile = failed_chain ? (int)blockIdx.x : (int)atomicAdd(ticket, 1ull);
    __syncthreads();
#pragma nounroll
    for (int tile = s_tile; tile < n_tiles;) {
    const int t0 = tile * T;
    const int nr = min(T, n - t0);
    if (failed_chain) {
        // failed chain: leave an all-zero row_ptr (every list empty) so that a dependent launch stays in bounds
        if (t < nr) { a.row_ptr[t0 + t] = 0; if ((t & 15) == 0) a.blkneed[(t0 + t) >> 4] = 0; }
        if (t0 + t == 0) a.row_ptr[n] = 0;
        tile += gridDim.x;                                  // (nothing to balance)
        continue;
    }
    if (t == 0) s_misc[2] = 0;
    if (t < 64) {
        unsigned long long m = t < SB_SLOTS ? a.sb[SB_MAXLEN + t] : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(m, o); m = y > m ? y : m; }
        if (t == 0) s_misc[1] = (uint32_t)m;
    }
    const int wlo = max(0, t0 - Cfg::HB), whi = min(n, t0 + nr + Cfg::HF);
    const int wn = whi - wlo;
    const uint32_t c0w = a.rowC[wlo];
    const uint32_t l0w = a.rowL[wlo];
    const int64_t cap = a.col_cap;
    int n1 = 0;                                // '-' rows in the window
    {
        // thread t builds window rows t*EPT .. (read straight from the row arrays): scan of the '-' count
        // (position inside the strand) and of the two running maxima, then the compacted stores
        unsigned long long comp[EPT], own_cl[EPT], r0 = 0, r1 = 0;
        uint32_t rr[EPT], str[EPT], c1 = 0;
        unsigned long long v0[EPT], v1[EPT];
        uint32_t cb[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = t * EPT + e;
            comp[e] = 0; own_cl[e] = 0; rr[e] = 0; str[e] = 0;
            cb[e] = c1;
            if (i < wn) {
                const uint32_t rc = a.rowC[wlo + i] - c0w, r2 = a.rowR2[wlo + i];
                own_cl[e] = ((unsigned long long)rc << 32) | a.rowL[wlo + i];
                rr[e] = r2 >> 1; str[e] = r2 & 1u;
                comp[e] = ((unsigned long long)rc << 32) | rr[e];
                if (str[e]) { r1 = comp[e] > r1 ? comp[e] : r1; ++c1; } else { r0 = comp[e] > r0 ? comp[e] : r0; }
            }
            v0[e] = r0; v1[e] = r1;
        }
        unsigned long long x0 = r0, x1 = r1;
        uint32_t xc = c1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long y0 = __shfl_up(x0, o), y1 = __shfl_up(x1, o);
            const uint32_t yc = __shfl_up(xc, o);
            if (lane >= o) { x0 = y0 > x0 ? y0 : x0; x1 = y1 > x1 ? y1 : x1; xc += yc; }
        }
        if (lane == 63) { wmax0[w] = x0; wmax1[w] = x1; wsum[w] = xc; }
        __syncthreads();
        unsigned long long p0 = __shfl_up(x0, 1), p1 = __shfl_up(x1, 1);      // exclusive over the threads before
        uint32_t pc = xc - c1;
        if (lane == 0) { p0 = 0; p1 = 0; }
#pragma unroll
        for (int k = 0; k < T / 64; ++k) {
            if (k < w) { p0 = wmax0[k] > p0 ? wmax0[k] : p0; p1 = wmax1[k] > p1 ? wmax1[k] : p1; pc += wsum[k]; }
            n1 += (int)wsum[k];
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = t * EPT + e;
            if (i < wn) {
                const int before1 = (int)(pc + cb[e]);             // '-' rows before window row i
                const int k = str[e] ? W - 1 - before1 : i - before1;
                const unsigned long long m = str[e] ? (v1[e] > p1 ? v1[e] : p1) : (v0[e] > p0 ? v0[e] : p0);
                cL[k] = own_cl[e]; pm[k] = m; wR[k] = rr[e]; rid[k] = (uint16_t)i; pos[i] = (uint16_t)k;
            }
        }
    }
    __syncthreads();                           // (wsum is reused by the degree scan below: all reads above are done)
    const uint32_t maxlen = s_misc[1];
    const int n0 = wn - n1;                    // '+' rows occupy [0, n0), '-' rows [W - n1, W)

    const int r = t0 + t;
    const int ri = r - wlo;
    const bool act = t < nr;
    int p = 0;
    unsigned long long own = 0;
    uint32_t rgt = 0;
    if (act) { p = pos[ri]; own = cL[p]; rgt = wR[p]; }
    const uint32_t st = p >= W - n1 ? 1u : 0u;
    const uint32_t l = (uint32_t)own, crel = (uint32_t)(own >> 32), c = crel + c0w;
    const unsigned long long tb = own;                                          // (chrom, left)
    const unsigned long long tf = (own & 0xffffffff00000000ull) | rgt;          // (chrom, right)
    // the window maxima decide the end of the backward walk iff no row before the window can reach `left`
    const bool exact = wlo == 0 || crel != 0 || l0w + maxlen < l;
    // strand-local direction of increasing row number, and the bounds of this strand's region
    const int dfw = st ? -1 : 1;
    const int first = st ? W - 1 : 0;                      // position of the strand's first window row
    const int last = st ? W - n1 : n0 - 1;                 // ... and of its last one

    uint32_t deg = 0;
    int far_b = r, far_f = r;
    bool redo = false;                         // the 16-bit list of this row is incomplete
    // rows reached outside the window (rare): distance may not fit 16 bits
    auto emit_far = [&](int q) {
        const int d = q - r;
        if (deg < (uint32_t)KMAX) {
            if (d >= -32768 && d <= 32767) tmp[deg * T + t] = (int16_t)d; else redo = true;
        }
        ++deg;
        if (d < 0) far_b = q; else far_f = q;
    };
    if (act && !SD_ABL(a, 64)) {
        // The loops are written without conditional statements (loads at a clamped position, a dummy 17th
        // list row for the non-hits, two rows per trip): as nested ifs they compiled to two serialised LDS
        // round trips and ~25 exec-mask instructions per row.
        int16_t* mytmp = tmp + t;
        auto take = [&](bool hit, int wi) {
            const uint32_t slot = (hit & (deg < (uint32_t)KMAX)) ? deg : (uint32_t)KMAX;
            mytmp[slot * T] = (int16_t)(wi - ri);
            deg += hit ? 1u : 0u;
            return hit ? wlo + wi : -1;
        };
        // in strand-local coordinates u = (position - first) * dfw  (0 = the strand's first window row)
        const int up = (p - first) * dfw, ulast = (last - first) * dfw;
        // ---- earlier rows of this strand, most recent first
        if (exact) {
            int u = up - 1;
            for (;;) {
                const int q0 = first + dfw * max(u, 0), q1 = first + dfw * max(u - 1, 0);
                const unsigned long long m0 = pm[q0], m1 = pm[q1];
                const uint32_t x0 = wR[q0], x1 = wR[q1];
                const int i0 = rid[q0], i1 = rid[q1];
                if (!((u >= 0) & (m0 >= tb))) break;
                const int f0 = take(x0 >= l, i0);              // (pm >= target: the row is of this chromosome)
                far_b = f0 >= 0 ? f0 : far_b;
                --u;
                if (!((u >= 0) & (m1 >= tb))) break;
                const int f1 = take(x1 >= l, i1);
                far_b = f1 >= 0 ? f1 : far_b;
                --u;
            }
            if (u < 0 && wlo > 0) walk_back_global(a, wlo - 1, c, l, st, maxlen, emit_far);
        } else {
            walk_back_global(a, r - 1, c, l, st, maxlen, emit_far);
        }
        // ---- later rows of this strand in order: every row up to the first with (chrom, left) > (c, right)
        {
            int u = up + 1;
            for (;;) {
                const int q0 = first + dfw * min(u, ulast), q1 = first + dfw * min(u + 1, ulast);
                const unsigned long long c0 = cL[q0], c1 = cL[q1];
                const int i0 = rid[q0], i1 = rid[q1];
                if (!((u <= ulast) & (c0 <= tf))) break;
                far_f = take(true, i0);
                ++u;
                if (!((u <= ulast) & (c1 <= tf))) break;
                far_f = take(true, i1);
                ++u;
            }
            if (u > ulast && whi < n) walk_fwd_global(a, whi, n, c, rgt, st, emit_far);
        }
        if (deg > (uint32_t)KMAX) redo = true;
    }
    uint32_t reach = (uint32_t)max(r - far_b, far_f - r);
    {
        // per 16-row block (tiles start at multiples of 16): how far the lists reach beyond the block on either side
        const int b16 = t & 15;
        const unsigned lo_n = act ? (unsigned)min(max(r - far_b - b16, 0), 255) : 0u;
        const unsigned hi_n = act ? (unsigned)min(max(far_f - r - (15 - b16), 0), 255) : 0u;
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        us2 pk = {(unsigned short)lo_n, (unsigned short)hi_n};
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const unsigned y = (unsigned)__shfl_xor((int)__builtin_bit_cast(unsigned, pk), o);
            pk = __builtin_elementwise_max(pk, __builtin_bit_cast(us2, y));
        }
        if (act && b16 == 0) a.blkneed[r >> 4] = (uint32_t)pk.x | ((uint32_t)pk.y << 8);
    }
    // ---- workgroup exclusive scan of the degrees, look-back for the global offset
    uint32_t x = deg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) reach = max(reach, (uint32_t)__shfl_xor((int)reach, o));
    if (lane == 63) wsum[w] = x;
    if (lane == 0 && reach) atomicMax(&s_misc[2], reach);
    __syncthreads();                           // (also: every walk over the window is done, the stage may overwrite it)
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int k = 0; k < T / 64; ++k) { const uint32_t v = wsum[k]; if (k < w) wbase += v; total += v; }
    const uint32_t loff = wbase + x - deg;
    if (w == 0) {
        const unsigned long long base = SD_ABL(a, 128) ? 0ull : lookback_exclusive(a.tile_state, tile, (unsigned long long)total, lane, a.sb, (a.ablate & 256) != 0);
        if (lane == 0) {
            s_base = base;
            if (s_misc[2]) atomicMax(&a.sb[SB_REACH + (tile % SB_SLOTS)], (unsigned long long)s_misc[2]);
        }
    }
    const bool staged = total <= (uint32_t)Cfg::STAGE;
    if (staged && act && deg) {
        if (!redo) {
            for (uint32_t k = 0; k < deg; ++k) stage[loff + k] = r + (int)tmp[k * T + t];
        } else {
            uint32_t k = loff;
            walk_back_global(a, r - 1, c, l, st, maxlen, [&](int q) { stage[k++] = q; });
            walk_fwd_global(a, r + 1, n, c, rgt, st, [&](int q) { stage[k++] = q; });
        }
    }
    __syncthreads();
    const int64_t base = (int64_t)s_base;
    if (act) a.row_ptr[r] = min(base + (int64_t)loff, cap);
    if (tile == n_tiles - 1 && t == 0) {
        a.row_ptr[n] = min(base + (int64_t)total, cap);
        a.sb[SB_NNZ] = (unsigned long long)(base + (int64_t)total);
        if (base + (int64_t)total > cap) { atomicOr(&a.sb[SB_FLAGS], (unsigned long long)ST_COL_OVERFLOW); atomicOr(&a.sb[SB_STICKY], (unsigned long long)ST_COL_OVERFLOW); }
    }
    if (staged) {
        for (uint32_t k = t; k < total; k += T)
            if (base + k < cap) a.col[base + k] = stage[k];
    } else if (act && deg) {                   // a very dense tile: straight to global memory
        int64_t k = base + loff;
        auto put = [&](int q) { if (k < cap) a.col[k] = q; ++k; };
        walk_back_global(a, r - 1, c, l, st, maxlen, put);
        walk_fwd_global(a, r + 1, n, c, rgt, st, put);
    }
    // next tile (the barrier also says: this tile's readers are done with the LDS)
    if (t == 0) s_tile = (int)atomicAdd(ticket, 1ull);
    __syncthreads();
    tile = s_tile;
    }   // tile loop
}

inline unsigned grid_for(int64_t n, int threads) { return (unsigned)sd_ceil_div(n, threads); }

struct FastPlan {
    FastArgs a;
    size_t zero_bytes;     // the region cleared before every run (status block .. tile states)
    void* zero_base;
    size_t k4_zero_bytes;  // part of it needed again when only the neighbour kernel is re-run
    int n_tiles;
};

constexpr int NB_T = 512;

int fast_plan(sdice_ctx* ctx, int64_t n, const int32_t* d_chrom, const int32_t* d_left, const int32_t* d_right,
              const int8_t* d_strand, int32_t* d_row_of, int64_t* d_row_ptr, FastPlan& pl) {
    FastArgs& a = pl.a;
    a.chrom = d_chrom; a.left = d_left; a.right = d_right; a.strand = d_strand; a.n = n;
    int64_t bucket_mean = ctx->param("cluster.bucket_mean", BUCKET_MEAN);
    if (bucket_mean < 256 || bucket_mean > BUCKET_MEAN) bucket_mean = BUCKET_MEAN;
    int64_t B = sd_ceil_div(n, bucket_mean);
    if (B < 1) B = 1;
    if (B > MAX_BUCKETS) B = MAX_BUCKETS;
    a.B = (int)B;
    // samples per bucket: 12 up to 2 M junctions (tighter bucket sizes: the largest bucket IS the sort kernel's time; -1.4 % on
    // the whole quant step at 1 M), 8 beyond (the sample is ranked by brute force, O(S^2))
    a.spb = (int)ctx->param("cluster.spb", 0);
    if (a.spb < 2 || a.spb > 64) a.spb = n <= ((int64_t)2 << 20) ? 12 : SPB;
    a.S = B > 1 ? (int)(B * a.spb) : 0;
    const int64_t mean = sd_ceil_div(n, B);
    a.slot_cap = B > 1 ? (mean * SLOT_FACTOR < n ? mean * SLOT_FACTOR : n) : n;
    int64_t lds_cap = ctx->param("cluster.lds_cap", 8192);      // (test knob: 0 = default; small values force the HBM sort)
    if (lds_cap <= 0 || lds_cap > 8192) lds_cap = 8192;
    if (lds_cap < 2) lds_cap = 2;
    a.lds_cap = (int)lds_cap;
    a.sample_sort = ctx->param("cluster.sample_sort", 1) != 0;
    pl.n_tiles = (int)sd_ceil_div(n, NB_T);
    const size_t nb64 = (size_t)sd_ceil_div(n, 64) + 2;
    // The status block is a small persistent allocation of the context (an asynchronous call is
    // resolved later, when the arena may already hold another call's scratch).
    if (!ctx->cluster_sb) {
        hipError_t e = hipMalloc((void**)&ctx->cluster_sb, (size_t)SB_ALLOC_WORDS * 8);
        if (e != hipSuccess) {
            sdice_set_error("sdice_cluster: hipMalloc of the status block failed: %s", hipGetErrorString(e));
            return SDICE_ERR_NOMEM;
        }
        SD_HIP(hipMemsetAsync(ctx->cluster_sb, 0, (size_t)SB_ALLOC_WORDS * 8, ctx->stream));
    }
    // zero region: [tile states] (needed by every neighbour run) then [cursor | rank | bmax64]
    const size_t zk4 = ((size_t)pl.n_tiles + 2) * 8;      // (+ the ticket counter of the tile loop behind the tile states)
    const size_t n_cd = (size_t)(a.S + 255) / 256 + 4;
    const size_t zrest = (size_t)a.B * 4 + (size_t)(a.S + 4) * 4 + nb64 * 4 + n_cd * 4;     // cursor | rank | bmax64 | col_done
    const size_t slots_bytes = (size_t)a.B * (size_t)a.slot_cap * 16;
    const size_t total = zk4 + zrest + slots_bytes + (size_t)n * 12 + (size_t)a.B * 16 + 16 * 4096;
    SD_TRY(ctx->arena.reserve(total, ctx->stream));
    Arena& A = ctx->arena;
    char* z = (char*)A.alloc(zk4 + zrest + 1024);
    a.slots = (uint4*)A.alloc(slots_bytes);
    a.rowC = (uint32_t*)A.alloc((size_t)n * 4);
    a.rowL = (uint32_t*)A.alloc((size_t)n * 4);
    a.rowR2 = (uint32_t*)A.alloc((size_t)n * 4);
    a.spl = (uint64_t*)A.alloc((size_t)a.B * 8 + 8);
    if (!z || !a.slots || !a.rowC || !a.rowL || !a.rowR2 || !a.spl) return SDICE_ERR_NOMEM;
    pl.zero_base = z;
    pl.k4_zero_bytes = zk4;
    pl.zero_bytes = zk4 + zrest;
    a.sb = (unsigned long long*)ctx->cluster_sb;
    a.tile_state = (unsigned long long*)z;
    a.cursor = (uint32_t*)(z + zk4);
    a.rank = a.cursor + a.B;
    a.bmax64 = a.rank + a.S + 4;
    a.col_done = a.bmax64 + nb64;
    a.row_of = d_row_of; a.row_ptr = d_row_ptr;
    a.ablate = (int)ctx->param("cluster.ablate", 0);
    return SDICE_OK;
}

int launch_neighbours(sdice_ctx* ctx, FastPlan& pl) {
    pl.a.col = ctx->d_col;
    pl.a.col_cap = ctx->col_cap;
    pl.a.blkneed = ctx->d_reach;
    // at most as many workgroups as the device holds at once (what the look-back's forward progress rests on); cluster.nb_grid
    // overrides it downwards (tests: a handful of workgroups walking many tiles each)
    static int per_cu = 0;
    if (per_cu == 0) {
        int occ = 0;
        SD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(neighbours_kernel<NB_T>), NB_T, 0));
        per_cu = occ > 0 ? occ : 1;
    }
    int64_t grid = std::min<int64_t>(pl.n_tiles, (int64_t)per_cu * ctx->n_cu);
    const int64_t forced = ctx->param("cluster.nb_grid", 0);
    if (forced > 0) grid = std::min<int64_t>(grid, forced);
    SD_LAUNCH(ctx, "neighbours_kernel", (neighbours_kernel<NB_T>), dim3((unsigned)std::max<int64_t>(grid, 1)), dim3(NB_T), 0, pl.a);
    return SDICE_OK;
}

int ensure_col(sdice_ctx* ctx, int64_t want) {
    if (want <= ctx->col_cap) return SDICE_OK;
    if (ctx->d_col) {
        SD_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_col);
    }
    ctx->d_col = nullptr;
    ctx->col_cap = 0;
    hipError_t e = hipMalloc((void**)&ctx->d_col, (size_t)want * 4);
    if (e != hipSuccess) {
        sdice_set_error("sdice_cluster: hipMalloc of %lld neighbour indices failed: %s", (long long)want, hipGetErrorString(e));
        return SDICE_ERR_NOMEM;
    }
    ctx->col_cap = want;
    return SDICE_OK;
}

int ensure_reach(sdice_ctx* ctx, int64_t n) {       // one word per 16 rows
    n = sd_ceil_div(n, 16) + 1;
    if (n <= ctx->reach_cap) return SDICE_OK;
    if (ctx->d_reach) {
        SD_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_reach);
    }
    ctx->d_reach = nullptr;
    ctx->reach_cap = 0;
    hipError_t e = hipMalloc((void**)&ctx->d_reach, (size_t)n * 4);
    if (e != hipSuccess) {
        sdice_set_error("sdice_cluster: hipMalloc of %lld block-reach entries failed: %s", (long long)n, hipGetErrorString(e));
        return SDICE_ERR_NOMEM;
    }
    ctx->reach_cap = n;
    return SDICE_OK;
}

}  // namespace

// Fetch the status block of the last fast-path run (synchronises).  Returns the flags.
// (also clears the sticky word: the caller reports what it held)
static int fast_fetch_status(sdice_ctx* ctx, unsigned long long* flags, int64_t* nnz, int64_t* reach, unsigned long long* sticky = nullptr) {
    int64_t* hp = ctx->h_pinned;
    SD_HIP(hipMemcpyAsync(hp, ctx->cluster_sb, (size_t)SB_ALLOC_WORDS * 8, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipMemsetAsync((unsigned long long*)ctx->cluster_sb + SB_STICKY, 0, 8, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    if (sticky) *sticky = (unsigned long long)hp[SB_STICKY];
    *flags = (unsigned long long)hp[SB_FLAGS];
    *nnz = hp[SB_NNZ];
    int64_t rc = 0;
    for (int i = 0; i < SB_SLOTS; ++i) rc = hp[SB_REACH + i] > rc ? hp[SB_REACH + i] : rc;
    *reach = rc;
    return SDICE_OK;
}

static int fast_flags_to_error(unsigned long long flags) {
    if (flags & ST_INVALID) {
        sdice_set_error("sdice_cluster: invalid junction (need 0 <= left <= right, chrom_rank >= 0, strand in {0,1})");
        return SDICE_ERR_ARG;
    }
    if (flags & ST_DUP) {
        sdice_set_error("sdice_cluster: duplicate junction (chrom, left, right, strand): the junctions must be distinct");
        return SDICE_ERR_ARG;
    }
    return SDICE_OK;
}

// Resolve the status of an asynchronous sdice_cluster_dev (nnz == NULL).  Called by sdice_sync,
// sdice_cluster_status and sdice_cluster_col_dev.
int sd_cluster_resolve(sdice_ctx* ctx) {
    if (!ctx->cluster_pending) return SDICE_OK;
    ctx->cluster_pending = false;
    unsigned long long flags = 0, sticky = 0;
    int64_t nnz = 0, reach = 0;
    SD_TRY(fast_fetch_status(ctx, &flags, &nnz, &reach, &sticky));
    ctx->nnz = 0;
    // the sticky word holds the failures of EVERY chain enqueued since the last report (the per-chain words only those of
    // the last one: a chain re-zeroes them)
    flags |= sticky;
    SD_TRY(fast_flags_to_error(flags));
    if (flags & (ST_SLOT_OVERFLOW | ST_LOOKBACK)) {
        sdice_set_error("sdice_cluster_dev (asynchronous): %s; call with nnz != NULL (synchronous) "
                        "to take the generic path", (flags & ST_LOOKBACK) ? "the look-back gave up waiting" : "a sort bucket overflowed");
        return SDICE_ERR_STATE;
    }
    if (flags & ST_COL_OVERFLOW) {
        sdice_set_error("sdice_cluster_dev (asynchronous): %lld neighbour entries exceed the list capacity %lld; "
                        "call once with nnz != NULL (synchronous) to size it", (long long)nnz, (long long)ctx->col_cap);
        return SDICE_ERR_STATE;
    }
    ctx->nnz = nnz;
    ctx->cluster_reach = (int)reach;
    return SDICE_OK;
}

extern "C" int sdice_cluster_status(sdice_ctx* ctx, int64_t* nnz, int32_t* reach) {
    SD_ARG(ctx, "ctx is NULL");
    SD_TRY(sd_cluster_resolve(ctx));
    if (nnz) *nnz = ctx->nnz;
    if (reach) *reach = ctx->cluster_reach;
    return SDICE_OK;
}

extern "C" int sdice_cluster_dev(sdice_ctx* ctx, int64_t n, const int32_t* d_chrom, const int32_t* d_left,
                                 const int32_t* d_right, const int8_t* d_strand, int32_t* d_row_of,
                                 int64_t* d_row_ptr, int64_t* nnz_out) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && n < ((int64_t)1 << 31), "n out of range");
    SD_HIP(hipSetDevice(ctx->device));
    // a synchronous call reports what earlier asynchronous chains left behind (it synchronises anyway); an asynchronous
    // one leaves their status pending: failures accumulate in the sticky word of the status block
    if (nnz_out && ctx->cluster_pending) SD_TRY(sd_cluster_resolve(ctx));
    ctx->reach_n = 0;
    const bool legacy = n > FAST_MAX_N || ctx->param("cluster.generic", 0) || ctx->param("cluster.legacy", 0);
    if ((legacy || n == 0) && ctx->cluster_pending) SD_TRY(sd_cluster_resolve(ctx));     // (the generic path synchronises)
    if (legacy || n == 0) return sd_cluster_legacy(ctx, n, d_chrom, d_left, d_right, d_strand, d_row_of, d_row_ptr, nnz_out);
    SD_ARG(d_chrom && d_left && d_right && d_strand && d_row_of && d_row_ptr, "NULL pointer");
    ctx->nnz = 0;
    ctx->cluster_reach = 0;
    if (nnz_out) *nnz_out = 0;

    FastPlan pl;
    SD_TRY(fast_plan(ctx, n, d_chrom, d_left, d_right, d_strand, d_row_of, d_row_ptr, pl));
    FastArgs& a = pl.a;
    // list capacity: what the last clustering needed, at least 16 entries per junction
    SD_TRY(ensure_col(ctx, 16 * n + 1024));
    SD_TRY(ensure_reach(ctx, n));

    SD_HIP(hipMemsetAsync(pl.zero_base, 0, pl.zero_bytes, ctx->stream));
    if (a.S == 0) SD_HIP(hipMemsetAsync(a.sb, 0, (size_t)SB_WORDS * 8, ctx->stream));   // (else cleared by the rank kernel)
    if (a.S > 0) {
        const unsigned g = grid_for(a.S, 256);
        SD_LAUNCH(ctx, "sample_rank_kernel", sample_rank_kernel, dim3(g, g), dim3(256), 0, a);
    }
    {
        const size_t lds = (size_t)a.B * 16;
        // tiles of 2048 keys (512 threads) for small inputs, 4096 (1024 threads) beyond
        if (n <= (1 << 19)) {      // (4096-key tiles halve the global atomics: 27.5 -> 21.4 us at 1 M keys)
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_scatter_kernel<512>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SD_LAUNCH(ctx, "bucket_scatter_kernel", (bucket_scatter_kernel<512>), dim3(grid_for(n, 512 * SC_KPT)), dim3(512), lds, a);
        } else {
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_scatter_kernel<1024>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SD_LAUNCH(ctx, "bucket_scatter_kernel", (bucket_scatter_kernel<1024>), dim3(grid_for(n, 1024 * SC_KPT)), dim3(1024), lds, a);
        }
    }
    {
        const size_t lds = SORT_LDS_BYTES;
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sort_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SD_LAUNCH(ctx, "bucket_sort_kernel", bucket_sort_kernel, dim3((unsigned)a.B), dim3(SORT_T), lds, a);
    }
    SD_TRY(launch_neighbours(ctx, pl));

    if (!nnz_out) {          // asynchronous: the status is resolved at the next synchronising call
        ctx->cluster_pending = true;
        ctx->reach_n = n;    // (a failed chain leaves empty lists and zero reach)
        return SDICE_OK;
    }
    unsigned long long flags = 0;
    int64_t nnz = 0, reach = 0;
    SD_TRY(fast_fetch_status(ctx, &flags, &nnz, &reach));
    SD_TRY(fast_flags_to_error(flags));
    if (flags & (ST_SLOT_OVERFLOW | ST_LOOKBACK))      // (> 8x the mean bucket size between two splitters, or an unexpected dispatch order)
        return sd_cluster_legacy(ctx, n, d_chrom, d_left, d_right, d_strand, d_row_of, d_row_ptr, nnz_out);
    if (flags & ST_COL_OVERFLOW) {
        SD_TRY(sd_cluster_check_nnz(ctx, nnz, false));
        SD_TRY(ensure_col(ctx, nnz + nnz / 8 + 1024));
        SD_HIP(hipMemsetAsync(pl.zero_base, 0, pl.k4_zero_bytes, ctx->stream));
        // flags, total and reach start over; the maximum length (words SB_MAXLEN..) stays
        SD_HIP(hipMemsetAsync(a.sb, 0, (size_t)SB_MAXLEN * 8, ctx->stream));
        SD_HIP(hipMemsetAsync(a.sb + SB_REACH, 0, (size_t)SB_SLOTS * 8, ctx->stream));
        SD_TRY(launch_neighbours(ctx, pl));
        SD_TRY(fast_fetch_status(ctx, &flags, &nnz, &reach));
        if (flags) {
            sdice_set_error("sdice_cluster: neighbour pass failed after resizing (flags %llu)", flags);
            return SDICE_ERR_STATE;
        }
    }
    ctx->nnz = nnz;
    ctx->cluster_reach = (int)reach;
    ctx->reach_n = n;
    *nnz_out = nnz;
    return SDICE_OK;
}
