// K5: two-group Wilcoxon rank-sum per junction row + per-group median / mean.
//
// Replaces the row loop of compare_sample_sets (compareSampleSets.py:216-232):
//   d1 = row[g1], d2 = row[g2]; drop NaNs; skip the row if either group has < 3 values;
//   scipy.stats.ranksums(d1, d2)  (average ranks, no tie / continuity correction,
//   p = 2*ndtr(-|z|)); np.median x2, np.mean x2 on float32; delta = med1 - med2.
//
// Arithmetic used here (exact restatement, not an approximation):
//   sum of ranks of group 1 = n1(n1+1)/2 + U,  U = #{(i,j): b_j < a_i} + 0.5 #{b_j == a_i}
//   => s - expected = U - n1*n2/2  (half-integers, exact), z = that / sqrt(n1 n2 (n1+n2+1)/12).
//   2U = sum_i (lower_bound(B, a_i) + upper_bound(B, a_i)) over the sorted second group.
//   np.mean(float32 vector) = numpy's pairwise summation in float32 (8 interleaved
//   accumulators per <=128-element block, halving recursion above) divided by n in float32;
//   it is reproduced operation for operation so the means are bit-identical.
//
// Kernels (auto dispatch: lane pair for groups <= 64, wave for <= 1024, block above):
//   ranksum_pair_kernel  (n1, n2 <= 64): the lane kernel's machinery with TWO lanes per row (one
//       per group), half the LDS per wave, twice the resident waves (see its comment).
//   ranksum_count_kernel (n1, n2 in 65..1024, rows of 3-decimal PS values -- what compare_sample_sets
//       always sees): one WAVE per row, the two groups become histograms over the 1001 possible
//       values, nothing is sorted (see its comment); other rows are left to
//   ranksum_wave_kernel  (n1, n2 <= 1024): one WAVE per row, 64*E-element bitonic network held
//       in VGPRs across the wave, no workgroup barriers (see the kernel's comment).
//   ranksum_lane_kernel  (n1, n2 <= 64): one LANE per row.  A wave stages 64 rows (only the
//       selected columns) into LDS with a coalesced copy, each lane then compacts its row,
//       sums it, sorts each group with a fully unrolled bitonic network held in VGPRs
//       (v_min/v_max pairs, no cross-lane traffic), writes the sorted groups back to LDS and
//       merges them for U.  All 64 lanes stay busy on serial per-row work.
//   ranksum_block_kernel (any n1, n2 that fit LDS): one workgroup per row, bitonic sort in
//       LDS, binary searches for U.
#include "common.h"
#include <math.h>

namespace {

struct RsOut {
    uint8_t* tested;
    double* p;
    double* z;
    float* med1;
    float* med2;
    float* mean1;
    float* mean2;
    float* delta;
};

__device__ __forceinline__ void rs_finish(int nv1, int nv2, long long u2, double& z, double& p) {
    // scipy ranksums: z = (s - n1(n1+n2+1)/2) / sqrt(n1 n2 (n1+n2+1)/12), s - expected = U - n1 n2/2
    const double num = 0.5 * (double)(u2 - (long long)nv1 * (long long)nv2);
    const double den = sqrt((double)((long long)nv1 * nv2 * (nv1 + nv2 + 1)) / 12.0);
    z = num / den;
    // scipy: p = 2 * ndtr(-|z|); cephes ndtr evaluates 0.5*erfc(|z|/sqrt2) for |z| >= 1 and
    // 0.5 + 0.5*erf(-|z|/sqrt2) below, i.e. erfc(|z|/sqrt2) up to one rounding of a value >= 0.3
    p = erfc(fabs(z) * 0.70710678118654752440);
}

// ------------------------------------------------------------------ lane-per-row variant
template <int P>
__device__ __forceinline__ void bitonic_regs(float (&a)[P]) {
#pragma unroll
    for (int k = 2; k <= P; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool asc = (i & k) == 0;
                    // plain v_min/v_max: no NaN reaches the network (compaction removed them), and
                    // fminf/fmaxf would add a canonicalising v_max per operand
                    float lo, hi;
                    asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(a[l]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a[i]), "v"(a[l]));
                    a[i] = asc ? lo : hi;
                    a[l] = asc ? hi : lo;
                }
            }
        }
    }
}

// One group of one row, handled by one lane.  row: this lane's slice of the LDS tile (group
// values at [0, cnt)).  Compacts in place (NaNs dropped, order kept), returns the number of
// valid values, the float32 pairwise mean, and leaves the group SORTED in row[0..nv).
template <int P>
__device__ __forceinline__ int lane_group(float* row, int cnt, float& mean) {
    // all LDS reads first (independent, pipelined), then the in-place compaction is a chain of
    // stores whose only dependency is the integer counter, then one more batch of reads
    float a[P];
    const float inf = __builtin_inff();
#pragma unroll
    for (int k = 0; k < P; ++k) a[k] = (k < cnt) ? row[k] : __builtin_nanf("");
    int nv = 0;
#pragma unroll
    for (int k = 0; k < P; ++k)
        if (a[k] == a[k]) { row[nv] = a[k]; ++nv; }
#pragma unroll
    for (int k = 0; k < P; ++k) a[k] = (k < nv) ? row[k] : inf;
    // numpy pairwise_sum (n <= 128 -> single block), npy loops_utils.h.src
    float res;
    if (nv < 8) {
        res = 0.f;
#pragma unroll
        for (int k = 0; k < (P < 8 ? P : 7); ++k)
            if (k < nv) res += a[k];
    } else {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        const int main_n = nv - (nv & 7);
#pragma unroll
        for (int k = 8; k < P; ++k)
            if (k < main_n) r[k & 7] += a[k];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int k = 8; k < P; ++k)
            if (k >= main_n && k < nv) res += a[k];
    }
    mean = nv > 0 ? res / (float)nv : 0.f;
    bitonic_regs<P>(a);
#pragma unroll
    for (int k = 0; k < P; ++k)
        if (k < cnt) row[k] = a[k];
    return nv;
}

__device__ __forceinline__ float median_sorted(const float* a, int nv) {
    // np.median: odd -> middle; even -> np.mean of the two middle values in float32
    const int h = nv >> 1;
    if (nv & 1) return a[h];
    return (a[h - 1] + a[h]) / 2.0f;
}

template <int P>
__global__ void __launch_bounds__(256) ranksum_lane_kernel(const float* __restrict__ ps, int64_t n, int s,
                                                           const int32_t* __restrict__ gsel1, const int32_t* __restrict__ gsel2, int n1, int n2,
                                                           int stride, RsOut o, int ablate) {
    extern __shared__ float smemf[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (wave-uniform: row indices and offsets stay scalar)
    const int waves_per_block = blockDim.x >> 6;
    float* tile = smemf + (size_t)wave * 64 * stride;
    const int nsel = n1 + n2;
    int* selL = reinterpret_cast<int*>(smemf + (size_t)waves_per_block * 64 * stride);
    for (int k = threadIdx.x; k < nsel; k += blockDim.x) selL[k] = k < n1 ? gsel1[k] : gsel2[k - n1];
    __syncthreads();
    const int64_t n_groups = (n + 63) >> 6;
    for (int64_t g = (int64_t)blockIdx.x * waves_per_block + wave; g < n_groups;
         g += (int64_t)gridDim.x * waves_per_block) {
        const int64_t row0 = g << 6;
        // ---- stage 64 rows x nsel selected columns.  Lane l owns selected columns l and l + 64
        // for every row, so a row is one or two coalesced wave loads with a 32-bit offset that
        // advances by s per row (no per-element index arithmetic); 16 rows are in flight at once.
        {
            const int rows_avail = (int)min((int64_t)64, n - row0);
            const float* gbase = ps + row0 * s;
            const bool act0 = lane < nsel, act1 = lane + 64 < nsel;
            const int sel0 = act0 ? selL[lane] : 0;
            const int sel1 = act1 ? selL[lane + 64] : 0;
            constexpr int RB = 16;
            for (int r0 = 0; r0 < 64; r0 += RB) {
                float x0[RB], x1[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int rc = min(r0 + q, rows_avail - 1);       // rows past the end re-read the last row
                    x0[q] = act0 ? gbase[rc * s + sel0] : 0.f;
                    x1[q] = act1 ? gbase[rc * s + sel1] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int r = r0 + q;
                    const bool live = r < rows_avail;
                    if (act0) tile[r * stride + lane] = live ? x0[q] : __builtin_nanf("");
                    if (act1) tile[r * stride + lane + 64] = live ? x1[q] : __builtin_nanf("");
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // ---- per-lane serial work on row (row0 + lane)
        float* row = tile + lane * stride;
        // both groups go through ONE copy of the (fully unrolled, register resident) group code:
        // two inlined copies of the 64-wide network do not fit the instruction cache
        float mean1 = 0.f, mean2 = 0.f;
        int nv1 = 0, nv2 = 0;
#pragma nounroll
        for (int gi = 0; gi < ((ablate & 2) ? 0 : 2); ++gi) {
            float m;
            const int nv = lane_group<P>(row + (gi ? n1 : 0), gi ? n2 : n1, m);
            if (gi == 0) { nv1 = nv; mean1 = m; } else { nv2 = nv; mean2 = m; }
        }
        const int64_t rr = row0 + lane;
        if (rr < n) {
            const bool tested = nv1 >= 3 && nv2 >= 3;
            float med1 = 0.f, med2 = 0.f, dl = 0.f;
            unsigned long long packed = 0;      // (2U, n1, n2) for ranksum_finish_kernel, as the wave kernel
            if (tested && !(ablate & 1)) {
                const float* A = row;
                const float* B = row + n1;
                med1 = median_sorted(A, nv1);
                med2 = median_sorted(B, nv2);
                dl = med1 - med2;
                // 2U = sum_i (lb_i + ub_i) = sum_i (2 ub_i - eq_i) by ONE sequential merge with a
                // fixed trip count nv1 + nv2 (no nested data-dependent loops: lanes stay in step).
                // Rule: take b while b <= a.  When a_i is taken, j = #{b <= a_i} = ub_i, and the
                // run of equal b's taken last tells eq_i = #{b == a_i}.
                long long u2 = 0;
                int i = 0, j = 0, run_len = 0;
                float run_val = 0.f;
                float av = A[0], bv = B[0];
                const int steps = nv1 + nv2;
                for (int t = 0; t < steps; ++t) {
                    const bool take_b = j < nv2 && (i >= nv1 || bv <= av);
                    if (take_b) {
                        run_len = (run_len > 0 && bv == run_val) ? run_len + 1 : 1;
                        run_val = bv;
                        ++j;
                        bv = B[j < nv2 ? j : nv2 - 1];
                    } else {
                        const int eq = (run_len > 0 && run_val == av) ? run_len : 0;
                        u2 += 2 * j - eq;
                        ++i;
                        av = A[i < nv1 ? i : nv1 - 1];
                    }
                }
                packed = (unsigned long long)(unsigned)u2 | ((unsigned long long)nv1 << 32) | ((unsigned long long)nv2 << 48);
            } else {
                mean1 = 0.f; mean2 = 0.f;
            }
            o.tested[rr] = tested ? 1 : 0;
            reinterpret_cast<unsigned long long*>(o.p)[rr] = packed;
            o.med1[rr] = med1; o.med2[rr] = med2;
            o.mean1[rr] = mean1; o.mean2[rr] = mean2;
            o.delta[rr] = dl;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ lane-pair variant
// Same per-lane machinery, but a row is shared by TWO lanes: lane 2r sorts group 1 of row r, lane
// 2r+1 group 2, so a wave stages 32 rows -- half the LDS per wave and twice the resident waves
// (the lane kernel's occupancy is set by its LDS tile).  For U the even lane merges with the rule
// "take b while b <= a" (sum of upper bounds), the odd lane with "b < a" (sum of lower bounds);
// the pair adds them: 2U = sum_i lb_i + ub_i.  No tie-run bookkeeping in either merge.
template <int P>
__global__ void __launch_bounds__(256) ranksum_pair_kernel(const float* __restrict__ ps, int64_t n, int s,
                                                           const int32_t* __restrict__ gsel1, const int32_t* __restrict__ gsel2, int n1, int n2,
                                                           int stride, RsOut o) {
    extern __shared__ float smemp[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (wave-uniform: row indices and offsets stay scalar)
    const int waves_per_block = blockDim.x >> 6;
    float* tile = smemp + (size_t)wave * 32 * stride;
    const int nsel = n1 + n2;
    int* selL = reinterpret_cast<int*>(smemp + (size_t)waves_per_block * 32 * stride);
    for (int k = threadIdx.x; k < nsel; k += blockDim.x) selL[k] = k < n1 ? gsel1[k] : gsel2[k - n1];
    __syncthreads();
    const int64_t n_groups = (n + 31) >> 5;
    for (int64_t g = (int64_t)blockIdx.x * waves_per_block + wave; g < n_groups;
         g += (int64_t)gridDim.x * waves_per_block) {
        const int64_t row0 = g << 5;
        {   // stage 32 rows x nsel selected columns (lane l owns selected columns l and l + 64)
            const int rows_avail = (int)min((int64_t)32, n - row0);
            const float* gbase = ps + row0 * s;
            const bool act0 = lane < nsel, act1 = lane + 64 < nsel;
            const int sel0 = act0 ? selL[lane] : 0;
            const int sel1 = act1 ? selL[lane + 64] : 0;
            constexpr int RB = 16;
            for (int r0 = 0; r0 < 32; r0 += RB) {
                float x0[RB], x1[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int rc = min(r0 + q, rows_avail - 1);
                    x0[q] = act0 ? gbase[rc * s + sel0] : 0.f;
                    x1[q] = act1 ? gbase[rc * s + sel1] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int r = r0 + q;
                    const bool live = r < rows_avail;
                    if (act0) tile[r * stride + lane] = live ? x0[q] : __builtin_nanf("");
                    if (act1) tile[r * stride + lane + 64] = live ? x1[q] : __builtin_nanf("");
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        const int r = lane >> 1, grp = lane & 1;
        float* row = tile + r * stride;
        float mean;
        const int nv = lane_group<P>(row + (grp ? n1 : 0), grp ? n2 : n1, mean);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();          // the partner's sorted group is in LDS
        const int nv_other = __shfl_xor(nv, 1);
        const int nv1 = grp ? nv_other : nv, nv2 = grp ? nv : nv_other;
        const bool tested = nv1 >= 3 && nv2 >= 3;
        const float* A = row;
        const float* B = row + n1;
        float med = 0.f;
        int part = 0;
        if (tested) {
            med = median_sorted(grp ? B : A, nv);
            int i = 0, j = 0;
            float av = A[0], bv = B[0];
            const int steps = nv1 + nv2;
            for (int t = 0; t < steps; ++t) {
                const bool before = grp ? (bv < av) : (bv <= av);
                const bool take_b = j < nv2 && (i >= nv1 || before);
                if (take_b) {
                    ++j;
                    bv = B[j < nv2 ? j : nv2 - 1];
                } else {
                    part += j;
                    ++i;
                    av = A[i < nv1 ? i : nv1 - 1];
                }
            }
        }
        const int u2 = part + __shfl_xor(part, 1);
        const float med_other = __shfl_xor(med, 1);
        const float mean_other = __shfl_xor(mean, 1);
        const int64_t rr = row0 + r;
        if (grp == 0 && rr < n) {
            const unsigned long long packed =
                tested ? ((unsigned long long)(unsigned)u2 | ((unsigned long long)nv1 << 32) | ((unsigned long long)nv2 << 48))
                       : 0ull;
            o.tested[rr] = tested ? 1 : 0;
            reinterpret_cast<unsigned long long*>(o.p)[rr] = packed;
            o.med1[rr] = med; o.med2[rr] = med_other;
            o.mean1[rr] = tested ? mean : 0.f; o.mean2[rr] = tested ? mean_other : 0.f;
            o.delta[rr] = med - med_other;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ lane-pair variant on 16-bit keys
// compare_sample_sets only ever sees 3-decimal PS values, float32(k / 1000) for k = 0..1000 (see the
// counting kernel below), so a value IS its integer k.  Same two-lanes-per-row scheme as above, but
//   * the staged tile holds 16-bit keys (0xFFFF = NaN): half the LDS per wave, twice the resident waves
//     for the latency-bound merge;
//   * a lane sorts its group as 2 x P/2 keys packed two per VGPR -- element i in the low half of
//     register i mod P/2, element i + P/2 in the high half -- with v_pk_min_u16 / v_pk_max_u16: one
//     instruction pair exchanges two comparators, half the instructions of the float network (only
//     the flip of the last merge crosses the halves and costs six instructions per register pair);
//   * floats come back as float32(k / 1000), computed (ps_of_key: no table, LDS instructions are the scarce resource
//     here): the numpy pairwise sum runs over them in original order, medians are those of the middle keys.
// A row holding any value that is not exactly float32(k / 1000) is marked RS_REDO and left to ranksum_wave_kernel.
constexpr unsigned char RS_REDO_Q = 0xFF;       // (same mark as RS_REDO below)
// layout of a staged row in dwords: group 1 in [0, 32), one spare, group 2 in [33, 65), one spare: row pitch 66.  ds_read_b32
// and every ds_write bank by (dword address) mod 32 and serve lanes 0..31 and 32..63 as separate groups (MI355X_MICROARCH.md,
// LDS): lane (row r, group g) reading dword idx of its group sits in bank (2 r + g + idx) mod 32 -- the 16 rows x 2 groups of
// a half wave in 32 different banks.  (Rounds 2-3 had group 2 at dword 32 and pitch 65, planned for 64 banks: every
// per-lane access of the two lanes of a row met in ONE bank, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.57.)
constexpr int RSQ_G2 = 66;                      // u16 offset of group 2 inside a staged row
constexpr int RSQ_STRIDE = 132;                 // u16 per staged row: 66 dwords
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
typedef uint32_t __attribute__((may_alias)) u32_alias;      // dword view of the staged 16-bit keys

__device__ __forceinline__ uint32_t pk_min16(uint32_t a, uint32_t b) {
    union { uint32_t u; v2u16 v; } x, y, r;
    x.u = a; y.u = b;
    r.v = __builtin_elementwise_min(x.v, y.v);
    return r.u;
}
__device__ __forceinline__ uint32_t pk_max16(uint32_t a, uint32_t b) {
    union { uint32_t u; v2u16 v; } x, y, r;
    x.u = a; y.u = b;
    r.v = __builtin_elementwise_max(x.v, y.v);
    return r.u;
}
__device__ __forceinline__ uint32_t swap16(uint32_t a) { return (a >> 16) | (a << 16); }
// float32(k / 1000.0) for k = 0..1000 without the table: the product with float32(0.001) plus one residual step is
// the correctly rounded quotient for every one of the 1001 values (checked exhaustively against the table's
// definition, tests/test_abi_and_host.py) -- three VALU instructions instead of an LDS look-up, which is what this
// kernel is short of
__device__ __forceinline__ float ps_of_key(float kf) {
    const float q = kf * 0.001f;
    return __builtin_fmaf(__builtin_fmaf(-q, 1000.0f, kf), 0.001f, q);
}
__device__ __forceinline__ uint32_t lo_hi(uint32_t lo_from, uint32_t hi_from) { return (lo_from & 0xffffu) | (hi_from & 0xffff0000u); }

// ascending sort of P 16-bit keys, k[r] = element r | element (r + P/2) << 16  (flip + disperse network)
template <int P>
__device__ __forceinline__ void sort_keys16(uint32_t (&k)[P / 2]) {
    constexpr int NR = P / 2;
#pragma unroll
    for (int kk = 2; kk <= P; kk <<= 1) {
        if (kk <= NR) {
#pragma unroll
            for (int a = 0; a < NR; ++a) {
                const int b = a ^ (kk - 1);
                if (b > a) { const uint32_t mn = pk_min16(k[a], k[b]), mx = pk_max16(k[a], k[b]); k[a] = mn; k[b] = mx; }
            }
        } else {
            // flip(P): element i <-> P-1-i, i.e. (k[a].lo, k[b].hi) and (k[b].lo, k[a].hi) with b = NR-1-a
#pragma unroll
            for (int a = 0; a < NR / 2; ++a) {
                const int b = NR - 1 - a;
                const uint32_t bs = swap16(k[b]);
                const uint32_t mn = pk_min16(k[a], bs), mx = pk_max16(k[a], bs);
                k[a] = lo_hi(mn, mx);
                k[b] = swap16(lo_hi(mx, mn));
            }
        }
#pragma unroll
        for (int j = kk >> 2; j >= 1; j >>= 1) {
#pragma unroll
            for (int a = 0; a < NR; ++a) {
                if ((a & j) == 0) { const uint32_t mn = pk_min16(k[a], k[a + j]), mx = pk_max16(k[a], k[a + j]); k[a] = mn; k[a + j] = mx; }
            }
        }
    }
}

template <int P>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) ranksum_pairq_kernel(const float* __restrict__ ps, int64_t n, int s,
                                                            const int32_t* __restrict__ gsel1, const int32_t* __restrict__ gsel2, int n1, int n2,
                                                            int stride /* u16 units, odd multiple ... */, RsOut o) {
    extern __shared__ float smemq[];
    constexpr int NR = P / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (wave-uniform: row indices and offsets stay scalar)
    const int waves_per_block = blockDim.x >> 6;
    const int nsel = n1 + n2;
    int* selL = reinterpret_cast<int*>(smemq);                          // [nsel]
    unsigned short* tile = reinterpret_cast<unsigned short*>(selL + ((nsel + 1) & ~1)) + (size_t)wave * 32 * stride;
    for (int k = threadIdx.x; k < nsel; k += blockDim.x) selL[k] = k < n1 ? gsel1[k] : gsel2[k - n1];
    // every slot of the tile starts as 0xFFFF ("no value"): the staging only ever writes the first n1 / n2 slots of a
    // group and the sorted write-back leaves 0xFFFF in the others, so the slots behind a group never need a bound check
    for (int k = lane; k < 16 * stride; k += 64) reinterpret_cast<u32_alias*>(tile)[k] = 0xFFFFFFFFu;
    __syncthreads();
    const int64_t n_groups = (n + 31) >> 5;
    for (int64_t g = (int64_t)blockIdx.x * waves_per_block + wave; g < n_groups;
         g += (int64_t)gridDim.x * waves_per_block) {
        const int64_t row0 = g << 5;
        unsigned redo_rows = 0;                        // bit r: row r holds a value that is not a 3-decimal PS
        {   // stage 32 rows x nsel selected columns as keys: lanes 0..31 own group 1's columns (2l, 2l + 1), lanes 32..63
            // group 2's -- the two keys of a lane are ONE dword of the row (dword index = lane: one conflict-free
            // 32-bit store per row and lane instead of two 16-bit stores; slots behind the group get 0xFFFF)
            const int rows_avail = (int)min((int64_t)32, n - row0);
            const float* gbase = ps + row0 * s;
            const int e0 = 2 * (lane & 31), e1 = e0 + 1;
            const int cntg = lane >= 32 ? n2 : n1, selbase = lane >= 32 ? n1 : 0;
            const bool act0 = e0 < cntg, act1 = e1 < cntg;
            const int sel0 = act0 ? selL[selbase + e0] : 0;
            const int sel1 = act1 ? selL[selbase + e1] : 0;
            u32_alias* tile32 = reinterpret_cast<u32_alias*>(tile);
            const int pitch32 = stride >> 1;
            // group 1 at the start of the row, group 2 at dword 33, row pitch 66 dwords (see RSQ_G2)
            constexpr int RB = 16;
            for (int r0 = 0; r0 < 32; r0 += RB) {
                float x0[RB], x1[RB];
                __builtin_amdgcn_s_setprio(3);          // (loads first: ~1 % -- see ps.hip)
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int rc = min(r0 + q, rows_avail - 1);
                    x0[q] = act0 ? __builtin_nontemporal_load(gbase + rc * s + sel0) : 0.f;      // (read once: -2 %)
                    x1[q] = act1 ? __builtin_nontemporal_load(gbase + rc * s + sel1) : 0.f;
                }
                __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int r = r0 + q;
                    const bool live = r < rows_avail;
                    // key of a value: k = rint(1000 x) when x == float32(k / 1000) exactly, 0xFFFF for NaN (and rows past the end)
                    const float a0 = x0[q], a1 = x1[q];
                    const float kf0 = __builtin_amdgcn_fmed3f(__builtin_rintf(a0 * 1000.0f), 0.0f, 1000.0f);     // (a NaN's key is never used)
                    const float kf1 = __builtin_amdgcn_fmed3f(__builtin_rintf(a1 * 1000.0f), 0.0f, 1000.0f);
                    const int k0 = (int)kf0, k1 = (int)kf1;
                    const bool nan0 = !(a0 == a0) || !live, nan1 = !(a1 == a1) || !live;
                    const bool bad0 = act0 && !nan0 && ps_of_key(kf0) != a0, bad1 = act1 && !nan1 && ps_of_key(kf1) != a1;
                    if (__ballot(bad0 || bad1)) redo_rows |= 1u << r;
                    const uint32_t kk0 = (act0 && !nan0) ? (uint32_t)k0 : 0xFFFFu, kk1 = (act1 && !nan1) ? (uint32_t)k1 : 0xFFFFu;
                    tile32[r * pitch32 + lane + (lane >> 5)] = kk0 | (kk1 << 16);      // (group 2 starts at dword 33)
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        const int r = lane >> 1, grp = lane & 1;
        unsigned short* row = tile + r * stride;
        unsigned short* grow = row + (grp ? RSQ_G2 : 0);
        // ---- compaction (NaNs dropped, order kept) through LDS, then the keys of this group in registers
        int nv = 0;
#pragma unroll
        for (int e0 = 0; e0 < P; e0 += 16) {       // 8 dword reads (16 keys) in flight, then their compacting stores (all to indices <= e)
            uint32_t w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = reinterpret_cast<const u32_alias*>(grow)[e0 / 2 + k];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned short v = (unsigned short)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
                if (v != 0xFFFF) { grow[nv] = v; ++nv; }
            }
        }
        // one dword = two compacted keys: the sorting network takes its input in any order, so dword d (values 2d and
        // 2d + 1) goes straight to register d (32 LDS reads instead of 64); entries behind the last value become 0xFFFF
        uint32_t key[NR];
        {
            const u32_alias* grow32 = reinterpret_cast<const u32_alias*>(grow);
#pragma unroll
            for (int d = 0; d < NR; ++d) {
                uint32_t v = grow32[d];
                v = 2 * d + 1 < nv ? v : (v | 0xFFFF0000u);
                key[d] = 2 * d < nv ? v : 0xFFFFFFFFu;
            }
        }
        // ---- numpy pairwise_sum over float32(key / 1000) in original order (n <= 128 -> single block), npy loops_utils.h.src
        float mean;
        {
            // (only read for e < nv; float32(k / 1000) by arithmetic, not from the table: LDS instructions are what the kernel is short of)
            auto val = [&](int e) -> float { return ps_of_key((float)((e & 1) ? (key[e >> 1] >> 16) : (key[e >> 1] & 0xffffu))); };
            float res;
            if (nv < 8) {
                res = 0.f;
#pragma unroll
                for (int e = 0; e < 7; ++e)
                    if (e < nv) res += val(e);
            } else {
                float rr8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) rr8[j] = val(j);
                const int main_n = nv - (nv & 7);
#pragma unroll
                for (int e = 8; e < P; ++e)
                    if (e < main_n) rr8[e & 7] += val(e);
                res = ((rr8[0] + rr8[1]) + (rr8[2] + rr8[3])) + ((rr8[4] + rr8[5]) + (rr8[6] + rr8[7]));
#pragma unroll
                for (int e = 8; e < P; ++e)
                    if (e >= main_n && e < nv) res += val(e);
            }
            mean = nv > 0 ? res / (float)nv : 0.f;
        }
        sort_keys16<P>(key);
        // sorted: rank i sits in the low half of key[i] (i < NR), rank i + NR in the high half of key[i].  Back to LDS as
        // 32 dwords (rank 2d | rank 2d + 1 << 16, one v_perm each) instead of 64 16-bit stores; all 64 slots of the
        // group are written, the padding (0xFFFF, ranks >= nv) is the sentinel behind the values (n1, n2 <= 63)
        {
            u32_alias* grow32 = reinterpret_cast<u32_alias*>(grow);
#pragma unroll
            for (int d = 0; d < NR; ++d) {
                const int i0 = 2 * d, i1 = 2 * d + 1;
                grow32[d] = i1 < NR ? __builtin_amdgcn_perm(key[i1], key[i0], 0x05040100u)
                                    : __builtin_amdgcn_perm(key[i1 - NR], key[i0 - NR], 0x07060302u);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();          // the partner's sorted group is in LDS
        const int nv_other = __shfl_xor(nv, 1);
        const int nv1 = grp ? nv_other : nv, nv2 = grp ? nv : nv_other;
        const bool tested = nv1 >= 3 && nv2 >= 3;
        const unsigned short* A = row;
        const unsigned short* B = row + RSQ_G2;
        float med = 0.f;
        int part = 0;
        if (tested) {
            const unsigned short* G = grp ? B : A;
            const int h = nv >> 1;
            med = (nv & 1) ? ps_of_key((float)G[h]) : (ps_of_key((float)G[h - 1]) + ps_of_key((float)G[h])) / 2.0f;      // np.median on float32
            // one merge walk; positions nv1 of A and nv2 of B hold 0xFFFF (the sorted padding or the sentinel), so an
            // exhausted side loses every comparison and no index has to be checked or clamped.  The even lane
            // counts "b <= a" (upper bounds), the odd lane "b < a": b goes first when b < a + adj.
            const unsigned adj = grp ? 0u : 1u;
            int jb = RSQ_G2;                          // row index of the next B element
            unsigned av = A[0], bv = B[0];
            const int steps = nv1 + nv2;
            for (int t = 0; t < steps; ++t) {
                const bool take_b = bv < av + adj;
                part += take_b ? 0 : jb;              // (+ 64 per A element: taken off below)
                jb += take_b ? 1 : 0;
                const int i = t + 1 + RSQ_G2 - jb;    // A elements taken so far
                const unsigned nx = row[take_b ? jb : i];
                av = take_b ? av : nx;
                bv = take_b ? nx : bv;
            }
            part -= RSQ_G2 * nv1;
        }
        const int u2 = part + __shfl_xor(part, 1);
        const float med_other = __shfl_xor(med, 1);
        const float mean_other = __shfl_xor(mean, 1);
        const int64_t rr = row0 + r;
        if (grp == 0 && rr < n) {
            const bool redo = (redo_rows >> r) & 1u;
            const unsigned long long packed =
                (tested && !redo) ? ((unsigned long long)(unsigned)u2 | ((unsigned long long)nv1 << 32) | ((unsigned long long)nv2 << 48))
                                  : 0ull;
            o.tested[rr] = redo ? RS_REDO_Q : (tested ? 1 : 0);
            reinterpret_cast<unsigned long long*>(o.p)[rr] = packed;
            o.med1[rr] = med; o.med2[rr] = med_other;
            o.mean1[rr] = tested ? mean : 0.f; o.mean2[rr] = tested ? mean_other : 0.f;
            o.delta[rr] = med - med_other;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ block-per-row variant
constexpr int RB_THREADS = 256;

// ordered compaction of the non-NaN values of ps[row, idx[0..cnt)] into dst; returns count
__device__ int block_compact(const float* __restrict__ prow, const int32_t* __restrict__ idx, int cnt,
                             float* dst, int* wcnt /* [RB_THREADS/64 + 1] shared */) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int base = 0;
    for (int c0 = 0; c0 < cnt; c0 += RB_THREADS) {
        const int k = c0 + tid;
        float x = __builtin_nanf("");
        if (k < cnt) x = prow[idx[k]];
        const bool valid = x == x;
        const unsigned long long m = __ballot(valid);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[w] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int q = 0; q < RB_THREADS / 64; ++q) {
            if (q < w) woff += wcnt[q];
            tot += wcnt[q];
        }
        if (valid) dst[base + woff + pre] = x;
        base += tot;
        __syncthreads();
    }
    return base;
}

// numpy's pairwise_sum recursion  `n <= 128 ? leaf : sum(a, n2) + sum(a + n2, n - n2)`,
// n2 = n/2 - (n/2) % 8, unrolled at compile time to PW_DEPTH levels.  The larger half is up to
// len/2 + 7.5, so 4096 values can need SIX levels (4095 -> 2055 -> 1031 -> 519 -> 263 -> 135 -> 71)
// and up to 64 leaves.  Every thread walks it redundantly with block-uniform arguments: no stacks,
// no single-lane section.
constexpr int PW_DEPTH = 6;

template <int DEPTH>
__device__ __forceinline__ void pw_leaves(int off, int len, int* leaf_off, int& nl, bool writer) {
    if (DEPTH == 0 || len <= 128) {
        if (writer) leaf_off[nl] = off;
        ++nl;
    } else {
        int n2 = len / 2;
        n2 -= n2 % 8;
        pw_leaves<(DEPTH > 0 ? DEPTH - 1 : 0)>(off, n2, leaf_off, nl, writer);
        pw_leaves<(DEPTH > 0 ? DEPTH - 1 : 0)>(off + n2, len - n2, leaf_off, nl, writer);
    }
}

template <int DEPTH>
__device__ __forceinline__ float pw_combine(int len, const float* leaf_sum, int& next) {
    if (DEPTH == 0 || len <= 128) return leaf_sum[next++];
    int n2 = len / 2;
    n2 -= n2 % 8;
    const float l = pw_combine<(DEPTH > 0 ? DEPTH - 1 : 0)>(n2, leaf_sum, next);
    const float r = pw_combine<(DEPTH > 0 ? DEPTH - 1 : 0)>(len - n2, leaf_sum, next);
    return l + r;
}

// numpy pairwise_sum over a[0..n) in float32 by the whole block; result to all threads
__device__ float block_pairwise_sum(const float* a, int n, int* leaf_off /*[LEAF_MAX+1]*/, float* leaf_sum,
                                    float* scratch8 /* [LEAF_MAX*8] */, int leaf_max) {
    const int tid = threadIdx.x;
    (void)leaf_max;
    int nl = 0;
    pw_leaves<PW_DEPTH>(0, n, leaf_off, nl, tid == 0);
    if (tid == 0) leaf_off[nl] = n;
    __syncthreads();
    for (int t = tid; t < nl * 8; t += blockDim.x) {
        const int L = t >> 3, j = t & 7;
        const int off = leaf_off[L], len = leaf_off[L + 1] - off;
        float r = 0.f;
        if (len >= 8) {
            r = a[off + j];
            for (int i = 8; i < len - (len % 8); i += 8) r += a[off + i + j];
        }
        scratch8[t] = r;
    }
    __syncthreads();
    for (int L = tid; L < nl; L += blockDim.x) {
        const int off = leaf_off[L], len = leaf_off[L + 1] - off;
        float res;
        if (len < 8) {
            res = 0.f;
            for (int i = 0; i < len; ++i) res += a[off + i];
        } else {
            const float* r = scratch8 + L * 8;
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (int i = len - (len % 8); i < len; ++i) res += a[off + i];
        }
        leaf_sum[L] = res;
    }
    __syncthreads();
    int next = 0;
    const float out = pw_combine<PW_DEPTH>(n, leaf_sum, next);
    __syncthreads();      // leaf_sum / leaf_off are reused by the next call
    return out;
}

__global__ void __launch_bounds__(RB_THREADS) ranksum_block_kernel(const float* __restrict__ ps, int64_t n, int s,
                                                                   const int32_t* __restrict__ g1, int n1,
                                                                   const int32_t* __restrict__ g2, int n2, int P1,
                                                                   int P2, int leaf_max, RsOut o) {
    extern __shared__ float smemf[];
    float* A = smemf;             // [P1]
    float* B = A + P1;            // [P2]
    float* leaf_sum = B + P2;     // [leaf_max]
    float* scratch8 = leaf_sum + leaf_max;  // [leaf_max*8]
    int* leaf_off = reinterpret_cast<int*>(scratch8 + leaf_max * 8);  // [leaf_max+1]
    int* wcnt = leaf_off + leaf_max + 1;    // [8]
    __shared__ long long u2_s;
    const int tid = threadIdx.x;
    const float inf = __builtin_inff();
    for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
        const float* prow = ps + row * s;
        const int nv1 = block_compact(prow, g1, n1, A, wcnt);
        const int nv2 = block_compact(prow, g2, n2, B, wcnt);
        const bool tested = nv1 >= 3 && nv2 >= 3;   // block-uniform
        if (!tested) {
            if (tid == 0) {
                o.tested[row] = 0; o.p[row] = 0.0;
                if (o.z) o.z[row] = 0.0;
                o.med1[row] = 0.f; o.med2[row] = 0.f; o.mean1[row] = 0.f; o.mean2[row] = 0.f; o.delta[row] = 0.f;
            }
            __syncthreads();
            continue;
        }
        const float sum1 = block_pairwise_sum(A, nv1, leaf_off, leaf_sum, scratch8, leaf_max);
        const float sum2 = block_pairwise_sum(B, nv2, leaf_off, leaf_sum, scratch8, leaf_max);
        for (int i = nv1 + tid; i < P1; i += RB_THREADS) A[i] = inf;
        for (int i = nv2 + tid; i < P2; i += RB_THREADS) B[i] = inf;
        if (tid == 0) u2_s = 0;
        __syncthreads();
        // bitonic sort of both groups, sharing the barriers
        const int PM = P1 > P2 ? P1 : P2;
        for (int k = 2; k <= PM; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < PM; i += RB_THREADS) {
                    const int l = i ^ j;
                    if (l > i) {
                        const bool asc = (i & k) == 0;
                        if (k <= P1 && l < P1) {
                            const float x = A[i], y = A[l];
                            if ((x > y) == asc) { A[i] = y; A[l] = x; }
                        }
                        if (k <= P2 && l < P2) {
                            const float x = B[i], y = B[l];
                            if ((x > y) == asc) { B[i] = y; B[l] = x; }
                        }
                    }
                }
                __syncthreads();
            }
        }
        // 2U = sum_i lower_bound(B, a_i) + upper_bound(B, a_i)
        long long local = 0;
        for (int i = tid; i < nv1; i += RB_THREADS) {
            const float ai = A[i];
            int lo = 0, hi = nv2;
            while (lo < hi) { const int m = (lo + hi) >> 1; if (B[m] < ai) lo = m + 1; else hi = m; }
            const int lb = lo;
            hi = nv2;
            while (lo < hi) { const int m = (lo + hi) >> 1; if (B[m] <= ai) lo = m + 1; else hi = m; }
            local += lb + lo;
        }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) local += __shfl_xor(local, ofs);
        if ((tid & 63) == 0 && local) atomicAdd((unsigned long long*)&u2_s, (unsigned long long)local);
        __syncthreads();
        if (tid == 0) {
            double z, p;
            rs_finish(nv1, nv2, u2_s, z, p);
            const float med1 = median_sorted(A, nv1), med2 = median_sorted(B, nv2);
            o.tested[row] = 1; o.p[row] = p;
            if (o.z) o.z[row] = z;
            o.med1[row] = med1; o.med2[row] = med2;
            o.mean1[row] = sum1 / (float)nv1; o.mean2[row] = sum2 / (float)nv2;
            o.delta[row] = med1 - med2;
        }
        __syncthreads();
    }
}

// number of set bits of a wave mask below this lane (v_mbcnt_lo + v_mbcnt_hi: two instructions, no 64-bit mask per lane)
__device__ __forceinline__ int lanes_below(unsigned long long m, int base = 0) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)base));
}

// ------------------------------------------------------------------ wave-per-row variant
// 64 < max(n1, n2) <= 64*E: one WAVE per row, E values of a group per lane, no workgroup
// barrier anywhere.  The sort is a bitonic network over 64*E elements in "blocked" layout
// (element lane*E + e lives in register e of that lane): compare-exchange distances below E
// are register-to-register, the others are one cross-lane exchange per register.  The
// per-wave LDS buffers hold the compacted (then sorted) groups for the mean, the medians and
// the binary searches of the U statistic.
#define SD_WAVE_SYNC()                                        \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

template <int E>
__device__ __forceinline__ void bitonic_wave(float (&a)[E], int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * E; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= E) {
                const int lj = j / E;
                const bool upper = (lane & lj) != 0;
                const bool asc = (lane & (k / E)) == 0;   // k/E == 64 on the last merge: ascending
                const bool take_min = asc != upper;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float other = __shfl_xor(a[e], lj);
                    float lo, hi;
                    asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(a[e]), "v"(other));
                    asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a[e]), "v"(other));
                    a[e] = take_min ? lo : hi;
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int l = e ^ j;
                    if (l > e) {
                        const bool asc = (k < E) ? ((e & k) == 0) : ((lane & (k / E)) == 0);
                        float lo, hi;
                        asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(a[e]), "v"(a[l]));
                        asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a[e]), "v"(a[l]));
                        a[e] = asc ? lo : hi;
                        a[l] = asc ? hi : lo;
                    }
                }
            }
        }
    }
}

// numpy pairwise_sum of C[0..nv) (nv <= 1024) by one wave: lane = leaf*8 + j owns accumulator j
// of its leaf; the 8 accumulators are folded with three xor-exchanges, which reproduces
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) because float addition is commutative.  The halving keeps
// multiples of 8 on the left, so the larger half is up to len/2 + 7.5: 1024 values need up to FOUR
// levels (e.g. 972 -> 492 -> 252 -> 132 -> 68) and up to 16 leaves -- two rounds of 8 leaves.
constexpr int PW_WAVE_DEPTH = 4;

__device__ __forceinline__ float wave_pairwise_sum(const float* C, int nv, int lane, int* leaf_off, float* leaf_sum) {
    int nl = 0;
    pw_leaves<PW_WAVE_DEPTH>(0, nv, leaf_off, nl, lane == 0);
    if (lane == 0) leaf_off[nl] = nv;
    SD_WAVE_SYNC();
    const int j = lane & 7;
    for (int base = 0; base < nl; base += 8) {          // wave-uniform trip count
        const int L = base + (lane >> 3);
        int off = 0, len = 0;
        if (L < nl) { off = leaf_off[L]; len = leaf_off[L + 1] - off; }
        const int main_n = len - (len & 7);
        float r = 0.f;
        if (len >= 8) {
            r = C[off + j];
            for (int i = 8; i < main_n; i += 8) r += C[off + i + j];
        }
        r = r + __shfl_xor(r, 1);
        r = r + __shfl_xor(r, 2);
        r = r + __shfl_xor(r, 4);
        for (int i = (len >= 8 ? main_n : 0); i < len; ++i) r += C[off + i];
        if (j == 0 && L < nl) leaf_sum[L] = r;
    }
    SD_WAVE_SYNC();
    int next = 0;
    const float out = pw_combine<PW_WAVE_DEPTH>(nv, leaf_sum, next);
    SD_WAVE_SYNC();
    return out;
}

constexpr unsigned char RS_REDO = 0xFF;     // `tested` mark: row left to the sorting kernel by ranksum_count_kernel

template <int E>
__global__ void __launch_bounds__(256) ranksum_wave_kernel(const float* __restrict__ ps, int64_t n, int s,
                                                           const int32_t* __restrict__ g1, int n1,
                                                           const int32_t* __restrict__ g2, int n2, int ch,
                                                           int redo_only, RsOut o) {
    extern __shared__ __align__(16) float smemw[];
    constexpr int N = 64 * E;
    constexpr int WSTRIDE = 2 * N + 40;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    float* SA = smemw + (size_t)wave * WSTRIDE;
    float* SB = SA + N;
    float* leaf_sum = SB + N;                                  // [<= 16]
    int* leaf_off = reinterpret_cast<int*>(leaf_sum + 16);     // [<= 17]
    const float inf = __builtin_inff();
    // the selected columns of this lane (striped: selection k = e*64 + lane), loaded once
    int idx1[E], idx2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = e * 64 + lane;
        idx1[e] = k < n1 ? g1[k] : -1;
        idx2[e] = k < n2 ? g2[k] : -1;
    }
    // A wave owns chunks of `ch` consecutive rows.  Lane i keeps the integer / float results of
    // the chunk's i-th row; after the chunk all `ch` lanes do the double precision finish (z, erfc)
    // at once and the outputs go out as contiguous runs, instead of one lane per row doing both.
    const int64_t n_chunks = (n + ch - 1) / ch;
    for (int64_t c = (int64_t)blockIdx.x * wpb + wave; c < n_chunks; c += (int64_t)gridDim.x * wpb) {
      const int64_t row0 = c * ch;
      const int rows_here = (int)min((int64_t)ch, n - row0);
      int s_nv1 = 0, s_nv2 = 0, s_u2 = 0;
      float s_med1 = 0.f, s_med2 = 0.f, s_sum1 = 0.f, s_sum2 = 0.f;
      // redo mode (after ranksum_count_kernel): only the rows it marked RS_REDO are computed and written
      unsigned long long todo = ~0ull;
      if (redo_only) {
          const bool marked = lane < rows_here && o.tested[row0 + lane] == RS_REDO;
          todo = __ballot(marked);
          if (todo == 0ull) continue;
      }
      for (int ri = 0; ri < rows_here; ++ri) {
        if (!((todo >> ri) & 1ull)) continue;
        const float* prow = ps + (row0 + ri) * s;
        float x[E], y[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            x[e] = idx1[e] >= 0 ? prow[idx1[e]] : __builtin_nanf("");
            y[e] = idx2[e] >= 0 ? prow[idx2[e]] : __builtin_nanf("");
        }
        SD_WAVE_SYNC();          // the previous row's readers are done with SA / SB
        // ordered compaction (NaNs dropped): position = valid values in earlier rounds + lower lanes
        int nv1 = 0, nv2 = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool v1 = x[e] == x[e];
            const unsigned long long m1 = __ballot(v1);
            if (v1) SA[lanes_below(m1, nv1)] = x[e];
            nv1 += __popcll(m1);
            const bool v2 = y[e] == y[e];
            const unsigned long long m2 = __ballot(v2);
            if (v2) SB[lanes_below(m2, nv2)] = y[e];
            nv2 += __popcll(m2);
        }
        SD_WAVE_SYNC();
        if (nv1 < 3 || nv2 < 3) {          // wave-uniform
            if (lane == ri) { s_nv1 = nv1; s_nv2 = nv2; }
            continue;
        }
        const float sum1 = wave_pairwise_sum(SA, nv1, lane, leaf_off, leaf_sum);
        const float sum2 = wave_pairwise_sum(SB, nv2, lane, leaf_off, leaf_sum);
        // sort group 2 then group 1 through one copy of the network; x ends up holding sorted A
#pragma nounroll
        for (int gi = 1; gi >= 0; --gi) {
            float* S = gi ? SB : SA;
            const int nv = gi ? nv2 : nv1;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float v = S[lane * E + e];
                x[e] = (lane * E + e) < nv ? v : inf;
            }
            bitonic_wave<E>(x, lane);
#pragma unroll
            for (int e = 0; e < E; ++e) S[lane * E + e] = x[e];
        }
        SD_WAVE_SYNC();
        // 2U = sum_i lower_bound(B, a_i) + upper_bound(B, a_i); B is +inf padded to N, so the
        // searches are fixed-trip and branch-free, E of them interleaved per lane
        int lb[E], ub[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { lb[e] = 0; ub[e] = 0; }
#pragma unroll
        for (int step = N / 2; step >= 1; step >>= 1) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float v = SB[lb[e] + step - 1];
                const float w = SB[ub[e] + step - 1];
                if (v < x[e]) lb[e] += step;
                if (w <= x[e]) ub[e] += step;
            }
        }
        int local = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (SB[lb[e]] < x[e]) ++lb[e];
            if (SB[ub[e]] <= x[e]) ++ub[e];
            if (lane * E + e < nv1) local += min(lb[e], nv2) + min(ub[e], nv2);
        }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) local += __shfl_xor(local, ofs);
        const float med1 = median_sorted(SA, nv1), med2 = median_sorted(SB, nv2);   // broadcast reads
        if (lane == ri) {
            s_nv1 = nv1; s_nv2 = nv2; s_u2 = local;
            s_med1 = med1; s_med2 = med2; s_sum1 = sum1; s_sum2 = sum2;
        }
      }
      if (lane < rows_here && ((todo >> lane) & 1ull)) {
        const int64_t row = row0 + lane;
        const bool tested = s_nv1 >= 3 && s_nv2 >= 3;
        float mean1 = 0.f, mean2 = 0.f;
        if (tested) {
            mean1 = s_sum1 / (float)s_nv1;
            mean2 = s_sum2 / (float)s_nv2;
        }
        // (2U, n1, n2) travel to ranksum_finish_kernel in the bits of p[row]: the double
        // precision tail (sqrt, divide, erfc) would otherwise set this kernel's VGPR budget
        const unsigned long long packed =
            tested ? ((unsigned long long)(unsigned)s_u2 | ((unsigned long long)s_nv1 << 32) | ((unsigned long long)s_nv2 << 48))
                   : 0ull;
        o.tested[row] = tested ? 1 : 0;
        reinterpret_cast<unsigned long long*>(o.p)[row] = packed;
        o.med1[row] = s_med1; o.med2[row] = s_med2;
        o.mean1[row] = mean1; o.mean2[row] = mean2;
        o.delta[row] = s_med1 - s_med2;
      }
    }
}

// The same sums for TWO arrays of at most 512 values each (at most 8 leaves each) in one pass: lanes 0..31 work on A,
// lanes 32..63 on B, four leaves per half and round -- half the LDS instructions of two calls (the counting kernel
// issues most of its LDS instructions here).
__device__ __forceinline__ void wave_pairwise_sum2(const float* CA, int nvA, const float* CB, int nvB, int lane, int* leaf_off /* [18] */,
                                                   float* leaf_sum /* [16] */, float& sumA, float& sumB) {
    const bool second = lane >= 32;
    const float* C = second ? CB : CA;
    const int nv = second ? nvB : nvA;
    int* lo = leaf_off + (second ? 9 : 0);
    float* ls = leaf_sum + (second ? 8 : 0);
    int nl = 0;
    pw_leaves<PW_WAVE_DEPTH>(0, nv, lo, nl, (lane & 31) == 0);
    if ((lane & 31) == 0) lo[nl] = nv;
    SD_WAVE_SYNC();
    const int nl_max = max(__shfl(nl, 0), __shfl(nl, 32));
    const int j = lane & 7;
    for (int base = 0; base < nl_max; base += 4) {      // wave-uniform trip count
        const int L = base + ((lane >> 3) & 3);
        int off = 0, len = 0;
        if (L < nl) { off = lo[L]; len = lo[L + 1] - off; }
        const int main_n = len - (len & 7);
        float r = 0.f;
        if (len >= 8) {
            r = C[off + j];
            for (int i = 8; i < main_n; i += 8) r += C[off + i + j];
        }
        r = r + __shfl_xor(r, 1);
        r = r + __shfl_xor(r, 2);
        r = r + __shfl_xor(r, 4);
        for (int i = (len >= 8 ? main_n : 0); i < len; ++i) r += C[off + i];
        if (j == 0 && L < nl) ls[L] = r;
    }
    SD_WAVE_SYNC();
    int next = 0;
    const float out = pw_combine<PW_WAVE_DEPTH>(nv, ls, next);
    SD_WAVE_SYNC();
    sumA = __shfl(out, 0);
    sumB = __shfl(out, 32);
}

// ------------------------------------------------------------------ counting variant (quantised PS)
// compare_sample_sets always sees PS values that went through the '.3f' text of _allPS.tsv
// (SPLICEDICE.py:353 -> compareSampleSets.py:202), i.e. float32(k / 1000.0) for k = 0..1000: a row's
// two groups are then two HISTOGRAMS over 1001 bins and nothing has to be sorted:
//   2U = sum_v a_v * (2 * #{b < v} + b_v),   medians = the bins where the cumulative counts cross n/2.
// One wave per row as in the wave kernel (same ordered compaction and numpy pairwise sums for the
// means); the histogram of both groups is one array of packed 16+16-bit counters in LDS, filled with
// LDS atomics (the two heavy bins 0.000 and 1.000 are counted with ballots instead), scanned with
// 16 bins per lane.  A row holding any value that is NOT exactly float32(k/1000) is left to the
// sorting kernel: it is marked RS_REDO in `tested` and ranksum_wave_kernel is run over the marked rows.
constexpr int RS_BINS = 1024;            // 1001 used
#ifndef RS_BALLOT_EXTREMES
#define RS_BALLOT_EXTREMES 1
#endif

// (occupancy hint: the kernel is latency bound; capping it at 128 VGPRs -- 4 waves per SIMD -- costs
//  E = 8 two spilled dwords and buys 25 %; E = 16 would spill 55 dwords and keeps its 2 waves)
template <int E>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(E <= 4 ? 6 : E <= 8 ? 5 : 3, 8))) ranksum_count_kernel(const float* __restrict__ ps, int64_t n, int s,
                                                            const int32_t* __restrict__ g1, int n1,
                                                            const int32_t* __restrict__ g2, int n2, int ch,
                                                            int abl /* timing experiments: 1 = always the ordered compaction */, RsOut o) {
    extern __shared__ __align__(16) float smemc[];
    constexpr int N = 64 * E;
    // for E >= 8 the histogram lives in the SA | SB area (1024 words): the compacted values are dead once the two pairwise sums
    // are taken, and 4 KB less per wave lets a fifth wave per SIMD in (the kernel is latency bound)
    constexpr bool H_ALIAS = 2 * N >= RS_BINS;
    constexpr int WSTRIDE = 2 * N + 40 + (H_ALIAS ? 0 : RS_BINS) + 8;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    float* SA = smemc + (size_t)wave * WSTRIDE;
    float* SB = SA + N;
    float* leaf_sum = SB + N;
    int* leaf_off = reinterpret_cast<int*>(leaf_sum + 16);
    unsigned* H = H_ALIAS ? reinterpret_cast<unsigned*>(SA) : reinterpret_cast<unsigned*>(SB + N + 40);    // [RS_BINS] a_v | b_v << 16
    int idx1[E], idx2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = e * 64 + lane;
        idx1[e] = k < n1 ? g1[k] : -1;
        idx2[e] = k < n2 ? g2[k] : -1;
    }
    const int64_t n_chunks = (n + ch - 1) / ch;
    for (int64_t c = (int64_t)blockIdx.x * wpb + wave; c < n_chunks; c += (int64_t)gridDim.x * wpb) {
      const int64_t row0 = c * ch;
      const int rows_here = (int)min((int64_t)ch, n - row0);
      int s_nv1 = 0, s_nv2 = 0, s_u2 = 0, s_redo = 0;
      float s_med1 = 0.f, s_med2 = 0.f, s_sum1 = 0.f, s_sum2 = 0.f;
      for (int ri = 0; ri < rows_here; ++ri) {
        const float* prow = ps + (row0 + ri) * s;
        float x[E], y[E];
        __builtin_amdgcn_s_setprio(3);                  // (loads first: ~1 % -- see ps.hip)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            x[e] = idx1[e] >= 0 ? __builtin_nontemporal_load(prow + idx1[e]) : __builtin_nanf("");
            y[e] = idx2[e] >= 0 ? __builtin_nontemporal_load(prow + idx2[e]) : __builtin_nanf("");
        }
        __builtin_amdgcn_s_setprio(0);
        // bin of every value and the check that the value IS its bin's float; kk[e] packs (bin + 1) of the
        // group-1 value in the low half and of the group-2 value in the high half, 0 = NaN
        unsigned kk[E];
        bool ok = true;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            unsigned packed_k = 0;
            {
                // (a value outside [0, 1] clamps to a key whose float it is not: no separate range check; arithmetic, no table)
                const float v = x[e];
                const float kf = __builtin_amdgcn_fmed3f(rintf(v * 1000.0f), 0.0f, 1000.0f);
                const int k = (int)kf;
                ok = ok && (v != v || ps_of_key(kf) == v);
                packed_k = (v != v) ? 0u : (unsigned)(k + 1);
            }
            {
                const float v = y[e];
                const float kf = __builtin_amdgcn_fmed3f(rintf(v * 1000.0f), 0.0f, 1000.0f);
                const int k = (int)kf;
                ok = ok && (v != v || ps_of_key(kf) == v);
                packed_k |= (v != v) ? 0u : ((unsigned)(k + 1) << 16);
            }
            kk[e] = packed_k;
        }
        if (__ballot(ok) != ~0ull) {                 // some value is not a 3-decimal float: the sorting kernel's row
            if (lane == ri) s_redo = 1;
            continue;
        }
        SD_WAVE_SYNC();          // the previous row's readers are done with the wave's LDS
        int nv1 = 0, nv2 = 0;
        bool hole = false;       // a selected value that is NaN
#pragma unroll
        for (int e = 0; e < E; ++e)
            hole = hole || (idx1[e] >= 0 && (kk[e] & 0xffffu) == 0u) || (idx2[e] >= 0 && (kk[e] >> 16) == 0u);
        if (!(abl & 1) && __ballot(hole) == 0ull) {
            // no NaN among the row's selected values (nearly every row): the compacted order IS the selection order --
            // plain stores, none of the 2 E ballots, prefix counts and popcounts of the ordered compaction
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (idx1[e] >= 0) SA[e * 64 + lane] = x[e];
                if (idx2[e] >= 0) SB[e * 64 + lane] = y[e];
            }
            nv1 = n1; nv2 = n2;
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool v1 = (kk[e] & 0xffffu) != 0u;
                const unsigned long long m1 = __ballot(v1);
                if (v1) SA[lanes_below(m1, nv1)] = x[e];
                nv1 += __popcll(m1);
                const bool v2 = (kk[e] >> 16) != 0u;
                const unsigned long long m2 = __ballot(v2);
                if (v2) SB[lanes_below(m2, nv2)] = y[e];
                nv2 += __popcll(m2);
            }
        }
        // clear the histogram: 16 words per lane (when it shares the SA | SB area: after the sums)
        if (!H_ALIAS) {
#pragma unroll
            for (int q = 0; q < RS_BINS / 64 / 4; ++q) reinterpret_cast<uint4*>(H)[lane * (RS_BINS / 64 / 4) + q] = make_uint4(0, 0, 0, 0);
        }
        SD_WAVE_SYNC();
        if (nv1 < 3 || nv2 < 3) {          // wave-uniform
            if (lane == ri) { s_nv1 = nv1; s_nv2 = nv2; }
            continue;
        }
        float sum1, sum2;
        if (E <= 8) {
            wave_pairwise_sum2(SA, nv1, SB, nv2, lane, leaf_off, leaf_sum, sum1, sum2);
        } else {
            sum1 = wave_pairwise_sum(SA, nv1, lane, leaf_off, leaf_sum);
            sum2 = wave_pairwise_sum(SB, nv2, lane, leaf_off, leaf_sum);
        }
        if (H_ALIAS) {
            SD_WAVE_SYNC();                 // every lane has read its values
#pragma unroll
            for (int q = 0; q < RS_BINS / 64 / 4; ++q) reinterpret_cast<uint4*>(H)[lane * (RS_BINS / 64 / 4) + q] = make_uint4(0, 0, 0, 0);
            SD_WAVE_SYNC();
        }
        // histogram: ballots for the two heavy bins, LDS atomics for the rest
        unsigned c_lo = 0, c_hi = 0;         // packed (group 1 | group 2 << 16) counts of bins 0 and 1000
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int kxe = (int)(kk[e] & 0xffffu) - 1, kye = (int)(kk[e] >> 16) - 1;     // -1 = NaN
            if (RS_BALLOT_EXTREMES) {
                c_lo += (unsigned)__popcll(__ballot(kxe == 0)) + ((unsigned)__popcll(__ballot(kye == 0)) << 16);
                c_hi += (unsigned)__popcll(__ballot(kxe == 1000)) + ((unsigned)__popcll(__ballot(kye == 1000)) << 16);
                if (kxe > 0 && kxe < 1000) atomicAdd(&H[kxe], 1u);
                if (kye > 0 && kye < 1000) atomicAdd(&H[kye], 0x10000u);
            } else {
                if (kxe >= 0) atomicAdd(&H[kxe], 1u);
                if (kye >= 0) atomicAdd(&H[kye], 0x10000u);
            }
        }
        if (RS_BALLOT_EXTREMES) {            // the two ballot-counted bins join the histogram (no atomics touched them)
            if (lane == 0) H[0] = c_lo;
            if (lane == 1) H[1000] = c_hi;
        }
        SD_WAVE_SYNC();
        // scan: lane owns bins [16 * lane, 16 * lane + 16)
        int local = 0, cumA0, cumB0, totA, totB;
        {
            unsigned w[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 t4 = reinterpret_cast<const uint4*>(H)[lane * 4 + q];
                w[4 * q] = t4.x; w[4 * q + 1] = t4.y; w[4 * q + 2] = t4.z; w[4 * q + 3] = t4.w;
            }
            unsigned tot = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += w[q];           // halves stay below 2^16 (at most 1024 values per group)
            unsigned pre = tot;
#pragma unroll
            for (int ofs = 1; ofs < 64; ofs <<= 1) {
                const unsigned up = __shfl_up(pre, ofs);
                if (lane >= ofs) pre += up;
            }
            pre -= tot;
            cumA0 = (int)(pre & 0xffffu); cumB0 = (int)(pre >> 16);
            totA = (int)(tot & 0xffffu); totB = (int)(tot >> 16);
            int cumB = cumB0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int a = (int)(w[q] & 0xffffu), b = (int)(w[q] >> 16);
                local += a * (2 * cumB + b);
                cumB += b;
            }
        }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) local += __shfl_xor(local, ofs);
        // medians: the bin where a group's cumulative count crosses a middle position.  The lane whose 16
        // bins contain the position is found with a ballot; its 16 counters are then examined by lanes
        // 0..15 together (prefix by shuffles, crossing by a second ballot) -- uniform, no divergent walk.
        auto find_bin = [&](int target, bool second) -> int {
            const int c0 = second ? cumB0 : cumA0, tt = second ? totB : totA;
            const int L = __ffsll((long long)__ballot(target >= c0 && target < c0 + tt)) - 1;
            const int base = __shfl(c0, L);
            const unsigned wq = lane < 16 ? H[L * 16 + lane] : 0u;
            int inc = second ? (int)(wq >> 16) : (int)(wq & 0xffffu);
#pragma unroll
            for (int ofs = 1; ofs < 16; ofs <<= 1) {
                const int up = __shfl_up(inc, ofs);
                if (lane >= ofs) inc += up;
            }
            return L * 16 + (__ffsll((long long)__ballot(lane < 16 && target < base + inc)) - 1);
        };
        const int hA = nv1 >> 1, hB = nv2 >> 1;
        const int binA1 = find_bin(hA, false), binB1 = find_bin(hB, true);
        const int binA0 = (nv1 & 1) ? binA1 : find_bin(hA - 1, false);      // wave-uniform branches
        const int binB0 = (nv2 & 1) ? binB1 : find_bin(hB - 1, true);
        // np.median: odd -> middle value; even -> (v[h-1] + v[h]) / 2 in float32
        const float a0 = ps_of_key((float)binA0), a1 = ps_of_key((float)binA1), b0 = ps_of_key((float)binB0), b1 = ps_of_key((float)binB1);
        const float med1 = (nv1 & 1) ? a1 : (a0 + a1) / 2.0f;
        const float med2 = (nv2 & 1) ? b1 : (b0 + b1) / 2.0f;
        if (lane == ri) {
            s_nv1 = nv1; s_nv2 = nv2; s_u2 = local;
            s_med1 = med1; s_med2 = med2; s_sum1 = sum1; s_sum2 = sum2;
        }
      }
      if (lane < rows_here) {
        const int64_t row = row0 + lane;
        const bool tested = s_nv1 >= 3 && s_nv2 >= 3;
        float mean1 = 0.f, mean2 = 0.f;
        if (tested) {
            mean1 = s_sum1 / (float)s_nv1;
            mean2 = s_sum2 / (float)s_nv2;
        }
        const unsigned long long packed =
            tested ? ((unsigned long long)(unsigned)s_u2 | ((unsigned long long)s_nv1 << 32) | ((unsigned long long)s_nv2 << 48))
                   : 0ull;
        o.tested[row] = s_redo ? RS_REDO : (tested ? 1 : 0);
        reinterpret_cast<unsigned long long*>(o.p)[row] = packed;
        o.med1[row] = s_med1; o.med2[row] = s_med2;
        o.mean1[row] = mean1; o.mean2[row] = mean2;
        o.delta[row] = s_med1 - s_med2;
      }
    }
}

__global__ void __launch_bounds__(256) ranksum_finish_kernel(int64_t n, double* __restrict__ p_io,
                                                             double* __restrict__ z_out) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    const unsigned long long packed = reinterpret_cast<const unsigned long long*>(p_io)[row];
    double z = 0.0, p = 0.0;
    if (packed) rs_finish((int)((packed >> 32) & 0xffff), (int)(packed >> 48), (long long)(packed & 0xffffffffull), z, p);
    p_io[row] = p;
    if (z_out) z_out[row] = z;
}

int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

template <int E>
int launch_wave(sdice_ctx* ctx, const float* d_ps, int64_t n, int s, const int32_t* g1, int n1, const int32_t* g2,
                int n2, RsOut o, bool counting) {
    const int waves = 4;
    const size_t lds = (size_t)waves * (2 * 64 * E + 40) * 4;
    // rows per chunk: as many as keeps every wave slot of the chip (32 per CU) busy twice over
    const int64_t slots = (int64_t)ctx->n_cu * 32;
    int ch = 64;
    while (ch > 1 && sd_ceil_div(n, ch) < 2 * slots) ch >>= 1;
    int64_t blocks = sd_ceil_div(sd_ceil_div(n, ch), waves);
    const int64_t cap = (int64_t)ctx->n_cu * 8;
    if (blocks > cap) blocks = cap;
    if (counting) {
        // histogram path for rows of 3-decimal PS values; the sorting kernel then takes the rows it marked
        const size_t lds_c = (size_t)(waves * (2 * 64 * E + 40 + (2 * 64 * E >= RS_BINS ? 0 : RS_BINS) + 8)) * 4;
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ranksum_count_kernel<E>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        SD_LAUNCH(ctx, "ranksum_count_kernel", (ranksum_count_kernel<E>), dim3((unsigned)blocks), dim3(waves * 64), lds_c,
                  d_ps, n, s, g1, n1, g2, n2, ch, (int)ctx->param("ranksum.ablate", 0), o);
    }
    SD_LAUNCH(ctx, "ranksum_wave_kernel", (ranksum_wave_kernel<E>), dim3((unsigned)blocks), dim3(waves * 64), lds, d_ps, n,
              s, g1, n1, g2, n2, ch, counting ? 1 : 0, o);
    SD_LAUNCH(ctx, "ranksum_finish_kernel", ranksum_finish_kernel, dim3((unsigned)sd_ceil_div(n, 256)), dim3(256), 0, n,
              o.p, o.z);
    return SDICE_OK;
}

template <int P>
int launch_lane(sdice_ctx* ctx, const float* d_ps, int64_t n, int s, const int32_t* g1, int n1, const int32_t* g2, int n2, RsOut o) {
    const int stride = (n1 + n2) | 1;
    int waves = 2;
    const size_t lds = (size_t)waves * 64 * stride * 4 + (size_t)(n1 + n2) * 4;
    int64_t groups = sd_ceil_div(n, 64);
    int64_t blocks = sd_ceil_div(groups, waves);
    const int64_t cap = (int64_t)ctx->n_cu * 16;
    if (blocks > cap) blocks = cap;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ranksum_lane_kernel<P>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "ranksum_lane_kernel", (ranksum_lane_kernel<P>), dim3((unsigned)blocks), dim3(waves * 64), lds, d_ps, n,
              s, g1, g2, n1, n2, stride, o, (int)ctx->param("ranksum.ablate", 0));
    SD_LAUNCH(ctx, "ranksum_finish_kernel", ranksum_finish_kernel, dim3((unsigned)sd_ceil_div(n, 256)), dim3(256), 0, n,
              o.p, o.z);
    return SDICE_OK;
}

template <int P>
int launch_pair(sdice_ctx* ctx, const float* d_ps, int64_t n, int s, const int32_t* g1, int n1, const int32_t* g2, int n2, RsOut o) {
    const int stride = (n1 + n2) | 1;
    const int waves = 2;
    const size_t lds = (size_t)waves * 32 * stride * 4 + (size_t)(n1 + n2) * 4;
    int64_t blocks = sd_ceil_div(sd_ceil_div(n, 32), waves);
    const int64_t cap = (int64_t)ctx->n_cu * 24;
    if (blocks > cap) blocks = cap;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ranksum_pair_kernel<P>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "ranksum_pair_kernel", (ranksum_pair_kernel<P>), dim3((unsigned)blocks), dim3(waves * 64), lds, d_ps, n,
              s, g1, g2, n1, n2, stride, o);
    SD_LAUNCH(ctx, "ranksum_finish_kernel", ranksum_finish_kernel, dim3((unsigned)sd_ceil_div(n, 256)), dim3(256), 0, n,
              o.p, o.z);
    return SDICE_OK;
}

// 16-bit-key lane-pair kernel, then the sorting wave kernel over the rows it marked (non 3-decimal values)
template <int P>
int launch_pairq(sdice_ctx* ctx, const float* d_ps, int64_t n, int s, const int32_t* g1, int n1,
                 const int32_t* g2, int n2, RsOut o) {
    const int stride = RSQ_STRIDE;                         // u16 units (n1, n2 <= 64)
    const int waves = 4;
    const int nsel = n1 + n2;
    const size_t lds = (size_t)((nsel + 1) & ~1) * 4 + (size_t)waves * 32 * stride * 2;
    int64_t blocks = sd_ceil_div(sd_ceil_div(n, 32), waves);
    const int64_t cap = (int64_t)ctx->n_cu * 16;
    if (blocks > cap) blocks = cap;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ranksum_pairq_kernel<P>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "ranksum_pairq_kernel", (ranksum_pairq_kernel<P>), dim3((unsigned)blocks), dim3(waves * 64), lds, d_ps, n,
              s, g1, g2, n1, n2, stride, o);
    {   // rows marked RS_REDO (a value that is not float32(k/1000)): the float sorting kernel, marked rows only
        const int ww = 4;
        const size_t lds_w = (size_t)ww * (2 * 64 + 40) * 4;
        const int64_t slots = (int64_t)ctx->n_cu * 32;
        int ch = 64;
        while (ch > 1 && sd_ceil_div(n, ch) < 2 * slots) ch >>= 1;
        int64_t wb = sd_ceil_div(sd_ceil_div(n, ch), ww);
        if (wb > (int64_t)ctx->n_cu * 8) wb = (int64_t)ctx->n_cu * 8;
        SD_LAUNCH(ctx, "ranksum_wave_kernel", (ranksum_wave_kernel<1>), dim3((unsigned)wb), dim3(ww * 64), lds_w, d_ps, n, s, g1,
                  n1, g2, n2, ch, 1, o);
    }
    SD_LAUNCH(ctx, "ranksum_finish_kernel", ranksum_finish_kernel, dim3((unsigned)sd_ceil_div(n, 256)), dim3(256), 0, n,
              o.p, o.z);
    return SDICE_OK;
}

}  // namespace

extern "C" int sdice_ranksum_dev(sdice_ctx* ctx, int64_t n, int32_t s, const float* d_ps, const int32_t* d_g1,
                                 int32_t n1, const int32_t* d_g2, int32_t n2, uint8_t* d_tested, double* d_p,
                                 double* d_z, float* d_med1, float* d_med2, float* d_mean1, float* d_mean2,
                                 float* d_delta) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0 && n1 >= 0 && n2 >= 0, "negative size");
    if (n == 0) return SDICE_OK;
    SD_ARG(d_tested && d_p && d_med1 && d_med2 && d_mean1 && d_mean2 && d_delta, "NULL output");
    SD_ARG((n1 == 0 || d_g1) && (n2 == 0 || d_g2), "NULL group index");
    SD_HIP(hipSetDevice(ctx->device));
    SD_TRY(ctx->arena.reset(ctx->stream));
    RsOut o{d_tested, d_p, d_z, d_med1, d_med2, d_mean1, d_mean2, d_delta};
    if (n1 < 3 || n2 < 3) {
        // no row can be tested (compareSampleSets.py:223)
        SD_HIP(hipMemsetAsync(d_tested, 0, (size_t)n, ctx->stream));
        SD_HIP(hipMemsetAsync(d_p, 0, (size_t)n * 8, ctx->stream));
        if (d_z) SD_HIP(hipMemsetAsync(d_z, 0, (size_t)n * 8, ctx->stream));
        float* f[5] = {d_med1, d_med2, d_mean1, d_mean2, d_delta};
        for (auto q : f) SD_HIP(hipMemsetAsync(q, 0, (size_t)n * 4, ctx->stream));
        return SDICE_OK;
    }
    SD_ARG(d_ps, "ps is NULL");
    const int64_t variant = ctx->param("ranksum.variant", 0);  // 0 auto, 1 lane, 2 block, 3 wave (sorting only), 4 lane pair, 5 counting + wave
    const bool lane_ok = n1 <= 64 && n2 <= 64;
    SD_ARG((variant != 1 && variant != 4) || lane_ok, "lane variants need n1, n2 <= 64");
    if ((variant == 0 && lane_ok) || variant == 1 || variant == 4) {
        // (the kernels read the two index lists themselves: two device-to-device copies into one list cost 9 us per call)
        const int big = n1 > n2 ? n1 : n2;
        if (variant == 0 && big > 16 && n1 <= 63) {     // auto, groups of 17..64 (group 1 <= 63: its sentinel slot): the lane-pair kernel on 16-bit keys
            if (big <= 32) return launch_pairq<32>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
            return launch_pairq<64>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
        }
        if (variant != 1) {      // groups <= 16 and variant 4: the float lane-pair kernel (1.45x the lane kernel at 50 v 50)
            switch (next_pow2(big < 8 ? 8 : big)) {
                case 8: return launch_pair<8>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
                case 16: return launch_pair<16>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
                case 32: return launch_pair<32>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
                default: return launch_pair<64>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
            }
        }
        switch (next_pow2(big < 8 ? 8 : big)) {
            case 8: return launch_lane<8>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
            case 16: return launch_lane<16>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
            case 32: return launch_lane<32>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
            default: return launch_lane<64>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o);
        }
    }
    const int big = n1 > n2 ? n1 : n2;
    SD_ARG((variant != 3 && variant != 5) || big <= 1024, "wave / counting variants need n1, n2 <= 1024");
    if ((variant == 0 && big <= 1024) || variant == 3 || variant == 5) {
        // auto and 5: histogram kernel for rows of 3-decimal values + sorting kernel for the rest; 3: sorting only
        const bool counting = variant != 3;
        switch (next_pow2(big < 64 ? 64 : big) / 64) {
            case 1: return launch_wave<1>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o, counting);
            case 2: return launch_wave<2>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o, counting);
            case 4: return launch_wave<4>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o, counting);
            case 8: return launch_wave<8>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o, counting);
            default: return launch_wave<16>(ctx, d_ps, n, s, d_g1, n1, d_g2, n2, o, counting);
        }
    }
    const int P1 = next_pow2(n1), P2 = next_pow2(n2);
    SD_ARG(big <= 4096, "group larger than 4096 samples is not supported");
    const int leaf_max = 72;             // <= 2^6 leaves of numpy's pairwise recursion (+ sentinel)
    const size_t lds = (size_t)(P1 + P2 + leaf_max * 9) * 4 + (size_t)(leaf_max + 1 + 8) * 4;
    SD_ARG(lds <= 150 * 1024, "groups too large for LDS");
    int64_t blocks = n;
    const int64_t cap = (int64_t)ctx->n_cu * 8;
    if (blocks > cap) blocks = cap;
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ranksum_block_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_LAUNCH(ctx, "ranksum_block_kernel", ranksum_block_kernel, dim3((unsigned)blocks), dim3(RB_THREADS), lds, d_ps, n,
              (int)s, d_g1, (int)n1, d_g2, (int)n2, P1, P2, leaf_max, o);
    return SDICE_OK;
}

extern "C" int sdice_ranksum(sdice_ctx* ctx, int64_t n, int32_t s, const float* ps, const int32_t* g1, int32_t n1,
                             const int32_t* g2, int32_t n2, uint8_t* tested, double* p, double* z, float* med1,
                             float* med2, float* mean1, float* mean2, float* delta) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0 && n1 >= 0 && n2 >= 0, "negative size");
    if (n == 0) return SDICE_OK;
    SD_ARG(tested && p && med1 && med2 && mean1 && mean2 && delta, "NULL output");
    for (int i = 0; i < n1; ++i) SD_ARG(g1[i] >= 0 && g1[i] < s, "g1 index out of range");
    for (int i = 0; i < n2; ++i) SD_ARG(g2[i] >= 0 && g2[i] < s, "g2 index out of range");
    float* d_ps = nullptr;
    int32_t *dg1 = nullptr, *dg2 = nullptr;
    uint8_t* dt = nullptr;
    double *dp = nullptr, *dz = nullptr;
    float* df = nullptr;
    int rc = sdice_dmalloc(ctx, n * s * 4, (void**)&d_ps);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, (int64_t)n1 * 4, (void**)&dg1);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, (int64_t)n2 * 4, (void**)&dg2);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n, (void**)&dt);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * 8, (void**)&dp);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * 8, (void**)&dz);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * 4 * 5, (void**)&df);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, d_ps, ps, n * s * 4);
    if (rc == SDICE_OK && n1) rc = sdice_h2d(ctx, dg1, g1, (int64_t)n1 * 4);
    if (rc == SDICE_OK && n2) rc = sdice_h2d(ctx, dg2, g2, (int64_t)n2 * 4);
    if (rc == SDICE_OK)
        rc = sdice_ranksum_dev(ctx, n, s, d_ps, dg1, n1, dg2, n2, dt, dp, dz, df, df + n, df + 2 * n, df + 3 * n,
                               df + 4 * n);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, tested, dt, n);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, p, dp, n * 8);
    if (rc == SDICE_OK && z) rc = sdice_d2h(ctx, z, dz, n * 8);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, med1, df, n * 4);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, med2, df + n, n * 4);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, mean1, df + 2 * n, n * 4);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, mean2, df + 3 * n, n * 4);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, delta, df + 4 * n, n * 4);
    sdice_dfree(ctx, d_ps); sdice_dfree(ctx, dg1); sdice_dfree(ctx, dg2); sdice_dfree(ctx, dt);
    sdice_dfree(ctx, dp); sdice_dfree(ctx, dz); sdice_dfree(ctx, df);
    return rc;
}
