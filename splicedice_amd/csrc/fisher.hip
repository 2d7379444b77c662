// K6: two-sided Fisher exact test for every sample pair of every junction.
//
// Replaces the inner loop of pairwise (pairwise_fisher.py:164-179):
//   table = [[incl_a, incl_b], [excl_a, excl_b]];  p = scipy.stats.fisher_exact(table)[1]
// scipy (1.15.3, _stats_py.py fisher_exact): any zero margin -> 1.0; otherwise, with
// n1 = a+b, n2 = c+d, n = a+c and the hypergeometric pmf over k = table[0][0], the
// two-sided p is the sum of pmf(k) over the support with pmf(k) <= pmf(a) * (1 + 1e-14)
// (its cdf/sf + binary-search branches compute exactly that set), clipped at 1.
//
// Here: pmf(a) from a log-factorial table in HBM/L2 (device lgamma beyond the table), and
// the RATIOS r_k = pmf(k)/pmf(a) by the exact one-step recurrence outward from k = a in both
// directions, so the tie test r_k <= 1 + 1e-12 is decided to ~1e-14, far tighter than a
// log-gamma difference could.  p = pmf(a) * (1 + sum of the accepted r_k).  Monotone tails are
// cut once the remaining mass is below 1e-13 of the sum (the pmf is log-concave).
// The kernel is f64-VALU bound (O(support) steps per p-value), not HBM bound.
#include "common.h"
#include <math.h>
#include <algorithm>

namespace {

struct LfTable {
    const double* lf;   // lf[k] = lgamma(k + 1), k in [0, n)
    long long n;
};

// beyond the table (margins above fisher.table_max): a real call, so that the nine look-ups of a
// p-value do not inline nine copies of lgamma and their registers into the kernels
__device__ __noinline__ double lgamma_beyond_table(double x) { return lgamma(x); }

__device__ __forceinline__ double logfact(const LfTable& t, long long k) {
    return k < t.n ? t.lf[k] : lgamma_beyond_table((double)k + 1.0);
}

// 1/y for y an integer-valued double in [1, 2^63): hardware seed + two Newton steps (no scaling or
// fix-up is ever needed in this range).  Accurate to ~1 ulp; the step ratios it feeds carry a
// relative error of a few 1e-16, against a tie slack of 1e-12 and a p-value tolerance of 1e-9.
// The IEEE division it replaces was ~2/3 of the instructions of a walk step.
__device__ __forceinline__ double rcp_pos(double y) {
    double x = __builtin_amdgcn_rcp(y);
    double e = fma(-y, x, 1.0);
    x = fma(x, e, x);
    e = fma(-y, x, 1.0);
    return fma(x, e, x);
}

// One-directional walk of the ratios r_k = pmf(k)/pmf(a) away from k = a.
//   up:   rho = (n1-k)(n-k) / ((k+1)(n2-n+k+1))        down: rho = k (n2-n+k) / ((n1-k+1)(n-k+1))
// Both are (u1 u2)/(v1 v2) with u falling and v rising by one per step, so one step routine serves both.
// No division per step: r_k = P_k/Q_k with P = prod(u1 u2), Q = prod(v1 v2), and the accepted ratios are
// summed over the common denominator, S_k = S_{k-1} (v1 v2) + [accepted] P_k, so the side's sum is S/Q --
// one reciprocal per p-value.  "Accepted" (pmf(k) <= pmf(a) (1 + 1e-12)) is P <= slack Q; the products carry
// a few 1e-16 of rounding per step.
// The step is BRANCH-FREE and needs no end test: N = u1 u2 and D = v1 v2 are quadratics in the step index,
// advanced by their first differences (N += dN, dN += 2: exact integers below 2^63), and at the end of the
// support one of u1, u2 is zero, so N = 0 there, P = 0 from then on and further steps leave S/Q unchanged (both
// take the same factor D).  A lane may therefore run past its end, and a lane with nothing to do may step on
// stale state: the state machine issues steps for the whole wave without touching the exec mask
// (10 VALU per step -- 2 mul, 1 fma, 4 add, 1 compare, 2 select -- against 15 + ~10 scalar for the exec-masked step
// with the four factors kept separately).
// Scaling (exact powers of two), once per TRIP of at most 8 steps (a step multiplies P and Q by less than 2^62,
// a trip by less than 2^496): Q is brought back into [1, 2) with P and S; P, which outgrows Q while the walk
// crosses the mode, is kept below 2^500 by an exponent count eP (true P = P 2^(500 eP)).  eP > 0 is only left
// standing when P >= 2^-3, i.e. true P / Q > 2^496 at the start of a trip: such a lane cannot come back under
// Q within the trip, so "accepted" is (eP == 0 at the start of the trip) && P <= Q.
struct Walk {
    double N, dN, D, dD, P, Q, S, dN_end;
    int eP;
    // P carries a factor 1/slack, so that "accepted" is simply P <= Q; sum() puts it back
    static constexpr double SLACK = 1.0 + 1e-12, INV_SLACK = 1.0 / (1.0 + 1e-12);
    static constexpr int TRIP = 8;            // steps between two rescale() calls, at most
    __device__ __forceinline__ void side(double u1, double u2, double v1, double v2, double steps) {
        N = u1 * u2; dN = 1.0 - (u1 + u2);
        D = v1 * v2; dD = v1 + v2 + 1.0;
        dN_end = dN + 2.0 * steps;
    }
    __device__ __forceinline__ void start(double u1, double u2, double v1, double v2, double steps) {
        side(u1, u2, v1, v2, steps);
        P = INV_SLACK; Q = 1.0; S = 0.0; eP = 0;
    }
    // the walk on the other side of a continues on the same denominator: r = 1 again means P = Q / slack,
    // and S / Q ends as the sum over both sides
    __device__ __forceinline__ void turn(double u1, double u2, double v1, double v2, double steps) {
        side(u1, u2, v1, v2, steps);
        P = Q * INV_SLACK; eP = 0;
    }
    // ok: eP == 0 at the last rescale()
    __device__ __forceinline__ void step(bool ok) {
        P *= N; Q *= D;
        S = fma(S, D, (ok && P <= Q) ? P : 0.0);
        N += dN; dN += 2.0;
        D += dD; dD += 2.0;
    }
    __device__ __forceinline__ bool at_end() const { return dN >= dN_end; }
    // what remains of this side is below 1e-13 of the sum: the next ratio of ratios is below 1/2 and falls
    // from here on (the pmf is log-concave), so the rest is less than the last accepted term.  (1e-13 against a
    // p-value tolerance of 1e-9 -- north_star asks for 1e-6 --; the earlier 1e-18 bought nothing but ~10 % more steps:
    // a Gaussian-like tail needs ~1.4 sigma more to fall from 1e-13 to 1e-18 of the peak.)
    __device__ __forceinline__ bool tail_negligible() const {
        return eP == 0 && P <= Q && N + N < D && P < 1e-13 * (Q + S);
    }
    // may TRIP more steps be taken without rescale()?  A step multiplies P and Q by less than 2^62 (a trip: 2^496) and
    // S <= 2^31 Q: yes while Q and P are below 2^490 (2^490 * 2^496 * 2^31 < 2^1023) -- with counts in the thousands Q
    // grows by ~2^20 per step and the answer is yes; a lane with eP > 0 keeps its per-trip invariant only through
    // rescale().  Wave-uniform answer.
    __device__ __forceinline__ bool rescale_due() const {
        return __ballot(Q > 0x1p490 || P > 0x1p490 || eP != 0) != 0ull;
    }
    __device__ __forceinline__ void rescale() {
        // Q >= 1 (a normal number): back into [1, 2) by the power of two whose exponent field mirrors Q's -- three
        // full-rate multiplications (v_frexp_exp / v_ldexp_f64 are quarter-rate: the same scaling cost 20 issue slots)
        const unsigned ef = ((unsigned)__double2hiint(Q) >> 20) & 0x7ffu;
        const double sc = __hiloint2double((int)((2046u - ef) << 20), 0);
        Q *= sc; S *= sc; P *= sc;
        const bool up = P > 0x1p500, dn = eP > 0 && P < 0x1p-3;
        if (__ballot(up || dn) != 0ull) {                       // (wave-uniform: a ratio beyond 2^500 is a rare guest)
            P = __builtin_ldexp(P, up ? -500 : dn ? 500 : 0);
            eP += up ? 1 : dn ? -1 : 0;
        }
    }
    // slack S / Q, Q in [1, 2): hardware seed (~2^-23) + one Newton step -> ~1e-14, against a tolerance of 1e-9
    __device__ __forceinline__ double sum() const {
        double x = __builtin_amdgcn_rcp(Q);
        const double e = fma(-Q, x, 1.0);
        x = fma(x, e, x);
        return (S * SLACK) * x;
    }
    // walks this side to its end or until the rest is negligible
    __device__ __forceinline__ void run() {
        while (true) {
            const bool ok = eP == 0;
#pragma unroll
            for (int k = 0; k < TRIP; ++k) step(ok);
            rescale();
            if (at_end() || tail_negligible()) break;
        }
    }
};

__device__ double fisher_two_sided(long long a, long long b, long long c, long long d, const LfTable& t) {
    const long long n1 = a + b, n2 = c + d, n = a + c, m = b + d;
    if (n1 == 0 || n2 == 0 || n == 0 || m == 0) return 1.0;   // a zero margin (scipy: p = 1)
    const long long M = n1 + n2;
    const long long lo = n - n2 > 0 ? n - n2 : 0;
    const long long hi = n1 < n ? n1 : n;
    const double logp = logfact(t, n1) + logfact(t, n2) + logfact(t, n) + logfact(t, M - n) - logfact(t, M) -
                        logfact(t, a) - logfact(t, n1 - a) - logfact(t, n - a) - logfact(t, n2 - n + a);
    const double pexact = exp(logp);
    const double da = (double)a, dn1 = (double)n1, dn2 = (double)n2, dn = (double)n;
    Walk w;
    w.start(da, dn2 - dn + da, dn1 - da + 1.0, dn - da + 1.0, (double)(a - lo));                 // down from k = a
    if (a > lo) w.run();
    w.turn(dn1 - da, dn - da, da + 1.0, dn2 - dn + da + 1.0, (double)(hi - a));                    // up from k = a
    if (hi > a) w.run();
    const double total = 1.0 + w.sum();
    const double p = pexact * total;
    return p < 1.0 ? p : 1.0;
}

__global__ void __launch_bounds__(256) lf_table_kernel(double* __restrict__ lf, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lf[i] = lgamma((double)i + 1.0);
}

__global__ void __launch_bounds__(256) fisher_tables_kernel(const int64_t* __restrict__ abcd, int64_t m,
                                                            double* __restrict__ p, LfTable t) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    p[i] = fisher_two_sided(abcd[4 * i], abcd[4 * i + 1], abcd[4 * i + 2], abcd[4 * i + 3], t);
}

// pair index q -> (i, j), i < j, row-major (pairwise_fisher.py:142-147): q = i*s - i(i+1)/2 + (j-i-1)
__device__ __forceinline__ void pair_of(int64_t q, int s, int& i, int& j) {
    const double bb = 2.0 * s - 1.0;
    i = (int)((bb - sqrt(bb * bb - 8.0 * (double)q)) * 0.5);
    if (i < 0) i = 0;
    if (i > s - 2) i = s - 2;
    while (i > 0 && (int64_t)i * s - (int64_t)i * (i + 1) / 2 > q) --i;
    while ((int64_t)(i + 1) * s - (int64_t)(i + 1) * (i + 2) / 2 <= q) ++i;
    j = (int)(q - ((int64_t)i * s - (int64_t)i * (i + 1) / 2)) + i + 1;
}

// (i << 16 | j) of every pair index: the same for every junction, built once per call shape
__global__ void __launch_bounds__(256) pair_table_kernel(unsigned* __restrict__ tab, int64_t n_pairs, int s) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_pairs) return;
    int i, j;
    pair_of(q, s, i, j);
    tab[q] = ((unsigned)i << 16) | (unsigned)j;
}

__device__ __forceinline__ double logfact_d(const LfTable& t, double k) {
    return k < (double)t.n ? t.lf[(int)k] : lgamma_beyond_table(k + 1.0);
}
// log pmf(a) of a table whose total is beyond the log-factorial table
__device__ __noinline__ double log_pmf_beyond_table(double a, double b, double c, double d) {
    const double n1 = a + b, n2 = c + d, nn = a + c, mm = b + d;
    return lgamma(n1 + 1.0) + lgamma(n2 + 1.0) + lgamma(nn + 1.0) + lgamma(mm + 1.0) - lgamma(n1 + n2 + 1.0) - lgamma(a + 1.0) -
           lgamma(b + 1.0) - lgamma(c + 1.0) - lgamma(d + 1.0);
}

// p = pmf(a) * sum for the 256 pairs from q0 on, whose sums wait in the ring: the whole wave, four pairs per lane, the
// p-values leave as contiguous 512-byte pieces.  No call in here: a table whose total lies beyond the log-factorial table
// (counts above fisher.table_max, 2^20 by default) leaves MINUS its sum (sums are >= 1) and raises the junction's flag;
// fisher_beyond_table_kernel finishes those junctions.  (A call to lgamma -- even one, in a cold branch -- makes the
// register allocator keep the walk state in the 48 callee-saved VGPRs and spill the rest inside the loop; a pmf pass in
// a function of its own saved and restored 112 bytes per lane per call: 14 GB of scratch traffic per launch.)
extern __shared__ double smd[];
__device__ __forceinline__ bool pmf_block(int s, double* __restrict__ out, const unsigned* __restrict__ pair_tab, LfTable tab,
                                          int q0, int n_pairs, int lane) {
    const double* inc = smd;
    const double* exc = smd + s;
    const double* ring = smd + 2 * s;
    bool beyond = false;
#pragma nounroll
    for (int t = 0; t < 4; ++t) {
        const int q = q0 + t * 64 + lane;
        if (q < n_pairs) {
            const double total = ring[q & 511];
            const unsigned ij = pair_tab[q];
            const int i = (int)(ij >> 16), j = (int)(ij & 0xffffu);
            const double a = inc[i], b = inc[j], c = exc[i], d = exc[j];
            const double n1 = a + b, n2 = c + d, nn = a + c, mm = b + d, M = n1 + n2;
            double pv = 1.0;                                           // a zero margin (scipy: p = 1)
            if (n1 != 0.0 && n2 != 0.0 && nn != 0.0 && mm != 0.0) {
                if (M < (double)tab.n) {                               // (the largest of the nine arguments)
                    const double* lf = tab.lf;
                    const double logp = lf[(int)n1] + lf[(int)n2] + lf[(int)nn] + lf[(int)mm] - lf[(int)M] - lf[(int)a] -
                                        lf[(int)b] - lf[(int)c] - lf[(int)d];
                    pv = exp(logp) * total;
                    pv = pv < 1.0 ? pv : 1.0;
                } else {
                    pv = -total;
                    beyond = true;
                }
            }
            out[q] = pv;
        }
    }
    return beyond;
}

// the junctions flagged by the pair kernel: p = pmf(a) * sum with lgamma for the pairs that carry a negative sum
__global__ void __launch_bounds__(256) fisher_beyond_table_kernel(const int32_t* __restrict__ incl, const int64_t* __restrict__ excl,
                                                                  int64_t n, int s, double* __restrict__ p,
                                                                  const unsigned* __restrict__ pair_tab,
                                                                  const unsigned char* __restrict__ row_flag) {
    const int64_t row = blockIdx.x;
    if (row >= n || !row_flag[row]) return;
    const int64_t n_pairs = (int64_t)s * (s - 1) / 2;
    double* out = p + row * n_pairs;
    for (int64_t q = threadIdx.x; q < n_pairs; q += 256) {
        const double v = out[q];
        if (v < 0.0) {
            const unsigned ij = pair_tab[q];
            const int i = (int)(ij >> 16), j = (int)(ij & 0xffffu);
            const double a = (double)incl[row * s + i], b = (double)incl[row * s + j];
            const double c = (double)excl[row * s + i], d = (double)excl[row * s + j];
            const double pv = exp(log_pmf_beyond_table(a, b, c, d)) * -v;
            out[q] = pv < 1.0 ? pv : 1.0;
        }
    }
}

// One WAVE per junction.  The walk lengths of the pairs of one junction differ by an order of magnitude (they follow the
// margins), and a loop "for each pair: walk" keeps a wave at the pace of its slowest lane: 27 of 64 lanes were active on
// average.  Here every lane is a small state machine -- idle / walking down / walking up -- and the wave issues walk steps
// for all lanes (see Walk).  Pairs are handed out IN ORDER: when `refill` lanes are idle they take the next pairs q = next,
// next + 1, ... (a lane's rank among the idle lanes picks its pair; the (i, j) of the next 128 pairs wait in two registers
// across the wave and come by a lane shuffle), so the lanes finish a junction within one pair of each other and the pairs
// in flight are a window of ~100 consecutive indices.  Finished sums go to a ring of 512 slots in LDS; as soon as a block
// of 256 consecutive pairs is complete, the whole wave applies pmf(a) -- nine log-factorial look-ups and an exp, all lanes
// busy -- and writes the 256 p-values as four contiguous 512-byte pieces.  A p-value is written exactly once and nothing
// is read back (the version with a statically assigned run of 8 pairs per lane wrote the sums, read them again for the
// pmf pass and wrote the p-values: 12.9 GB of HBM traffic for 4 GB of results).
template <int UNROLL, bool COUNT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8)))
fisher_pairs_kernel(const int32_t* __restrict__ incl, const int64_t* __restrict__ excl, int64_t n, int s,
                    double* __restrict__ p, LfTable tab, const unsigned* __restrict__ pair_tab, int refill,
                    unsigned long long* __restrict__ row_counter, unsigned char* __restrict__ row_flag) {
    // counts staged as doubles (exact below 2^53): the set-up of a pair is four LDS reads and a dozen f64 operations
    double* inc = smd;
    double* exc = smd + s;
    double* ring = smd + 2 * s;                                        // [512] finished sums, slot = q mod 512
    const int lane = threadIdx.x;
    const int n_pairs = (int)((int64_t)s * (s - 1) / 2);               // (s <= 8192: fewer than 2^25 pairs)
    const int n_blk = (n_pairs + 255) >> 8;
    static_assert(UNROLL >= 1 && UNROLL <= 24, "unroll");
    // COUNT (fisher.count_steps, a measurement build): lane-steps issued and lane-steps that advanced a live walk
    // inside its support -- row_counter[1], [2]
    unsigned long long n_useful = 0, n_trips = 0;
    while (true) {
        // junctions are handed out one at a time: the grid is the set of resident waves, and a wave that
        // drew cheap junctions takes more of them
        unsigned long long row_u = 0;
        if (lane == 0) row_u = atomicAdd(row_counter, 1ull);
        const int64_t row = (int64_t)__shfl((long long)row_u, 0);
        if (row >= n) break;
        __syncthreads();
        for (int k = lane; k < s; k += 64) {
            inc[k] = (double)incl[row * s + k];
            exc[k] = (double)excl[row * s + k];
        }
        __syncthreads();
        double* out = p + row * (int64_t)n_pairs;

        int next = 0, fin_blk = 0;             // wave-uniform: next pair to hand out, next block for the pmf pass
        int done0 = 0, done1 = 0;              // wave-uniform: finished pairs of the even / odd block in the ring
        int tab_base = 0;                      // pair_tab[tab_base + lane] / [tab_base + 64 + lane] sit in tab_cur / tab_nxt
        unsigned tab_cur = lane < n_pairs ? pair_tab[lane] : 0u;
        unsigned tab_nxt = 64 + lane < n_pairs ? pair_tab[64 + lane] : 0u;
        int phase = 0, q_cur = 0;              // 0 idle, 1 walking down from a, 2 walking up from a
        Walk w;
        w.start(1.0, 1.0, 1.0, 1.0, 0.0);
        // the table [[a, b], [c, d]] of the current pair: down from k = a the ratio is a d / ((b + 1)(c + 1)) over
        // min(a, d) steps, up it is b c / ((a + 1)(d + 1)) over min(b, c) steps.  A zero margin leaves no step on
        // either side (the sum comes out as 1 and the pmf pass sets p = 1, as scipy does).
        double ta = 0.0, tb = 0.0, tc = 0.0, td = 0.0;
        while (true) {
            // ---- p = pmf(a) * sum for a complete block of 256 pairs, all lanes busy
            if (fin_blk < n_blk) {
                const int need = min(256, n_pairs - (fin_blk << 8));
                if (((fin_blk & 1) ? done1 : done0) == need) {
                    if (__ballot(pmf_block(s, out, pair_tab, tab, fin_blk << 8, n_pairs, lane)) != 0ull && lane == 0) row_flag[row] = 1;
                    if (fin_blk & 1) done1 = 0; else done0 = 0;
                    fin_blk += 1;
                }
            }
            // ---- hand out pairs (the ring holds the two blocks from fin_blk on)
            const unsigned long long idle_m = __ballot(phase == 0);
            const int lim = min(n_pairs, (fin_blk + 2) << 8);
            if (idle_m == ~0ull && next >= n_pairs && fin_blk >= n_blk) break;
            const int n_idle = __popcll(idle_m);
            if ((n_idle >= refill || idle_m == ~0ull) && next < lim) {
                const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle_m, 0u));
                const int q = next + r;
                const int dq = q - tab_base;                              // < 128 for a lane that takes a pair
                const unsigned e0 = (unsigned)__shfl((int)tab_cur, dq & 63), e1 = (unsigned)__shfl((int)tab_nxt, dq & 63);
                if (phase == 0 && q < lim) {
                    const unsigned ij = dq < 64 ? e0 : e1;
                    const int i = (int)(ij >> 16), j = (int)(ij & 0xffffu);
                    ta = inc[i]; tb = inc[j]; tc = exc[i]; td = exc[j];
                    q_cur = q;
                    w.start(ta, td, tb + 1.0, tc + 1.0, ta < td ? ta : td);   // down from k = a (possibly no step at all)
                    phase = 1;
                }
                next = min(next + n_idle, lim);
                if (next >= tab_base + 64) {
                    tab_base += 64;
                    tab_cur = tab_nxt;
                    tab_nxt = tab_base + 64 + lane < n_pairs ? pair_tab[tab_base + 64 + lane] : 0u;
                }
            }
            // ---- one trip: every lane steps, whatever its phase (see Walk); the exec mask is not touched
            if (COUNT) {
                const double left = (w.dN_end - w.dN) * 0.5;
                if (phase != 0 && left > 0.0) n_useful += (unsigned long long)(left < (double)UNROLL ? left : (double)UNROLL);
                n_trips += 1;
            }
            bool ok = w.eP == 0;
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                w.step(ok);
                // between the halves of a long trip the scaling is only due when some lane's products have left the first
                // half of the exponent range (wave-uniform test: 3 instructions instead of ~14)
                if (k % Walk::TRIP == Walk::TRIP - 1 && k + 1 < UNROLL && w.rescale_due()) { w.rescale(); ok = w.eP == 0; }
            }
            w.rescale();
            const bool fin = phase != 0 && (w.at_end() || w.tail_negligible());
            const double up_steps = tb < tc ? tb : tc;
            const bool dep = fin && !(phase == 1 && up_steps > 0.0);
            const unsigned long long dep1 = __ballot(dep && (q_cur & 256) != 0), dep_m = __ballot(dep);
            done1 += __popcll(dep1);
            done0 += __popcll(dep_m) - __popcll(dep1);
            if (fin) {
                if (!dep) {
                    w.turn(tb, tc, ta + 1.0, td + 1.0, up_steps);                                   // up from k = a
                    phase = 2;
                } else {
                    ring[q_cur & 511] = 1.0 + w.sum();
                    phase = 0;
                }
            }
        }
    }
    if (COUNT) {
        atomicAdd(row_counter + 1, n_useful);
        if (lane == 0) atomicAdd(row_counter + 2, n_trips * (unsigned long long)(UNROLL * 64));
    }
}

}  // namespace

// log-factorial table owned by the context (built on first use)
static int get_lf_table(sdice_ctx* ctx, LfTable* out) {
    long long want = ctx->param("fisher.table_max", 1 << 20);
    if (want < 2) want = 2;
    if (ctx->d_lf && ctx->lf_n == want) {
        out->lf = ctx->d_lf;
        out->n = want;
        return SDICE_OK;
    }
    if (ctx->d_lf) {
        SD_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_lf);
        ctx->d_lf = nullptr;
        ctx->lf_n = 0;
    }
    double* lf = nullptr;
    hipError_t e = hipMalloc((void**)&lf, (size_t)want * 8);
    if (e != hipSuccess) {
        sdice_set_error("fisher: hipMalloc of log-factorial table failed: %s", hipGetErrorString(e));
        return SDICE_ERR_NOMEM;
    }
    SD_LAUNCH(ctx, "lf_table_kernel", lf_table_kernel, dim3((unsigned)sd_ceil_div(want, 256)), dim3(256), 0, lf, want);
    ctx->d_lf = lf;
    ctx->lf_n = want;
    out->lf = lf;
    out->n = want;
    return SDICE_OK;
}

extern "C" int sdice_fisher_pairs_dev(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* d_incl,
                                      const int64_t* d_excl, double* d_p) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    if (n == 0 || s < 2) return SDICE_OK;
    SD_ARG(d_incl && d_excl && d_p, "NULL pointer");
    SD_ARG(s <= 8192, "more than 8192 samples per junction is not supported");
    SD_HIP(hipSetDevice(ctx->device));
    LfTable t;
    SD_TRY(get_lf_table(ctx, &t));
    int refill = (int)ctx->param("fisher.refill", 12);
    if (refill < 1) refill = 1;
    if (refill > 64) refill = 64;
    // steps per trip: 16 (153 VGPRs, three waves per SIMD, nothing spilled) 16.0 ms per 25 000 x 19 900; 8 (114 VGPRs, four
    // waves) 16.6 ms; 16 under a 128-VGPR cap spills 100 bytes inside the loop: 18.5 ms; under 96 (five waves): 34-43 ms
    const int unroll = (int)ctx->param("fisher.unroll", 16);
    const int64_t n_pairs = (int64_t)s * (s - 1) / 2;
    SD_TRY(ctx->arena.reserve((size_t)n_pairs * 4 + (size_t)n + 8192, ctx->stream));
    unsigned* pair_tab = (unsigned*)ctx->arena.alloc((size_t)n_pairs * 4);
    unsigned long long* row_counter = (unsigned long long*)ctx->arena.alloc(24);     // + the two step counters
    unsigned char* row_flag = (unsigned char*)ctx->arena.alloc((size_t)n);
    if (!pair_tab || !row_counter || !row_flag) return SDICE_ERR_NOMEM;
    SD_HIP(hipMemsetAsync(row_counter, 0, 24, ctx->stream));
    SD_HIP(hipMemsetAsync(row_flag, 0, (size_t)n, ctx->stream));
    SD_LAUNCH(ctx, "pair_table_kernel", pair_table_kernel, dim3((unsigned)sd_ceil_div(n_pairs, (int64_t)256)), dim3(256), 0,
              pair_tab, n_pairs, (int)s);
    const size_t lds = (size_t)s * 16 + 512 * 8;
    const bool count = ctx->param("fisher.count_steps", 0) != 0;
    auto kern = count ? (unroll <= 4 ? fisher_pairs_kernel<4, true> : unroll <= 8 ? fisher_pairs_kernel<8, true> :
                         unroll <= 12 ? fisher_pairs_kernel<12, true> : unroll <= 16 ? fisher_pairs_kernel<16, true> :
                         unroll <= 20 ? fisher_pairs_kernel<20, true> : fisher_pairs_kernel<24, true>)
                      : (unroll <= 4 ? fisher_pairs_kernel<4, false> : unroll <= 8 ? fisher_pairs_kernel<8, false> :
                         unroll <= 12 ? fisher_pairs_kernel<12, false> : unroll <= 16 ? fisher_pairs_kernel<16, false> :
                         unroll <= 20 ? fisher_pairs_kernel<20, false> : fisher_pairs_kernel<24, false>);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    SD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64, lds));
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = std::min<int64_t>(n, (int64_t)ctx->n_cu * per_cu);     // the resident waves; junctions by counter
    SD_LAUNCH(ctx, "fisher_pairs_kernel", kern, dim3((unsigned)blocks), dim3(64), lds, d_incl, d_excl, n, (int)s, d_p, t,
              pair_tab, refill, row_counter, row_flag);
    // junctions with a table beyond the log-factorial table (their workgroups return at once otherwise)
    SD_LAUNCH(ctx, "fisher_beyond_table_kernel", fisher_beyond_table_kernel, dim3((unsigned)n), dim3(256), 0, d_incl, d_excl, n,
              (int)s, d_p, pair_tab, row_flag);
    if (count) {
        SD_HIP(hipMemcpyAsync(ctx->fisher_steps, row_counter + 1, 16, hipMemcpyDeviceToHost, ctx->stream));
        SD_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SDICE_OK;
}

extern "C" int sdice_fisher_step_stats(sdice_ctx* ctx, uint64_t* useful, uint64_t* issued) {
    SD_ARG(ctx && useful && issued, "NULL pointer");
    *useful = ctx->fisher_steps[0];
    *issued = ctx->fisher_steps[1];
    return SDICE_OK;
}

extern "C" int sdice_fisher_pairs(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* incl, const int64_t* excl,
                                  double* p) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && s >= 0, "negative size");
    if (n == 0 || s < 2) return SDICE_OK;
    SD_ARG(incl && excl && p, "NULL pointer");
    for (int64_t i = 0; i < n * s; ++i) SD_ARG(incl[i] >= 0 && excl[i] >= 0, "counts must be non-negative");
    const int64_t n_pairs = (int64_t)s * (s - 1) / 2;
    int32_t* di = nullptr;
    int64_t* de = nullptr;
    double* dp = nullptr;
    int rc = sdice_dmalloc(ctx, n * s * 4, (void**)&di);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * s * 8, (void**)&de);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, n * n_pairs * 8, (void**)&dp);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, di, incl, n * s * 4);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, de, excl, n * s * 8);
    if (rc == SDICE_OK) rc = sdice_fisher_pairs_dev(ctx, n, s, di, de, dp);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, p, dp, n * n_pairs * 8);
    sdice_dfree(ctx, di); sdice_dfree(ctx, de); sdice_dfree(ctx, dp);
    return rc;
}

extern "C" int sdice_fisher_tables(sdice_ctx* ctx, int64_t m, const int64_t* abcd, double* p) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(m >= 0, "negative size");
    if (m == 0) return SDICE_OK;
    SD_ARG(abcd && p, "NULL pointer");
    for (int64_t i = 0; i < 4 * m; ++i) SD_ARG(abcd[i] >= 0, "table entries must be non-negative");
    SD_HIP(hipSetDevice(ctx->device));
    LfTable t;
    SD_TRY(get_lf_table(ctx, &t));
    int64_t* dt = nullptr;
    double* dp = nullptr;
    int rc = sdice_dmalloc(ctx, m * 32, (void**)&dt);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, m * 8, (void**)&dp);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, dt, abcd, m * 32);
    if (rc == SDICE_OK) {
        hipLaunchKernelGGL(fisher_tables_kernel, dim3((unsigned)sd_ceil_div(m, 256)), dim3(256), 0, ctx->stream, dt, m, dp,
                           t);
        if (hipGetLastError() != hipSuccess) { sdice_set_error("fisher_tables_kernel launch failed"); rc = SDICE_ERR_HIP; }
    }
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, p, dp, m * 8);
    sdice_dfree(ctx, dt); sdice_dfree(ctx, dp);
    return rc;
}
