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
    __device__ __forceinline__ void rescale() {
        const int e = 1 - __builtin_amdgcn_frexp_exp(Q);        // Q >= 1: back into [1, 2)
        Q = __builtin_ldexp(Q, e); S = __builtin_ldexp(S, e); P = __builtin_ldexp(P, e);
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

// One WAVE per junction.  The walk lengths of the pairs of one junction differ by an order of
// magnitude (they follow the margins), and a loop "for each pair: walk" keeps a wave at the pace of its
// slowest lane: 27 of 64 lanes were active on average.  Here every lane runs its own stream of pairs
// q = lane, lane + 64, ... as a small state machine -- idle / walking down / walking up -- and the wave
// executes walk steps for whoever is walking.  Lanes that finish a pair wait until `refill` of them are
// idle (the set-up of a pair is ~100 instructions that the whole wave issues), then fetch their next
// pair together.  Over the ~300 pairs of a lane the lengths average out, so the lanes finish a
// junction within a few per cent of each other.  The state machine only produces the sum of the ratios;
// pmf(a) -- nine log-factorial look-ups and an exp -- is applied afterwards in a pass with all lanes busy.
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

template <int UNROLL, bool COUNT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 8)))
fisher_pairs_kernel(const int32_t* __restrict__ incl, const int64_t* __restrict__ excl, int64_t n, int s,
                    double* __restrict__ p, LfTable tab, const unsigned* __restrict__ pair_tab, int refill,
                    unsigned long long* __restrict__ row_counter) {
    // counts staged as doubles (exact below 2^53): the set-up of a pair is then a table look-up, four LDS
    // reads and a dozen f64 operations, no integer -> double conversions
    extern __shared__ double smd[];
    double* inc = smd;
    double* exc = smd + s;
    const int lane = threadIdx.x;
    double* run = smd + 2 * s + lane * 10;         // this lane's run of finished sums (80-byte pitch: 16 B aligned, 4-way bank spread)
    const int64_t n_pairs = (int64_t)s * (s - 1) / 2;
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
        double* out = p + row * n_pairs;

        // ---- sums of the ratios: out[q] = 1 + sum over the accepted k != a of pmf(k)/pmf(a)
        // a lane owns runs of 8 consecutive pairs (q = 512 b + 8 lane + j): finished sums wait in 64 B of LDS and
        // leave as one contiguous 64-byte piece -- sums written one by one as the lanes drift apart left the L2
        // as partial lines (PMC: 21.6 GB written for 4 GB of p-values)
        const bool out16 = ((uintptr_t)out & 15) == 0;                 // (row base; run starts are multiples of 64 B from it)
        auto deposit = [&](int q, double v) {
            run[q & 7] = v;
            if ((q & 7) == 7) {
                double* o = out + (q - 7);
                if (out16) {
                    const double2* r2 = reinterpret_cast<const double2*>(run);
                    double2* o2 = reinterpret_cast<double2*>(o);
                    const double2 a0 = r2[0], a1 = r2[1], a2 = r2[2], a3 = r2[3];
                    o2[0] = a0; o2[1] = a1; o2[2] = a2; o2[3] = a3;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = run[j];
                }
            } else if (q + 1 >= (int)n_pairs) {                        // the row's last, incomplete run
                double* o = out + (q & ~7);
                for (int j = 0; j <= (q & 7); ++j) o[j] = run[j];
            }
        };
        int q_next = lane * 8, q_cur = 0;                              // (s <= 8192: fewer than 2^25 pairs)
        const int n_pairs_i = (int)n_pairs;
        unsigned ij_next = q_next < n_pairs_i ? pair_tab[q_next] : 0u; // always one entry ahead: its latency hides behind a walk
        int phase = 0;                         // 0 idle, 1 walking down from a, 2 walking up from a
        Walk w;
        w.start(1.0, 1.0, 1.0, 1.0, 0.0);
        // the table [[a, b], [c, d]] of the current pair: down from k = a the ratio is a d / ((b + 1)(c + 1)) over
        // min(a, d) steps, up it is b c / ((a + 1)(d + 1)) over min(b, c) steps.  A zero margin leaves no step on
        // either side (the sum comes out as 1 and the pmf pass sets p = 1, as scipy does).
        double ta = 0.0, tb = 0.0, tc = 0.0, td = 0.0;
        while (true) {
            const bool can_fetch = phase == 0 && q_next < n_pairs_i;
            const unsigned long long idle_m = __ballot(phase == 0), fetch_m = __ballot(can_fetch);
            if (idle_m == ~0ull && fetch_m == 0ull) break;
            if ((__popcll(fetch_m) >= refill || idle_m == ~0ull) && can_fetch) {
                const int i = (int)(ij_next >> 16), j = (int)(ij_next & 0xffffu);
                ta = inc[i]; tb = inc[j]; tc = exc[i]; td = exc[j];
                q_cur = q_next;
                q_next += (q_next & 7) == 7 ? 512 - 7 : 1;
                if (q_next < n_pairs_i) ij_next = pair_tab[q_next];
                w.start(ta, td, tb + 1.0, tc + 1.0, ta < td ? ta : td);     // down from k = a (possibly no step at all)
                phase = 1;
            }
            // one trip: every lane steps, whatever its phase (see Walk); the exec mask is not touched
            if (COUNT) {
                const double left = (w.dN_end - w.dN) * 0.5;
                if (phase != 0 && left > 0.0) n_useful += (unsigned long long)(left < (double)UNROLL ? left : (double)UNROLL);
                n_trips += 1;
            }
            bool ok = w.eP == 0;
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                w.step(ok);
                if (k % Walk::TRIP == Walk::TRIP - 1 && k + 1 < UNROLL) { w.rescale(); ok = w.eP == 0; }
            }
            w.rescale();
            if (phase != 0 && (w.at_end() || w.tail_negligible())) {
                const double up_steps = tb < tc ? tb : tc;
                if (phase == 1 && up_steps > 0.0) {
                    w.turn(tb, tc, ta + 1.0, td + 1.0, up_steps);                                   // up from k = a
                    phase = 2;
                } else {
                    deposit(q_cur, 1.0 + w.sum());
                    phase = 0;
                }
            }
        }
        __syncthreads();

        // ---- p = pmf(a) * sum, all lanes busy
        for (int64_t q = lane; q < n_pairs; q += 64) {
            const double total = out[q];
            const unsigned ij = pair_tab[q];
            const int i = (int)(ij >> 16), j = (int)(ij & 0xffffu);
            const double a = inc[i], b = inc[j], c = exc[i], d = exc[j];
            const double n1 = a + b, n2 = c + d, nn = a + c, mm = b + d, M = n1 + n2;
            double pv = 1.0;                                           // a zero margin (scipy: p = 1)
            if (n1 != 0.0 && n2 != 0.0 && nn != 0.0 && mm != 0.0) {
                const double logp = logfact_d(tab, n1) + logfact_d(tab, n2) + logfact_d(tab, nn) + logfact_d(tab, mm) -
                                    logfact_d(tab, M) - logfact_d(tab, a) - logfact_d(tab, b) - logfact_d(tab, c) -
                                    logfact_d(tab, d);
                pv = exp(logp) * total;
                pv = pv < 1.0 ? pv : 1.0;
            }
            out[q] = pv;
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
    int refill = (int)ctx->param("fisher.refill", 16);
    if (refill < 1) refill = 1;
    if (refill > 64) refill = 64;
    const int unroll = (int)ctx->param("fisher.unroll", 8);
    const int64_t n_pairs = (int64_t)s * (s - 1) / 2;
    SD_TRY(ctx->arena.reserve((size_t)n_pairs * 4 + 8192, ctx->stream));
    unsigned* pair_tab = (unsigned*)ctx->arena.alloc((size_t)n_pairs * 4);
    unsigned long long* row_counter = (unsigned long long*)ctx->arena.alloc(24);     // + the two step counters
    if (!pair_tab || !row_counter) return SDICE_ERR_NOMEM;
    SD_HIP(hipMemsetAsync(row_counter, 0, 24, ctx->stream));
    SD_LAUNCH(ctx, "pair_table_kernel", pair_table_kernel, dim3((unsigned)sd_ceil_div(n_pairs, (int64_t)256)), dim3(256), 0,
              pair_tab, n_pairs, (int)s);
    const size_t lds = (size_t)s * 16 + 64 * 10 * 8;
    const bool count = ctx->param("fisher.count_steps", 0) != 0;
    auto kern = count ? (unroll <= 4 ? fisher_pairs_kernel<4, true> : unroll <= 6 ? fisher_pairs_kernel<6, true> : unroll <= 8 ? fisher_pairs_kernel<8, true> :
                         unroll <= 12 ? fisher_pairs_kernel<12, true> : fisher_pairs_kernel<16, true>)
                      : (unroll <= 4 ? fisher_pairs_kernel<4, false> : unroll <= 6 ? fisher_pairs_kernel<6, false> : unroll <= 8 ? fisher_pairs_kernel<8, false> :
                         unroll <= 12 ? fisher_pairs_kernel<12, false> : fisher_pairs_kernel<16, false>);
    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    SD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64, lds));
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = std::min<int64_t>(n, (int64_t)ctx->n_cu * per_cu);     // the resident waves; junctions by counter
    SD_LAUNCH(ctx, "fisher_pairs_kernel", kern, dim3((unsigned)blocks), dim3(64), lds, d_incl, d_excl, n, (int)s, d_p, t,
              pair_tab, refill, row_counter);
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
