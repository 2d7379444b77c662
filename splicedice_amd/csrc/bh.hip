// K7: Benjamini-Hochberg FDR.
//
// Replaces statsmodels.stats.multitest.multipletests(p, method="fdr_bh")[1]
// (compareSampleSets.py:235; pairwise_fisher.py:185,190): sort ascending,
// p_(i) / (i/m), running minimum from the largest rank down, clip at 1, unsort.
// (statsmodels is not installable in the build image: "parity unpinned" for this call,
// cross-checked against scipy.stats.false_discovery_control.)
//
// Device: radix sort of the IEEE-754 bit patterns (non-negative doubles order like unsigned
// integers) with the original index as payload, an elementwise kernel that writes
// p_(i)/(i/m) in REVERSED order, an inclusive min-scan (on bit patterns, again order
// preserving), and a scatter back to the original positions.
// Everything is written for `segs` equally long segments so that the per-pair-column mode of
// `pairwise` (pairwise_fisher.py:187-191: BH down each of the S(S-1)/2 columns) is one batched
// pass over the transposed table instead of one sort per column.
#include "common.h"
#include <algorithm>

int sd_inclusive_min_scan_u64(sdice_ctx* ctx, int64_t n, const uint64_t* d_in, uint64_t* d_out);
// bh_cols.hip
size_t sd_bh_cols_scratch(int64_t m, int64_t segs);
bool sd_bh_cols_supported(int64_t m, int64_t segs);
int sd_bh_cols_samplesort(sdice_ctx* ctx, int64_t m, int64_t segs, double* d_rm, int64_t pitch);
bool sd_bh_vector_supported(int64_t n);
size_t sd_bh_vector_scratch(sdice_ctx* ctx, int64_t n);
int sd_bh_vector_samplesort(sdice_ctx* ctx, int64_t n, const double* d_p, const uint8_t* d_tested, bool masked, double* d_q);

namespace {

__global__ void __launch_bounds__(256) bh_keys_kernel(const double* __restrict__ p, int64_t m, int64_t total,
                                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    double v = p[i];
    if (v == 0.0) v = 0.0;   // -0.0 -> +0.0
    keys[i] = (uint64_t)__double_as_longlong(v);
    idx[i] = (uint32_t)(i % m);          // position inside its segment
}

// Masked variant (one segment): entries with tested == 0 (or, without a mask, with p < 0) are absent --
// they sort behind every p-value (all-ones key) and are counted out of m.  Used for the gathered,
// padded per-junction table of a sharded compare (BH ranks the TESTED junctions only,
// compareSampleSets.py:223-235) without compacting it on the host.
// 1024 threads x 8 entries per workgroup and ONE atomic per workgroup for the count of present entries (an atomic
// per wave on the one counter serialised at ~12 ns each: 190 us of the 420 us that a masked vector of 1 M took)
__global__ void __launch_bounds__(1024) bh_keys_masked_kernel(const double* __restrict__ p,
                                                              const uint8_t* __restrict__ tested, int64_t n,
                                                              uint64_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                              unsigned long long* __restrict__ m_eff) {
    __shared__ unsigned wcount[16];
    unsigned here = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int64_t i = ((int64_t)blockIdx.x * 8 + q) * 1024 + threadIdx.x;
        if (i < n) {
            double v = p[i];
            const bool present = tested ? tested[i] != 0 : !(v < 0.0);
            if (v == 0.0) v = 0.0;
            keys[i] = present ? (uint64_t)__double_as_longlong(v) : ~0ull;
            idx[i] = (uint32_t)i;
            here += present ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) here += (unsigned)__shfl_xor((int)here, o);
    if ((threadIdx.x & 63) == 0) wcount[threadIdx.x >> 6] = here;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < 16; ++w) tot += wcount[w];
        if (tot) atomicAdd(m_eff, (unsigned long long)tot);
    }
}

__global__ void __launch_bounds__(256) bh_raw_masked_kernel(const uint64_t* __restrict__ sorted, int64_t n,
                                                            const unsigned long long* __restrict__ m_eff,
                                                            uint64_t* __restrict__ raw_rev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t m = (int64_t)*m_eff;
    uint64_t out = 0x7ff0000000000000ull;                 // +inf: absent entries never lower the running minimum
    if (i < m) {
        const double ps = __longlong_as_double((long long)sorted[i]);
        out = (uint64_t)__double_as_longlong(ps / ((double)(i + 1) / (double)m));
    }
    raw_rev[n - 1 - i] = out;
}

__global__ void __launch_bounds__(256) bh_scatter_masked_kernel(const uint64_t* __restrict__ cummin_rev,
                                                                const uint32_t* __restrict__ idx, int64_t n,
                                                                const unsigned long long* __restrict__ m_eff,
                                                                double* __restrict__ q) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = 0.0;                                        // absent entries get 0 (as the host code leaves them)
    if (i < (int64_t)*m_eff) {
        v = __longlong_as_double((long long)cummin_rev[n - 1 - i]);
        if (v > 1.0) v = 1.0;
    }
    q[idx[i]] = v;
}

// raw_rev[seg][m-1-i] = p_(i) / ((i+1)/m)
__global__ void __launch_bounds__(256) bh_raw_kernel(const uint64_t* __restrict__ sorted, int64_t m, int64_t total,
                                                     uint64_t* __restrict__ raw_rev) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int64_t seg = g / m, i = g - seg * m;
    const double ps = __longlong_as_double((long long)sorted[g]);
    const double ecdf = (double)(i + 1) / (double)m;
    const double raw = ps / ecdf;
    raw_rev[seg * m + (m - 1 - i)] = (uint64_t)__double_as_longlong(raw);
}

// inclusive running minimum inside every segment (one workgroup per segment, 2048 values per round)
__global__ void __launch_bounds__(256) seg_minscan_kernel(const uint64_t* __restrict__ in, int64_t m,
                                                          uint64_t* __restrict__ out) {
    __shared__ uint64_t wmin[4];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t* src = in + (int64_t)blockIdx.x * m;
    uint64_t* dst = out + (int64_t)blockIdx.x * m;
    if (tid == 0) carry_s = ~0ull;
    __syncthreads();
    for (int64_t base = 0; base < m; base += 256 * 8) {
        uint64_t v[8];
        const int64_t b = base + (int64_t)tid * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (b + q < m) ? src[b + q] : ~0ull;
#pragma unroll
        for (int q = 1; q < 8; ++q) v[q] = v[q] < v[q - 1] ? v[q] : v[q - 1];
        uint64_t x = v[7];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned lo = (unsigned)(x & 0xffffffffu), hi = (unsigned)(x >> 32);
            lo = __shfl_up(lo, o); hi = __shfl_up(hi, o);
            const uint64_t y = ((uint64_t)hi << 32) | lo;
            if (lane >= o) x = y < x ? y : x;
        }
        if (lane == 63) wmin[w] = x;
        __syncthreads();
        uint64_t pre = carry_s;
        for (int k = 0; k < w; ++k) pre = wmin[k] < pre ? wmin[k] : pre;
        unsigned lo = (unsigned)(x & 0xffffffffu), hi = (unsigned)(x >> 32);
        lo = __shfl_up(lo, 1); hi = __shfl_up(hi, 1);
        uint64_t prev = ((uint64_t)hi << 32) | lo;       // inclusive minimum of the lanes before this one
        if (lane == 0) prev = ~0ull;
        const uint64_t tpre = prev < pre ? prev : pre;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (b + q < m) dst[b + q] = v[q] < tpre ? v[q] : tpre;
        __syncthreads();
        if (tid == 255) {
            uint64_t tot = pre;
            tot = x < tot ? x : tot;
            carry_s = tot;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) bh_scatter_kernel(const uint64_t* __restrict__ cummin_rev,
                                                         const uint32_t* __restrict__ idx, int64_t m, int64_t total,
                                                         double* __restrict__ q) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int64_t seg = g / m, i = g - seg * m;
    double v = __longlong_as_double((long long)cummin_rev[seg * m + (m - 1 - i)]);
    if (v > 1.0) v = 1.0;
    q[seg * m + idx[g]] = v;
}

__global__ void __launch_bounds__(256) transpose_f64_kernel(const double* __restrict__ in, int64_t rows, int64_t cols,
                                                            double* __restrict__ out) {
    // out[c, r] = in[r, c]; 32x32 tiles through LDS
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) tile[k][tx] = in[r * cols + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][k];
    }
}

// out[c, r] = in[r, c] for an in of `rows` x `cols` with row pitch in_pitch, out row pitch out_pitch (elements)
__global__ void __launch_bounds__(256) transpose_f64_pitched_kernel(const double* __restrict__ in, int64_t rows, int64_t cols,
                                                                    int64_t in_pitch, double* __restrict__ out,
                                                                    int64_t out_pitch) {
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) tile[k][tx] = in[r * in_pitch + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) out[c * out_pitch + r] = tile[tx][k];
    }
}

int transpose(sdice_ctx* ctx, const double* in, int64_t rows, int64_t cols, double* out) {
    // the grid's y extent is limited to 65535 blocks: tall matrices go in row chunks
    const int64_t chunk = (int64_t)65535 * 32;
    for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
        const int64_t rc = std::min(chunk, rows - r0);
        const int64_t gx = sd_ceil_div(cols, 32), gy = sd_ceil_div(rc, 32);
        if (gx > 0x7fffffff) { sdice_set_error("transpose: too many columns"); return SDICE_ERR_ARG; }
        SD_LAUNCH(ctx, "transpose_f64_kernel", transpose_f64_pitched_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0,
                  in + r0 * cols, rc, cols, cols, out + r0, rows);
    }
    return SDICE_OK;
}

}  // namespace

// BH inside each of `segs` contiguous segments of m p-values: d_p[seg*m + i] -> d_q[seg*m + i]
static int bh_segments(sdice_ctx* ctx, int64_t m, int64_t segs, const double* d_p, double* d_q) {
    Arena& A = ctx->arena;
    const int64_t total = m * segs;
    const size_t M = (size_t)total;
    uint64_t* kA = (uint64_t*)A.alloc(M * 8);
    uint64_t* kB = (uint64_t*)A.alloc(M * 8);
    uint64_t* kC = (uint64_t*)A.alloc(M * 8);
    uint32_t* vA = (uint32_t*)A.alloc(M * 4);
    uint32_t* vB = (uint32_t*)A.alloc(M * 4);
    uint32_t* vC = (uint32_t*)A.alloc(M * 4);
    if (!kA || !kB || !kC || !vA || !vB || !vC) return SDICE_ERR_NOMEM;
    const unsigned g = (unsigned)sd_ceil_div(total, 256);
    SD_LAUNCH(ctx, "bh_keys_kernel", bh_keys_kernel, dim3(g), dim3(256), 0, d_p, m, total, kA, vA);
    // p-values live in [0, 1] (or NaN): the sign bit never varies, every other digit may
    SD_TRY(sd_radix_sort_pairs_segmented(ctx, m, segs, kA, vA, kB, vB, kC, vC, 0x7fffffffffffffffull));
    SD_LAUNCH(ctx, "bh_raw_kernel", bh_raw_kernel, dim3(g), dim3(256), 0, kB, m, total, kA);
    if (segs == 1) {
        SD_TRY(sd_inclusive_min_scan_u64(ctx, m, kA, kC));      // one long vector: the grid-wide 3-launch scan
    } else {
        SD_LAUNCH(ctx, "seg_minscan_kernel", seg_minscan_kernel, dim3((unsigned)segs), dim3(256), 0, kA, m, kC);
    }
    SD_LAUNCH(ctx, "bh_scatter_kernel", bh_scatter_kernel, dim3(g), dim3(256), 0, kC, vB, m, total, d_q);
    return SDICE_OK;
}

extern "C" int sdice_bh_dev(sdice_ctx* ctx, int64_t m, const double* d_p, double* d_q) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(m >= 0 && m < ((int64_t)1 << 32), "m out of range");
    if (m == 0) return SDICE_OK;
    SD_ARG(d_p && d_q, "NULL pointer");
    SD_HIP(hipSetDevice(ctx->device));
    // bh.vector_path: 0 = by size, 1 = radix path, 2 = sample-sort path (bh_cols.hip, four launches)
    const int64_t vpath = ctx->param("bh.vector_path", 0);
    if (vpath != 1 && sd_bh_vector_supported(m)) {
        SD_TRY(ctx->arena.reserve(sd_bh_vector_scratch(ctx, m), ctx->stream));
        return sd_bh_vector_samplesort(ctx, m, d_p, nullptr, false, d_q);
    }
    SD_ARG(vpath != 2, "bh.vector_path = 2 needs 16384 <= m <= 2 Mi values");
    SD_TRY(ctx->arena.reserve((size_t)m * 37 + (size_t)(m / 3072 + 2) * 1024 + (1 << 16), ctx->stream));
    return bh_segments(ctx, m, 1, d_p, d_q);
}

extern "C" int sdice_bh_masked_dev(sdice_ctx* ctx, int64_t n, const double* d_p, const uint8_t* d_tested, double* d_q) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && n < ((int64_t)1 << 32), "n out of range");
    if (n == 0) return SDICE_OK;
    SD_ARG(d_p && d_q, "NULL pointer");
    SD_HIP(hipSetDevice(ctx->device));
    const int64_t vpath = ctx->param("bh.vector_path", 0);
    if (vpath != 1 && sd_bh_vector_supported(n)) {
        SD_TRY(ctx->arena.reserve(sd_bh_vector_scratch(ctx, n), ctx->stream));
        return sd_bh_vector_samplesort(ctx, n, d_p, d_tested, true, d_q);
    }
    SD_ARG(vpath != 2, "bh.vector_path = 2 needs 16384 <= n <= 2 Mi values");
    SD_TRY(ctx->arena.reserve((size_t)n * 37 + (size_t)(n / 3072 + 2) * 1024 + (1 << 16), ctx->stream));
    Arena& A = ctx->arena;
    const size_t M = (size_t)n;
    uint64_t* kA = (uint64_t*)A.alloc(M * 8);
    uint64_t* kB = (uint64_t*)A.alloc(M * 8);
    uint64_t* kC = (uint64_t*)A.alloc(M * 8);
    uint32_t* vA = (uint32_t*)A.alloc(M * 4);
    uint32_t* vB = (uint32_t*)A.alloc(M * 4);
    uint32_t* vC = (uint32_t*)A.alloc(M * 4);
    unsigned long long* m_eff = (unsigned long long*)A.alloc(8);
    if (!kA || !kB || !kC || !vA || !vB || !vC || !m_eff) return SDICE_ERR_NOMEM;
    SD_HIP(hipMemsetAsync(m_eff, 0, 8, ctx->stream));
    const unsigned g = (unsigned)sd_ceil_div(n, 256);
    SD_LAUNCH(ctx, "bh_keys_masked_kernel", bh_keys_masked_kernel, dim3((unsigned)sd_ceil_div(n, (int64_t)8192)), dim3(1024), 0, d_p,
              d_tested, n, kA, vA, m_eff);
    SD_TRY(sd_radix_sort_pairs(ctx, n, kA, vA, kB, vB, kC, vC, ~0ull));
    SD_LAUNCH(ctx, "bh_raw_masked_kernel", bh_raw_masked_kernel, dim3(g), dim3(256), 0, kB, n, m_eff, kA);
    SD_TRY(sd_inclusive_min_scan_u64(ctx, n, kA, kC));
    SD_LAUNCH(ctx, "bh_scatter_masked_kernel", bh_scatter_masked_kernel, dim3(g), dim3(256), 0, kC, vB, n, m_eff, d_q);
    return SDICE_OK;
}

extern "C" int sdice_bh(sdice_ctx* ctx, int64_t m, const double* p, double* q) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(m >= 0, "negative size");
    if (m == 0) return SDICE_OK;
    SD_ARG(p && q, "NULL pointer");
    double *dp = nullptr, *dq = nullptr;
    int rc = sdice_dmalloc(ctx, m * 8, (void**)&dp);
    if (rc == SDICE_OK) rc = sdice_dmalloc(ctx, m * 8, (void**)&dq);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, dp, p, m * 8);
    if (rc == SDICE_OK) rc = sdice_bh_dev(ctx, m, dp, dq);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, q, dq, m * 8);
    sdice_dfree(ctx, dp); sdice_dfree(ctx, dq);
    return rc;
}

// BH down each column of a row-major [n, cols] device table, in place.
// Columns of up to 2^18 values: sample-sort path (bh_cols.hip), which reads and writes the row-major table itself;
// column groups when the scratch (28 B per value) does not fit what is free.
static int bh_columns_samplesort(sdice_ctx* ctx, int64_t n, int64_t cols, int64_t pitch, double* d_p_inout) {
    size_t free_b = 0, total_b = 0;
    SD_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t arena_b = 0;
    for (auto& c : ctx->arena.chunks) arena_b += c.cap;
    const int64_t budget = (int64_t)((free_b + arena_b) / 10 * 9);
    int64_t group = std::max<int64_t>(1, (budget - (1 << 20)) / (n * 28 + 49 * 1025));
    if (group > cols) group = cols;
    if (group < cols && group > 16) group &= ~(int64_t)15;      // whole 128-byte lines of the row-major table per group
    for (int64_t c0 = 0; c0 < cols; c0 += group) {
        const int64_t gc = std::min(group, cols - c0);
        int rc = ctx->arena.reserve(sd_bh_cols_scratch(n, group), ctx->stream);
        if (rc == SDICE_ERR_NOMEM && group > 1) {            // less memory than hipMemGetInfo promised
            group = (group + 1) / 2;
            c0 -= group;
            continue;
        }
        if (rc != SDICE_OK) return rc;
        SD_TRY(sd_bh_cols_samplesort(ctx, n, gc, d_p_inout + c0, pitch));
    }
    return SDICE_OK;
}

extern "C" int sdice_bh_columns_dev(sdice_ctx* ctx, int64_t n, int64_t cols, double* d_p_inout) {
    return sdice_bh_columns_pitched_dev(ctx, n, cols, cols, d_p_inout);
}

extern "C" int sdice_bh_columns_pitched_dev(sdice_ctx* ctx, int64_t n, int64_t cols, int64_t pitch, double* d_p_inout) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && cols >= 0, "negative size");
    if (n == 0 || cols == 0) return SDICE_OK;
    SD_ARG(d_p_inout, "NULL pointer");
    SD_ARG(pitch >= cols, "pitch must be at least the number of columns");
    SD_HIP(hipSetDevice(ctx->device));
    // bh.columns_path: 0 = by size, 1 = generic radix path, 2 = sample-sort path
    const int64_t path = ctx->param("bh.columns_path", 0);
    if (path != 1 && sd_bh_cols_supported(n, cols)) return bh_columns_samplesort(ctx, n, cols, pitch, d_p_inout);
    SD_ARG(path != 2, "bh.columns_path = 2 needs columns of at most 2^18 values");
    // columns per group from what is free right now: per value 2 x 8 B (transposed in / out) + 8 B (dense slab
    // when the columns go in groups) + 37 B of sort scratch + histograms; a failed reservation halves the group
    size_t free_b = 0, total_b = 0;
    SD_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t arena_b = 0;
    for (auto& c : ctx->arena.chunks) arena_b += c.cap;         // the arena's own chunk is reused
    const int64_t budget = (int64_t)((free_b + arena_b) / 10 * 9);
    int64_t group = budget / (n * 64) > 0 ? budget / (n * 64) : 1;
    if (group > cols) group = cols;
    if (group > 65535) group = 65535;
    int rc = SDICE_OK;
    for (int64_t c0 = 0; c0 < cols && rc == SDICE_OK; c0 += group) {
        const int64_t gc = std::min(group, cols - c0);
        // everything comes from the context arena (kept between calls: no multi-GB hipMalloc per call):
        // transposed in/out (+ a dense slab when the columns go in groups), 3 x u64 + 3 x u32 per value
        // for the sort, per-segment histograms (256 x tiles x 4 B) and bin totals
        const size_t vals = (size_t)n * (size_t)group;
        const bool slab = group < cols || pitch != cols;     // a column group of the table, or a pitched table: gathered dense first
        rc = ctx->arena.reserve(vals * (16 + (slab ? 8 : 0) + 37) +
                                    (size_t)group * ((size_t)(n / 3072 + 2) * 1024 + 1024) + (1 << 16), ctx->stream);
        if (rc == SDICE_ERR_NOMEM && group > 1) {            // less memory than hipMemGetInfo promised: fewer columns at once
            group = (group + 1) / 2;
            c0 -= group;                                      // (the loop increment adds it back: retry this column)
            rc = SDICE_OK;
            continue;
        }
        if (rc != SDICE_OK) break;
        double* d_cm = (double*)ctx->arena.alloc(vals * 8);
        double* d_q = (double*)ctx->arena.alloc(vals * 8);
        double* d_slab = slab ? (double*)ctx->arena.alloc(vals * 8) : nullptr;
        if (!d_cm || !d_q || (slab && !d_slab)) return SDICE_ERR_NOMEM;
        const double* src = d_p_inout;
        int64_t src_cols = cols;
        if (rc == SDICE_OK && slab) {
            // gather the column group [c0, c0+gc) into a dense [n, gc] slab first
            rc = hipMemcpy2DAsync(d_slab, (size_t)gc * 8, d_p_inout + c0, (size_t)pitch * 8, (size_t)gc * 8, (size_t)n,
                                  hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? SDICE_OK : SDICE_ERR_HIP;
            src = d_slab;
            src_cols = gc;
        }
        if (rc == SDICE_OK) rc = transpose(ctx, src, n, src_cols, d_cm);
        if (rc == SDICE_OK) rc = bh_segments(ctx, n, gc, d_cm, d_q);
        if (rc == SDICE_OK) {
            if (slab) {
                rc = transpose(ctx, d_q, gc, n, d_slab);
                if (rc == SDICE_OK)
                    rc = hipMemcpy2DAsync(d_p_inout + c0, (size_t)pitch * 8, d_slab, (size_t)gc * 8, (size_t)gc * 8, (size_t)n,
                                          hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? SDICE_OK : SDICE_ERR_HIP;
            } else {
                rc = transpose(ctx, d_q, gc, n, d_p_inout);
            }
        }
    }
    return rc;
}

extern "C" int sdice_bh_columns(sdice_ctx* ctx, int64_t n, int64_t cols, double* p_inout) {
    SD_ARG(ctx, "ctx is NULL");
    SD_ARG(n >= 0 && cols >= 0, "negative size");
    if (n == 0 || cols == 0) return SDICE_OK;
    SD_ARG(p_inout, "NULL pointer");
    double* d_rm = nullptr;
    int rc = sdice_dmalloc(ctx, n * cols * 8, (void**)&d_rm);
    if (rc == SDICE_OK) rc = sdice_h2d(ctx, d_rm, p_inout, n * cols * 8);
    if (rc == SDICE_OK) rc = sdice_bh_columns_dev(ctx, n, cols, d_rm);
    if (rc == SDICE_OK) rc = sdice_d2h(ctx, p_inout, d_rm, n * cols * 8);
    sdice_dfree(ctx, d_rm);
    return rc;
}
