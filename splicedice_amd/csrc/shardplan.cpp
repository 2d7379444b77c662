// Junction-axis shard plan for multi-GPU runs (new; the reference is single-process) -- the
// `sdice_shard_plan` of SURVEY 8(b).  Host only, no device, no context.
//
// Rows (junctions in output order) are cut into `world` contiguous ranges.  A cut before row r is CLEAN
// when no CSR edge joins a row < r with a row >= r; overlap clusters are gene sized, so a clean cut
// almost always lies within a few rows of the ideal position k * n / world and the shard then needs no
// halo.  When none lies within max_shift_frac * n / world of it, the ideal position is kept and the shard
// is extended by the rows its own rows reference (read-only halo: ext_lo / ext_hi).
#include <stdint.h>
#include <algorithm>
#include <exception>
#include <vector>
#include "sdice.h"

void sdice_set_error(const char* fmt, ...);

// lo[r] / hi[r]: nearest / furthest row that row r's list references (r itself when the list is empty) -> plan
static void plan_from_reach(int64_t n, std::vector<int64_t>& lo, std::vector<int64_t>& hi, int32_t world, double max_shift_frac,
                            int64_t* plan) {
    // clean[r], r in [0, n]: furthest row referenced by the rows < r lies before r, nearest row referenced
    // by the rows >= r is not before r
    std::vector<uint8_t> clean((size_t)n + 1, 1);
    std::vector<int64_t> pmax(hi), smin(lo);
    if (n > 0) {
        for (int64_t r = 1; r < n; ++r) pmax[(size_t)r] = std::max(pmax[(size_t)r], pmax[(size_t)r - 1]);        // prefix maximum
        for (int64_t r = n - 2; r >= 0; --r) smin[(size_t)r] = std::min(smin[(size_t)r], smin[(size_t)r + 1]);   // suffix minimum
        for (int64_t r = 1; r < n; ++r) clean[(size_t)r] = (pmax[(size_t)r - 1] < r && smin[(size_t)r] >= r) ? 1 : 0;
    }
    const int64_t max_shift = std::max<int64_t>(1, (int64_t)(max_shift_frac * (double)n / (double)world));
    std::vector<int64_t> bounds((size_t)world + 1, 0);
    for (int32_t k = 1; k < world; ++k) {
        const int64_t ideal = (int64_t)k * n / world;
        // nearest clean position on either side of the ideal one (the left one wins a tie)
        int64_t left = ideal - 1, right = ideal;
        while (left >= 0 && !clean[(size_t)left]) --left;
        while (right <= n && !clean[(size_t)right]) ++right;
        int64_t best = ideal;
        const int64_t dl = left >= 0 ? ideal - left : INT64_MAX, dr = right <= n ? right - ideal : INT64_MAX;
        if (dl != INT64_MAX || dr != INT64_MAX) best = dl <= dr ? left : right;
        if (std::llabs(best - ideal) > max_shift) best = ideal;
        bounds[(size_t)k] = std::max(best, bounds[(size_t)k - 1]);
    }
    bounds[(size_t)world] = n;
    for (int32_t k = 0; k < world; ++k) {
        const int64_t lo_r = bounds[(size_t)k], hi_r = bounds[(size_t)k + 1];
        int64_t elo = lo_r, ehi = hi_r;
        for (int64_t r = lo_r; r < hi_r; ++r) {
            elo = std::min(elo, lo[(size_t)r]);
            ehi = std::max(ehi, hi[(size_t)r] + 1);
        }
        plan[4 * k] = lo_r; plan[4 * k + 1] = hi_r; plan[4 * k + 2] = elo; plan[4 * k + 3] = ehi;
    }
}

extern "C" int sdice_shard_plan(int64_t n, const int64_t* row_ptr, const int32_t* col, int32_t world, double max_shift_frac,
                                int64_t* plan /* [world][4] = own_lo, own_hi, ext_lo, ext_hi */) try {
    if (n < 0 || world < 1 || !row_ptr || !plan || (row_ptr[n] > 0 && !col)) {
        sdice_set_error("sdice_shard_plan: bad arguments");
        return SDICE_ERR_ARG;
    }
    std::vector<int64_t> lo((size_t)n), hi((size_t)n);
    for (int64_t r = 0; r < n; ++r) {
        int64_t a = r, b = r;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
            const int64_t c = col[k];
            if (c < 0 || c >= n) { sdice_set_error("sdice_shard_plan: column index out of range"); return SDICE_ERR_ARG; }
            a = std::min(a, c); b = std::max(b, c);
        }
        lo[(size_t)r] = a; hi[(size_t)r] = b;
    }
    plan_from_reach(n, lo, hi, world, max_shift_frac, plan);
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_shard_plan: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_shard_plan: unknown exception");
    return SDICE_ERR_STATE;
}

// The same plan from the junction coordinates alone (rows in OUTPUT order: sorted by (chrom, left, right, strand)), so
// that no rank has to cluster the whole junction set to learn where to cut it: junctions i < j of one (chrom, strand)
// are joined iff left[j] <= right[i] (SPLICEDICE.py:237-250, the sweep's inclusive test), hence row r's list reaches
// back to the first earlier row of its (chrom, strand) whose running maximum of `right` is >= left[r], and forward to the
// last later row of its (chrom, strand) with left <= right[r].
extern "C" int sdice_shard_plan_junctions(int64_t n, const int32_t* chrom, const int32_t* left, const int32_t* right,
                                          const int8_t* strand, int32_t world, double max_shift_frac, int64_t* plan) try {
    if (n < 0 || world < 1 || !plan || (n > 0 && (!chrom || !left || !right || !strand))) {
        sdice_set_error("sdice_shard_plan_junctions: bad arguments");
        return SDICE_ERR_ARG;
    }
    for (int64_t r = 1; r < n; ++r) {
        const bool ordered = chrom[r - 1] < chrom[r] || (chrom[r - 1] == chrom[r] && (left[r - 1] < left[r] ||
                             (left[r - 1] == left[r] && (right[r - 1] < right[r] || (right[r - 1] == right[r] && strand[r - 1] < strand[r])))));
        if (!ordered) { sdice_set_error("sdice_shard_plan_junctions: rows must be distinct and sorted by (chrom, left, right, strand)"); return SDICE_ERR_ARG; }
    }
    std::vector<int64_t> lo((size_t)n), hi((size_t)n);
    std::vector<int64_t> rows;          // rows of one (chrom, strand), in row order (= sorted by left)
    std::vector<int32_t> pmax;          // running maximum of right over them
    for (int64_t c0 = 0; c0 < n;) {
        int64_t c1 = c0;
        while (c1 < n && chrom[c1] == chrom[c0]) ++c1;
        for (int sd = 0; sd < 2; ++sd) {
            rows.clear(); pmax.clear();
            for (int64_t r = c0; r < c1; ++r) {
                if ((strand[r] != 0) != (sd != 0)) continue;
                rows.push_back(r);
                pmax.push_back(pmax.empty() ? right[r] : std::max(pmax.back(), right[r]));
            }
            const size_t m = rows.size();
            for (size_t q = 0; q < m; ++q) {
                const int64_t r = rows[q];
                // first earlier member whose running maximum of right reaches left[r]
                const size_t a = (size_t)(std::lower_bound(pmax.begin(), pmax.begin() + (std::ptrdiff_t)q, left[r]) - pmax.begin());
                lo[(size_t)r] = a < q ? rows[a] : r;
                // last later member with left <= right[r]
                size_t b0 = q + 1, b1 = m;
                while (b0 < b1) { const size_t mid = (b0 + b1) >> 1; if (left[rows[mid]] <= right[r]) b0 = mid + 1; else b1 = mid; }
                hi[(size_t)r] = b0 - 1 > q ? rows[b0 - 1] : r;
            }
        }
        c0 = c1;
    }
    plan_from_reach(n, lo, hi, world, max_shift_frac, plan);
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_shard_plan_junctions: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_shard_plan_junctions: unknown exception");
    return SDICE_ERR_STATE;
}
