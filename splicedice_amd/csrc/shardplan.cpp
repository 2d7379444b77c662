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

extern "C" int sdice_shard_plan(int64_t n, const int64_t* row_ptr, const int32_t* col, int32_t world, double max_shift_frac,
                                int64_t* plan /* [world][4] = own_lo, own_hi, ext_lo, ext_hi */) try {
    if (n < 0 || world < 1 || !row_ptr || !plan || (row_ptr[n] > 0 && !col)) {
        sdice_set_error("sdice_shard_plan: bad arguments");
        return SDICE_ERR_ARG;
    }
    // clean[r], r in [0, n]: furthest row referenced by the rows < r lies before r, nearest row referenced
    // by the rows >= r is not before r
    std::vector<uint8_t> clean((size_t)n + 1, 1);
    if (n > 0) {
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
        for (int64_t r = 1; r < n; ++r) hi[(size_t)r] = std::max(hi[(size_t)r], hi[(size_t)r - 1]);        // prefix maximum
        for (int64_t r = n - 2; r >= 0; --r) lo[(size_t)r] = std::min(lo[(size_t)r], lo[(size_t)r + 1]);   // suffix minimum
        for (int64_t r = 1; r < n; ++r) clean[(size_t)r] = (hi[(size_t)r - 1] < r && lo[(size_t)r] >= r) ? 1 : 0;
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
        if (hi_r > lo_r)
            for (int64_t q = row_ptr[lo_r]; q < row_ptr[hi_r]; ++q) {
                elo = std::min<int64_t>(elo, col[q]);
                ehi = std::max<int64_t>(ehi, (int64_t)col[q] + 1);
            }
        plan[4 * k] = lo_r; plan[4 * k + 1] = hi_r; plan[4 * k + 2] = elo; plan[4 * k + 3] = ehi;
    }
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_shard_plan: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_shard_plan: unknown exception");
    return SDICE_ERR_STATE;
}
