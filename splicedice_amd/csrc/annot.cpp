// Host side of the GTF annotation join of compare_sample_sets (compareSampleSets.py:238-264): for every
// tested event the reference walks ALL gene intervals of the event's (chromosome, strand) in a Python loop
// and keeps those that contain the event's start or stop.  Here: the same scan, threaded, over flat arrays;
// the result is a CSR of interval indices in the reference's order (the dict order of the intervals).
#include "sdice.h"

int sd_default_threads();   // textio.cpp: hardware threads capped by the cgroup's CPU quota
#include <algorithm>
#include <cstdint>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

void sdice_set_error(const char* fmt, ...);

namespace {
// An exception inside a worker (or std::system_error from a thread that cannot be started) must not reach
// std::terminate: workers catch into `err`, every started thread is joined, and the first exception is rethrown on
// the caller's thread, where the extern "C" function-try-block turns it into SDICE_ERR_*.
template <class F>
void for_blocks(int64_t n, int threads, F f) {
    int used = threads > 0 ? threads : sd_default_threads();
    if (used < 1) used = 1;
    if (used > 64) used = 64;
    if (n < 4096) used = 1;
    std::vector<std::thread> pool;
    std::exception_ptr err;
    std::mutex err_mu;
    auto guarded = [&](int64_t a, int64_t b) {
        try { f(a, b); }
        catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
    };
    const int64_t per = (n + used - 1) / used;
    try {
        for (int t = 0; t < used; ++t) {
            const int64_t a = std::min<int64_t>(n, t * per), b = std::min<int64_t>(n, a + per);
            if (a < b) pool.emplace_back(guarded, a, b);
        }
    } catch (...) {
        std::lock_guard<std::mutex> g(err_mu);
        if (!err) err = std::current_exception();
    }
    for (auto& th : pool) th.join();
    if (err) std::rethrow_exception(err);
}
}  // namespace

// Event e belongs to group ev_group[e] (-1 = its (chromosome, strand) has no intervals) and has the two
// positions ev_a[e], ev_b[e]; group g owns the intervals grp_ptr[g] .. grp_ptr[g+1] of lo[] / hi[].
// Interval k matches when lo <= a <= hi or lo <= b <= hi.  Two calls: with out_idx == NULL the function
// fills out_ptr[0..n_events] (prefix sums of the match counts); with out_idx (capacity out_cap >= out_ptr[n])
// it also writes the matching interval indices, per event in increasing k.
extern "C" int sdice_interval_overlaps(int64_t n_events, const int32_t* ev_group, const int64_t* ev_a, const int64_t* ev_b,
                                       int32_t n_groups, const int64_t* grp_ptr, const int64_t* lo, const int64_t* hi,
                                       int64_t* out_ptr, int64_t* out_idx, int64_t out_cap, int threads) try {
    if (n_events < 0 || n_groups < 0 || !out_ptr || (n_events > 0 && (!ev_group || !ev_a || !ev_b)) ||
        (n_groups > 0 && (!grp_ptr || (grp_ptr[n_groups] > 0 && (!lo || !hi))))) {
        sdice_set_error("sdice_interval_overlaps: bad arguments");
        return SDICE_ERR_ARG;
    }
    for (int64_t e = 0; e < n_events; ++e)
        if (ev_group[e] < -1 || ev_group[e] >= n_groups) {
            sdice_set_error("sdice_interval_overlaps: group index out of range");
            return SDICE_ERR_ARG;
        }
    for (int32_t g = 0; g < n_groups; ++g)
        if (grp_ptr[g] < 0 || grp_ptr[g + 1] < grp_ptr[g]) {
            sdice_set_error("sdice_interval_overlaps: grp_ptr must be non-decreasing from 0");
            return SDICE_ERR_ARG;
        }
    if (!out_idx) {
        out_ptr[0] = 0;
        for_blocks(n_events, threads, [&](int64_t a, int64_t b) {
            for (int64_t e = a; e < b; ++e) {
                int64_t c = 0;
                const int32_t g = ev_group[e];
                if (g >= 0) {
                    const int64_t x = ev_a[e], y = ev_b[e];
                    for (int64_t k = grp_ptr[g]; k < grp_ptr[g + 1]; ++k)
                        c += ((lo[k] <= x) & (x <= hi[k])) | ((lo[k] <= y) & (y <= hi[k]));
                }
                out_ptr[e + 1] = c;
            }
        });
        for (int64_t e = 0; e < n_events; ++e) out_ptr[e + 1] += out_ptr[e];
        return SDICE_OK;
    }
    if (out_cap < out_ptr[n_events]) {
        sdice_set_error("sdice_interval_overlaps: out_idx too small");
        return SDICE_ERR_ARG;
    }
    for_blocks(n_events, threads, [&](int64_t a, int64_t b) {
        for (int64_t e = a; e < b; ++e) {
            const int32_t g = ev_group[e];
            if (g < 0) continue;
            int64_t w = out_ptr[e];
            const int64_t w_end = out_ptr[e + 1];
            const int64_t x = ev_a[e], y = ev_b[e];
            for (int64_t k = grp_ptr[g]; k < grp_ptr[g + 1] && w < w_end; ++k)
                if ((lo[k] <= x && x <= hi[k]) || (lo[k] <= y && y <= hi[k])) out_idx[w++] = k;
        }
    });
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_interval_overlaps: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_interval_overlaps: unknown exception");
    return SDICE_ERR_STATE;
}
