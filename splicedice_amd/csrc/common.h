// Internal definitions shared by the HIP translation units of libsplicedice_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>
#include "sdice.h"

void sdice_set_error(const char* fmt, ...);

#define SD_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            sdice_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,              \
                            hipGetErrorString(_e));                                    \
            return SDICE_ERR_HIP;                                                      \
        }                                                                              \
    } while (0)

#define SD_TRY(expr)                                                                   \
    do {                                                                               \
        int _s = (expr);                                                               \
        if (_s != SDICE_OK) return _s;                                                 \
    } while (0)

#define SD_ARG(cond, msg)                                                              \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            sdice_set_error("%s: %s", __func__, msg);                                  \
            return SDICE_ERR_ARG;                                                      \
        }                                                                              \
    } while (0)

struct ProfPending {
    int name_id;
    hipEvent_t e0, e1;
};

struct ProfStat {
    int64_t launches = 0;
    double total_ms = 0.0;
};

// Bump arena for per-call device scratch.  Chunks are only released at reset() when the
// call needed more than one, after which a single chunk of the combined size is kept.
struct Arena {
    struct Chunk { char* p; size_t cap; size_t off; };
    std::vector<Chunk> chunks;
    void* alloc(size_t bytes);   // nullptr on failure (error set)
    int reset(hipStream_t s);
    // reset + make sure ONE chunk of at least `total` bytes exists (a call that knows its scratch
    // need up front avoids the grow-by-chunks path and its multi-GB free/alloc at the next reset)
    int reserve(size_t total, hipStream_t s);
    void release();
};

struct sdice_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 0;
    int64_t hbm_bytes = 0;
    char dev_name[128] = {0};

    Arena arena;

    // clustering result kept for sdice_cluster_col*
    int32_t* d_col = nullptr;
    int64_t col_cap = 0;
    int64_t nnz = 0;
    int cluster_reach = 0;   // max |row(neighbour) - row| of the last clustering (PS halo hint)
    // reach of the lists in d_col per block of 16 rows (fast clustering path): low byte = rows the block's lists reach
    // below its first row, next byte = rows beyond its last row, both saturating at 255; the PS kernel sizes a tile's
    // halo from it
    uint32_t* d_reach = nullptr;
    int64_t reach_cap = 0;   // words
    int64_t reach_n = 0;     // rows d_reach describes; 0 = not valid for d_col (generic clustering path, no clustering yet)
    int64_t* h_pinned = nullptr;  // small pinned buffer for scalar read-backs
    void* cluster_sb = nullptr;   // device status block of the fast clustering path (persistent, 768 B)
    bool cluster_pending = false; // an asynchronous sdice_cluster_dev has not been resolved yet

    // log-factorial table of the Fisher kernel: lf[k] = lgamma(k+1)
    double* d_lf = nullptr;
    int64_t lf_n = 0;
    uint64_t fisher_steps[2] = {0, 0};   // fisher.count_steps: useful / issued lane-steps of the last launch

    // profiling
    bool prof_on = false;
    int prof_mode = 0;   // 1 = every kernel, 2 = only the dominant kernel of each path
    std::vector<std::string> prof_names;
    std::map<std::string, int> prof_ids;
    std::vector<ProfStat> prof_stats;
    std::vector<ProfPending> prof_pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t t0 = nullptr, t1 = nullptr;

    // tuning parameters
    std::map<std::string, int64_t> params;

    // RCCL
    void* rccl_lib = nullptr;
    void* comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t comm_stream = nullptr;   // second stream for collectives between sdice_comm_fork and sdice_comm_join
    hipEvent_t comm_ev = nullptr;
    bool comm_forked = false;
    hipStream_t coll_stream() const { return comm_forked ? comm_stream : stream; }

    int64_t param(const char* name, int64_t dflt) const {
        auto it = params.find(name);
        return it == params.end() ? dflt : it->second;
    }
};

int sd_prof_begin(sdice_ctx* ctx, const char* name);
void sd_prof_end(sdice_ctx* ctx, int token);
int sd_prof_drain(sdice_ctx* ctx);

// Launch `kernel` on the context stream, bracketed by HIP events when profiling is on.
#define SD_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                          \
    do {                                                                               \
        int _tok = sd_prof_begin((ctx), (name));                                       \
        hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (ctx)->stream, __VA_ARGS__); \
        sd_prof_end((ctx), _tok);                                                      \
        SD_HIP(hipGetLastError());                                                     \
    } while (0)

static inline int64_t sd_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// status of an asynchronous sdice_cluster_dev (synchronises when one is pending)
int sd_cluster_resolve(sdice_ctx* ctx);
// SDICE_ERR_NOMEM (message naming nnz) when a neighbour list of nnz entries exceeds param cluster.max_nnz or, when that
// is 0, what fits the free device memory (+ the list buffer the context already holds); `host`: also what half of the
// host's physical memory holds.  Called BEFORE the list is allocated anywhere.
int sd_cluster_check_nnz(sdice_ctx* ctx, int64_t nnz, bool host);

// ---- internal device-level primitives shared between translation units ----
// stable LSD radix sort of (key64, val32) pairs; only the bits set in `bit_mask`
// (bits that differ between keys) are sorted on.  Result ends in keys_out/vals_out.
int sd_radix_sort_pairs(sdice_ctx* ctx, int64_t n, const uint64_t* d_keys_in, const uint32_t* d_vals_in,
                        uint64_t* d_keys_out, uint32_t* d_vals_out, uint64_t* d_keys_tmp, uint32_t* d_vals_tmp,
                        uint64_t bit_mask);
// the same for `segs` equally long segments of n keys, each sorted independently (contiguous layout)
int sd_radix_sort_pairs_segmented(sdice_ctx* ctx, int64_t n, int64_t segs, const uint64_t* d_keys_in,
                                  const uint32_t* d_vals_in, uint64_t* d_keys_out, uint32_t* d_vals_out,
                                  uint64_t* d_keys_tmp, uint32_t* d_vals_tmp, uint64_t bit_mask);
// exclusive scan of int64 values (in place allowed): out[i] = sum_{j<i} in[j]; total to d_total if non-null
int sd_exclusive_scan_i64(sdice_ctx* ctx, int64_t n, const int64_t* d_in, int64_t* d_out, int64_t* d_total);
