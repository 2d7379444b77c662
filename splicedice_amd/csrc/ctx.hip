// Context lifecycle, device memory, per-kernel HIP-event profiling, parameters.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[1024] = "";

void sdice_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sdice_last_error(void) { return g_err; }
extern "C" int sdice_version(void) { return SDICE_ABI_VERSION; }

// ------------------------------------------------------------------------------- arena
void* Arena::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    if (!chunks.empty()) {
        Chunk& c = chunks.back();
        if (c.off + bytes <= c.cap) {
            void* p = c.p + c.off;
            c.off += bytes;
            return p;
        }
    }
    size_t cap = bytes;
    if (!chunks.empty() && chunks.back().cap * 2 > cap) cap = chunks.back().cap * 2;
    if (cap < (size_t(1) << 20)) cap = size_t(1) << 20;
    char* p = nullptr;
    hipError_t e = hipMalloc((void**)&p, cap);
    if (e != hipSuccess) {
        cap = bytes;
        e = hipMalloc((void**)&p, cap);
        if (e != hipSuccess) {
            sdice_set_error("arena: hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
            return nullptr;
        }
    }
    chunks.push_back({p, cap, bytes});
    return p;
}

int Arena::reset(hipStream_t s) {
    if (chunks.size() > 1) {
        SD_HIP(hipStreamSynchronize(s));
        size_t total = 0;
        for (auto& c : chunks) { total += c.cap; (void)hipFree(c.p); }
        chunks.clear();
        char* p = nullptr;
        if (hipMalloc((void**)&p, total) == hipSuccess) chunks.push_back({p, total, 0});
    } else if (chunks.size() == 1) {
        chunks[0].off = 0;
    }
    return SDICE_OK;
}

int Arena::reserve(size_t total, hipStream_t s) {
    total = (total + 4095) & ~size_t(4095);
    if (chunks.size() == 1 && chunks[0].cap >= total) {
        chunks[0].off = 0;
        return SDICE_OK;
    }
    if (!chunks.empty()) {
        SD_HIP(hipStreamSynchronize(s));
        release();
    }
    char* p = nullptr;
    hipError_t e = hipMalloc((void**)&p, total);
    if (e != hipSuccess) {
        sdice_set_error("arena: hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
        return SDICE_ERR_NOMEM;
    }
    chunks.push_back({p, total, 0});
    return SDICE_OK;
}

void Arena::release() {
    for (auto& c : chunks) (void)hipFree(c.p);
    chunks.clear();
}

// ----------------------------------------------------------------------------- context
extern "C" int sdice_ctx_create(int device_ordinal, sdice_ctx** out) {
    SD_ARG(out != nullptr, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        sdice_set_error("sdice_ctx_create: no HIP device available (%s); there is no CPU backend",
                        e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return SDICE_ERR_HIP;
    }
    SD_ARG(device_ordinal >= 0 && device_ordinal < count, "device ordinal out of range");
    SD_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    SD_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        sdice_set_error("sdice_ctx_create: device %d is %s; this library ships gfx950 (MI355X) code only",
                        device_ordinal, prop.gcnArchName);
        return SDICE_ERR_HIP;
    }
    sdice_ctx* ctx = new sdice_ctx();
    ctx->device = device_ordinal;
    ctx->n_cu = prop.multiProcessorCount;
    ctx->hbm_bytes = (int64_t)prop.totalGlobalMem;
    snprintf(ctx->dev_name, sizeof(ctx->dev_name), "%s (%s)", prop.name, prop.gcnArchName);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->t0) != hipSuccess || hipEventCreate(&ctx->t1) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_pinned, 16384) != hipSuccess) {
        sdice_set_error("sdice_ctx_create: stream/event/pinned allocation failed");
        delete ctx;
        return SDICE_ERR_HIP;
    }
    *out = ctx;
    return SDICE_OK;
}

extern "C" int sdice_ctx_destroy(sdice_ctx* ctx) {
    if (!ctx) return SDICE_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    sdice_comm_destroy(ctx);
    for (auto& p : ctx->prof_pending) { ctx->event_pool.push_back(p.e0); ctx->event_pool.push_back(p.e1); }
    for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    if (ctx->d_col) (void)hipFree(ctx->d_col);
    if (ctx->d_reach) (void)hipFree(ctx->d_reach);
    if (ctx->d_lf) (void)hipFree(ctx->d_lf);
    if (ctx->cluster_sb) (void)hipFree(ctx->cluster_sb);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    ctx->arena.release();
    if (ctx->comm_stream) { (void)hipStreamSynchronize(ctx->comm_stream); (void)hipStreamDestroy(ctx->comm_stream); }
    if (ctx->comm_ev) (void)hipEventDestroy(ctx->comm_ev);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SDICE_OK;
}

extern "C" int sdice_sync(sdice_ctx* ctx) {
    SD_ARG(ctx, "ctx is NULL");
    if (ctx->comm_forked) SD_TRY(sdice_comm_join(ctx));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return sd_cluster_resolve(ctx);      // (reports the deferred status of an asynchronous clustering)
}

extern "C" int sdice_trim(sdice_ctx* ctx) {
    SD_ARG(ctx, "ctx is NULL");
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    ctx->arena.release();
    return sd_cluster_resolve(ctx);
}

extern "C" int sdice_device_info(sdice_ctx* ctx, char* name, int name_cap, int* compute_units,
                                 int64_t* hbm_bytes) {
    SD_ARG(ctx, "ctx is NULL");
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", ctx->dev_name);
    if (compute_units) *compute_units = ctx->n_cu;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return SDICE_OK;
}

// ------------------------------------------------------------------------ device memory
extern "C" int sdice_dmalloc(sdice_ctx* ctx, int64_t bytes, void** dptr) {
    SD_ARG(ctx && dptr && bytes >= 0, "bad arguments");
    *dptr = nullptr;
    SD_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes > 0 ? (size_t)bytes : 256);
    if (e != hipSuccess) {
        sdice_set_error("sdice_dmalloc(%lld): %s", (long long)bytes, hipGetErrorString(e));
        return SDICE_ERR_NOMEM;
    }
    return SDICE_OK;
}

extern "C" int sdice_dfree(sdice_ctx* ctx, void* dptr) {
    SD_ARG(ctx, "ctx is NULL");
    if (!dptr) return SDICE_OK;
    SD_HIP(hipStreamSynchronize(ctx->stream));
    SD_HIP(hipFree(dptr));
    return SDICE_OK;
}

extern "C" int sdice_h2d(sdice_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes) {
    SD_ARG(ctx && bytes >= 0, "bad arguments");
    if (bytes == 0) return SDICE_OK;
    SD_ARG(dst_dev && src_host, "NULL pointer");
    SD_HIP(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SDICE_OK;
}

extern "C" int sdice_d2h(sdice_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes) {
    SD_ARG(ctx && bytes >= 0, "bad arguments");
    if (bytes == 0) return SDICE_OK;
    SD_ARG(dst_host && src_dev, "NULL pointer");
    SD_HIP(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    return SDICE_OK;
}

extern "C" int sdice_dmemset(sdice_ctx* ctx, void* dptr, int value, int64_t bytes) {
    SD_ARG(ctx && bytes >= 0, "bad arguments");
    if (bytes == 0) return SDICE_OK;
    SD_HIP(hipMemsetAsync(dptr, value, (size_t)bytes, ctx->stream));
    return SDICE_OK;
}

// --------------------------------------------------------------------------- profiling
static hipEvent_t pool_get(sdice_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

int sd_prof_begin(sdice_ctx* ctx, const char* name) {
    if (!ctx->prof_on) return -1;
    if (ctx->prof_mode == 2) {          // only the dominant kernel of each path
        static const char* const kDominant[] = {"ps_tile_v3_kernel", "ps_tile_kernel", "ranksum_wave_kernel",
                                                "ranksum_count_kernel", "ranksum_pair_kernel", "ranksum_pairq_kernel",
                                                "ranksum_lane_kernel", "ranksum_block_kernel", "fisher_pairs_kernel", "rccl_allgather"};
        bool hit = false;
        for (const char* k : kDominant) hit = hit || strcmp(name, k) == 0;
        if (!hit) return -1;
    }
    if (ctx->prof_pending.size() >= 8192 && sd_prof_drain(ctx) != SDICE_OK) return -1;
    int id;
    auto it = ctx->prof_ids.find(name);
    if (it == ctx->prof_ids.end()) {
        id = (int)ctx->prof_names.size();
        ctx->prof_names.push_back(name);
        ctx->prof_ids[name] = id;
        ctx->prof_stats.push_back(ProfStat());
    } else {
        id = it->second;
    }
    ProfPending p;
    p.name_id = id;
    p.e0 = pool_get(ctx);
    p.e1 = pool_get(ctx);
    if (!p.e0 || !p.e1) return -1;
    (void)hipEventRecord(p.e0, ctx->stream);
    ctx->prof_pending.push_back(p);
    return (int)ctx->prof_pending.size() - 1;
}

void sd_prof_end(sdice_ctx* ctx, int token) {
    if (token < 0) return;
    (void)hipEventRecord(ctx->prof_pending[token].e1, ctx->stream);
}

int sd_prof_drain(sdice_ctx* ctx) {
    if (ctx->prof_pending.empty()) return SDICE_OK;
    SD_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& p : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            ctx->prof_stats[p.name_id].launches += 1;
            ctx->prof_stats[p.name_id].total_ms += ms;
        }
        ctx->event_pool.push_back(p.e0);
        ctx->event_pool.push_back(p.e1);
    }
    ctx->prof_pending.clear();
    return SDICE_OK;
}

extern "C" int sdice_prof_enable(sdice_ctx* ctx, int on) {
    SD_ARG(ctx, "ctx is NULL");
    if (!on) SD_TRY(sd_prof_drain(ctx));
    ctx->prof_on = on != 0;
    ctx->prof_mode = on;
    return SDICE_OK;
}

extern "C" int sdice_prof_reset(sdice_ctx* ctx) {
    SD_ARG(ctx, "ctx is NULL");
    SD_TRY(sd_prof_drain(ctx));
    for (auto& s : ctx->prof_stats) s = ProfStat();
    return SDICE_OK;
}

extern "C" int sdice_prof_query(sdice_ctx* ctx, const char* name, int64_t* launches, double* total_ms) {
    SD_ARG(ctx && name, "bad arguments");
    SD_TRY(sd_prof_drain(ctx));
    auto it = ctx->prof_ids.find(name);
    if (launches) *launches = 0;
    if (total_ms) *total_ms = 0.0;
    if (it == ctx->prof_ids.end()) return SDICE_OK;
    if (launches) *launches = ctx->prof_stats[it->second].launches;
    if (total_ms) *total_ms = ctx->prof_stats[it->second].total_ms;
    return SDICE_OK;
}

extern "C" int sdice_prof_report(sdice_ctx* ctx, char* buf, int cap) {
    SD_ARG(ctx && buf && cap > 0, "bad arguments");
    SD_TRY(sd_prof_drain(ctx));
    int off = 0;
    buf[0] = 0;
    for (size_t i = 0; i < ctx->prof_names.size(); ++i) {
        if (ctx->prof_stats[i].launches == 0) continue;
        int w = snprintf(buf + off, cap - off, "%s %lld %.6f\n", ctx->prof_names[i].c_str(),
                         (long long)ctx->prof_stats[i].launches, ctx->prof_stats[i].total_ms);
        if (w < 0 || w >= cap - off) break;
        off += w;
    }
    return SDICE_OK;
}

extern "C" int sdice_timer_start(sdice_ctx* ctx) {
    SD_ARG(ctx, "ctx is NULL");
    SD_HIP(hipEventRecord(ctx->t0, ctx->stream));
    return SDICE_OK;
}

extern "C" int sdice_timer_stop(sdice_ctx* ctx, double* elapsed_ms) {
    SD_ARG(ctx && elapsed_ms, "bad arguments");
    SD_HIP(hipEventRecord(ctx->t1, ctx->stream));
    SD_HIP(hipEventSynchronize(ctx->t1));
    float ms = 0.f;
    SD_HIP(hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
    *elapsed_ms = ms;
    return SDICE_OK;
}

extern "C" int sdice_set_param(sdice_ctx* ctx, const char* name, int64_t value) {
    SD_ARG(ctx && name, "bad arguments");
    static const char* known[] = {"ps.lds_bytes", "ps.tile_rows", "ps.threads", "ps.chunk_cols",
                                  "ps.xcd_remap", "ps.halo_rows", "cluster.generic", "cluster.legacy", "cluster.lds_cap", "cluster.ablate", "cluster.nb_grid", "cluster.sample_sort", "cluster.bucket_mean", "cluster.spb", "cluster.max_nnz", "ps.ablate", "ps.quantize3", "ps.prio", "ps.nt_loads", "ps.gen1", "ps.use_reach", "sort.rounds", "ranksum.variant", "ranksum.ablate",
                                  "fisher.table_max", "fisher.refill", "fisher.unroll", "fisher.count_steps", "bh.columns_path", "bh.vector_path", "bhv.mean", "bhv.cap", "bh.reg_cap", "bh.mean", "bh.rows_per_block", "bh.fused_count", "bh.finish_cols", "bh.finish_nt", "bh.wg", "bh.big_wg", "bh.spb", nullptr};
    for (int i = 0; known[i]; ++i)
        if (strcmp(known[i], name) == 0) {
            ctx->params[name] = value;
            return SDICE_OK;
        }
    sdice_set_error("sdice_set_param: unknown parameter '%s'", name);
    return SDICE_ERR_ARG;
}
