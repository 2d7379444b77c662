// C1: RCCL all-gather of row shards over xGMI (one process per GPU, one communicator per
// context).  The reference is single-process; this step is new.
//
// librccl.so.1 is dlopen'ed on first use so that single-GPU runs never touch RCCL.  The
// 128-byte unique id is created on rank 0 and distributed by the caller (bench.py and the
// host package use torch.distributed / any byte channel for that).
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef struct { char internal[128]; } sd_ncclUniqueId;
typedef int (*fn_GetUniqueId)(sd_ncclUniqueId*);
typedef int (*fn_CommInitRank)(void**, int, sd_ncclUniqueId, int);
typedef int (*fn_CommDestroy)(void*);
typedef const char* (*fn_GetErrorString)(int);
typedef int (*fn_AllGather)(const void*, void*, size_t, int /*ncclDataType_t*/, void*, hipStream_t);
typedef int (*fn_Send)(const void*, size_t, int, int /*peer*/, void*, hipStream_t);
typedef int (*fn_Recv)(void*, size_t, int, int /*peer*/, void*, hipStream_t);
typedef int (*fn_Group)(void);

struct Rccl {
    void* lib = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
    fn_AllGather AllGather = nullptr;
    fn_Send Send = nullptr;
    fn_Recv Recv = nullptr;
    fn_Group GroupStart = nullptr;
    fn_Group GroupEnd = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return SDICE_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
    void* h = nullptr;
    for (int i = 0; names[i] && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        sdice_set_error("cannot dlopen librccl.so.1: %s", dlerror());
        return SDICE_ERR_COMM;
    }
    g_rccl.GetUniqueId = (fn_GetUniqueId)dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (fn_CommInitRank)dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (fn_CommDestroy)dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (fn_GetErrorString)dlsym(h, "ncclGetErrorString");
    g_rccl.AllGather = (fn_AllGather)dlsym(h, "ncclAllGather");
    g_rccl.Send = (fn_Send)dlsym(h, "ncclSend");
    g_rccl.Recv = (fn_Recv)dlsym(h, "ncclRecv");
    g_rccl.GroupStart = (fn_Group)dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (fn_Group)dlsym(h, "ncclGroupEnd");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.Send ||
        !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
        sdice_set_error("librccl is missing a required symbol");
        dlclose(h);
        return SDICE_ERR_COMM;
    }
    g_rccl.lib = h;
    return SDICE_OK;
}

int nccl_fail(const char* what, int code) {
    sdice_set_error("%s failed: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "?", code);
    return SDICE_ERR_COMM;
}

}  // namespace

extern "C" int sdice_comm_unique_id(sdice_ctx* ctx, void* id_out) {
    SD_ARG(ctx && id_out, "bad arguments");
    SD_TRY(load_rccl());
    sd_ncclUniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id_out, &id, SDICE_COMM_ID_BYTES);
    return SDICE_OK;
}

extern "C" int sdice_comm_init(sdice_ctx* ctx, const void* id, int rank, int world) {
    SD_ARG(ctx && id, "bad arguments");
    SD_ARG(world >= 1 && rank >= 0 && rank < world, "bad rank/world");
    SD_ARG(ctx->comm == nullptr, "communicator already initialised");
    SD_TRY(load_rccl());
    SD_HIP(hipSetDevice(ctx->device));
    sd_ncclUniqueId uid;
    memcpy(&uid, id, SDICE_COMM_ID_BYTES);
    void* comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, world, uid, rank);
    if (rc != 0) return nccl_fail("ncclCommInitRank", rc);
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world;
    return SDICE_OK;
}

extern "C" int sdice_comm_destroy(sdice_ctx* ctx) {
    if (!ctx || !ctx->comm) return SDICE_OK;
    (void)hipStreamSynchronize(ctx->stream);
    g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->rank = 0;
    ctx->world = 1;
    return SDICE_OK;
}

// Collectives beside compute: between sdice_comm_fork and sdice_comm_join the collectives of this context are enqueued
// on a SECOND stream.  fork: that stream waits for everything enqueued on the main stream so far (call it again before
// each collective whose input the main stream has just produced); join: the main stream waits for the collectives issued
// since, and collectives go back to the main stream.  (`pairwise`, correction per pair column: the all-to-all that takes
// a column group's corrected values home runs while the next group is being corrected.)
extern "C" int sdice_comm_fork(sdice_ctx* ctx) {
    SD_ARG(ctx, "ctx is NULL");
    SD_HIP(hipSetDevice(ctx->device));
    if (!ctx->comm_stream) {
        SD_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        SD_HIP(hipEventCreateWithFlags(&ctx->comm_ev, hipEventDisableTiming));
    }
    SD_HIP(hipEventRecord(ctx->comm_ev, ctx->stream));
    SD_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ev, 0));
    ctx->comm_forked = true;
    return SDICE_OK;
}

extern "C" int sdice_comm_join(sdice_ctx* ctx) {
    SD_ARG(ctx, "ctx is NULL");
    if (!ctx->comm_forked) return SDICE_OK;
    SD_HIP(hipEventRecord(ctx->comm_ev, ctx->comm_stream));
    SD_HIP(hipStreamWaitEvent(ctx->stream, ctx->comm_ev, 0));
    ctx->comm_forked = false;
    return SDICE_OK;
}

extern "C" int sdice_allgather_dev(sdice_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes_per_rank) {
    SD_ARG(ctx && bytes_per_rank >= 0, "bad arguments");
    if (bytes_per_rank == 0) return SDICE_OK;
    SD_ARG(d_send && d_recv, "NULL pointer");
    if (ctx->world == 1 && !ctx->comm) {
        if ((const void*)d_send != d_recv)
            SD_HIP(hipMemcpyAsync(d_recv, d_send, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice, ctx->coll_stream()));
        return SDICE_OK;
    }
    SD_ARG(ctx->comm, "communicator not initialised (sdice_comm_init)");
    int tok = sd_prof_begin(ctx, "rccl_allgather");
    const int rc = g_rccl.AllGather(d_send, d_recv, (size_t)bytes_per_rank, 0 /* ncclInt8 */, ctx->comm, ctx->coll_stream());
    sd_prof_end(ctx, tok);
    if (rc != 0) return nccl_fail("ncclAllGather", rc);
    return SDICE_OK;
}

// All-to-all of equal blocks (rows -> columns transpose of the pairwise p-value matrix, SURVEY
// 8(e) K6): block q of d_send goes to rank q, block r of d_recv comes from rank r.  Grouped
// ncclSend/ncclRecv pairs: on the xGMI mesh every pair has its own direct link.
extern "C" int sdice_alltoall_dev(sdice_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes_per_peer) {
    SD_ARG(ctx && bytes_per_peer >= 0, "bad arguments");
    if (bytes_per_peer == 0) return SDICE_OK;
    SD_ARG(d_send && d_recv && d_send != d_recv, "NULL or aliased buffers");
    if (ctx->world == 1 && !ctx->comm) {
        SD_HIP(hipMemcpyAsync(d_recv, d_send, (size_t)bytes_per_peer, hipMemcpyDeviceToDevice, ctx->coll_stream()));
        return SDICE_OK;
    }
    SD_ARG(ctx->comm, "communicator not initialised (sdice_comm_init)");
    int tok = sd_prof_begin(ctx, "rccl_alltoall");
    int rc = g_rccl.GroupStart();
    for (int q = 0; q < ctx->world && rc == 0; ++q) {
        rc = g_rccl.Send((const char*)d_send + (size_t)q * bytes_per_peer, (size_t)bytes_per_peer, 0 /* ncclInt8 */, q,
                         ctx->comm, ctx->coll_stream());
        if (rc == 0)
            rc = g_rccl.Recv((char*)d_recv + (size_t)q * bytes_per_peer, (size_t)bytes_per_peer, 0, q, ctx->comm,
                             ctx->coll_stream());
    }
    const int rc_end = g_rccl.GroupEnd();
    sd_prof_end(ctx, tok);
    if (rc != 0) return nccl_fail("ncclSend/ncclRecv", rc);
    if (rc_end != 0) return nccl_fail("ncclGroupEnd", rc_end);
    return SDICE_OK;
}

// Strided device copy: `rows` rows of `width_bytes` from a matrix with row pitch spitch into one
// with row pitch dpitch (packing / unpacking a column block of a row-major matrix).
extern "C" int sdice_copy2d_dev(sdice_ctx* ctx, void* d_dst, int64_t dpitch, const void* d_src, int64_t spitch,
                                int64_t width_bytes, int64_t rows) {
    SD_ARG(ctx && width_bytes >= 0 && rows >= 0, "bad arguments");
    if (width_bytes == 0 || rows == 0) return SDICE_OK;
    SD_ARG(d_dst && d_src && dpitch >= width_bytes && spitch >= width_bytes, "bad pointers or pitches");
    SD_HIP(hipSetDevice(ctx->device));
    SD_HIP(hipMemcpy2DAsync(d_dst, (size_t)dpitch, d_src, (size_t)spitch, (size_t)width_bytes, (size_t)rows,
                            hipMemcpyDeviceToDevice, ctx->stream));
    return SDICE_OK;
}
