// K0: sorted set of 64-bit keys (the junction union of `quant`).
//
// Replaces the `self.junctions = set(); ... add(junction)` accumulation over every sample file and
// the later `sorted(self.junctions)` (SPLICEDICE.py:147-228, :96): the host packs each admitted
// junction into one order-preserving 64-bit key (chrom rank | left | right - left | strand, i.e.
// the tuple order (chrom, left, right, strand)), this entry point sorts the concatenation of all
// files and drops duplicates.  Radix sort (radix.hip) + adjacent-difference flags + exclusive scan
// (scan.hip) + compaction.  16 algorithmic bytes per input key.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) unique_flag_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                          int64_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ void __launch_bounds__(256) unique_compact_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                             const int64_t* __restrict__ pos,
                                                             uint64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && (i == 0 || keys[i] != keys[i - 1])) out[pos[i]] = keys[i];
}

}  // namespace

extern "C" int sdice_sort_unique_u64(sdice_ctx* ctx, int64_t n, uint64_t* keys_inout, int64_t* n_unique) {
    SD_ARG(ctx && n >= 0 && n_unique, "bad arguments");
    *n_unique = 0;
    if (n == 0) return SDICE_OK;
    SD_ARG(keys_inout, "NULL keys");
    SD_ARG(n < ((int64_t)1 << 32), "more than 2^32 keys");
    SD_HIP(hipSetDevice(ctx->device));
    SD_TRY(ctx->arena.reserve((size_t)n * (8 * 3 + 4 * 3 + 8) + (1 << 20) + (size_t)sd_ceil_div(n, 3072) * 1024 + 65536,
                              ctx->stream));
    uint64_t* kA = (uint64_t*)ctx->arena.alloc((size_t)n * 8);
    uint64_t* kB = (uint64_t*)ctx->arena.alloc((size_t)n * 8);
    uint64_t* kC = (uint64_t*)ctx->arena.alloc((size_t)n * 8);
    uint32_t* vA = (uint32_t*)ctx->arena.alloc((size_t)n * 4);
    uint32_t* vB = (uint32_t*)ctx->arena.alloc((size_t)n * 4);
    uint32_t* vC = (uint32_t*)ctx->arena.alloc((size_t)n * 4);
    int64_t* pos = (int64_t*)ctx->arena.alloc((size_t)n * 8);
    if (!kA || !kB || !kC || !vA || !vB || !vC || !pos) return SDICE_ERR_NOMEM;
    SD_HIP(hipMemcpyAsync(kA, keys_inout, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    // only the digits that vary need a pass: OR / AND of all keys on the host is one cheap sweep
    uint64_t any = 0, all = ~0ull;
    for (int64_t i = 0; i < n; ++i) { any |= keys_inout[i]; all &= keys_inout[i]; }
    SD_HIP(hipMemsetAsync(vA, 0, (size_t)n * 4, ctx->stream));
    SD_TRY(sd_radix_sort_pairs(ctx, n, kA, vA, kB, vB, kC, vC, any & ~all));
    const unsigned blocks = (unsigned)sd_ceil_div(n, 256);
    SD_LAUNCH(ctx, "unique_flag_kernel", unique_flag_kernel, dim3(blocks), dim3(256), 0, kB, n, pos);
    int64_t* d_total = (int64_t*)ctx->arena.alloc(8);
    if (!d_total) return SDICE_ERR_NOMEM;
    SD_TRY(sd_exclusive_scan_i64(ctx, n, pos, pos, d_total));      // pos[i] = output slot, total = distinct keys
    SD_LAUNCH(ctx, "unique_compact_kernel", unique_compact_kernel, dim3(blocks), dim3(256), 0, kB, n, pos, kC);
    int64_t count = 0;
    SD_HIP(hipMemcpyAsync(&count, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(hipStreamSynchronize(ctx->stream));
    SD_HIP(hipMemcpy(keys_inout, kC, (size_t)count * 8, hipMemcpyDeviceToHost));
    *n_unique = count;
    return SDICE_OK;
}
