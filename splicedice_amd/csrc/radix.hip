// Stable LSD radix sort of (u64 key, u32 value) pairs, 8-bit digits, 3 launches per digit:
//
//   1. radix_hist_kernel    one 256-thread workgroup per tile of TILE keys: 256-bin histogram in
//                           LDS -> hist[bin][tile] (bin-major)
//   2. radix_binscan_kernel one workgroup per bin: exclusive scan of hist[bin][*] in place,
//                           bin total -> bin_total[bin]
//   3. radix_scatter_kernel one workgroup per tile.  Each of its 4 waves owns a contiguous
//                           quarter of the tile (kept in registers).  Per-wave histograms give the
//                           wave bases; every wave then walks its quarter 64 keys at a time, the
//                           stable rank of a key among equal digits of the wave comes from eight
//                           64-bit ballots (wave-wide match), running per-digit offsets live in LDS.
//                           Keys are first placed into an LDS staging buffer at their position in
//                           the tile's digit-sorted order, then written out so that consecutive
//                           lanes write consecutive addresses inside each digit run (coalesced
//                           instead of a 64-way scatter per wave).
// Digits whose bits are constant over all keys (bit_mask) are skipped entirely.
#include "common.h"

namespace {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
// ROUNDS = keys per lane (template parameter): a workgroup owns a tile of THREADS * ROUNDS keys and
// each of its waves a contiguous run of 64 * ROUNDS.  12 (3072-key tiles) is the default; 4
// (param sort.rounds) was tried for small sorts: a 1M-key pass is three ~5 us kernels at the
// launch/drain floor either way, so smaller tiles do not help there.

// Every kernel takes a segment index in blockIdx.y: `segs` equally long, independently sorted
// segments of n keys each (segs = 1 for a plain sort; BH per pair column sorts 19 900 at once).
template <int ROUNDS>
__global__ void __launch_bounds__(THREADS) radix_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                             int n_tiles, uint32_t* __restrict__ hist) {
    constexpr int TILE = THREADS * ROUNDS;
    __shared__ uint32_t h[RADIX];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    keys += (int64_t)blockIdx.y * n;
    hist += (int64_t)blockIdx.y * RADIX * n_tiles;
    h[tid] = 0;
    __syncthreads();
    const int64_t beg = (int64_t)tile * TILE;
    const int64_t end = min(n, beg + TILE);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = beg + r * THREADS + tid;
        if (i < end) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    hist[(int64_t)tid * n_tiles + tile] = h[tid];
}

__global__ void __launch_bounds__(256) radix_binscan_kernel(uint32_t* __restrict__ hist, int n_tiles,
                                                            uint32_t* __restrict__ bin_total) {
    // exclusive scan of hist[bin][0..n_tiles) in place
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    const int bin = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    hist += (int64_t)blockIdx.y * RADIX * n_tiles;
    bin_total += (int64_t)blockIdx.y * RADIX;
    uint32_t* row = hist + (int64_t)bin * n_tiles;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 256) {
        const int i = base + tid;
        const uint32_t v = i < n_tiles ? row[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        uint32_t wbase = 0;
        for (int k = 0; k < w; ++k) wbase += wsum[k];
        const uint32_t carry = carry_s;
        if (i < n_tiles) row[i] = carry + wbase + x - v;
        __syncthreads();
        if (tid == 255) carry_s = carry + wbase + x;
        __syncthreads();
    }
    if (tid == 0) bin_total[bin] = carry_s;
}

// Same scan for short segments (n_tiles small): one workgroup per SEGMENT, thread b walks the
// n_tiles counters of bin b -- instead of 256 workgroups per segment with a handful of values each.
__global__ void __launch_bounds__(RADIX) radix_binscan_small_kernel(uint32_t* __restrict__ hist, int n_tiles,
                                                                   uint32_t* __restrict__ bin_total) {
    hist += (int64_t)blockIdx.x * RADIX * n_tiles;
    uint32_t* row = hist + (int64_t)threadIdx.x * n_tiles;
    uint32_t run = 0;
    for (int i = 0; i < n_tiles; ++i) {
        const uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
    bin_total[(int64_t)blockIdx.x * RADIX + threadIdx.x] = run;
}

template <int ROUNDS>
__global__ void __launch_bounds__(THREADS) radix_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,
                                                                uint64_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                                int n_tiles, const uint32_t* __restrict__ hist,
                                                                const uint32_t* __restrict__ bin_total) {
    constexpr int TILE = THREADS * ROUNDS;
    constexpr int WAVE_KEYS = 64 * ROUNDS;
    __shared__ uint64_t stage_k[TILE];
    __shared__ uint32_t stage_v[TILE];
    __shared__ uint32_t off[WAVES][RADIX];     // per-wave histogram, then running LDS position per digit
    __shared__ uint32_t digit_start[RADIX];    // start of digit d's run inside the staged tile
    __shared__ uint32_t gbase[RADIX];          // global position of the first key of this tile with digit d
    __shared__ uint32_t wsum[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
    {
        const int64_t seg_base = (int64_t)blockIdx.y * n;
        keys_in += seg_base; vals_in += seg_base; keys_out += seg_base; vals_out += seg_base;
        hist += (int64_t)blockIdx.y * RADIX * n_tiles;
        bin_total += (int64_t)blockIdx.y * RADIX;
    }
    const int64_t beg = (int64_t)tile * TILE;
    const int64_t end = min(n, beg + TILE);
    const int64_t wbeg = beg + (int64_t)w * WAVE_KEYS;

    // ---- this wave's contiguous quarter of the tile -> registers (one memory round trip)
    uint64_t rk[ROUNDS];
    uint32_t rv[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wbeg + r * 64 + lane;
        rk[r] = 0; rv[r] = 0;
        if (i < end) { rk[r] = keys_in[i]; rv[r] = vals_in[i]; }
    }
    // digit bases over the whole array: exclusive scan of the 256 bin totals (one bin per thread)
    const uint32_t bt = bin_total[tid];
    const uint32_t tile_off = hist[(int64_t)tid * n_tiles + tile];
#pragma unroll
    for (int q = 0; q < WAVES; ++q) off[q][tid] = 0;
    __syncthreads();
    // ---- per-wave histograms
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wbeg + r * 64 + lane;
        if (i < end) atomicAdd(&off[w][(uint32_t)(rk[r] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    {
        // thread d: digit d.  (a) global base = scan(bin_total)[d] + offset of this tile within the bin;
        // (b) tile-local run start = exclusive scan over digits of the tile's digit counts;
        // (c) per-wave starting positions inside the run.
        uint32_t c[WAVES], tot = 0;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) { c[q] = off[q][tid]; tot += c[q]; }
        uint32_t x = bt, y = tot;     // two inclusive block scans at once
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t xs = __shfl_up(x, o), ys = __shfl_up(y, o);
            if (lane >= o) { x += xs; y += ys; }
        }
        __shared__ uint32_t wsum2[WAVES];
        if (lane == 63) { wsum[w] = x; wsum2[w] = y; }
        __syncthreads();
        uint32_t px = 0, py = 0;
        for (int q = 0; q < w; ++q) { px += wsum[q]; py += wsum2[q]; }
        const uint32_t base_excl = px + x - bt;       // exclusive scan of bin totals
        const uint32_t start_excl = py + y - tot;     // exclusive scan of the tile's digit counts
        gbase[tid] = base_excl + tile_off;
        digit_start[tid] = start_excl;
        uint32_t run = start_excl;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) { off[q][tid] = run; run += c[q]; }
    }
    __syncthreads();
    // ---- each wave ranks its own keys round by round and stages them in digit-sorted order
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wbeg + r * 64 + lane;
        if (wbeg + r * 64 >= end) break;        // wave-uniform
        const bool active = i < end;
        const uint32_t d = (uint32_t)(rk[r] >> shift) & (RADIX - 1);
        uint64_t peers = __ballot(active);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        uint32_t pos = 0;
        if (active) pos = off[w][d] + (uint32_t)__popcll(peers & lt_mask);
        __builtin_amdgcn_wave_barrier();          // every lane has read off[w][*] before the leaders bump it
        if (active && (peers & lt_mask) == 0) off[w][d] += (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (active) { stage_k[pos] = rk[r]; stage_v[pos] = rv[r]; }
    }
    __syncthreads();
    // ---- write the staged tile: consecutive lanes -> consecutive addresses inside a digit run
    const int count = (int)(end - beg);
    for (int i = tid; i < count; i += THREADS) {
        const uint64_t k = stage_k[i];
        const uint32_t d = (uint32_t)(k >> shift) & (RADIX - 1);
        const uint32_t g = gbase[d] + ((uint32_t)i - digit_start[d]);
        keys_out[g] = k;
        vals_out[g] = stage_v[i];
    }
}

}  // namespace

template <int ROUNDS>
static int radix_sort_passes(sdice_ctx* ctx, int64_t n, int64_t segs, const uint64_t* d_keys_in, const uint32_t* d_vals_in,
                             uint64_t* d_keys_out, uint32_t* d_vals_out, uint64_t* d_keys_tmp, uint32_t* d_vals_tmp,
                             const int* shifts, int np) {
    constexpr int TILE = THREADS * ROUNDS;
    const int64_t n_tiles = sd_ceil_div(n, TILE);
    uint32_t* hist = (uint32_t*)ctx->arena.alloc((size_t)segs * RADIX * n_tiles * 4);
    uint32_t* bin_total = (uint32_t*)ctx->arena.alloc((size_t)segs * RADIX * 4);
    if (!hist || !bin_total) return SDICE_ERR_NOMEM;
    // ping-pong so that the last pass lands in *_out (in, out and tmp must be distinct buffers)
    const uint64_t* kin = d_keys_in;
    const uint32_t* vin = d_vals_in;
    for (int p = 0; p < np; ++p) {
        const bool to_out = ((np - 1 - p) % 2) == 0;
        uint64_t* kout = to_out ? d_keys_out : d_keys_tmp;
        uint32_t* vout = to_out ? d_vals_out : d_vals_tmp;
        SD_LAUNCH(ctx, "radix_hist_kernel", (radix_hist_kernel<ROUNDS>), dim3((unsigned)n_tiles, (unsigned)segs),
                  dim3(THREADS), 0, kin, n, shifts[p], (int)n_tiles, hist);
        if (n_tiles <= 64 && segs >= 64) {
            SD_LAUNCH(ctx, "radix_binscan_small_kernel", radix_binscan_small_kernel, dim3((unsigned)segs), dim3(RADIX), 0,
                      hist, (int)n_tiles, bin_total);
        } else {
            SD_LAUNCH(ctx, "radix_binscan_kernel", radix_binscan_kernel, dim3(RADIX, (unsigned)segs), dim3(256), 0, hist,
                      (int)n_tiles, bin_total);
        }
        SD_LAUNCH(ctx, "radix_scatter_kernel", (radix_scatter_kernel<ROUNDS>), dim3((unsigned)n_tiles, (unsigned)segs),
                  dim3(THREADS), 0, kin, vin, kout, vout, n, shifts[p], (int)n_tiles, hist, bin_total);
        kin = kout;
        vin = vout;
    }
    return SDICE_OK;
}

int sd_radix_sort_pairs_segmented(sdice_ctx* ctx, int64_t n, int64_t segs, const uint64_t* d_keys_in,
                                  const uint32_t* d_vals_in, uint64_t* d_keys_out, uint32_t* d_vals_out,
                                  uint64_t* d_keys_tmp, uint32_t* d_vals_tmp, uint64_t bit_mask) {
    if (n <= 0 || segs <= 0) return SDICE_OK;
    if (n >= ((int64_t)1 << 32) || segs > 65535) {
        sdice_set_error("radix sort: segment longer than 2^32 keys or more than 65535 segments");
        return SDICE_ERR_ARG;
    }
    // digits that actually vary
    int shifts[8], np = 0;
    for (int d = 0; d < 8; ++d)
        if ((bit_mask >> (8 * d)) & 0xffull) shifts[np++] = 8 * d;
    if (np == 0) {
        if (d_keys_in != d_keys_out) {
            SD_HIP(hipMemcpyAsync(d_keys_out, d_keys_in, (size_t)(n * segs) * 8, hipMemcpyDeviceToDevice, ctx->stream));
            SD_HIP(hipMemcpyAsync(d_vals_out, d_vals_in, (size_t)(n * segs) * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return SDICE_OK;
    }
    int64_t rounds = ctx->param("sort.rounds", 0);
    if (rounds != 4 && rounds != 12) rounds = 12;   // measured: 4 is no faster at 1M keys and 16% slower at 5M
    if (rounds == 4)
        return radix_sort_passes<4>(ctx, n, segs, d_keys_in, d_vals_in, d_keys_out, d_vals_out, d_keys_tmp, d_vals_tmp,
                                    shifts, np);
    return radix_sort_passes<12>(ctx, n, segs, d_keys_in, d_vals_in, d_keys_out, d_vals_out, d_keys_tmp, d_vals_tmp, shifts,
                                 np);
}

int sd_radix_sort_pairs(sdice_ctx* ctx, int64_t n, const uint64_t* d_keys_in, const uint32_t* d_vals_in,
                        uint64_t* d_keys_out, uint32_t* d_vals_out, uint64_t* d_keys_tmp, uint32_t* d_vals_tmp,
                        uint64_t bit_mask) {
    return sd_radix_sort_pairs_segmented(ctx, n, 1, d_keys_in, d_vals_in, d_keys_out, d_vals_out, d_keys_tmp, d_vals_tmp,
                                         bit_mask);
}
