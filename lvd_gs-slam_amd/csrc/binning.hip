// Grouping of the (Gaussian, tile) pairs by tile: by counting (default), or pair emission + tile ranges for the radix path.
#include "common.hpp"
#include "device_utils.hpp"

namespace lvdgs {

namespace {

// One lane per Gaussian, in id order.  A Gaussian's pairs occupy [slot_base[i], slot_base[i] + tiles) of the
// unsorted pair list, the kept tiles of its rectangle (common.hpp: rect_keeps) in row-major order.
__global__ void __launch_bounds__(256) emit_pairs_kernel(int N, int gx, const uint32_t *__restrict__ slot_base,
                                                         const uint32_t *__restrict__ tiles_touched, const uint4 *__restrict__ rect,
                                                         uint32_t *__restrict__ tile_keys, uint32_t *__restrict__ ids,
                                                         uint32_t capacity) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (uint32_t)N) return;
    if (tiles_touched[id] == 0) return;
    const uint4 r = rect[id];
    const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
    const int area = (x1 - x0) * (y1 - y0);
    uint32_t o = slot_base[id];
    int k = 0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++, k++) {
            if (!rect_keeps(r, k, area)) continue;
            if (o < capacity) {  // pairs beyond the caller's capacity are dropped (the caller is told and re-runs)
                tile_keys[o] = (uint32_t)(y * gx + x);
                ids[o] = id;
            }
            o++;
        }
}

// ---- grouping by tile without a sort --------------------------------------------------------------------------
// A workgroup owns a chunk of consecutive Gaussians (1024 x PER, see group_per_thread_for) and keeps one counter per tile in LDS.
//   count:    counter[tile] += 1 for every (Gaussian, tile) pair of the chunk        -> hist[chunk][tile]
//   colscan:  per tile, exclusive prefix of hist over the chunks, and the tile total
//   tilescan: exclusive scan of the totals                                           -> ranges[tile]
//   scatter:  counter[tile] = ranges[tile].begin + hist[chunk][tile]; every pair takes the next slot of its tile
//             with one returning LDS atomic and writes its Gaussian id there.
// Rectangles larger than a wave's worth of tiles are walked by the whole wave, so one screen-filling Gaussian does
// not serialise thousands of atomics on one lane.
constexpr int GROUP_THREADS = 1024;
// Gaussians per workgroup = 1024 x PER.  Few, large chunks keep the [chunk][tile] count matrix and its scan small (2 M
// Gaussians: 489 chunks of 4096); many, small ones spread the counting and scattering over the chip (200 k Gaussians
// are 49 chunks of 4096 on 256 CUs, 98 of 2048).  Measured (same box): 100k / 640x480 0.2235 -> 0.2088 ms per tracking
// iteration with 2048, KITTI geometry 0.2881 -> 0.2795, config 3 0.6247 -> 0.6230, 2 M / 1920x1280 1.49 -> 1.52.
__host__ __device__ constexpr int group_per_thread_for(int N) { return N <= (1 << 20) ? 2 : 4; }
constexpr int GROUP_MAX_TILES = 16384;  // 64 KiB of LDS counters
constexpr int GROUP_BIG_RECT = 64;

template <int PER, typename F>
__device__ __forceinline__ void for_each_pair_of_chunk(int N, int gx, const uint4 *__restrict__ rect, F visit) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = blockIdx.x * (GROUP_THREADS * PER) + k * GROUP_THREADS + (int)threadIdx.x;
        const uint4 r = i < N ? rect[i] : make_uint4(0u, 0u, 0u, 0u);
        const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
        const int w = x1 - x0, area = w * (y1 - y0);
        if (area > 0 && area <= GROUP_BIG_RECT) {
            // only the tiles the Gaussian can reach (common.hpp: rect_keeps); bit t of the mask is tile t of the rectangle
            uint64_t m = (uint64_t)r.z | ((uint64_t)r.w << 32);
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++, m >>= 1)
                    if (m & 1ull) visit(y * gx + x, (uint32_t)i);
        }
        uint64_t big = __ballot(area > GROUP_BIG_RECT);
        while (big) {
            const int src = __builtin_ctzll(big);
            big &= big - 1;
            const int bx0 = __shfl(x0, src, 64), by0 = __shfl(y0, src, 64), bw = __shfl(w, src, 64), barea = __shfl(area, src, 64);
            const uint32_t bi = (uint32_t)__shfl(i, src, 64);
            for (int t = lane; t < barea; t += 64) visit((by0 + t / bw) * gx + bx0 + t % bw, bi);
        }
    }
}
static_assert(GROUP_BIG_RECT == RECT_MASK_TILES, "rectangles walked by the whole wave are the ones without a tile mask");

// Exclusive scan of one value per thread over a workgroup of 1024 threads (wave shifts, then the 16 wave totals by wave
// 0: two barriers); *total = sum over the workgroup.  s_scan: 33 words.
__device__ __forceinline__ uint32_t scan_1024(uint32_t v, uint32_t *s_scan, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane >= off) inc += x;
    }
    if (lane == 63) s_scan[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < 16 ? s_scan[lane] : 0u;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t x = (uint32_t)__shfl_up((int)w, off, 64);
            if (lane >= off) w += x;
        }
        if (lane < 16) s_scan[16 + lane] = w;   // inclusive over the waves
    }
    __syncthreads();
    *total = s_scan[31];
    return inc - v + (wave ? s_scan[16 + wave - 1] : 0u);
}

// (Also clears what the later kernels of the frame accumulate into: n_touched and the tile-sort queue -- two memset
// launches less -- and, when `tt` is given, makes slot_base[i] = exclusive scan of tiles_touched and the pair total from
// the sums preprocess_fwd left per 256 Gaussians: the launch of the separate slot scan less.  Every workgroup adds up the
// block sums in front of its chunk -- at most a few thousand values -- and scans its own Gaussians.)
template <int PER>
__global__ void __launch_bounds__(GROUP_THREADS) count_pairs_kernel(int N, int gx, int T, const uint4 *__restrict__ rect,
                                                                   uint32_t *__restrict__ hist, int32_t *__restrict__ n_touched,
                                                                   uint32_t *__restrict__ queue_counts,
                                                                   const uint32_t *__restrict__ tt, const uint32_t *__restrict__ blocksums,
                                                                   uint32_t *__restrict__ slot_base, uint32_t *__restrict__ total_out) {
    static_assert(GROUP_THREADS == 1024, "scan_1024");
    constexpr int GROUP_PER_THREAD = PER, GROUP_CHUNK = GROUP_THREADS * PER;
    extern __shared__ uint32_t s_tile[];
    __shared__ uint32_t s_scan[33];
    for (int t = threadIdx.x; t < T; t += GROUP_THREADS) s_tile[t] = 0u;
#pragma unroll
    for (int k = 0; k < GROUP_PER_THREAD; k++) {
        const int i = blockIdx.x * GROUP_CHUNK + k * GROUP_THREADS + (int)threadIdx.x;
        if (i < N) n_touched[i] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) queue_counts[threadIdx.x] = 0u;
    if (tt) {
        uint32_t before = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x * (GROUP_CHUNK / 256); b += GROUP_THREADS) before += blocksums[b];
        uint32_t prefix;
        scan_1024(before, s_scan, &prefix);  // only the total is of interest
        __syncthreads();                     // s_scan is used again
        const int base = blockIdx.x * GROUP_CHUNK + (int)threadIdx.x * GROUP_PER_THREAD;
        uint32_t v[GROUP_PER_THREAD], mine = 0;
#pragma unroll
        for (int k = 0; k < GROUP_PER_THREAD; k++) { v[k] = base + k < N ? tt[base + k] : 0u; mine += v[k]; }
        uint32_t total;
        uint32_t run = scan_1024(mine, s_scan, &total) + prefix;
#pragma unroll
        for (int k = 0; k < GROUP_PER_THREAD; k++) {
            if (base + k < N) slot_base[base + k] = run;
            run += v[k];
        }
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = prefix + total;
    }
    __syncthreads();
    for_each_pair_of_chunk<PER>(N, gx, rect, [&](int tile, uint32_t) { atomicAdd(&s_tile[tile], 1u); });
    __syncthreads();
    uint32_t *row = hist + (size_t)blockIdx.x * T;
    for (int t = threadIdx.x; t < T; t += GROUP_THREADS) row[t] = s_tile[t];
}

// Exclusive prefix of hist over the chunks, per tile.  A workgroup covers 16 tiles; its 16 thread groups split the
// chunks between them (loads of one group are independent and 64-byte coalesced over the 16 tiles), meet in LDS
// for the group offsets, then write the prefixes.
// (16 tiles x 16 chunk groups: 510 workgroups at 1080p; 32 x 8 measured 2.5 us slower at config 3, 8 x 32 4.5 us)
constexpr int COLSCAN_TILES = 16, COLSCAN_GROUPS = 16;
__global__ void __launch_bounds__(COLSCAN_TILES * COLSCAN_GROUPS) group_colscan_kernel(int T, int nchunks, uint32_t *__restrict__ hist,
                                                                                      uint32_t *__restrict__ totals) {
    __shared__ uint32_t s_part[COLSCAN_GROUPS][COLSCAN_TILES];
    const int tl = threadIdx.x % COLSCAN_TILES, cg = threadIdx.x / COLSCAN_TILES;
    const int t = blockIdx.x * COLSCAN_TILES + tl;
    const int per = (nchunks + COLSCAN_GROUPS - 1) / COLSCAN_GROUPS;
    const int c0 = cg * per, c1 = min(nchunks, c0 + per);
    uint32_t sum = 0;
    if (t < T)
        for (int c = c0; c < c1; c++) sum += hist[(size_t)c * T + t];
    s_part[cg][tl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int g = 0; g < COLSCAN_GROUPS; g++) {
        const uint32_t v = s_part[g][tl];
        run += g < cg ? v : 0u;
        total += v;
    }
    if (t < T) {
        for (int c = c0; c < c1; c++) {
            const uint32_t v = hist[(size_t)c * T + t];
            hist[(size_t)c * T + t] = run;
            run += v;
        }
        if (cg == 0) totals[t] = total;
    }
}

// one workgroup: ranges[t] = [sum of totals before t, + totals[t]), clamped to the pair capacity
// tile_order (optional, T entries): the tiles by descending list length (in steps of 8 entries; ties in arrival order --
// it only decides which workgroup of a blend kernel takes which tile, never a result).
__global__ void __launch_bounds__(1024) group_tilescan_kernel(int T, const uint32_t *__restrict__ totals, uint32_t capacity,
                                                              uint2 *__restrict__ ranges, uint32_t *__restrict__ tile_order,
                                                              uint32_t *__restrict__ order_valid) {
    __shared__ uint32_t s_scan[1024];
    constexpr int PER = GROUP_MAX_TILES / 1024;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int t = (int)threadIdx.x * PER + k;
        v[k] = t < T ? totals[t] : 0u;
        sum += v[k];
    }
    // inclusive scan of the 1024 thread sums: inside every wave by lane shifts, then the 16 wave totals by wave 0
    // (two barriers; the log-step scan over all 1024 threads it replaces needed twenty)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane >= off) inc += x;
    }
    if (lane == 63) s_scan[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < 16 ? s_scan[lane] : 0u;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t x = (uint32_t)__shfl_up((int)w, off, 64);
            if (lane >= off) w += x;
        }
        if (lane < 16) s_scan[16 + lane] = w;   // inclusive over the waves
    }
    __syncthreads();
    inc += wave ? s_scan[16 + wave - 1] : 0u;
    uint32_t run = inc - sum;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int t = (int)threadIdx.x * PER + k;
        // empty tiles keep (0, 0), as after the radix path's memset
        if (t < T) ranges[t] = v[k] ? make_uint2(min(run, capacity), min(run + v[k], capacity)) : make_uint2(0u, 0u);
        run += v[k];
    }
    if (!tile_order) return;
    // counting sort of the tiles by bucket 1023 - min(length / 8, 1023): bucket 0 holds the longest lists
    auto bucket = [](uint32_t len) { return 1023u - min(len >> 3, 1023u); };
    __syncthreads();
    s_scan[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++)
        if ((int)threadIdx.x * PER + k < T) atomicAdd(&s_scan[bucket(v[k])], 1u);
    __syncthreads();
    const uint32_t mine = s_scan[threadIdx.x];
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += x;
    }
    __shared__ uint32_t s_wave[32];
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < 16 ? s_wave[lane] : 0u;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t x = (uint32_t)__shfl_up((int)w, off, 64);
            if (lane >= off) w += x;
        }
        if (lane < 16) s_wave[16 + lane] = w;
    }
    __syncthreads();
    s_scan[threadIdx.x] = incl - mine + (wave ? s_wave[16 + wave - 1] : 0u);   // first position of this bucket
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int t = (int)threadIdx.x * PER + k;
        if (t < T) tile_order[atomicAdd(&s_scan[bucket(v[k])], 1u)] = (uint32_t)t;
    }
    if (threadIdx.x == 0) *order_valid = 1u;   // (cleared with the tile-sort queue at the start of every frame)
}

template <int PER>
__global__ void __launch_bounds__(GROUP_THREADS) scatter_pairs_kernel(int N, int gx, int T, const uint4 *__restrict__ rect,
                                                                     const uint32_t *__restrict__ hist,
                                                                     const uint2 *__restrict__ ranges, uint32_t capacity,
                                                                     const uint32_t *__restrict__ depth_bits,
                                                                     unsigned long long *__restrict__ keys64) {
    extern __shared__ uint32_t s_tile[];
    const uint32_t *row = hist + (size_t)blockIdx.x * T;
    for (int t = threadIdx.x; t < T; t += GROUP_THREADS) s_tile[t] = ranges[t].x + row[t];  // (empty tiles are never visited)
    __syncthreads();
    for_each_pair_of_chunk<PER>(N, gx, rect, [&](int tile, uint32_t id) {
        const uint32_t pos = atomicAdd(&s_tile[tile], 1u);
        // the tile sort's key, so that it need not gather depths; beyond the caller's capacity: dropped (the caller
        // is told and re-runs)
        if (pos < capacity) keys64[pos] = ((unsigned long long)depth_bits[id] << 32) | (unsigned long long)id;
    });
}

__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint32_t *__restrict__ tile_keys, int64_t D_cap,
                                                          const uint32_t *__restrict__ D_dev, uint2 *__restrict__ ranges) {
    const int64_t D = D_dev ? min((int64_t)*D_dev, D_cap) : D_cap;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= D) return;
    const uint32_t t = tile_keys[k];
    if (k == 0 || tile_keys[k - 1] != t) ranges[t].x = (uint32_t)k;
    if (k == D - 1 || tile_keys[k + 1] != t) ranges[t].y = (uint32_t)(k + 1);
}

}  // namespace

int group_max_tiles() { return GROUP_MAX_TILES; }
size_t group_hist_entries(int N, int num_tiles) {
    return (size_t)cdiv(N > 0 ? N : 1, GROUP_THREADS * group_per_thread_for(N)) * (size_t)num_tiles;
}

int launch_group_count(const lvdgs_args &a, const GeomView &g, const ImageView &im, const RenderScratch &w, bool slot_scan, hipStream_t s) {
    const int N = a.num_gaussians;
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE, T = gx * gy;
    if (N == 0 || T == 0) return LVDGS_OK;
    const int per = group_per_thread_for(N), nchunks = cdiv(N, GROUP_THREADS * per);
    const size_t lds = (size_t)T * sizeof(uint32_t);
    static unsigned char count_done2[16], count_done4[16];
    const uint32_t *tt = slot_scan ? (const uint32_t *)g.tiles_touched : nullptr;
    ProfScope ps("group_count", s);
    if (per == 2) {
        if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&count_pairs_kernel<2>), GROUP_MAX_TILES * 4, count_done2)) return e;
        hipLaunchKernelGGL(count_pairs_kernel<2>, dim3(nchunks), dim3(GROUP_THREADS), lds, s, N, gx, T, (const uint4 *)g.rect, w.group_hist,
                           a.n_touched, im.long_count, tt, (const uint32_t *)w.blocksums, g.slot_base, g.total);
    } else {
        if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&count_pairs_kernel<4>), GROUP_MAX_TILES * 4, count_done4)) return e;
        hipLaunchKernelGGL(count_pairs_kernel<4>, dim3(nchunks), dim3(GROUP_THREADS), lds, s, N, gx, T, (const uint4 *)g.rect, w.group_hist,
                           a.n_touched, im.long_count, tt, (const uint32_t *)w.blocksums, g.slot_base, g.total);
    }
    LVDGS_LAUNCH_CHECK("group_count", a.debug, s);
    return LVDGS_OK;
}

int launch_group_scatter(const lvdgs_args &a, const GeomView &g, const ImageView &im, const RenderScratch &w, unsigned long long *keys64,
                         int64_t capacity, hipStream_t s) {
    const int N = a.num_gaussians;
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE, T = gx * gy;
    if (N == 0 || T == 0) return LVDGS_OK;
    const int per = group_per_thread_for(N), nchunks = cdiv(N, GROUP_THREADS * per);
    const size_t lds = (size_t)T * sizeof(uint32_t);
    static unsigned char scatter_done2[16], scatter_done4[16];
    if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&scatter_pairs_kernel<2>), GROUP_MAX_TILES * 4, scatter_done2)) return e;
    if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&scatter_pairs_kernel<4>), GROUP_MAX_TILES * 4, scatter_done4)) return e;
    {
        ProfScope ps("group_scan", s);
        hipLaunchKernelGGL(group_colscan_kernel, dim3(cdiv(T, COLSCAN_TILES)), dim3(COLSCAN_TILES * COLSCAN_GROUPS), 0, s, T, nchunks, w.group_hist,
                           w.group_totals);
        hipLaunchKernelGGL(group_tilescan_kernel, dim3(1), dim3(1024), 0, s, T, (const uint32_t *)w.group_totals, (uint32_t)capacity,
                           im.ranges, tile_order_in_use(T) ? im.long_tiles + T : nullptr, im.long_count + 1);
        LVDGS_LAUNCH_CHECK("group_scan", a.debug, s);
    }
    {
        ProfScope ps("group_scatter", s);
        if (per == 2)
            hipLaunchKernelGGL(scatter_pairs_kernel<2>, dim3(nchunks), dim3(GROUP_THREADS), lds, s, N, gx, T, (const uint4 *)g.rect,
                               (const uint32_t *)w.group_hist, (const uint2 *)im.ranges, (uint32_t)capacity,
                               (const uint32_t *)g.depth_bits, keys64);
        else
            hipLaunchKernelGGL(scatter_pairs_kernel<4>, dim3(nchunks), dim3(GROUP_THREADS), lds, s, N, gx, T, (const uint4 *)g.rect,
                               (const uint32_t *)w.group_hist, (const uint2 *)im.ranges, (uint32_t)capacity,
                               (const uint32_t *)g.depth_bits, keys64);
        LVDGS_LAUNCH_CHECK("group_scatter", a.debug, s);
    }
    return LVDGS_OK;
}

int launch_emit_pairs(const lvdgs_args &a, const GeomView &g, uint32_t *tile_keys, uint32_t *ids, int64_t capacity, hipStream_t s) {
    const int N = a.num_gaussians;
    if (N == 0) return LVDGS_OK;
    const int gx = (a.image_width + TILE - 1) / TILE;
    ProfScope ps("emit_pairs", s);
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, N, gx, g.slot_base, g.tiles_touched,
                       (const uint4 *)g.rect, tile_keys, ids, (uint32_t)capacity);
    LVDGS_LAUNCH_CHECK("emit_pairs", a.debug, s);
    return LVDGS_OK;
}

int launch_tile_ranges(const uint32_t *tile_keys, int64_t D, const uint32_t *D_dev, const ImageView &im, int num_tiles, int dbg,
                       hipStream_t s) {
    // empty tiles keep (0, 0); the counter of over-long segments sits right behind the ranges and is cleared with them
    const size_t bytes = (size_t)((char *)(im.long_count + 64) - (char *)im.ranges);
    if (int e = check_hip(hipMemsetAsync(im.ranges, 0, bytes, s), "memset ranges")) return e;
    if (D == 0) return LVDGS_OK;
    ProfScope ps("tile_ranges", s);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(cdiv(D, 256)), dim3(256), 0, s, tile_keys, D, D_dev, im.ranges);
    LVDGS_LAUNCH_CHECK("tile_ranges", dbg, s);
    return LVDGS_OK;
}

}  // namespace lvdgs
