// Grouping of the (Gaussian, tile) pairs by tile: by counting (default), or pair emission + tile ranges for the radix path.
#include <stdlib.h>

#include "common.hpp"
#include "binning.hpp"
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
    const uint32_t o0 = slot_base[id];
    int k = 0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++, k++) {
            if (!rect_keeps(r, k, area)) continue;
            const uint32_t o = o0 + rect_rank(r, k, area);
            if (o < capacity) {  // pairs beyond the caller's capacity are dropped (the caller is told and re-runs)
                tile_keys[o] = (uint32_t)(y * gx + x);
                ids[o] = id;
            }
        }
}

// ---- grouping by tile without a sort --------------------------------------------------------------------------
// A workgroup owns a chunk of consecutive Gaussians (1024 x PER, binning.hpp) and keeps one counter per tile in LDS.
//   count:    counter[tile] += 1 for every (Gaussian, tile) pair of the chunk        -> hist[chunk][tile]
//             (lvdgs_forward: done by the projection kernel itself, preprocess.hip: preprocess_count_kernel)
//   colscan:  per tile, exclusive prefix of hist over the chunks, and the tile total
//   tilescan: exclusive scan of the totals                                           -> ranges[tile], the pair count,
//             the queue of over-long segments for the tile sort, the tiles by list length (small grids);
//             on small frames the last workgroup of the scatter's launch (LVDGS_SCAN_IN_SCATTER below), a kernel of its own otherwise
//   scatter:  counter[tile] = ranges[tile].begin + hist[chunk][tile]; every pair takes the next slot of its tile
//             with one returning LDS atomic and writes its sort key there; also makes slot_base (lvdgs_forward).
// Rectangles larger than a wave's worth of tiles are walked by the whole wave, so one screen-filling Gaussian does
// not serialise thousands of atomics on one lane.

// (Two-call API: clears what the later kernels of the frame accumulate into, n_touched and the tile-sort queue.)
template <int GROUP_THREADS, int OWNERS, int PER>
__global__ void __launch_bounds__(GROUP_THREADS) count_pairs_kernel(int N, int gx, int T, const uint4 *__restrict__ rect,
                                                                   uint32_t *__restrict__ hist, int32_t *__restrict__ n_touched,
                                                                   uint32_t *__restrict__ queue_counts) {
    constexpr int GROUP_CHUNK = OWNERS * PER;
    constexpr bool HELPERS = GROUP_THREADS > OWNERS;
    static_assert(!HELPERS || PER == 1, "helper waves: one Gaussian per owner thread");
    extern __shared__ uint32_t s_tile[];
    __shared__ BigRectQueue s_big;
    for (int t = threadIdx.x; t < T; t += GROUP_THREADS) s_tile[t] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 64) queue_counts[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_big.count = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = blockIdx.x * GROUP_CHUNK + k * OWNERS + (int)threadIdx.x;
        const bool mine = (int)threadIdx.x < OWNERS && i < N;
        if (mine) n_touched[i] = 0;
        const uint4 r = mine ? rect[i] : make_uint4(0u, 0u, 0u, 0u);
        if constexpr (HELPERS) for_each_pair_of_rect_wg(r, i, gx, 0u, s_big, [&](int tile, uint32_t, uint32_t) { atomicAdd(&s_tile[tile], 1u); });
        else for_each_pair_of_rect(r, i, gx, 0u, [&](int tile, uint32_t, uint32_t) { atomicAdd(&s_tile[tile], 1u); });
    }
    __syncthreads();
    uint32_t *row = hist + (size_t)blockIdx.x * T;
    for (int t = threadIdx.x; t < T; t += GROUP_THREADS) row[t] = s_tile[t];
}

// Exclusive prefix of hist over the chunks, per tile.  A workgroup covers 64 tiles, one per lane -- every load and store
// of a wave is one 256-byte run of a matrix row -- and its waves split the chunks between them: a wave holds its R rows
// in registers (all loads in flight together: the 16 dependent round trips of a row-by-row walk were what this kernel
// spent its 12 us on), the waves meet in LDS for their offsets, then write the prefixes from the registers.
// R = 0: more than 32 rows per wave (maps beyond 2 M Gaussians): the rows are read twice, eight at a time.
constexpr int COLSCAN_TILES = 64;
template <int R>
__device__ __forceinline__ void group_colscan_body(int T, int nchunks, uint32_t *__restrict__ hist, uint32_t *__restrict__ totals, bool write_prefixes = true) {
    __shared__ uint32_t s_part[16][COLSCAN_TILES];
    const int tl = threadIdx.x & 63, cg = threadIdx.x >> 6, groups = blockDim.x >> 6;
    const int t = blockIdx.x * COLSCAN_TILES + tl;
    const int per = (nchunks + groups - 1) / groups;
    const int c0 = cg * per, c1 = min(nchunks, c0 + per);
    uint32_t sum = 0;
    uint32_t v[R > 0 ? R : 1];
    if constexpr (R > 0) {
#pragma unroll
        for (int k = 0; k < R; k++) v[k] = (t < T && c0 + k < c1) ? hist[(size_t)(c0 + k) * T + t] : 0u;
#pragma unroll
        for (int k = 0; k < R; k++) sum += v[k];
    } else if (t < T) {
        for (int c = c0; c < c1; c += 8) {
            uint32_t u[8];
#pragma unroll
            for (int k = 0; k < 8; k++) u[k] = c + k < c1 ? hist[(size_t)(c + k) * T + t] : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) sum += u[k];
        }
    }
    s_part[cg][tl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (int g = 0; g < groups; g++) {
        const uint32_t x = s_part[g][tl];
        run += g < cg ? x : 0u;
        total += x;
    }
    if (t >= T) return;
    if (!write_prefixes) {   // (two-level grouping: nothing scatters into the TILE grid's segments; only the totals are wanted of it)
        if (cg == 0) totals[t] = total;
        return;
    }
    if constexpr (R > 0) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            if (c0 + k < c1) hist[(size_t)(c0 + k) * T + t] = run;
            run += v[k];
        }
    } else {
        for (int c = c0; c < c1; c += 8) {
            uint32_t u[8];
#pragma unroll
            for (int k = 0; k < 8; k++) u[k] = c + k < c1 ? hist[(size_t)(c + k) * T + t] : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (c + k < c1) hist[(size_t)(c + k) * T + t] = run;
                run += u[k];
            }
        }
    }
    if (cg == 0) totals[t] = total;
}
template <int R>
__global__ void __launch_bounds__(1024) group_colscan_kernel(int T, int nchunks, uint32_t *__restrict__ hist, uint32_t *__restrict__ totals) {
    group_colscan_body<R>(T, nchunks, hist, totals);
}
// lvdgs_forward_batch: the count matrices of several views (blockIdx.y: the view)
struct ColscanBatch { uint32_t *hist[FWD_BATCH_VIEWS], *totals[FWD_BATCH_VIEWS]; };
template <int R>
__global__ void __launch_bounds__(1024) group_colscan_batch_kernel(int T, int nchunks, ColscanBatch b) {
    group_colscan_body<R>(T, nchunks, b.hist[blockIdx.y], b.totals[blockIdx.y]);
}

// one workgroup: ranges[t] = [sum of totals before t, + totals[t]), clamped to the pair capacity; the pair count;
// the queue of segments too long for one wave to sort (tilesort.hip) -- filled here, where every length is at hand, so
// that the sort needs no pass of its own to find them;
// tile_order (optional, tiles [t_lo, t_hi)): those tiles by descending list length (in steps of 8 entries; ties in
// arrival order -- it only decides which workgroup of a blend kernel takes which tile, never a result).
template <int THREADS>
__device__ __forceinline__ void group_tilescan_body(int T, const uint32_t *__restrict__ totals, uint32_t capacity,
                                                    uint2 *__restrict__ ranges, uint32_t *__restrict__ total_out,
                                                    uint32_t long_limit, uint32_t *__restrict__ queue_count, uint32_t *__restrict__ queue,
                                                    uint32_t *__restrict__ tile_order, int t_lo, int t_hi, uint32_t *__restrict__ order_valid,
                                                    uint32_t *host_out, uint32_t host_seq) {
    static_assert(THREADS == 1024 || THREADS == 512, "the grouping kernels' workgroup sizes (the scan rides in the scatter's launch)");
    __shared__ uint32_t s_scan[1024];
    __shared__ uint32_t s_q, s_longest, s_total;
    constexpr int PER = GROUP_MAX_TILES / THREADS;   // 16 or 32 tiles per thread: 16-byte loads, 16-byte stores
    constexpr int WAVES = THREADS / 64;
    uint32_t v[PER], sum = 0;
    if (threadIdx.x == 0) { s_q = 0u; s_longest = 0u; }
    {
        const int t0 = (int)threadIdx.x * PER;
        const uint4 *src = reinterpret_cast<const uint4 *>(totals + t0);   // (256-byte aligned; the tail is guarded below)
#pragma unroll
        for (int q = 0; q < PER / 4; q++) {
            uint4 x = make_uint4(0u, 0u, 0u, 0u);
            if (t0 + 4 * q + 3 < T) x = src[q];
            else {
                if (t0 + 4 * q + 0 < T) x.x = totals[t0 + 4 * q + 0];
                if (t0 + 4 * q + 1 < T) x.y = totals[t0 + 4 * q + 1];
                if (t0 + 4 * q + 2 < T) x.z = totals[t0 + 4 * q + 2];
            }
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int k = 0; k < PER; k++) sum += v[k];
    }
    // inclusive scan of the thread sums: inside every wave by lane shifts, then the up to 16 wave totals by wave 0
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
        uint32_t w = lane < WAVES ? s_scan[lane] : 0u;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t x = (uint32_t)__shfl_up((int)w, off, 64);
            if (lane >= off) w += x;
        }
        if (lane < 16) s_scan[16 + lane] = w;   // inclusive over the waves
    }
    __syncthreads();
    inc += wave ? s_scan[16 + wave - 1] : 0u;
    if (threadIdx.x == THREADS - 1) {
        s_total = inc;
        if (total_out) total_out[0] = inc;   // the frame's pair count (before clamping)
    }
    uint32_t run = inc - sum;
    uint32_t longest = 0;
#pragma unroll
    for (int k = 0; k < PER; k += 2) {
        const int t = (int)threadIdx.x * PER + k;
        // empty tiles keep (0, 0), as after the radix path's memset
        const uint2 r0 = v[k] ? make_uint2(min(run, capacity), min(run + v[k], capacity)) : make_uint2(0u, 0u);
        run += v[k];
        const uint2 r1 = v[k + 1] ? make_uint2(min(run, capacity), min(run + v[k + 1], capacity)) : make_uint2(0u, 0u);
        run += v[k + 1];
        if (t + 1 < T) *reinterpret_cast<uint4 *>(ranges + t) = make_uint4(r0.x, r0.y, r1.x, r1.y);
        else if (t < T) ranges[t] = r0;
        if (r0.y - r0.x > long_limit) queue[atomicAdd(&s_q, 1u)] = (uint32_t)t;
        if (r1.y - r1.x > long_limit) queue[atomicAdd(&s_q, 1u)] = (uint32_t)(t + 1);
        longest = max(longest, max(r0.y - r0.x, r1.y - r1.x));
    }
    if (longest > long_limit) atomicMax(&s_longest, longest);
    __syncthreads();
    if (threadIdx.x == 0) {
        *queue_count = s_q;
        if (total_out) { total_out[1] = s_longest; total_out[2] = s_q; }   // longest queued segment (0: none) and the queue's length, read back with the pair count
        // The host's copy, written straight into its (pinned, device-visible) memory: the pair count and the two hints, then --
        // behind a system-scope fence -- the sequence number of the call, which is what the host spins on.  (A device-to-host
        // copy enqueued behind this kernel is a blit kernel of its own: 3.6 us on every frame's critical path.)
        if (host_out) {
            host_out[0] = s_total; host_out[1] = s_longest; host_out[2] = s_q;
            __threadfence_system();
            __hip_atomic_store(host_out + 3, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (!tile_order) return;
    // counting sort of the tiles by bucket 1023 - min(length / 8, 1023): bucket 0 holds the longest lists
    auto bucket = [](uint32_t len) { return 1023u - min(len >> 3, 1023u); };
    constexpr int BPT = 1024 / THREADS;   // buckets per thread (consecutive ones)
    __syncthreads();   // (s_scan's wave totals have been read)
#pragma unroll
    for (int q = 0; q < BPT; q++) s_scan[BPT * threadIdx.x + q] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int t = (int)threadIdx.x * PER + k;
        if (t >= t_lo && t < t_hi) atomicAdd(&s_scan[bucket(v[k])], 1u);
    }
    __syncthreads();
    uint32_t cnt[BPT], mine = 0u;
#pragma unroll
    for (int q = 0; q < BPT; q++) { cnt[q] = s_scan[BPT * threadIdx.x + q]; mine += cnt[q]; }
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
        uint32_t w = lane < WAVES ? s_wave[lane] : 0u;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t x = (uint32_t)__shfl_up((int)w, off, 64);
            if (lane >= off) w += x;
        }
        if (lane < 16) s_wave[16 + lane] = w;
    }
    __syncthreads();
    {
        uint32_t first = incl - mine + (wave ? s_wave[16 + wave - 1] : 0u);   // first position of this thread's first bucket
#pragma unroll
        for (int q = 0; q < BPT; q++) { s_scan[BPT * threadIdx.x + q] = first; first += cnt[q]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int t = (int)threadIdx.x * PER + k;
        if (t >= t_lo && t < t_hi) tile_order[atomicAdd(&s_scan[bucket(v[k])], 1u)] = (uint32_t)t;
    }
    if (threadIdx.x == 0) *order_valid = 1u;   // (cleared with the tile-sort queue at the start of every frame)
}
__global__ void __launch_bounds__(1024) group_tilescan_kernel(int T, const uint32_t *__restrict__ totals, uint32_t capacity,
                                                              uint2 *__restrict__ ranges, uint32_t *__restrict__ total_out,
                                                              uint32_t long_limit, uint32_t *__restrict__ queue_count, uint32_t *__restrict__ queue,
                                                              uint32_t *__restrict__ tile_order, int t_lo, int t_hi, uint32_t *__restrict__ order_valid,
                                                              uint32_t *host_out, uint32_t host_seq) {
    group_tilescan_body<1024>(T, totals, capacity, ranges, total_out, long_limit, queue_count, queue, tile_order, t_lo, t_hi, order_valid, host_out, host_seq);
}
// lvdgs_forward_batch: one workgroup per view (blockIdx.x); view k's pair count and hints go to host_out + 4 k
struct TilescanView { const uint32_t *totals; uint32_t capacity; uint2 *ranges; uint32_t *total_out, *queue_count, *queue, *tile_order, *order_valid; };
struct TilescanBatch { TilescanView v[FWD_BATCH_VIEWS]; };
__global__ void __launch_bounds__(1024) group_tilescan_batch_kernel(int T, TilescanBatch b, uint32_t long_limit, int t_lo, int t_hi, uint32_t *host_out,
                                                                    uint32_t host_seq) {
    const TilescanView &v = b.v[blockIdx.x];
    group_tilescan_body<1024>(T, v.totals, v.capacity, v.ranges, v.total_out, long_limit, v.queue_count, v.queue, v.tile_order, t_lo, t_hi, v.order_valid,
                              host_out + 4 * blockIdx.x, host_seq);
}

// Two-level grouping: the tile grid's and the super-tile grid's scans side by side -- blockIdx.y picks the grid (column scan), blockIdx.x
// the grid (range scan: one workgroup each) -- instead of two more launches on every frame's critical path.
template <int R>
__global__ void __launch_bounds__(1024) group_colscan_pair_kernel(int T0, int T1, int nchunks, uint32_t *__restrict__ hist0, uint32_t *__restrict__ totals0,
                                                                  uint32_t *__restrict__ hist1, uint32_t *__restrict__ totals1) {
    const int T = blockIdx.y ? T1 : T0;
    if ((int)blockIdx.x * COLSCAN_TILES >= T) return;
    group_colscan_body<R>(T, nchunks, blockIdx.y ? hist1 : hist0, blockIdx.y ? totals1 : totals0, blockIdx.y != 0);   // (grid 0: the tiles' -- totals only)
}
struct TilescanPairView { int T; const uint32_t *totals; uint2 *ranges; uint32_t *total_out, *queue_count, *queue, *tile_order; int t_lo, t_hi; uint32_t *order_valid, *host_out; uint32_t host_seq; };
__global__ void __launch_bounds__(1024) group_tilescan_pair_kernel(TilescanPairView a, TilescanPairView b, uint32_t capacity, uint32_t long_limit) {
    const TilescanPairView &v = blockIdx.x ? b : a;
    group_tilescan_body<1024>(v.T, v.totals, capacity, v.ranges, v.total_out, long_limit, v.queue_count, v.queue, v.tile_order, v.t_lo, v.t_hi, v.order_valid,
                              v.host_out, v.host_seq);
}

// lvdgs_forward_batch with two-level grouping: both grids of every view (blockIdx.y = 2 * view + grid; blockIdx.x likewise for the range scan)
struct ColscanPairBatch { uint32_t *hist[2 * FWD_BATCH_VIEWS], *totals[2 * FWD_BATCH_VIEWS]; };
template <int R>
__global__ void __launch_bounds__(1024) group_colscan_pair_batch_kernel(int T0, int T1, int nchunks, ColscanPairBatch b) {
    const int grid = blockIdx.y & 1, T = grid ? T1 : T0;
    if ((int)blockIdx.x * COLSCAN_TILES >= T) return;
    group_colscan_body<R>(T, nchunks, b.hist[blockIdx.y], b.totals[blockIdx.y], grid != 0);
}
struct TilescanPairBatch { TilescanView v[2 * FWD_BATCH_VIEWS]; };
static_assert(sizeof(TilescanPairBatch) <= 3600, "kernel arguments");
__global__ void __launch_bounds__(1024) group_tilescan_pair_batch_kernel(int T0, int T1, TilescanPairBatch b, uint32_t long_limit, int t_lo, int t_hi,
                                                                         uint32_t *host_out, uint32_t host_seq, uint32_t *host_super) {
    const TilescanView &v = b.v[blockIdx.x];
    const int grid = blockIdx.x & 1, view = blockIdx.x >> 1;
    // (the tile grid's workgroup writes the view's four pinned words -- count, hints, the sequence number the host waits for; of the
    // super grids the first view's leaves its hints -- longest queued list, queue length -- for the next call's sort launch)
    group_tilescan_body<1024>(grid ? T1 : T0, v.totals, v.capacity, v.ranges, v.total_out, long_limit, v.queue_count, v.queue, v.tile_order,
                              grid ? 0 : t_lo, grid ? T1 : t_hi, v.order_valid, grid ? (view == 0 ? host_super : nullptr) : host_out + 4 * view, grid ? 0u : host_seq);
}

// SLOT_SCAN (lvdgs_forward): also makes slot_base[i] = exclusive scan of tiles_touched in id order (the backward's
// gradient slots) from the pair totals the projection kernel left per chunk: every workgroup adds up the totals in front
// of its chunk -- at most a few hundred values -- and scans its own Gaussians.  (The launch of a separate slot scan less.)
#ifndef LVDGS_SCATTER_XCD
#define LVDGS_SCATTER_XCD 1   // A/B builds: 0 = workgroup b takes chunk b
#endif
// The tile scan rides in the scatter's launch (round 5).  A scatter workgroup needs one number per tile from it -- where the tile's
// list begins -- and has always read all T of them: it now scans the T tile totals itself (4 T bytes out of L2 instead of the 8 T of
// ranges[], one workgroup scan), and ONE extra workgroup of the launch does what the tile-scan kernel did (ranges[], the pair count
// and its copy in the host's pinned words, the tile sort's queue, the tile order) beside the others instead of in front of them: a
// launch less on the critical path of every frame small enough for it (SCAN_IN_SCATTER_RUN; KITTI geometry: 3.4 of 199 us per tracking
// iteration -- the scan kernel was 7.4 us, half of which the scatter's workgroups now spend on their own scans).
#ifndef LVDGS_SCAN_IN_SCATTER
#define LVDGS_SCAN_IN_SCATTER 1   // A/B builds: 0 = the tile scan as a launch of its own in front of the scatter
#endif
struct TileScanArgs {   // group_tilescan_body's arguments (ranges == nullptr: no tile scan in this launch)
    uint2 *ranges; uint32_t *total_out; uint32_t long_limit; uint32_t *queue_count, *queue, *tile_order; int t_lo, t_hi; uint32_t *order_valid;
    uint32_t *host_out; uint32_t host_seq;
};
template <int GROUP_THREADS, int OWNERS, int PER, bool SLOT_SCAN>
__device__ __forceinline__ void scatter_pairs_body(int chunk, int N, int gx, int T, const uint4 *__restrict__ rect,
                                                   const uint32_t *__restrict__ hist,
                                                   const uint2 *__restrict__ ranges, const uint32_t *__restrict__ totals, uint32_t capacity,
                                                   const uint32_t *__restrict__ depth_bits,
                                                   unsigned long long *__restrict__ keys64,
                                                   const uint32_t *__restrict__ tt, const uint32_t *__restrict__ chunk_sums,
                                                   uint32_t *__restrict__ slot_base, uint8_t *__restrict__ pair_valid) {
    constexpr int GROUP_CHUNK = OWNERS * PER;
    constexpr bool HELPERS = GROUP_THREADS > OWNERS;   // waves without a Gaussian of their own: they help with the large rectangles
    static_assert(!HELPERS || PER == 1, "helper waves: one Gaussian per owner thread");
    extern __shared__ uint32_t s_tile[];
    __shared__ uint32_t s_scan[33];
    __shared__ uint32_t s_slots[2];
    __shared__ BigRectQueue s_big;
    const bool owner = (int)threadIdx.x < OWNERS;
    if (threadIdx.x == 0) s_big.count = 0u;
    const uint32_t *row = hist + (size_t)chunk * T;
    // this thread's Gaussians: requested before anything else (their first use is behind several workgroup barriers, which
    // the compiler does not move loads across: the round trip would otherwise start after the slot scan)
    uint4 my_rect[PER];
    uint32_t my_depth[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = chunk * GROUP_CHUNK + k * OWNERS + (int)threadIdx.x;
        my_rect[k] = (owner && i < N) ? rect[i] : make_uint4(0u, 0u, 0u, 0u);
        my_depth[k] = (owner && i < N) ? depth_bits[i] : 0u;
    }
    if (LVDGS_SCAN_IN_SCATTER && totals) {
        // where every tile's list begins: the exclusive scan of the tile totals, made here (a thread takes a run of consecutive
        // tiles; positions at or beyond the capacity are dropped below, as the clamped ranges[] made them be)
        // (frames of up to SCAN_IN_SCATTER_RUN tiles per thread only -- launch_group_scatter: a thread's run of tiles is read straight
        // from memory, 16 bytes of a line per lane; at eight tiles per thread (1080p) that cost 5 us over 489 workgroups, more than the
        // launch it saved, and staged through LDS -- two more barriers -- it was a wash at every size: same-box A/B in DESIGN section 2)
        const int per_t = (T + GROUP_THREADS - 1) / GROUP_THREADS;
        const int t0 = min(T, (int)threadIdx.x * per_t), t1 = min(T, t0 + per_t);
        uint32_t sum = 0;
        for (int t = t0; t < t1; t++) sum += totals[t];
        uint32_t all;
        uint32_t run = scan_workgroup<GROUP_THREADS>(sum, s_scan, &all);
        for (int t = t0; t < t1; t++) { s_tile[t] = run + row[t]; run += totals[t]; }
        __syncthreads();   // (s_scan is used again below)
    } else {
        for (int t = threadIdx.x; t < T; t += GROUP_THREADS) s_tile[t] = ranges[t].x + row[t];  // (empty tiles are never visited)
    }
    uint32_t slots_lo = 0, slots_hi = 0;   // the chunk's Gaussians' gradient slots: [lo, hi)
    if constexpr (SLOT_SCAN) {
        uint32_t before = 0;
        for (int b = threadIdx.x; b < chunk; b += GROUP_THREADS) before += chunk_sums[b];
        const int base = chunk * GROUP_CHUNK + (int)threadIdx.x * PER;
        uint32_t v[PER], mine = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) { v[k] = (owner && base + k < N) ? tt[base + k] : 0u; mine += v[k]; }
        uint32_t prefix;
        scan_workgroup<GROUP_THREADS>(before, s_scan, &prefix);  // only the total is of interest
        __syncthreads();                     // s_scan is used again
        uint32_t total;
        uint32_t run = scan_workgroup<GROUP_THREADS>(mine, s_scan, &total) + prefix;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (owner && base + k < N) slot_base[base + k] = run;
            run += v[k];
        }
        slots_lo = prefix; slots_hi = prefix + total;
    } else {
        if (threadIdx.x == 0) {
            const int first = chunk * GROUP_CHUNK, last = min(N, first + GROUP_CHUNK) - 1;
            s_slots[0] = slot_base[first]; s_slots[1] = slot_base[last] + tt[last];
        }
        __syncthreads();
        slots_lo = s_slots[0]; slots_hi = s_slots[1];
    }
    // nothing of this frame's backward has been written yet: clear the chunk's stretch of pair_valid (16 bytes per store
    // from the first 16-byte boundary; the edges byte by byte -- a neighbouring chunk owns the rest of those words)
    if (pair_valid) {
        slots_hi = min(slots_hi, capacity);
        const uint32_t a0 = min((slots_lo + 15u) & ~15u, slots_hi), a1 = max(a0, slots_hi & ~15u);
        for (uint32_t q = slots_lo + threadIdx.x; q < a0; q += GROUP_THREADS) pair_valid[q] = 0;
        for (uint32_t q = a0 + 16u * threadIdx.x; q < a1; q += 16u * GROUP_THREADS) *reinterpret_cast<uint4 *>(pair_valid + q) = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t q = a1 + threadIdx.x; q < slots_hi; q += GROUP_THREADS) pair_valid[q] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = chunk * GROUP_CHUNK + k * OWNERS + (int)threadIdx.x;
        const uint4 r = my_rect[k];
        auto place = [&](int tile, uint32_t id, uint32_t depth) {
            const uint32_t pos = atomicAdd(&s_tile[tile], 1u);
            // the tile sort's key, so that it need not gather depths; beyond the caller's capacity: dropped (the caller
            // is told and re-runs).  (id, depth: this thread's own Gaussian, or the one its wave / workgroup walks together.)
#ifdef LVDGS_DIAG_SCATTER_NO_STORE   // diagnostic build: what the walk and the LDS atomics cost without the key stores (results are garbage)
            if (pos == 0xffffffffu) keys64[0] = ((unsigned long long)depth << 32) | (unsigned long long)id;
#else
            if (pos < capacity) keys64[pos] = ((unsigned long long)depth << 32) | (unsigned long long)id;
#endif
        };
        if constexpr (HELPERS) for_each_pair_of_rect_wg(r, i, gx, my_depth[k], s_big, place);
        else for_each_pair_of_rect(r, i, gx, my_depth[k], place);
    }
}
// SCAN_RIDES: the launch's last workgroup runs the tile scan (small frames only, SCAN_IN_SCATTER_RUN).  A template parameter, so that the
// instantiations large frames use -- up to 64 KB of dynamic LDS for the tile counters -- do not carry the scan's static LDS (s_scan[1024] and the
// wave arrays, ~4.3 KB) and its code.
template <int GROUP_THREADS, int OWNERS, int PER, bool SLOT_SCAN, bool SCAN_RIDES>
__global__ void __launch_bounds__(GROUP_THREADS) scatter_pairs_kernel(int N, int gx, int T, const uint4 *__restrict__ rect,
                                                                     const uint32_t *__restrict__ hist,
                                                                     const uint2 *__restrict__ ranges, const uint32_t *__restrict__ totals, uint32_t capacity,
                                                                     const uint32_t *__restrict__ depth_bits,
                                                                     unsigned long long *__restrict__ keys64,
                                                                     const uint32_t *__restrict__ tt, const uint32_t *__restrict__ chunk_sums,
                                                                     uint32_t *__restrict__ slot_base, uint8_t *__restrict__ pair_valid, TileScanArgs ts) {
    // (the launch's LAST workgroup is the tile scan's when ts.ranges is set: a workgroup's XCD is its index mod 8, which the chunks'
    // XCD-contiguous order counts on)
    const int nchunks = (int)gridDim.x - (SCAN_RIDES ? 1 : 0);
    if constexpr (SCAN_RIDES && (GROUP_THREADS == 1024 || GROUP_THREADS == 512)) {
        if ((int)blockIdx.x == nchunks) {
            group_tilescan_body<GROUP_THREADS>(T, totals, capacity, ts.ranges, ts.total_out, ts.long_limit, ts.queue_count, ts.queue, ts.tile_order, ts.t_lo, ts.t_hi,
                                               ts.order_valid, ts.host_out, ts.host_seq);
            return;
        }
    }
    const int chunk = LVDGS_SCATTER_XCD ? xcd_contiguous_chunk((int)blockIdx.x, nchunks) : (int)blockIdx.x;
    scatter_pairs_body<GROUP_THREADS, OWNERS, PER, SLOT_SCAN>(chunk, N, gx, T, rect, hist, ranges, totals, capacity, depth_bits, keys64, tt, chunk_sums, slot_base, pair_valid);
}
// lvdgs_forward_batch (blockIdx.y: the view).  The grid's x extent is the chunk count rounded up to a multiple of 8, so that a
// workgroup's XCD is its x index mod 8 whatever the view (workgroups are dealt to the XCDs by their linear index) and an XCD takes
// consecutive chunks as in the single-view launch; the up to 7 workgroups without a chunk leave at once.
struct ScatterView {
    const uint4 *rect; const uint32_t *hist; const uint2 *ranges; uint32_t capacity; const uint32_t *depth_bits; unsigned long long *keys64;
    const uint32_t *tt, *chunk_sums; uint32_t *slot_base; uint8_t *pair_valid;
};
struct ScatterBatch { ScatterView v[FWD_BATCH_VIEWS]; };
// (The views' tile scans stay a launch of their own here -- one workgroup per view in front of this one: with ten views' scatter
// workgroups each scanning the tile totals the window was 1.645 / 1.77-1.80 ms against 1.62-1.64 / 1.75-1.76, same box.)
template <int GROUP_THREADS, int OWNERS, int PER>
__global__ void __launch_bounds__(GROUP_THREADS) scatter_pairs_batch_kernel(int N, int gx, int T, int nchunks, ScatterBatch b) {
    const int chunk = LVDGS_SCATTER_XCD ? xcd_contiguous_chunk((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    if (chunk >= nchunks) return;
    const ScatterView &v = b.v[blockIdx.y];
    scatter_pairs_body<GROUP_THREADS, OWNERS, PER, true>(chunk, N, gx, T, v.rect, v.hist, v.ranges, nullptr, v.capacity, v.depth_bits, v.keys64, v.tt, v.chunk_sums,
                                                         v.slot_base, v.pair_valid);
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


// ---- two-level grouping (LVDGS_FLAG_SUPER_TILES) ------------------------------------------------------------------------------------
// Opaque surfaces of large flat Gaussians -- the regime SLAM maps live in -- list a Gaussian on 70-80 tiles: 7.2 M (Gaussian, tile)
// pairs for 100 k Gaussians at 1080p, and the scatter of their 8-byte sort keys (94 us) and the per-tile depth sorts (74 us) are
// what the forward pass then spends its time on outside the blend.  Depth order is a property of the GAUSSIAN, not of the pair: the
// pairs are therefore emitted, scattered and depth-sorted per SUPER-TILE of SUPER x SUPER tiles (64 x 64 pixels) -- a tenth of the pairs
// on that scene (tools/supertile_model.py) -- and every tile's list is then read off its super-tile's sorted list: the entries whose
// kept-tile mask has the tile's bit set, in the list's order (expand_super_kernel).  The same (depth bits, id) order, the same kept
// tiles: point_list, ranges and everything downstream are the one-level path's, bit for bit.
//   count_super:  rect_s[i] = the Gaussian's rectangle in super-tile units + which of them hold a listed tile; hist_s[chunk][super-tile]
//   colscan / tilescan on (hist_s, Ts): ranges_s, the queue of long super lists (the SAME kernels on a grid of Ts "tiles")
//   scatter on (rect_s, Ts, hist_s, ranges_s): the keys into the super segments; slot_base and pair_valid as ever
//   tile sort on ranges_s: sorted ids into super_list
//   expand: one workgroup per super-tile, one WAVE per tile of it
// The tile-level count (the projection kernel's), colscan and tilescan run as always: they make ranges[] and the pair count.
// Which of the SUPER x SUPER tiles of the super-tile whose first tile is (stx, sty) does the Gaussian list?  Bit j * SUPER + i: tile
// (stx + i, sty + j) -- rect_keeps for the sixteen tiles at once, the rectangle decoded once.
__device__ __forceinline__ uint32_t kept_in_super(const uint4 r, int stx, int sty) {
    const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
    const int w = x1 - x0, h = y1 - y0, area = w * h;
    const uint64_t m = (uint64_t)r.z | ((uint64_t)r.w << 32);
    uint32_t out = 0u;
    if (area <= 0) return 0u;
    if (area <= RECT_MASK_TILES) {
#pragma unroll
        for (int j = 0; j < SUPER; j++) {
            const int ty = sty + j - y0;
            if ((unsigned)ty >= (unsigned)h) continue;
            const uint64_t row = m >> (ty * w);
#pragma unroll
            for (int i = 0; i < SUPER; i++) {
                const int tx = stx + i - x0;
                if ((unsigned)tx < (unsigned)w) out |= (uint32_t)((row >> tx) & 1ull) << (j * SUPER + i);
            }
        }
    } else {
        const RectBlocks g(w, h);
        const float inv_bw = __builtin_amdgcn_rcpf((float)g.bw), inv_bh = __builtin_amdgcn_rcpf((float)g.bh);
        int bc[SUPER], br[SUPER];
#pragma unroll
        for (int i = 0; i < SUPER; i++) {
            const int tx = stx + i - x0, ty = sty + i - y0;
            bc[i] = (unsigned)tx < (unsigned)w ? div_by(tx, g.bw, inv_bw) : -1;
            br[i] = (unsigned)ty < (unsigned)h ? div_by(ty, g.bh, inv_bh) : -1;
        }
#pragma unroll
        for (int j = 0; j < SUPER; j++)
#pragma unroll
            for (int i = 0; i < SUPER; i++)
                if (bc[i] >= 0 && br[j] >= 0) out |= (uint32_t)((m >> (br[j] * 8 + bc[i])) & 1ull) << (j * SUPER + i);
    }
    return out;
}

template <int GROUP_THREADS, int OWNERS, int PER>
__global__ void __launch_bounds__(GROUP_THREADS) count_super_kernel(int N, int gxs, int Ts, const uint4 *__restrict__ rect, uint4 *__restrict__ rect_s,
                                                                   uint32_t *__restrict__ hist, uint32_t *__restrict__ queue_counts) {
    constexpr int GROUP_CHUNK = OWNERS * PER;
    constexpr bool HELPERS = GROUP_THREADS > OWNERS;
    extern __shared__ uint32_t s_tile[];
    __shared__ BigRectQueue s_big;
    for (int t = threadIdx.x; t < Ts; t += GROUP_THREADS) s_tile[t] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 64) queue_counts[threadIdx.x] = 0u;   // the super lists' own tile-sort queue and order flag
    if (threadIdx.x == 0) s_big.count = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = blockIdx.x * GROUP_CHUNK + k * OWNERS + (int)threadIdx.x;
        const bool mine = (int)threadIdx.x < OWNERS && i < N;
        const uint4 r = mine ? super_rect_of(rect[i]) : make_uint4(0u, 0u, 0u, 0u);
        if (mine) rect_s[i] = r;
        if constexpr (HELPERS) for_each_pair_of_rect_wg(r, i, gxs, 0u, s_big, [&](int tile, uint32_t, uint32_t) { atomicAdd(&s_tile[tile], 1u); });
        else for_each_pair_of_rect(r, i, gxs, 0u, [&](int tile, uint32_t, uint32_t) { atomicAdd(&s_tile[tile], 1u); });
    }
    __syncthreads();
    uint32_t *row = hist + (size_t)blockIdx.x * Ts;
    for (int t = threadIdx.x; t < Ts; t += GROUP_THREADS) row[t] = s_tile[t];
}

// One workgroup per super-tile, one wave per tile of it (SUPER * SUPER = 16 waves): the super-tile's sorted list is staged through LDS
// 1024 entries at a time -- a thread per entry loads the id, gathers the Gaussian's tile rectangle + kept-tile mask and works out ONCE
// which of the sixteen tiles list the Gaussian (kept_in_super: the bits the projection kernel set, the very test the tile counts were
// taken with) -- then every wave walks the staged entries 64 at a time and appends the ids whose bit for its tile is set to the tile's
// segment, in list order.
__device__ __forceinline__ void expand_super_body(int gx, int gy, int gxs, const uint2 *__restrict__ ranges_s, const uint32_t *__restrict__ super_list,
                                                  const uint4 *__restrict__ rect, const uint2 *__restrict__ ranges, uint32_t *__restrict__ point_list) {
    constexpr int STAGE = 64 * SUPER * SUPER;
    __shared__ uint32_t s_id[STAGE];
    __shared__ uint32_t s_keep[STAGE];
    const int st = (int)blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stx = (st % gxs) * SUPER, sty = (st / gxs) * SUPER;
    const int tx = stx + (wave % SUPER), ty = sty + (wave / SUPER);
    const bool valid = tx < gx && ty < gy;
    const uint2 seg = ranges_s[st];
    const uint2 mine = valid ? ranges[ty * gx + tx] : make_uint2(0u, 0u);
    uint32_t out = mine.x;
    const int n = (int)(seg.y - seg.x);
    // (the first stage's ids are on their way before anything else)
    uint32_t id = (int)threadIdx.x < n ? super_list[seg.x + threadIdx.x] : 0u;
    for (int base = 0; base < n; base += STAGE) {
        const int m = min(STAGE, n - base);
        if ((int)threadIdx.x < m) {
            s_id[threadIdx.x] = id;
            s_keep[threadIdx.x] = kept_in_super(rect[id], stx, sty);
        }
        if (base + STAGE + (int)threadIdx.x < n) id = super_list[seg.x + base + STAGE + threadIdx.x];   // next stage's id: in flight under this stage's walk
        __syncthreads();
        if (mine.y > mine.x) {
            for (int j = 0; j < m; j += 64) {
                const int e = j + lane;
                const bool keep = e < m && ((s_keep[e] >> wave) & 1u);
                const uint64_t kept = __ballot(keep);
                if (keep) {
                    const uint32_t pos = out + (uint32_t)__popcll(kept & ((1ull << lane) - 1ull));
                    if (pos < mine.y) point_list[pos] = s_id[e];   // (positions at or beyond the pair capacity: dropped, as the clamped ranges say)
                }
                out += (uint32_t)__popcll(kept);
            }
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(64 * SUPER * SUPER) expand_super_kernel(int gx, int gy, int gxs, const uint2 *__restrict__ ranges_s,
                                                                        const uint32_t *__restrict__ super_list, const uint4 *__restrict__ rect,
                                                                        const uint2 *__restrict__ ranges, uint32_t *__restrict__ point_list) {
    expand_super_body(gx, gy, gxs, ranges_s, super_list, rect, ranges, point_list);
}

// lvdgs_forward_batch (blockIdx.y: the view)
struct ExpandView { const uint2 *ranges_s; const uint32_t *super_list; const uint4 *rect; const uint2 *ranges; uint32_t *point_list; };
struct ExpandBatch { ExpandView v[FWD_BATCH_VIEWS]; };
__global__ void __launch_bounds__(64 * SUPER * SUPER) expand_super_batch_kernel(int gx, int gy, int gxs, ExpandBatch b) {
    const ExpandView &v = b.v[blockIdx.y];
    expand_super_body(gx, gy, gxs, v.ranges_s, v.super_list, v.rect, v.ranges, v.point_list);
}

}  // namespace

#ifndef LVDGS_GROUP_CHUNK
#define LVDGS_GROUP_CHUNK 0   // A/B builds: 256, 512, 1024, 2048 or 4096 Gaussians per grouping workgroup whatever the map's size
#endif
GroupShape group_shape_for(int N) {
    if (LVDGS_GROUP_CHUNK) return LVDGS_GROUP_CHUNK <= 1024 ? GroupShape{LVDGS_GROUP_CHUNK, 1} : GroupShape{1024, LVDGS_GROUP_CHUNK / 1024};
    return group_shape_default(N);
}
int group_max_tiles() { return GROUP_MAX_TILES; }
static int chunk_of(int N) { const GroupShape g = group_shape_for(N); return g.threads * g.per; }
size_t group_chunks(int N) { return (size_t)cdiv(N > 0 ? N : 1, chunk_of(N)); }
size_t group_hist_entries(int N, int num_tiles) { return group_chunks(N) * (size_t)num_tiles; }

int launch_group_count(const lvdgs_args &a, const GeomView &g, const ImageView &im, const RenderScratch &w, hipStream_t s) {
    const int N = a.num_gaussians;
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE, T = gx * gy;
    if (N == 0 || T == 0) return LVDGS_OK;
    const int nchunks = (int)group_chunks(N);
    const size_t lds = (size_t)T * sizeof(uint32_t);
    static unsigned char done[GROUP_SHAPES][16];
    ProfScope ps("group_count", s);
    if (int e = group_dispatch(group_shape_for(N), [&](auto threads_, auto owners_, auto per_, int d) {
            constexpr int THREADS = decltype(threads_)::value, OWNERS = decltype(owners_)::value, PER = decltype(per_)::value;
            if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&count_pairs_kernel<THREADS, OWNERS, PER>), GROUP_MAX_TILES * 4, done[d])) return e;
            hipLaunchKernelGGL((count_pairs_kernel<THREADS, OWNERS, PER>), dim3(nchunks), dim3(THREADS), lds, s, N, gx, T, (const uint4 *)g.rect, w.group_hist,
                               a.n_touched, im.long_count);
            return (int)LVDGS_OK;
        })) return e;
    LVDGS_LAUNCH_CHECK("group_count", a.debug, s);
    return LVDGS_OK;
}

// colscan + tilescan: tile ranges, the pair count (total_out, may be null), the tile sort's queue, the tile order
constexpr int SCAN_IN_SCATTER_RUN = 4;   // tiles per scatter thread up to which the tile scan rides in the scatter's launch
static bool scan_rides_in_scatter(const lvdgs_args &a, int N, int T) {
    if (super_tiles_in_use(a)) return false;   // (two-level grouping: the tile-level scatter, whose launch the scan would ride in, does not run)
    const GroupShape g = group_shape_for(N);
    const int threads = g.threads == 256 ? GROUP_HELPER_THREADS : g.threads;   // (every shape launches 512 or 1024 threads)
    return LVDGS_SCAN_IN_SCATTER != 0 && (threads == 512 || threads == 1024) && cdiv(T, threads) <= SCAN_IN_SCATTER_RUN;
}

int launch_group_scan(const lvdgs_args &a, const ImageView &im, const RenderScratch &w, int64_t capacity, uint32_t *total_out, hipStream_t s,
                      uint32_t *host_out, uint32_t host_seq) {
    const int N = a.num_gaussians;
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE, T = gx * gy;
    if (N == 0 || T == 0) return LVDGS_OK;
    const int nchunks = (int)group_chunks(N);
    int row0, row1;
    tile_row_band(a, &row0, &row1);
    ProfScope ps("group_scan", s);
    // waves per workgroup: as many as leave every wave at most 16 (then 32) matrix rows to hold in registers
    const int wg_tiles = cdiv(T, COLSCAN_TILES);
    if (super_tiles_in_use(a)) {
        // two-level grouping: the super-tile grid's count matrix (launch_super_count has run) is scanned in the same two launches
        const SuperView &sv = w.super;
        const int Ts = super_tiles_of(a.image_width, a.image_height);
        const dim3 grid(wg_tiles, 2);
        if (nchunks <= 8 * 16) hipLaunchKernelGGL(group_colscan_pair_kernel<8>, grid, dim3(1024), 0, s, T, Ts, nchunks, w.group_hist, w.group_totals, sv.hist, sv.totals);
        else if (nchunks <= 16 * 16) hipLaunchKernelGGL(group_colscan_pair_kernel<16>, grid, dim3(1024), 0, s, T, Ts, nchunks, w.group_hist, w.group_totals, sv.hist, sv.totals);
        else if (nchunks <= 32 * 16) hipLaunchKernelGGL(group_colscan_pair_kernel<32>, grid, dim3(1024), 0, s, T, Ts, nchunks, w.group_hist, w.group_totals, sv.hist, sv.totals);
        else hipLaunchKernelGGL(group_colscan_pair_kernel<0>, grid, dim3(1024), 0, s, T, Ts, nchunks, w.group_hist, w.group_totals, sv.hist, sv.totals);
        // (the super grid's hints -- longest queued list, queue length -- go to the pinned words behind the call's own four: for the NEXT
        // frame's sort launch, read whenever; only the tile grid's workgroup writes the sequence number the host waits for)
        const TilescanPairView tv{T, w.group_totals, im.ranges, total_out, im.long_count, im.long_tiles, tile_order_in_use(T) ? im.long_tiles + T : nullptr,
                                  row0 * gx, row1 * gx, im.long_count + 1, host_out, host_seq};
        const TilescanPairView sv2{Ts, sv.totals, sv.ranges, sv.total, sv.long_count, sv.long_tiles, tile_order_in_use(Ts) ? sv.long_tiles + Ts : nullptr,
                                   0, Ts, sv.long_count + 1, host_out ? host_out + 8 : nullptr, 0u};
        hipLaunchKernelGGL(group_tilescan_pair_kernel, dim3(2), dim3(1024), 0, s, tv, sv2, (uint32_t)capacity, (uint32_t)tile_sort_wave_limit());
        LVDGS_LAUNCH_CHECK("group_scan (two-level)", a.debug, s);
        return LVDGS_OK;
    }
    if (nchunks <= 8 * 16) hipLaunchKernelGGL(group_colscan_kernel<8>, dim3(wg_tiles), dim3(1024), 0, s, T, nchunks, w.group_hist, w.group_totals);
    else if (nchunks <= 16 * 16) hipLaunchKernelGGL(group_colscan_kernel<16>, dim3(wg_tiles), dim3(1024), 0, s, T, nchunks, w.group_hist, w.group_totals);
    else if (nchunks <= 32 * 16) hipLaunchKernelGGL(group_colscan_kernel<32>, dim3(wg_tiles), dim3(1024), 0, s, T, nchunks, w.group_hist, w.group_totals);
    else hipLaunchKernelGGL(group_colscan_kernel<0>, dim3(wg_tiles), dim3(1024), 0, s, T, nchunks, w.group_hist, w.group_totals);
    if (!scan_rides_in_scatter(a, N, T))   // (else: launch_group_scatter's last workgroup)
        hipLaunchKernelGGL(group_tilescan_kernel, dim3(1), dim3(1024), 0, s, T, (const uint32_t *)w.group_totals, (uint32_t)capacity, im.ranges,
                           total_out, (uint32_t)tile_sort_wave_limit(), im.long_count, im.long_tiles,
                           tile_order_in_use(T) ? im.long_tiles + T : nullptr, row0 * gx, row1 * gx, im.long_count + 1, host_out, host_seq);
    LVDGS_LAUNCH_CHECK("group_scan", a.debug, s);
    return LVDGS_OK;
}

int launch_group_scatter(const lvdgs_args &a, const GeomView &g, const ImageView &im, const RenderScratch &w, unsigned long long *keys64,
                         int64_t capacity, bool slot_scan, uint8_t *pair_valid, hipStream_t s, uint32_t *total_out, uint32_t *host_out, uint32_t host_seq) {
    const int N = a.num_gaussians;
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE, T = gx * gy;
    if (N == 0 || T == 0) return LVDGS_OK;
    const int nchunks = (int)group_chunks(N);
    const size_t lds = (size_t)T * sizeof(uint32_t);
    static unsigned char done[4 * GROUP_SHAPES][16];
    ProfScope ps("group_scatter", s);
    TileScanArgs ts{};
    if (scan_rides_in_scatter(a, N, T)) {   // the tile scan as this launch's last workgroup (launch_group_scan has made the tile totals)
        int row0, row1;
        tile_row_band(a, &row0, &row1);
        ts = TileScanArgs{im.ranges, total_out, (uint32_t)tile_sort_wave_limit(), im.long_count, im.long_tiles, tile_order_in_use(T) ? im.long_tiles + T : nullptr,
                          row0 * gx, row1 * gx, im.long_count + 1, host_out, host_seq};
    }
    auto launch = [&](auto kernel, int d, int threads) {
        if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), GROUP_MAX_TILES * 4, done[d])) return e;
        hipLaunchKernelGGL(kernel, dim3(nchunks + (ts.ranges ? 1 : 0)), dim3(threads), lds, s, N, gx, T, (const uint4 *)g.rect, (const uint32_t *)w.group_hist,
                           (const uint2 *)im.ranges, ts.ranges ? (const uint32_t *)w.group_totals : nullptr, (uint32_t)capacity, (const uint32_t *)g.depth_bits, keys64,
                           (const uint32_t *)g.tiles_touched, (const uint32_t *)w.chunk_sums, g.slot_base, pair_valid, ts);
        return (int)LVDGS_OK;
    };
    if (int e = group_dispatch(group_shape_for(N), [&](auto threads_, auto owners_, auto per_, int d) {
            constexpr int THREADS = decltype(threads_)::value, OWNERS = decltype(owners_)::value, PER = decltype(per_)::value;
            if (ts.ranges)   // (scan_rides_in_scatter: every shape launches 512 or 1024 threads there)
                return slot_scan ? launch(&scatter_pairs_kernel<THREADS, OWNERS, PER, true, true>, 4 * d, THREADS) : launch(&scatter_pairs_kernel<THREADS, OWNERS, PER, false, true>, 4 * d + 1, THREADS);
            return slot_scan ? launch(&scatter_pairs_kernel<THREADS, OWNERS, PER, true, false>, 4 * d + 2, THREADS) : launch(&scatter_pairs_kernel<THREADS, OWNERS, PER, false, false>, 4 * d + 3, THREADS);
        })) return e;
    LVDGS_LAUNCH_CHECK("group_scatter", a.debug, s);
    return LVDGS_OK;
}

// ---- two-level grouping: count_super -> scans -> scatter on the super grid (the tile sort and the expansion: api.hip calls them) ----
bool super_tiles_in_use(const lvdgs_args &a) {
    if (!(a.flags & LVDGS_FLAG_SUPER_TILES) || (a.flags & LVDGS_FLAG_LIST_ALL_TILES)) return false;
    if (a.tile_row_begin != 0 || a.tile_row_end != 0) return false;   // (bands: a super-tile would straddle the band's edge)
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE;
    return a.num_gaussians > 0 && gx * gy <= GROUP_MAX_TILES && gx * gy >= 4 * SUPER * SUPER;
}
int super_tiles_of(int W, int H) { return cdiv(cdiv(W, TILE), SUPER) * cdiv(cdiv(H, TILE), SUPER); }

int launch_super_count(const lvdgs_args &a, const GeomView &g, const SuperView &sv, hipStream_t s) {
    const int N = a.num_gaussians;
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE;
    const int gxs = cdiv(gx, SUPER), Ts = gxs * cdiv(gy, SUPER);
    if (N == 0 || Ts == 0) return LVDGS_OK;
    const int nchunks = (int)group_chunks(N);
    const size_t lds = (size_t)Ts * sizeof(uint32_t);
    static unsigned char done[GROUP_SHAPES][16];
    ProfScope ps("super_count", s);
    if (int e = group_dispatch(group_shape_for(N), [&](auto threads_, auto owners_, auto per_, int d) {
            constexpr int THREADS = decltype(threads_)::value, OWNERS = decltype(owners_)::value, PER = decltype(per_)::value;
            if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&count_super_kernel<THREADS, OWNERS, PER>), GROUP_MAX_TILES * 4, done[d])) return e;
            hipLaunchKernelGGL((count_super_kernel<THREADS, OWNERS, PER>), dim3(nchunks), dim3(THREADS), lds, s, N, gxs, Ts, (const uint4 *)g.rect, sv.rect, sv.hist,
                               sv.long_count);
            return (int)LVDGS_OK;
        })) return e;
    LVDGS_LAUNCH_CHECK("super_count", a.debug, s);
    return LVDGS_OK;
}

int launch_super_scatter(const lvdgs_args &a, const GeomView &g, const SuperView &sv, const RenderScratch &w, unsigned long long *keys64, int64_t capacity,
                         bool slot_scan, uint8_t *pair_valid, hipStream_t s) {
    const int N = a.num_gaussians;
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE;
    const int gxs = cdiv(gx, SUPER), Ts = gxs * cdiv(gy, SUPER);
    if (N == 0 || Ts == 0) return LVDGS_OK;
    const int nchunks = (int)group_chunks(N);
    const size_t lds = (size_t)Ts * sizeof(uint32_t);
    static unsigned char done[2 * GROUP_SHAPES][16];
    ProfScope ps("super_scatter", s);
    TileScanArgs ts{};
    auto launch = [&](auto kernel, int d, int threads) {
        if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), GROUP_MAX_TILES * 4, done[d])) return e;
        hipLaunchKernelGGL(kernel, dim3(nchunks), dim3(threads), lds, s, N, gxs, Ts, (const uint4 *)sv.rect, (const uint32_t *)sv.hist, (const uint2 *)sv.ranges,
                           (const uint32_t *)nullptr, (uint32_t)capacity, (const uint32_t *)g.depth_bits, keys64, (const uint32_t *)g.tiles_touched,
                           (const uint32_t *)w.chunk_sums, g.slot_base, pair_valid, ts);
        return (int)LVDGS_OK;
    };
    if (int e = group_dispatch(group_shape_for(N), [&](auto threads_, auto owners_, auto per_, int d) {
            constexpr int THREADS = decltype(threads_)::value, OWNERS = decltype(owners_)::value, PER = decltype(per_)::value;
            return slot_scan ? launch(&scatter_pairs_kernel<THREADS, OWNERS, PER, true, false>, 2 * d, THREADS)
                             : launch(&scatter_pairs_kernel<THREADS, OWNERS, PER, false, false>, 2 * d + 1, THREADS);
        })) return e;
    LVDGS_LAUNCH_CHECK("super_scatter", a.debug, s);
    return LVDGS_OK;
}

int launch_super_expand(const lvdgs_args &a, const GeomView &g, const SuperView &sv, const ImageView &im, const uint32_t *super_list, uint32_t *point_list,
                        hipStream_t s) {
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE;
    const int gxs = cdiv(gx, SUPER), Ts = gxs * cdiv(gy, SUPER);
    if (a.num_gaussians == 0 || Ts == 0) return LVDGS_OK;
    ProfScope ps("super_expand", s);
    hipLaunchKernelGGL(expand_super_kernel, dim3(Ts), dim3(64 * SUPER * SUPER), 0, s, gx, gy, gxs, (const uint2 *)sv.ranges, super_list, (const uint4 *)g.rect,
                       (const uint2 *)im.ranges, point_list);
    LVDGS_LAUNCH_CHECK("super_expand", a.debug, s);
    return LVDGS_OK;
}

// ---- lvdgs_forward_batch: the same stages for n views of one map and one image size, one launch each ----
int launch_group_scan_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, const int64_t *caps, int n,
                            uint32_t *host_words, uint32_t host_seq, hipStream_t s) {
    const int N = a[0]->num_gaussians;
    const int gx = (a[0]->image_width + TILE - 1) / TILE, gy = (a[0]->image_height + TILE - 1) / TILE, T = gx * gy;
    if (N == 0 || T == 0 || n == 0) return LVDGS_OK;
    const int nchunks = (int)group_chunks(N);
    int row0, row1;
    tile_row_band(*a[0], &row0, &row1);
    ColscanBatch cb{};
    TilescanBatch tb{};
    for (int k = 0; k < n; k++) {
        cb.hist[k] = w[k].group_hist; cb.totals[k] = w[k].group_totals;
        tb.v[k] = TilescanView{w[k].group_totals, (uint32_t)caps[k], im[k].ranges, g[k].total, im[k].long_count, im[k].long_tiles,
                               tile_order_in_use(T) ? im[k].long_tiles + T : nullptr, im[k].long_count + 1};
    }
    ProfScope ps("group_scan", s);
    if (super_tiles_in_use(*a[0])) {
        // two-level grouping: the tile grid's and the super-tile grid's count matrices of every view in the same two launches
        const int Ts = super_tiles_of(a[0]->image_width, a[0]->image_height);
        ColscanPairBatch cpb{};
        TilescanPairBatch tpb{};
        for (int k = 0; k < n; k++) {
            const SuperView &sv = w[k].super;
            cpb.hist[2 * k] = w[k].group_hist; cpb.totals[2 * k] = w[k].group_totals;
            cpb.hist[2 * k + 1] = sv.hist; cpb.totals[2 * k + 1] = sv.totals;
            tpb.v[2 * k] = TilescanView{w[k].group_totals, (uint32_t)caps[k], im[k].ranges, g[k].total, im[k].long_count, im[k].long_tiles,
                                        tile_order_in_use(T) ? im[k].long_tiles + T : nullptr, im[k].long_count + 1};
            tpb.v[2 * k + 1] = TilescanView{sv.totals, (uint32_t)caps[k], sv.ranges, sv.total, sv.long_count, sv.long_tiles,
                                            tile_order_in_use(Ts) ? sv.long_tiles + Ts : nullptr, sv.long_count + 1};
        }
        const dim3 grid2(cdiv(T, COLSCAN_TILES), 2 * n);
        if (nchunks <= 8 * 16) hipLaunchKernelGGL(group_colscan_pair_batch_kernel<8>, grid2, dim3(1024), 0, s, T, Ts, nchunks, cpb);
        else if (nchunks <= 16 * 16) hipLaunchKernelGGL(group_colscan_pair_batch_kernel<16>, grid2, dim3(1024), 0, s, T, Ts, nchunks, cpb);
        else if (nchunks <= 32 * 16) hipLaunchKernelGGL(group_colscan_pair_batch_kernel<32>, grid2, dim3(1024), 0, s, T, Ts, nchunks, cpb);
        else hipLaunchKernelGGL(group_colscan_pair_batch_kernel<0>, grid2, dim3(1024), 0, s, T, Ts, nchunks, cpb);
        hipLaunchKernelGGL(group_tilescan_pair_batch_kernel, dim3(2 * n), dim3(1024), 0, s, T, Ts, tpb, (uint32_t)tile_sort_wave_limit(), row0 * gx, row1 * gx, host_words,
                           host_seq, host_words ? host_words - 8 : nullptr);   // (the call's pinned words: [8..11] the super grid's hints, [16 + 4 k..] view k's)
        LVDGS_LAUNCH_CHECK("group_scan (batch, two-level)", a[0]->debug, s);
        return LVDGS_OK;
    }
    const dim3 grid(cdiv(T, COLSCAN_TILES), n);
    if (nchunks <= 8 * 16) hipLaunchKernelGGL(group_colscan_batch_kernel<8>, grid, dim3(1024), 0, s, T, nchunks, cb);
    else if (nchunks <= 16 * 16) hipLaunchKernelGGL(group_colscan_batch_kernel<16>, grid, dim3(1024), 0, s, T, nchunks, cb);
    else if (nchunks <= 32 * 16) hipLaunchKernelGGL(group_colscan_batch_kernel<32>, grid, dim3(1024), 0, s, T, nchunks, cb);
    else hipLaunchKernelGGL(group_colscan_batch_kernel<0>, grid, dim3(1024), 0, s, T, nchunks, cb);
    hipLaunchKernelGGL(group_tilescan_batch_kernel, dim3(n), dim3(1024), 0, s, T, tb, (uint32_t)tile_sort_wave_limit(), row0 * gx, row1 * gx, host_words, host_seq);
    LVDGS_LAUNCH_CHECK("group_scan (batch)", a[0]->debug, s);
    return LVDGS_OK;
}

int launch_group_scatter_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, const BinView *b,
                               const int64_t *caps, int n, hipStream_t s) {
    const int N = a[0]->num_gaussians;
    const int gx = (a[0]->image_width + TILE - 1) / TILE, gy = (a[0]->image_height + TILE - 1) / TILE, T = gx * gy;
    if (N == 0 || T == 0 || n == 0) return LVDGS_OK;
    const int nchunks = (int)group_chunks(N);
    const bool super = super_tiles_in_use(*a[0]);   // two-level grouping: the same kernel on the super-tile grid (its rectangles, counts and ranges)
    const int gxs = cdiv(gx, SUPER), Ts = gxs * cdiv(gy, SUPER);
    const size_t lds = (size_t)(super ? Ts : T) * sizeof(uint32_t);
    ScatterBatch sb{};
    for (int k = 0; k < n; k++)
        sb.v[k] = super ? ScatterView{(const uint4 *)w[k].super.rect, w[k].super.hist, w[k].super.ranges, (uint32_t)caps[k], g[k].depth_bits, (unsigned long long *)w[k].keys,
                                      g[k].tiles_touched, w[k].chunk_sums, g[k].slot_base, b[k].pair_valid}
                        : ScatterView{(const uint4 *)g[k].rect, w[k].group_hist, im[k].ranges, (uint32_t)caps[k], g[k].depth_bits, (unsigned long long *)w[k].keys,
                                      g[k].tiles_touched, w[k].chunk_sums, g[k].slot_base, b[k].pair_valid};
    static unsigned char done[GROUP_SHAPES][16];
    ProfScope ps(super ? "super_scatter" : "group_scatter", s);
    if (int e = group_dispatch(group_shape_for(N), [&](auto threads_, auto owners_, auto per_, int d) {
            constexpr int THREADS = decltype(threads_)::value, OWNERS = decltype(owners_)::value, PER = decltype(per_)::value;
            if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&scatter_pairs_batch_kernel<THREADS, OWNERS, PER>), GROUP_MAX_TILES * 4, done[d])) return e;
            hipLaunchKernelGGL((scatter_pairs_batch_kernel<THREADS, OWNERS, PER>), dim3((nchunks + 7) & ~7, n), dim3(THREADS), lds, s, N, super ? gxs : gx, super ? Ts : T,
                               nchunks, sb);
            return (int)LVDGS_OK;
        })) return e;
    LVDGS_LAUNCH_CHECK("group_scatter (batch)", a[0]->debug, s);
    return LVDGS_OK;
}

int launch_super_expand_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, const BinView *b, int n, hipStream_t s) {
    const int gx = (a[0]->image_width + TILE - 1) / TILE, gy = (a[0]->image_height + TILE - 1) / TILE;
    const int gxs = cdiv(gx, SUPER), Ts = gxs * cdiv(gy, SUPER);
    if (a[0]->num_gaussians == 0 || Ts == 0 || n == 0) return LVDGS_OK;
    ExpandBatch eb{};
    for (int k = 0; k < n; k++) eb.v[k] = ExpandView{(const uint2 *)w[k].super.ranges, b[k].tile_keys, (const uint4 *)g[k].rect, (const uint2 *)im[k].ranges, b[k].point_list};
    ProfScope ps("super_expand", s);
    hipLaunchKernelGGL(expand_super_batch_kernel, dim3(Ts, n), dim3(64 * SUPER * SUPER), 0, s, gx, gy, gxs, eb);
    LVDGS_LAUNCH_CHECK("super_expand (batch)", a[0]->debug, s);
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
