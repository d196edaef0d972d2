// Depth order inside every tile's list.
//
// The (Gaussian, tile) pairs are grouped by tile by counting (binning.hip; a stable radix sort on the tile id for images
// of more than 16384 tiles), in arbitrary order inside a tile.  This file sorts every tile's segment by the 64-bit key
// (view-depth bits << 32 | id): exactly the order a stable sort of (tile << 32 | depth) keys over id-ordered pairs gives
// (ties between equal depths go to the smaller id), without ever sorting the N Gaussians or the D pairs globally -- the
// segments are a few hundred entries each and sort independently.
//
// Size classes (bitonic networks throughout):
//   <= 1024 entries: ONE WAVE per tile, keys in registers (4, 8 or 16 per lane): strides inside a lane are register
//      compare-exchanges, strides across lanes one 64-bit lane shuffle per key; no LDS array, no barrier.  This
//      is the common case by far and ~8x fewer instructions than a workgroup-wide LDS network, whose threads mostly wait
//      at barriers;
//   longer: queued -- by the tile-range scan on the counting path, by the wave kernel itself on the radix path -- and
//      sorted by the LAST workgroups of the same launch (256 threads, LDS up to 2048 entries, in place on global memory
//      beyond), or, when the previous frame on this device had such segments: up to 2048 entries by workgroups at the HEAD of
//      the same launch, the four waves of one sorting a segment together in registers (8 keys per lane, the three
//      stages that cross waves through LDS); beyond by a kernel launched behind the tile sort, a 1024-thread workgroup on
//      128 KiB of LDS (<= 16384), global memory beyond.  (Round 2 launched the 1024-thread kernel on every frame: ~6 us for
//      an empty queue.  Until round 4 segments of 1025-2048 entries had a kernel of their own behind the tile sort, one wave
//      each with 32 keys per lane: 184 registers, two waves per SIMD, and 1100 keys padded to 2048 by ONE wave -- 52 us on top
//      of the wave kernel's 57 on the opaque-surface frame, whose lists are 900-1130 entries.)
#include "common.hpp"
#include "device_utils.hpp"

namespace lvdgs {
namespace {

typedef unsigned long long u64;

// Where a segment's keys come from: already scattered as 64-bit keys (counting path), or gathered from the ids in
// point_list and the depths in the records (radix path).  The choice is uniform over the launch.
struct KeySource {
    const float *rec;
    const uint32_t *point_list;
    const u64 *keys;  // null: gather
    __device__ __forceinline__ u64 load(uint32_t pos) const {
        if (keys) return keys[pos];
        const uint32_t id = point_list[pos];
        return ((u64)__float_as_uint(rec[(size_t)id * REC_FLOATS + 9]) << 32) | (u64)id;
    }
};

// All-ascending bitonic network (every merge starts with a mirrored compare, so no direction flags).  With the
// tail beyond n treated as +infinity an exchange whose partner lies beyond n can never swap, so a segment of
// any length sorts in place without padding.
// Thread t owns compare-exchange t (+ multiples of the workgroup size) of every stage; for strides j <= 64 a
// wave's 64 exchanges stay inside one aligned block of 128 elements, the same block in every such stage, so
// those stages only need the wave's own LDS accesses ordered -- a workgroup barrier is paid for j >= 128 only
// (1 of 36 stages at 256 elements, 10 of 66 at 2048).
template <bool GLOBAL, typename Ptr>
__device__ __forceinline__ void bitonic_sort_ascending(Ptr s, int n, int tid, int nthreads) {
    int lpad = 0;
    while ((1 << lpad) < n) lpad++;
    const int half = (1 << lpad) >> 1;
    for (int lk = 1; lk <= lpad; lk++) {
        const int k = 1 << lk;
        for (int lj = lk - 1; lj >= 0; lj--) {
            const int j = 1 << lj;
            const bool mirror = lj == lk - 1;
            for (int t = tid; t < half; t += nthreads) {
                const int blk = t >> lj, w = t & (j - 1);
                const int i = (blk << (lj + 1)) | w;
                const int p = mirror ? (blk << lk) + (k - 1 - w) : i + j;
                if (p < n) {
                    const u64 a = s[i], b = s[p];
                    if (a > b) { s[i] = b; s[p] = a; }
                }
            }
            if (GLOBAL) {
                __threadfence_block();
                __syncthreads();
            } else if (j >= 128) {
                __syncthreads();
            } else {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // the next merge starts with a mirrored exchange over blocks of 2k: if that crosses waves, meet first
        if (!GLOBAL && k >= 128 && k < (1 << lpad)) __syncthreads();
    }
}

// value of lane ^ MASK.  Distances 1, 2, 4, 8 are DPP moves inside a row of 16 lanes (no LDS round trip):
// quad_perm for 1 and 2, row_ror:8 for 8, and 4 = 7 ^ 3 (row_half_mirror, then quad_perm [3,2,1,0]);
// 16 and 32 go through the LDS crossbar (ds_bpermute).
template <int MASK>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v, int lane) {
    if (MASK == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    if (MASK == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    if (MASK == 4) {
        const int t = __builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);                 // row_half_mirror: i ^ 7
        return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x1B, 0xf, 0xf, true);                    // quad_perm [3,2,1,0]: i ^ 3
    }
    if (MASK == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, true);  // row_ror:8
    return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, (int)v);
}

template <int MASK>
__device__ __forceinline__ u64 lane_xor64(u64 v, int lane) {
    return ((u64)lane_xor32<MASK>((uint32_t)(v >> 32), lane) << 32) | (u64)lane_xor32<MASK>((uint32_t)v, lane);
}

// Classic bitonic network over LANES * E keys held E per lane (element e = g * E + r), ascending.  Keys are
// distinct (the id is part of the key), so "take the other key" is one comparison xor a per-stage lane mask.
// LANES = 64: one wave (g = lane).  LANES = 256: the four waves of a workgroup sort one segment together (g = thread index);
// the three stages whose partner sits in another wave -- same lane, same register -- go through xbuf (16 KB of LDS, eight
// registers at a time) between two workgroup barriers: every wave of the workgroup must be in the same sort.
template <int E, int LANES, int K, int J>
__device__ __forceinline__ void wave_bitonic_stage(u64 (&key)[E], int g, u64 *xbuf) {
    const int lane = g & 63;
    if constexpr (J >= 64 * E) {
        static_assert(LANES > 64, "partner in another wave");
        const bool ascending = K >= LANES * E || (g & (K / E)) == 0;
        const bool keep_min = ((g & (J / E)) == 0) == ascending;
        const int w = g >> 6, pw = w ^ (J / (64 * E));
#pragma unroll
        for (int h = 0; h < E; h += 8) {
            __syncthreads();   // (the previous exchange has been read)
#pragma unroll
            for (int r = 0; r < 8 && h + r < E; r++) xbuf[(w * 8 + r) * 64 + lane] = key[h + r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8 && h + r < E; r++) {
                const u64 other = xbuf[(pw * 8 + r) * 64 + lane];
                key[h + r] = ((other < key[h + r]) == keep_min) ? other : key[h + r];
            }
        }
    } else if constexpr (J >= E) {
        // partner element e ^ J lives in lane ^ (J / E), same register; the direction of the K-block and which side
        // of the pair this lane is on are the same for all its registers
        const bool ascending = K >= LANES * E || (g & (K / E)) == 0;
        const bool keep_min = ((g & (J / E)) == 0) == ascending;
        u64 other[E];  // all the lane exchanges first: they are independent, the selects then follow without stalls
#pragma unroll
        for (int r = 0; r < E; r++) other[r] = lane_xor64<J / E>(key[r], lane);
#pragma unroll
        for (int r = 0; r < E; r++) key[r] = ((other[r] < key[r]) == keep_min) ? other[r] : key[r];
    } else {
        const bool lane_ascending = K >= LANES * E || (g & (K / E)) == 0;  // used when K >= E
#pragma unroll
        for (int r = 0; r < E; r++) {
            if ((r & J) == 0) {
                const bool ascending = K < E ? (r & K) == 0 : lane_ascending;
                const u64 a = key[r], b = key[r | J];
                const bool swap = (a > b) == ascending;
                key[r] = swap ? b : a;
                key[r | J] = swap ? a : b;
            }
        }
    }
    if constexpr (J > 1) wave_bitonic_stage<E, LANES, K, J / 2>(key, g, xbuf);
}

template <int E, int LANES, int K>
__device__ __forceinline__ void wave_bitonic_merge(u64 (&key)[E], int g, u64 *xbuf) {
    wave_bitonic_stage<E, LANES, K, K / 2>(key, g, xbuf);
    if constexpr (K < LANES * E) wave_bitonic_merge<E, LANES, 2 * K>(key, g, xbuf);
}

// g: the lane (LANES = 64) or the thread index in a workgroup of 256 whose four waves all make this call (LANES = 256)
template <int E, int LANES = 64>
__device__ __forceinline__ void wave_sort_segment(const KeySource &src, uint32_t first, uint32_t *__restrict__ seg, int n, int g, u64 *xbuf = nullptr) {
    u64 key[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int i = r * LANES + g;  // any assignment of elements to slots will do before sorting
        key[r] = i < n ? src.load(first + i) : ~0ull;
    }
    wave_bitonic_merge<E, LANES, 2>(key, g, xbuf);
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = g * E + r;
        if (e < n) seg[e] = (uint32_t)key[r];
    }
}

constexpr int CLASS_W = 1024;  // longest segment one wave sorts in registers (16 keys per lane)
constexpr int CLASS_L = 2048;  // longest segment a workgroup of the same launch sorts in (static) LDS: 16 KB
constexpr int CLASS_B = 16384; // longest segment the separate 1024-thread kernel sorts in (dynamic) LDS
constexpr int SORT_WAVES = 4;  // tiles per workgroup (16 were measured: the same at 8160 tiles, twice the time at 1848 -- too few workgroups)
constexpr int LONG_WGS = 256;  // workgroups at the end of the grid that take the queue of over-long segments
constexpr int CLASS_G = 2048;  // longest segment a workgroup's four waves sort together in registers (8 keys per lane; 16 per lane for 4096 entries would take the whole kernel from 69 to 97 registers)
static_assert(CLASS_L * sizeof(u64) >= 4 * 8 * 64 * sizeof(u64), "the four waves' exchange buffer is the in-launch LDS sort's array");

// `count` queued segments, one workgroup per segment in turn: in LDS up to `lds_cap` entries, in place on
// global memory beyond (64-bit key scratch).  Only segments of more than `take_from` and at most `take_upto` entries
// (other launches take the rest).
__device__ __forceinline__ void sort_queued_segments(u64 *s_keys, int lds_cap, int take_from, int take_upto, const uint2 *__restrict__ ranges, const KeySource &src,
                                                     uint32_t *__restrict__ point_list, int count, const uint32_t *__restrict__ queue, u64 *keys,
                                                     int first, int stride) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int q = first; q < count; q += stride) {
        const uint2 r = ranges[queue[q]];
        const int n = (int)(r.y - r.x);
        if (n <= take_from || n > take_upto) continue;
        if (n <= lds_cap) {
            for (int i = tid; i < n; i += nt) s_keys[i] = src.load(r.x + i);
            __syncthreads();
            bitonic_sort_ascending<false>(s_keys, n, tid, nt);
            __syncthreads();
            for (int i = tid; i < n; i += nt) point_list[r.x + i] = (uint32_t)s_keys[i];
        } else {
            volatile u64 *g = keys + r.x;
            if (!src.keys) {  // otherwise the keys are already in place
                for (int i = tid; i < n; i += nt) g[i] = src.load(r.x + i);
                __threadfence_block();
                __syncthreads();
            }
            bitonic_sort_ascending<true>(g, n, tid, nt);
            for (int i = tid; i < n; i += nt) point_list[r.x + i] = (uint32_t)g[i];
        }
        __syncthreads();
    }
}

// One wave per tile (four tiles per workgroup); segments of more than CLASS_W entries are left to whole workgroups:
//  * counting path (QUEUED): the tile-range scan has queued them already (binning.hip) and the LAST LONG_WGS workgroups
//    of this very launch take them -- no launch of their own on every frame's critical path for a queue that is nearly
//    always empty -- in LDS up to CLASS_L entries, in place on global memory beyond.  When the previous frame on this
//    device had such segments the host puts group_wgs workgroups at the HEAD of this launch (a segment of up to CLASS_G
//    entries each, sorted by the four waves together) and a 1024-thread kernel on 128 KiB of LDS behind it, and tells the
//    last workgroups which lengths are still theirs (take_from, take_upto).  A hint: results never depend on it;
//  * radix path: queued here with one atomic each, sorted by the launch that follows.
// tile_order (small grids, or null): the tiles by descending list length -- the waves of a workgroup then sort
// segments of similar length, and the long ones start first.  Only tiles [t_lo, t_hi) are looked at (the band being
// rendered).
#ifndef LVDGS_SORT_SOLO
#define LVDGS_SORT_SOLO 2   // A/B builds: 0 = one wave per segment up to 1024 entries whatever the grid; 1 = a workgroup per segment of 513-1024 entries on small grids; 2 = of 257-1024
#endif
#ifndef LVDGS_SORT_OCC
#define LVDGS_SORT_OCC 1
#endif
template <bool QUEUED>
__device__ __forceinline__ void tile_depth_sort_wave_body(const uint2 *__restrict__ ranges, int t_lo, int t_hi, const KeySource src,
                                                          uint32_t *__restrict__ point_list, uint32_t *queue_count,
                                                          uint32_t *__restrict__ queue, const uint32_t *__restrict__ tile_order,
                                                          u64 *keys, int group_wgs, int solo_wgs, int sort_wgs, int take_from, int take_upto) {
    __shared__ u64 s_long[QUEUED ? CLASS_L : 1];
    if (QUEUED && (int)blockIdx.x < group_wgs) {
        // the FIRST group_wgs workgroups (the host launches them when the previous frame queued segments): a queued segment of
        // up to CLASS_G entries per workgroup in turn, its four waves sorting it together
        const int count = (int)*queue_count;
        for (int q = blockIdx.x; q < count; q += group_wgs) {
            const uint2 r = ranges[queue[q]];
            const int n = (int)(r.y - r.x);
            if (n > CLASS_W && n <= CLASS_G) wave_sort_segment<CLASS_G / 256, 256>(src, r.x, point_list + r.x, n, (int)threadIdx.x, s_long);
        }
        return;
    }
    if (QUEUED && (int)blockIdx.x < group_wgs + solo_wgs) {
        // small grids (solo_wgs = the band's tiles, else 0): a workgroup per tile for the segments of 257-1024 entries, its four
        // waves sorting together with 2 or 4 keys per lane -- a KITTI frame's 800 tiles of 513-1024 entries are otherwise one wave's
        // 3 800-instruction chain each on a chip with more SIMDs than the frame has tiles (tile sort 18.6 -> 14.0 us; with the
        // 257-512 class too: 12.6; 640x480 / 100 k Gaussians, whose lists are 130-310 entries: 10.8 -> 8.5); the workgroups of the
        // other tiles leave at once
        const int slot = (int)blockIdx.x - group_wgs;
        const uint2 r = ranges[tile_order ? (int)tile_order[slot] : t_lo + slot];
        const int n = (int)(r.y - r.x);
        if (n > 512 && n <= CLASS_W) wave_sort_segment<CLASS_W / 256, 256>(src, r.x, point_list + r.x, n, (int)threadIdx.x, s_long);
        else if (LVDGS_SORT_SOLO > 1 && n > 256 && n <= 512) wave_sort_segment<2, 256>(src, r.x, point_list + r.x, n, (int)threadIdx.x, s_long);
        return;
    }
    const int block = (int)blockIdx.x - (QUEUED ? group_wgs + solo_wgs : 0);
    if (QUEUED && block >= sort_wgs) {
        sort_queued_segments(s_long, CLASS_L, take_from, take_upto, ranges, src, point_list, (int)*queue_count, queue, keys,
                             block - sort_wgs, LONG_WGS);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int slot = block * SORT_WAVES + (threadIdx.x >> 6);
    if (slot >= t_hi - t_lo) return;
    const int tile = tile_order ? (int)tile_order[slot] : t_lo + slot;
    const uint2 r = ranges[tile];
    const int n = (int)(r.y - r.x);
    if (n < 2) {
        if (n == 1 && src.keys && lane == 0) point_list[r.x] = (uint32_t)src.keys[r.x];  // nothing to sort, but the id must land
        return;
    }
    if (n <= 256) wave_sort_segment<4>(src, r.x, point_list + r.x, n, lane);
    else if (n <= 512) { if (!QUEUED || solo_wgs == 0 || LVDGS_SORT_SOLO < 2) wave_sort_segment<8>(src, r.x, point_list + r.x, n, lane); }
    else if (n <= CLASS_W) { if (!QUEUED || solo_wgs == 0) wave_sort_segment<16>(src, r.x, point_list + r.x, n, lane); }
    else if (!QUEUED && lane == 0) queue[atomicAdd(queue_count, 1u)] = (uint32_t)tile;
}
template <bool QUEUED>
__global__ void __launch_bounds__(64 * SORT_WAVES, LVDGS_SORT_OCC) tile_depth_sort_wave_kernel(const uint2 *__restrict__ ranges, int t_lo, int t_hi, KeySource src,
                                                                    uint32_t *__restrict__ point_list, uint32_t *queue_count,
                                                                    uint32_t *__restrict__ queue, const uint32_t *__restrict__ tile_order,
                                                                    u64 *keys, int group_wgs, int solo_wgs, int sort_wgs, int take_from, int take_upto) {
    tile_depth_sort_wave_body<QUEUED>(ranges, t_lo, t_hi, src, point_list, queue_count, queue, tile_order, keys, group_wgs, solo_wgs, sort_wgs, take_from, take_upto);
}
// lvdgs_forward_batch: the tile sorts of several views in one launch (blockIdx.y: the view; counting path only)
struct TileSortView { const uint2 *ranges; KeySource src; uint32_t *point_list, *queue_count, *queue; const uint32_t *tile_order; u64 *keys; };
struct TileSortBatch { TileSortView v[FWD_BATCH_VIEWS]; };
__global__ void __launch_bounds__(64 * SORT_WAVES, LVDGS_SORT_OCC) tile_depth_sort_wave_batch_kernel(TileSortBatch b, int t_lo, int t_hi, int group_wgs, int solo_wgs,
                                                                                                 int sort_wgs, int take_from, int take_upto) {
    const TileSortView &v = b.v[blockIdx.y];
    tile_depth_sort_wave_body<true>(v.ranges, t_lo, t_hi, v.src, v.point_list, v.queue_count, v.queue, v.tile_order, v.keys, group_wgs, solo_wgs, sort_wgs,
                                    take_from, take_upto);
}

// queued segments: in 128 KiB of LDS up to CLASS_B entries, in place on global memory beyond; segments of up to
// `done_below` entries were sorted by the launches before
__global__ void __launch_bounds__(1024) tile_depth_sort_long_kernel(const uint2 *__restrict__ ranges, KeySource src,
                                                                    uint32_t *__restrict__ point_list, const uint32_t *long_count,
                                                                    const uint32_t *__restrict__ long_tiles, u64 *keys, int done_below) {
    extern __shared__ u64 s_dyn[];
    const int tid = threadIdx.x;
    const int count = (int)*long_count;
    for (int q = blockIdx.x; q < count; q += gridDim.x) {
        const uint2 r = ranges[long_tiles[q]];
        const int n = (int)(r.y - r.x);
        if (n <= done_below) continue;
        if (n <= CLASS_B) {
            for (int i = tid; i < n; i += 1024) s_dyn[i] = src.load(r.x + i);
            __syncthreads();
            bitonic_sort_ascending<false>(s_dyn, n, tid, 1024);
            __syncthreads();
            for (int i = tid; i < n; i += 1024) point_list[r.x + i] = (uint32_t)s_dyn[i];
        } else {
            volatile u64 *g = keys + r.x;
            if (!src.keys) {  // otherwise the keys are already in place
                for (int i = tid; i < n; i += 1024) g[i] = src.load(r.x + i);
                __threadfence_block();
                __syncthreads();
            }
            bitonic_sort_ascending<true>(g, n, tid, 1024);
            for (int i = tid; i < n; i += 1024) point_list[r.x + i] = (uint32_t)g[i];
        }
        __syncthreads();
    }
}

}  // namespace

int tile_sort_wave_limit() { return CLASS_W; }

int launch_tile_depth_sort(const ImageView &im, int num_tiles, int t_lo, int t_hi, const float *rec, uint32_t *point_list, void *keys64,
                           bool keys_ready, int longest_expected, int queue_expected, int dbg, hipStream_t s) {
    if (num_tiles == 0 || t_hi <= t_lo) return LVDGS_OK;
    static unsigned char lds_done[16];
    if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&tile_depth_sort_long_kernel), CLASS_B * 8, lds_done)) return e;
    const KeySource src{rec, point_list, keys_ready ? (const u64 *)keys64 : nullptr};
    const int sort_wgs = cdiv(t_hi - t_lo, SORT_WAVES);
    // longest_expected: the longest segment the previous frame on this device queued (0: none; negative: unknown, assume the
    // worst).  Segments that turn up against the expectation are still sorted -- by the launch's own last workgroups.
    // grouped: workgroups at the head of the launch for the queue the previous frame leads to expect (one segment each, a quarter
    // more than it had; whatever the queue turns out to hold is sorted all the same, by them in turn or by the launch's last workgroups)
    const bool grouped = keys_ready && longest_expected > CLASS_W, big = !keys_ready || longest_expected > CLASS_G || longest_expected < 0;
    const int group_wgs = grouped ? min(8192, max(64, (int)((int64_t)queue_expected * 5 / 4))) : 0;
    const int solo_wgs = (LVDGS_SORT_SOLO && keys_ready && tile_order_in_use(num_tiles)) ? t_hi - t_lo : 0;
    {
        ProfScope ps("tile_sort", s);
        if (keys_ready)   // counting path: the queue is there already
            hipLaunchKernelGGL(tile_depth_sort_wave_kernel<true>, dim3(group_wgs + solo_wgs + sort_wgs + LONG_WGS), dim3(64 * SORT_WAVES), 0, s, (const uint2 *)im.ranges, t_lo, t_hi, src,
                               point_list, im.long_count, im.long_tiles, tile_order_in_use(num_tiles) ? im.long_tiles + num_tiles : nullptr,
                               (u64 *)keys64, group_wgs, solo_wgs, sort_wgs, grouped ? CLASS_G : CLASS_W, big ? CLASS_G : 0x7fffffff);
        else
            hipLaunchKernelGGL(tile_depth_sort_wave_kernel<false>, dim3(sort_wgs), dim3(64 * SORT_WAVES), 0, s, (const uint2 *)im.ranges, t_lo, t_hi, src,
                               point_list, im.long_count, im.long_tiles, (const uint32_t *)nullptr, (u64 *)keys64, 0, 0, sort_wgs, 0, 0);
        LVDGS_LAUNCH_CHECK("tile_sort", dbg, s);
    }
    if (big) {
        ProfScope ps("tile_sort_long", s);
        hipLaunchKernelGGL(tile_depth_sort_long_kernel, dim3(256), dim3(1024), CLASS_B * 8, s, (const uint2 *)im.ranges, src, point_list,
                           (const uint32_t *)im.long_count, (const uint32_t *)im.long_tiles, (unsigned long long *)keys64, keys_ready ? CLASS_G : 0);
        LVDGS_LAUNCH_CHECK("tile_sort_long", dbg, s);
    }
    return LVDGS_OK;
}

int launch_tile_depth_sort_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, const BinView *b, int n,
                                 int longest_expected, int queue_expected, hipStream_t s) {
    const bool super = super_tiles_in_use(*a[0]);   // two-level grouping: the super-tile lists are what is sorted, into b.tile_keys (binning.hip)
    const int gx = cdiv(a[0]->image_width, TILE), num_tiles = super ? super_tiles_of(a[0]->image_width, a[0]->image_height) : gx * cdiv(a[0]->image_height, TILE);
    int row0, row1;
    tile_row_band(*a[0], &row0, &row1);
    const int t_lo = super ? 0 : row0 * gx, t_hi = super ? num_tiles : row1 * gx;
    if (num_tiles == 0 || t_hi <= t_lo || n == 0) return LVDGS_OK;
    static unsigned char lds_done[16];
    if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&tile_depth_sort_long_kernel), CLASS_B * 8, lds_done)) return e;
    // (as launch_tile_depth_sort on the counting path: the expectation is the window's -- the longest queued segment / the longest
    // queue any recent view on this device had)
    const int sort_wgs = cdiv(t_hi - t_lo, SORT_WAVES);
    const bool grouped = longest_expected > CLASS_W, big = longest_expected > CLASS_G || longest_expected < 0;
    const int group_wgs = grouped ? min(8192, max(64, (int)((int64_t)queue_expected * 5 / 4))) : 0;
    const int solo_wgs = (LVDGS_SORT_SOLO && tile_order_in_use(num_tiles)) ? t_hi - t_lo : 0;
    TileSortBatch batch{};
    for (int k = 0; k < n; k++)
        batch.v[k] = super ? TileSortView{(const uint2 *)w[k].super.ranges, KeySource{g[k].rec, b[k].tile_keys, (const u64 *)w[k].keys}, b[k].tile_keys, w[k].super.long_count,
                                          w[k].super.long_tiles, tile_order_in_use(num_tiles) ? w[k].super.long_tiles + num_tiles : nullptr, (u64 *)w[k].keys}
                           : TileSortView{(const uint2 *)im[k].ranges, KeySource{g[k].rec, b[k].point_list, (const u64 *)w[k].keys}, b[k].point_list, im[k].long_count,
                                          im[k].long_tiles, tile_order_in_use(num_tiles) ? im[k].long_tiles + num_tiles : nullptr, (u64 *)w[k].keys};
    {
        ProfScope ps("tile_sort", s);
        hipLaunchKernelGGL(tile_depth_sort_wave_batch_kernel, dim3(group_wgs + solo_wgs + sort_wgs + LONG_WGS, n), dim3(64 * SORT_WAVES), 0, s, batch, t_lo, t_hi,
                           group_wgs, solo_wgs, sort_wgs, grouped ? CLASS_G : CLASS_W, big ? CLASS_G : 0x7fffffff);
        LVDGS_LAUNCH_CHECK("tile_sort (batch)", a[0]->debug, s);
    }
    if (big)   // segments beyond what the launch above takes (rare: a kernel per view, as in the single-view call)
        for (int k = 0; k < n; k++) {
            ProfScope ps("tile_sort_long", s);
            hipLaunchKernelGGL(tile_depth_sort_long_kernel, dim3(256), dim3(1024), CLASS_B * 8, s, batch.v[k].ranges, batch.v[k].src, batch.v[k].point_list,
                               (const uint32_t *)batch.v[k].queue_count, (const uint32_t *)batch.v[k].queue, (unsigned long long *)w[k].keys, CLASS_G);
            LVDGS_LAUNCH_CHECK("tile_sort_long", a[0]->debug, s);
        }
    return LVDGS_OK;
}

}  // namespace lvdgs
