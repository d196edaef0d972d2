// Pieces the grouping-by-counting kernels share (binning.hip, and the projection kernel that counts as it goes,
// preprocess.hip).
#pragma once
#include "common.hpp"

namespace lvdgs {

// A workgroup owns a chunk of consecutive Gaussians (THREADS x PER) and keeps one counter per tile in LDS.
// Few, large chunks keep the [chunk][tile] count matrix and its scan small (2 M Gaussians: 489 chunks of 4096); many, small
// ones spread the projection, counting and scattering over the chip (500 k Gaussians: 489 chunks of 1024 for 256 CUs).
// Measured (same box, round 3, ms per tracking iteration with chunks of 1024 / 2048 / 4096): KITTI geometry (200 k) 0.2557 /
// 0.2683 / -, config 3 (500 k) 0.5587 / 0.5638 / - (with the scatter's chunks dealt XCD-contiguously; 0.5667 / 0.5678
// before), 2 M / 1920x1280 - / 1.472 / 1.477.
// Small maps take smaller workgroups (round 4): 100 k Gaussians in chunks of 1024 are 98 workgroups for 256 CUs, and when the
// Gaussians are the large flat ones real maps are made of -- 80 listed tiles each on the opaque-surface workload -- the
// projection's reach tests, the counting and the scatter are all work per Gaussian on a chip 60 % idle.  The chunk is chosen so
// that a map of up to 2^19 Gaussians makes 256-512 workgroups (the scan of the count matrix holds up to 512 rows in registers).
constexpr int GROUP_THREADS_MAX = 1024;
#ifndef LVDGS_GROUP_HELPER_THREADS
#define LVDGS_GROUP_HELPER_THREADS 1024   // A/B builds: 256 = no helper waves
#endif
constexpr int GROUP_HELPER_THREADS = LVDGS_GROUP_HELPER_THREADS;   // threads launched per workgroup of the 256-Gaussian shape
template <int V> struct template_int { static constexpr int value = V; };
struct GroupShape { int threads, per; };   // Gaussians per workgroup = threads x per
__host__ __device__ constexpr GroupShape group_shape_default(int N) {
    return N <= (1 << 17) ? GroupShape{256, 1} : N <= (1 << 18) ? GroupShape{512, 1} : N <= (1 << 19) ? GroupShape{1024, 1}
         : N <= (1 << 20) ? GroupShape{1024, 2} : GroupShape{1024, 4};
}
GroupShape group_shape_for(int N);   // binning.hip: the default, or a -DLVDGS_GROUP_CHUNK=256/512/1024/2048/4096 build (A/B measurements)
// the instantiations of the grouping kernels: f(THREADS launched, OWNERS = threads that hold a Gaussian, PER as integral constants,
// index of the instantiation).  The smallest shape launches 1024 threads for its 256 Gaussians: the twelve waves without a
// Gaussian of their own are HELPERS in the walks over rectangles of more than 64 tiles (for_each_pair_of_rect_wg).
template <typename F>
inline int group_dispatch(GroupShape g, F &&f) {
    template_int<1> one; template_int<2> two; template_int<4> four;
    if (g.threads == 256) return f(template_int<GROUP_HELPER_THREADS>{}, template_int<256>{}, one, 0);
    if (g.threads == 512) return f(template_int<512>{}, template_int<512>{}, one, 1);
    if (g.per == 1) return f(template_int<1024>{}, template_int<1024>{}, one, 2);
    if (g.per == 2) return f(template_int<1024>{}, template_int<1024>{}, two, 3);
    return f(template_int<1024>{}, template_int<1024>{}, four, 4);
}
constexpr int GROUP_SHAPES = 5;
constexpr int GROUP_MAX_TILES = 16384;  // 64 KiB of LDS counters
constexpr int GROUP_BIG_RECT = 64;
static_assert(GROUP_BIG_RECT == RECT_MASK_TILES, "rectangles walked by the whole wave are the ones without a tile mask");

// visit(tile, id, payload) for every listed (Gaussian, tile) pair of the rectangles the wave's lanes hold (r: the lane's
// rect[], all zero for a lane without a Gaussian; payload: a word of the lane's Gaussian that travels with its pairs -- the
// scatter's depth bits).  Must be reached by all lanes of the wave: rectangles of more than 64 tiles (one mask bit per block of
// tiles, common.hpp: RectBlocks) are walked by the whole wave, so that one screen-filling Gaussian does not serialise thousands
// of LDS atomics on one lane -- the owner's payload is handed to the 64 lanes with its rectangle (until round 4 the scatter
// re-read the depth from memory in every one of those rounds: a dependent load per 64 pairs).
template <typename F>
__device__ __forceinline__ void for_each_pair_of_rect(const uint4 r, int i, int gx, uint32_t payload, F visit) {
    const int lane = threadIdx.x & 63;
    const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
    const int w = x1 - x0, area = w * (y1 - y0);
    if (area > 0 && area <= GROUP_BIG_RECT) {
        // only the tiles the Gaussian can reach (common.hpp: rect_keeps); bit t of the mask is tile t of the rectangle
        uint64_t m = (uint64_t)r.z | ((uint64_t)r.w << 32);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++, m >>= 1)
                if (m & 1ull) visit(y * gx + x, (uint32_t)i, payload);
    }
    uint64_t big = __ballot(area > GROUP_BIG_RECT);
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const int bx0 = __shfl(x0, src, 64), by0 = __shfl(y0, src, 64), bw = __shfl(w, src, 64), barea = __shfl(area, src, 64);
        const uint32_t bi = (uint32_t)__shfl(i, src, 64), bp = (uint32_t)__shfl((int)payload, src, 64);
        const uint64_t bm = (uint64_t)(uint32_t)__shfl((int)r.z, src, 64) | ((uint64_t)(uint32_t)__shfl((int)r.w, src, 64) << 32);
        const RectBlocks g(bw, barea / bw);
        const float inv_w = __builtin_amdgcn_rcpf((float)bw), inv_gw = __builtin_amdgcn_rcpf((float)g.bw), inv_gh = __builtin_amdgcn_rcpf((float)g.bh);
        for (int t = lane; t < barea; t += 64) {
            const int ty = div_by(t, bw, inv_w), tx = t - ty * bw;
            const int block = div_by(ty, g.bh, inv_gh) * 8 + div_by(tx, g.bw, inv_gw);   // = g.block_of(tx, ty)
            if ((bm >> block) & 1ull) visit((by0 + ty) * gx + bx0 + tx, bi, bp);
        }
    }
}

// The same for a workgroup with HELPER waves (more threads than Gaussians: maps of up to 2^17 Gaussians, whose 256-Gaussian chunks
// are four waves each -- 100 k Gaussians are 1563 waves for 1024 SIMDs, and when they are large, a wave's walk over its
// rectangles of hundreds of tiles, one Gaussian after the other with a returning LDS atomic per round, is part of what the kernel's
// time is: scatter 98.6 -> 91.1 us, projection + counting 62.0 -> 57.4 us on the opaque-surface workload with the helpers, same box).  Rectangles of up to 64 tiles are walked by their owner as
// above; the larger ones are queued in LDS and taken by ALL the workgroup's waves in turn.  Reached by every thread of the
// workgroup (one barrier); the queue holds one item per owner and is used ONCE per kernel (one Gaussian per owner thread): the
// caller zeroes q.count in front of a barrier of its own.
struct BigRectQueue {
    uint32_t count;
    struct Item { uint32_t rx, ry, mlo, mhi, id, payload; } items[256];
};
template <typename F>
__device__ __forceinline__ void for_each_pair_of_rect_wg(const uint4 r, int i, int gx, uint32_t payload, BigRectQueue &q, F visit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
    const int w = x1 - x0, area = w * (y1 - y0);
    if (area > GROUP_BIG_RECT) {
        const uint32_t at = atomicAdd(&q.count, 1u);
        q.items[at] = BigRectQueue::Item{r.x, r.y, r.z, r.w, (uint32_t)i, payload};
    }
    if (area > 0 && area <= GROUP_BIG_RECT) {
        uint64_t m = (uint64_t)r.z | ((uint64_t)r.w << 32);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++, m >>= 1)
                if (m & 1ull) visit(y * gx + x, (uint32_t)i, payload);
    }
    __syncthreads();
    const int n = (int)q.count;
    for (int it = wave; it < n; it += waves) {
        const BigRectQueue::Item b = q.items[it];
        const int bx0 = (int)(b.rx & 0xffffu), by0 = (int)(b.ry & 0xffffu), bw = (int)(b.rx >> 16) - bx0, bh = (int)(b.ry >> 16) - by0;
        const int barea = bw * bh;
        const uint64_t bm = (uint64_t)b.mlo | ((uint64_t)b.mhi << 32);
        const RectBlocks g(bw, bh);
        const float inv_w = __builtin_amdgcn_rcpf((float)bw), inv_gw = __builtin_amdgcn_rcpf((float)g.bw), inv_gh = __builtin_amdgcn_rcpf((float)g.bh);
        for (int t = lane; t < barea; t += 64) {
            const int ty = div_by(t, bw, inv_w), tx = t - ty * bw;
            const int block = div_by(ty, g.bh, inv_gh) * 8 + div_by(tx, g.bw, inv_gw);
            if ((bm >> block) & 1ull) visit((by0 + ty) * gx + bx0 + tx, b.id, b.payload);
        }
    }
}

// Two-level grouping (binning.hip): a Gaussian's tile rectangle + kept-tile mask -> its rectangle in super-tile units + which of those
// super-tiles hold a listed tile (same encoding; a rectangle of more than 64 super-tiles lists them all).
__device__ __forceinline__ uint4 super_rect_of(const uint4 r) {
    const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
    const int w = x1 - x0, h = y1 - y0, area = w * h;
    const uint64_t m = (uint64_t)r.z | ((uint64_t)r.w << 32);
    if (area <= 0 || m == 0ull) return make_uint4(0u, 0u, 0u, 0u);
    static_assert(SUPER == 4, "the shifts below");
    const int sx0 = x0 >> 2, sx1 = ((x1 - 1) >> 2) + 1, sy0 = y0 >> 2, sy1 = ((y1 - 1) >> 2) + 1;
    const int ws = sx1 - sx0, area_s = ws * (sy1 - sy0);
    uint64_t ms = 0ull;
    if (area_s > RECT_MASK_TILES) {
        ms = ~0ull;   // (a rectangle of more than 64 super-tiles -- over 1000 tiles: every super-tile of it is listed; the expansion drops what no tile keeps)
    } else if (area <= RECT_MASK_TILES) {
        // a bit per tile: row by row, the row's kept tiles folded into the super-columns they lie in (no walk over the set bits: a lane's
        // 64-iteration loop of dependent bit tricks was what the first version of this kernel spent 39 us on)
        const uint64_t rowmask = w < 64 ? ((1ull << w) - 1ull) : ~0ull;
        for (int ty = 0; ty < h; ty++) {
            const uint64_t row = (m >> (ty * w)) & rowmask;
            if (!row) continue;
            uint64_t cols = 0ull;
            for (int c = 0; c < ws; c++) {
                const int lo = max(0, ((sx0 + c) << 2) - x0), hi = min(w, ((sx0 + c + 1) << 2) - x0);   // tile columns [lo, hi) of super-column c
                if (row & (((1ull << (hi - lo)) - 1ull) << lo)) cols |= 1ull << c;
            }
            ms |= cols << ((((y0 + ty) >> 2) - sy0) * ws);
        }
    } else {
        // a bit per block of tiles (8 x 8 grid): the super-columns every block column covers, then block row by block row
        const RectBlocks g(w, h);
        uint64_t colbits[8];
#pragma unroll
        for (int bc = 0; bc < 8; bc++) {
            const int wd = g.width(bc), bx0 = x0 + bc * g.bw;
            const int ca = (bx0 >> 2) - sx0, cb = ((bx0 + wd - 1) >> 2) - sx0;
            colbits[bc] = wd > 0 ? (((2ull << (cb - ca)) - 1ull) << ca) : 0ull;
        }
#pragma unroll
        for (int br = 0; br < 8; br++) {
            const int hg = g.height(8 * br);
            const uint32_t bits = (uint32_t)(m >> (8 * br)) & 0xffu;
            if (hg <= 0 || !bits) continue;
            uint64_t cols = 0ull;
#pragma unroll
            for (int bc = 0; bc < 8; bc++) cols |= ((bits >> bc) & 1u) ? colbits[bc] : 0ull;
            const int by0 = y0 + br * g.bh;
            for (int cy = (by0 >> 2) - sy0; cy <= ((by0 + hg - 1) >> 2) - sy0; cy++) ms |= cols << (cy * ws);
        }
    }
    return make_uint4((uint32_t)sx0 | ((uint32_t)sx1 << 16), (uint32_t)sy0 | ((uint32_t)sy1 << 16), (uint32_t)ms, (uint32_t)(ms >> 32));
}

// Workgroup b of n -> the chunk it takes, such that the workgroups of one XCD (b mod 8: workgroups are dealt round-robin over
// the 8 XCDs) take CONSECUTIVE chunks.  Consecutive chunks place their pairs of a tile in neighbouring 8-byte slots of
// the tile's segment; written from one XCD at about the same time, the slots of a 64-byte line meet in that XCD's L2 and
// leave it as one full line instead of eight partial ones from eight L2s.
__device__ __forceinline__ int xcd_contiguous_chunk(int b, int n) {
    const int x = b & 7, k = b >> 3, q = n >> 3, r = n & 7;
    return x * q + min(x, r) + k;
}

// Exclusive scan of one value per thread over a workgroup of THREADS threads (wave shifts, then the up to 16 wave totals by
// wave 0: two barriers); *total = sum over the workgroup.  s_scan: 33 words.
template <int THREADS>
__device__ __forceinline__ uint32_t scan_workgroup(uint32_t v, uint32_t *s_scan, uint32_t *total) {
    constexpr int WAVES = THREADS / 64;
    static_assert(WAVES >= 1 && WAVES <= 16 && THREADS % 64 == 0, "up to 16 whole waves");
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
        uint32_t w = lane < WAVES ? s_scan[lane] : 0u;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t x = (uint32_t)__shfl_up((int)w, off, 64);
            if (lane >= off) w += x;
        }
        if (lane < 16) s_scan[16 + lane] = w;   // inclusive over the waves
    }
    __syncthreads();
    *total = s_scan[16 + WAVES - 1];
    return inc - v + (wave ? s_scan[16 + wave - 1] : 0u);
}

}  // namespace lvdgs
