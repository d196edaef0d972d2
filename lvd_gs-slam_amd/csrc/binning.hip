// (Gaussian, tile) pair emission in depth-rank order, and per-tile ranges of the sorted list.
#include "common.hpp"
#include "device_utils.hpp"

namespace lvdgs {

namespace {

// One lane per Gaussian, in id order.  A Gaussian's pairs occupy [slot_base[i], slot_base[i] + tiles) of the
// unsorted pair list, tiles in row-major order of its rectangle.
__global__ void __launch_bounds__(256) emit_pairs_kernel(int N, int gx, int gy, const uint32_t *__restrict__ slot_base,
                                                         const uint32_t *__restrict__ tiles_touched, const float *__restrict__ rec,
                                                         uint32_t *__restrict__ tile_keys, uint32_t *__restrict__ ids,
                                                         uint32_t capacity) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (uint32_t)N) return;
    const uint32_t cnt = tiles_touched[id];
    if (cnt == 0) return;
    const uint32_t first = slot_base[id];
    const float4 *r4 = reinterpret_cast<const float4 *>(rec + (size_t)id * REC_FLOATS);
    const float4 r0 = r4[0];
    const int rad = __float_as_int(r4[2].w);
    const float px = r0.x, py = r0.y;
    int x0 = (int)((px - (float)rad) / (float)TILE), y0 = (int)((py - (float)rad) / (float)TILE);
    int x1 = (int)((px + (float)rad + (float)(TILE - 1)) / (float)TILE);
    int y1 = (int)((py + (float)rad + (float)(TILE - 1)) / (float)TILE);
    x0 = min(gx, max(0, x0)); x1 = min(gx, max(0, x1));
    y0 = min(gy, max(0, y0)); y1 = min(gy, max(0, y1));
    uint32_t o = first;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            if (o < capacity) {  // pairs beyond the caller's capacity are dropped (the caller is told and re-runs)
                tile_keys[o] = (uint32_t)(y * gx + x);
                ids[o] = id;
            }
            o++;
        }
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

int launch_emit_pairs(const lvdgs_args &a, const GeomView &g, uint32_t *tile_keys, uint32_t *ids, int64_t capacity, hipStream_t s) {
    const int N = a.num_gaussians;
    if (N == 0) return LVDGS_OK;
    const int gx = (a.image_width + TILE - 1) / TILE, gy = (a.image_height + TILE - 1) / TILE;
    ProfScope ps("emit_pairs", s);
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, N, gx, gy, g.slot_base, g.tiles_touched,
                       g.rec, tile_keys, ids, (uint32_t)capacity);
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
