// Shared declarations of the gfx950 rasterizer library (see include/lvdgs.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lvdgs.h"

namespace lvdgs {

// ---- constants of the splatting algorithm (UNPINNED against upstream) ----
constexpr int TILE = 16;                 // tile edge in pixels; one 256-thread workgroup per tile
constexpr float NEAR_CULL = 0.2f;        // view z at or below this is culled
constexpr float HOMOG_EPS = 0.0000001f;  // added to w before the perspective divide
constexpr float FOV_GUARD = 1.3f;        // x/z, y/z clamp (times tan(fov/2)) in the EWA Jacobian
constexpr float LOWPASS = 0.3f;          // added to the 2D covariance diagonal
constexpr float LAMBDA_FLOOR = 0.1f;     // floor under the eigenvalue discriminant
constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;
constexpr float T_TOUCH = 0.5f;

constexpr int WAVE = 64;
// Per-Gaussian 2D record: x, y, conic a, b | conic c, opacity, r, g | b, view depth, (unused), radius | tile rectangle and
// kept-tile mask (a copy of rect[i], for the backward pass's slot arithmetic).  64 bytes, 64-byte aligned: a gathered
// record lies in one cache line (the 48-byte record of earlier rounds straddled two lines for a third of the ids, and
// the rectangle was a second gather).  The forward blend reads the first three quarters only.
#ifndef LVDGS_REC_FLOATS
#define LVDGS_REC_FLOATS 16
#endif
constexpr int REC_FLOATS = LVDGS_REC_FLOATS;
static_assert(REC_FLOATS == 12 || REC_FLOATS == 16, "12: the record without the rectangle copy (A/B builds)");
constexpr int PAIR_FLOATS = 10;  // per-(Gaussian, tile) partial gradient record: 40 bytes (48 with two pad floats until round 3)
constexpr int PAIR_FLOATS_POSE = 6;  // ... of a pose-only backward (LVDGS_FLAG_POSE_ONLY): d/d(2-D mean, conic, view depth), 24 bytes

constexpr int SORT_THREADS = 256;
constexpr int SORT_IPT = 16;                       // elements per thread per radix pass
constexpr int SORT_CHUNK = SORT_THREADS * SORT_IPT; // elements per workgroup
constexpr int SORT_MAX_BITS = 8;                   // digit width limit: <= 256 bins keeps same-digit runs long enough
                                                   // for the scatter's stores to coalesce

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
// the tile rows a call renders (lvdgs_args.tile_row_begin / _end; both 0 = all of them)
inline void tile_row_band(const lvdgs_args &a, int *row0, int *row1) {
    const int gy = (a.image_height + TILE - 1) / TILE;
    if (a.tile_row_begin == 0 && a.tile_row_end == 0) { *row0 = 0; *row1 = gy; return; }
    *row0 = a.tile_row_begin < 0 ? 0 : (a.tile_row_begin > gy ? gy : a.tile_row_begin);
    *row1 = a.tile_row_end < *row0 ? *row0 : (a.tile_row_end > gy ? gy : a.tile_row_end);
}
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- error plumbing ----
void set_error(const char *fmt, ...);
int check_hip(hipError_t e, const char *what);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per device and kernel (the attribute is per device; one process
// per GPU is the intended use, but a process may still touch several devices)
int allow_dynamic_lds(const void *kernel, int bytes, unsigned char (&done)[16]);

// ---- profiling hooks (no-ops unless lvdgs_profile_enable(1)) ----
struct ProfScope {
    int slot;
    hipStream_t stream;
    ProfScope(const char *name, hipStream_t s);
    ~ProfScope();
};

#define LVDGS_LAUNCH_CHECK(name, dbg, stream)                                   \
    do {                                                                        \
        if (int _e = ::lvdgs::check_hip(hipGetLastError(), name)) return _e;    \
        if (dbg) {                                                              \
            if (int _e = ::lvdgs::check_hip(hipStreamSynchronize(stream), name)) return _e; \
        }                                                                       \
    } while (0)

// Can a Gaussian (mean m, conic a,b,c, opacity op) reach alpha >= 1/255 anywhere in the pixel
// rectangle [x0,x1] x [y0,y1]?  Exact minimum of q(d) = 1/2 (a dx^2 + c dy^2) + b dx dy over the
// rectangle (convex: interior point or one of four clamped edge minima), compared against
// ln(255 op) with a safety margin that covers float rounding of the per-pixel evaluation.
__device__ __forceinline__ bool reaches_rect(float mx, float my, float a, float b, float c, float op, float x0, float y0,
                                             float x1, float y1) {
    if (!(op >= ALPHA_MIN)) return false;
    if (!(a > 0.f && c > 0.f && a * c - b * b > 0.f)) return true;  // not an ellipse: let the pixel test decide
    const float dx_lo = mx - x1, dx_hi = mx - x0, dy_lo = my - y1, dy_hi = my - y0;  // d = mean - pixel
    if (dx_lo <= 0.f && dx_hi >= 0.f && dy_lo <= 0.f && dy_hi >= 0.f) return true;
    const float inv_a = __builtin_amdgcn_rcpf(a), inv_c = __builtin_amdgcn_rcpf(c);
    auto along_y = [&](float dx) {
        const float dy = fminf(fmaxf(-b * dx * inv_c, dy_lo), dy_hi);
        return 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
    };
    auto along_x = [&](float dy) {
        const float dx = fminf(fmaxf(-b * dy * inv_a, dx_lo), dx_hi);
        return 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
    };
    const float qmin = fminf(fminf(along_y(dx_lo), along_y(dx_hi)), fminf(along_x(dy_lo), along_x(dy_hi)));
    const float dxm = fmaxf(fabsf(dx_lo), fabsf(dx_hi)), dym = fmaxf(fabsf(dy_lo), fabsf(dy_hi));
    const float margin = 0.02f + 2e-5f * (a * dxm * dxm + c * dym * dym);
    return qmin <= __logf(op * 255.f) + margin;
}

// reaches_rect() for the tiles of one Gaussian's rectangle, with the per-Gaussian part hoisted: the same bound (exact
// minimum of q over the tile against ln(255 op) plus the rounding margin, here the margin of the whole rectangle), the
// minimum taken over the two edges that face the mean -- a convex q with its minimum at the mean cannot have its
// rectangle minimum on an edge that faces away.
struct TileReach {
    float mx, my, a, b, c, nb_inv_a, nb_inv_c, bound;
    bool never, always;
    __device__ __forceinline__ TileReach(float mx_, float my_, float a_, float b_, float c_, float op, float far_x, float far_y)
        : mx(mx_), my(my_), a(a_), b(b_), c(c_) {
        never = !(op >= ALPHA_MIN);
        always = !(a > 0.f && c > 0.f && a * c - b * b > 0.f);   // not an ellipse: let the pixel test decide
        nb_inv_a = -b * __builtin_amdgcn_rcpf(a); nb_inv_c = -b * __builtin_amdgcn_rcpf(c);
        // far_x, far_y: the largest |mean - pixel| over the rectangle
        bound = __logf(op * 255.f) + 0.02f + 2e-5f * (a * far_x * far_x + c * far_y * far_y);
    }
    // pixel rectangle [x0, x0 + 15] x [y0, y0 + 15]
    __device__ __forceinline__ bool tile(float x0, float y0) const { return rect(x0, y0, (float)(TILE - 1), (float)(TILE - 1)); }
    // pixel rectangle [x0, x0 + w] x [y0, y0 + h] (a block of tiles: the same bound, the rectangle lies inside the one
    // the margin was taken over)
    __device__ __forceinline__ bool rect(float x0, float y0, float w, float h) const {
        const float dx_hi = mx - x0, dx_lo = dx_hi - w, dy_hi = my - y0, dy_lo = dy_hi - h;
        const float dxn = __builtin_amdgcn_fmed3f(0.f, dx_lo, dx_hi), dyn = __builtin_amdgcn_fmed3f(0.f, dy_lo, dy_hi);
        const float dy1 = __builtin_amdgcn_fmed3f(nb_inv_c * dxn, dy_lo, dy_hi);   // along the edge dx = dxn
        const float dx2 = __builtin_amdgcn_fmed3f(nb_inv_a * dyn, dx_lo, dx_hi);   // along the edge dy = dyn
        const float q1 = 0.5f * (a * dxn * dxn + c * dy1 * dy1) + b * dxn * dy1;
        const float q2 = 0.5f * (a * dx2 * dx2 + c * dyn * dyn) + b * dx2 * dyn;
        return (fminf(q1, q2) <= bound || always) && !never;   // (mean inside the rectangle: q1 = q2 = 0)
    }
};

// ---- the tile rectangle of a Gaussian and which of its tiles it can reach ----
// rect[i] = (x0 | x1 << 16, y0 | y1 << 16, mask lo, mask hi): the rectangle of 16x16 tiles covered by the 3-sigma
// radius (what the reference pairs the Gaussian with) and 64 bits that say which of its tiles are LISTED:
//  * rectangles of up to 64 tiles: one bit per tile in row-major order, set when TileReach cannot rule out alpha >= 1/255
//    on some pixel of the tile;
//  * larger rectangles: one bit per BLOCK of tiles -- the rectangle is cut into an 8 x 8 grid of blocks of
//    ceil(w / 8) x ceil(h / 8) tiles (bit 8 * block row + block column; fewer blocks where the sizes do not divide) --
//    set when the Gaussian can reach some pixel of the block; every tile of a kept block is listed.  (Until round 3 such
//    rectangles listed every tile: the large-footprint Gaussians real maps are made of got no culling at all.)
// Only listed (Gaussian, tile) pairs exist downstream; a dropped pair contributes nothing to any pixel, so every output is
// unchanged.  A Gaussian's pairs are numbered 0 .. tiles_touched - 1 (its gradient slots behind slot_base[i]): row-major
// over the kept tiles, resp. kept blocks in bit order and row-major inside a block (rect_rank).
constexpr int RECT_MASK_TILES = 64;
// n / d for 0 <= n < 2^23, 0 < d, with inv_d = 1 / (float)d to an ulp: the float quotient is off by at most one, one
// correction step makes it exact.  (The compiler's unsigned division is some 25 instructions; the walks over large
// rectangles divide three times per pair.)
__device__ __forceinline__ int div_by(int n, int d, float inv_d) {
    int q = (int)((float)n * inv_d);
    const int r = n - q * d;
    q += (r >= d) - (r < 0);
    return q;
}
struct RectBlocks {   // the block grid of a rectangle of more than RECT_MASK_TILES tiles
    int w, h, bw, bh;
    __device__ __forceinline__ RectBlocks(int w_, int h_) : w(w_), h(h_), bw((w_ + 7) >> 3), bh((h_ + 7) >> 3) {}
    __device__ __forceinline__ int block_of(int tx, int ty) const { return (ty / bh) * 8 + tx / bw; }   // tx, ty relative to the rectangle
    __device__ __forceinline__ int width(int b) const { return min(bw, w - (b & 7) * bw); }              // <= 0: no such block
    __device__ __forceinline__ int height(int b) const { return min(bh, h - (b >> 3) * bh); }
};
__device__ __forceinline__ bool rect_keeps(const uint4 r, int k, int area) {   // k: row-major tile index inside the rectangle
    const uint64_t m = (uint64_t)r.z | ((uint64_t)r.w << 32);
    if (area <= RECT_MASK_TILES) return (m >> k) & 1ull;
    const int w = (int)(r.x >> 16) - (int)(r.x & 0xffffu);
    const RectBlocks g(w, area / w);
    return (m >> g.block_of(k % w, k / w)) & 1ull;
}
// position of tile k of the rectangle among the listed ones (the pair's slot behind slot_base[i])
__device__ __forceinline__ uint32_t rect_rank(const uint4 r, int k, int area) {
    const uint64_t m = (uint64_t)r.z | ((uint64_t)r.w << 32);
    if (area <= RECT_MASK_TILES) return (uint32_t)__popcll(m & ((1ull << k) - 1ull));
    const int w = (int)(r.x >> 16) - (int)(r.x & 0xffffu);
    const RectBlocks g(w, area / w);
    const int tx = k % w, ty = k / w, b = g.block_of(tx, ty);
    // The tiles of the kept blocks in front of block b, in closed form (until round 4: a walk over the up to 63 mask bits below
    // b, which the backward blend pass paid for every staged entry of a large-footprint Gaussian).  Blocks are bw x bh tiles
    // except in the grid's last column (wl wide) and last row; every block row in front of b's is a full-height one.
    const int bc = b & 7, br = b >> 3;
    const int ncols = (g.w + g.bw - 1) / g.bw, wl = g.w - (ncols - 1) * g.bw;          // columns of the grid, width of the last one
    const uint64_t rows_before = br ? (~0ull >> (64 - 8 * br)) : 0ull;                // every bit of the block rows in front of br
    const uint64_t last_col = 0x0101010101010101ull << (ncols - 1);
    const uint64_t full_cols = (0x0101010101010101ull * (uint64_t)((1u << (ncols - 1)) - 1u));
    const uint32_t row_bits = (uint32_t)(m >> (8 * br)) & ((1u << bc) - 1u);           // kept blocks of b's row in front of it: full width
    const uint32_t before = (uint32_t)g.bh * ((uint32_t)g.bw * (uint32_t)__popcll(m & rows_before & full_cols) +
                                              (uint32_t)wl * (uint32_t)__popcll(m & rows_before & last_col)) +
                            (uint32_t)g.height(b) * (uint32_t)g.bw * (uint32_t)__popc(row_bits);
    return before + (uint32_t)((ty % g.bh) * g.width(b) + tx % g.bw);
}

// ---- state layouts ----
struct GeomView {
    float *rec;              // N * REC_FLOATS (16)
    uint32_t *tiles_touched; // N: tiles of the rectangle the Gaussian can reach (= pairs listed for it)
    uint32_t *depth_bits;    // N: view depth as ordered bits (positive floats compare like unsigned integers)
    uint4 *rect;             // N: tile rectangle x0 | x1 << 16, y0 | y1 << 16 (empty for culled Gaussians) and the kept-tile mask
    uint32_t *slot_base;     // N: exclusive scan of tiles_touched in id order: first pair / gradient slot of a Gaussian
    uint32_t *total;         // 1: pair count D of this frame (device copy)
};
struct PrepScratch {
    uint32_t *blocksums; // scan block sums
};
// Two-level grouping (LVDGS_FLAG_SUPER_TILES, binning.hip): the grid of SUPER x SUPER-tile super-tiles and its own counting state
constexpr int SUPER = 4;
struct SuperView {
    uint4 *rect;           // N: a Gaussian's rectangle in super-tile units + which of them hold a listed tile
    uint32_t *hist;        // [Gaussian chunk][super-tile]
    uint32_t *totals;      // Ts (+4)
    uint2 *ranges;         // Ts
    uint32_t *long_count;  // 64: queue length, order-valid flag (as ImageView::long_count)
    uint32_t *long_tiles;  // 2 Ts: queue of super-tiles with lists beyond one wave's sort, the super-tiles by list length
    uint32_t *total;       // 4: super pair count, longest queued list, queue length
};
struct RenderScratch {
    uint32_t *blocksums; // same place as PrepScratch::blocksums: the counting path's first kernel still reads them
    uint32_t *keys;  // D (alternate tile-key buffer of the radix path; 64-bit keys of over-long tile segments)
    uint32_t *vals;  // D (alternate id buffer)
    uint32_t *hist;         // radix path only
    uint32_t *totals;       // radix path only
    uint32_t *group_hist;   // counting path: [Gaussian chunk][tile] pair counts, then exclusive prefixes over the chunks
    uint32_t *group_totals; // counting path: pairs per tile
    uint32_t *chunk_sums;   // counting path: pairs per Gaussian chunk (left by the projection kernel that counts, read by the slot scan in the scatter)
    SuperView super;        // counting path: the two-level grouping's state
};
struct BinView {
    uint32_t *point_list; // D
    uint32_t *tile_keys;  // D
    uint8_t *pair_valid;  // D: 1 once the backward blend pass has written the pair's gradient record (cleared by the forward pass).
                          // Pairs behind a tile's last contributor -- nine tenths of the lists on opaque surfaces -- are
                          // neither staged nor written by blend_bwd, and preprocess_bwd passes over their slots.
};
struct ImageView {
    uint2 *ranges;       // T
    uint32_t *long_count; // [0]: length of the queue of over-long segments, [1]: tile order valid; zeroed with ranges
    uint32_t *long_tiles; // T: the queue (tile ids); T more: the tiles by descending list length (small grids)
    float *final_T;      // P
    uint32_t *n_contrib; // P
};
struct BwdScratch {
    float *pair_grads; // D*10
    float *tau_part;   // nblk*6
};

size_t geom_layout(int N, GeomView *v, void *base);
size_t prep_scratch_layout(int N, PrepScratch *v, void *base);
size_t bin_layout(int64_t D, BinView *v, void *base);
int64_t bin_view(const lvdgs_args *a, BinView *v);   // the views of a->binning_state (laid out by its size); returns the pairs it holds
size_t image_layout(int W, int H, ImageView *v, void *base);
size_t render_scratch_layout(int N, int64_t D, int W, int H, RenderScratch *v, void *base);
size_t bwd_scratch_layout(int N, int64_t D, BwdScratch *v, void *base);
int tile_sort_bits(int W, int H);  // number of key bits to sort for the tile id

// ---- launchers (one per kernel family; all enqueue on `stream` and return a status) ----
int launch_preprocess_fwd(const lvdgs_args &a, const GeomView &g, uint32_t *blocksums /* one per 256 Gaussians */, hipStream_t s);
// pair_total / pair_capacity (lvdgs_forward_backward_fused_loss): the pass does nothing when *pair_total exceeds the capacity
int launch_preprocess_bwd_views(const lvdgs_args *const *a, const GeomView *g, const BwdScratch *w, const BinView *b, int n, hipStream_t s);
int launch_preprocess_bwd(const lvdgs_args &a, const GeomView &g, const BwdScratch &b, const uint8_t *pair_valid, hipStream_t s,
                          const uint32_t *pair_total = nullptr, uint32_t pair_capacity = 0);
int launch_mark_visible(int N, const float *means3D, const float *view, uint8_t *present, hipStream_t s);

// stable LSD radix sort of (key, val) pairs on key bits [0, total_bits); result lands in
// (keys_a, vals_a) if *result_in_a, else in (keys_b, vals_b).  n may be 0.
int radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, int64_t n, int total_bits,
                     uint32_t *hist, uint32_t *totals, bool *result_in_a, int dbg, hipStream_t s,
                     const uint32_t *n_dev = nullptr);
int radix_num_passes(int total_bits);
size_t radix_hist_entries(int64_t n);

// slot_base[i] = exclusive scan over i of tiles_touched[i]; *total_dev = the sum (pair count D)
bool tile_order_in_use(int num_tiles);   // api.hip: the blend kernels take their tiles from ImageView::long_tiles + T

// Pieces lvdgs_tracking_tail (pose.hip) puts into one launch.
struct LossTail {            // what photometric_finish_kernel<2> reads and writes (loss.hip)
    const float *partial; int nblk; int P; float w_rgb, w_d; float *loss, *d_a, *d_b;
};
int loss_tail_params(const lvdgs_loss_args *a, bool partials_per_tile, LossTail *out);   // validates like lvdgs_photometric_loss_value_and_grad
struct LossParams;           // photometric.hpp
int loss_fused_params(const lvdgs_loss_args *a, LossParams *out);   // for the backward blend pass: partial = a->scratch, 4 per tile
int launch_blend_bwd_fused_loss(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, const BwdScratch &w,
                                const LossParams &loss, int propagate_opacity, hipStream_t s);

int launch_slot_scan(const uint32_t *tiles_touched, uint32_t *slot_base, uint32_t *blocksums, uint32_t *total_dev, int N, int dbg,
                     hipStream_t s);

// Grouping of the (Gaussian, tile) pairs by tile without sorting (binning.hip): counts per (Gaussian chunk, tile) in
// LDS, prefix sums, then every pair is placed with one LDS atomic.  The order inside a tile's segment is arbitrary;
// the tile sort makes it canonical.  Only for images of at most group_max_tiles() tiles (LDS counters).
int group_max_tiles();
size_t group_hist_entries(int N, int num_tiles);
size_t group_chunks(int N);
// lvdgs_forward: projection + per-chunk tile counts in one kernel (preprocess.hip); the two-call API projects in
// lvdgs_forward_prepare and counts with launch_group_count.
int launch_preprocess_count(const lvdgs_args &a, const GeomView &g, const ImageView &im, const RenderScratch &w, hipStream_t s);
// lvdgs_forward_batch: the stages of up to FWD_BATCH_VIEWS views of one map and one image size in one launch each (the view is
// blockIdx.y; every view its own state and scratch buffers).  caps[k]: view k's pair capacity; host_words: 4 pinned words per view
// as the device addresses them (pair count, longest queued segment, queue length, the call's sequence number).
constexpr int FWD_BATCH_VIEWS = 10;
int launch_preprocess_count_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, int n, hipStream_t s);
int launch_group_scan_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, const int64_t *caps, int n,
                            uint32_t *host_words, uint32_t host_seq, hipStream_t s);
int launch_group_scatter_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, const BinView *b,
                               const int64_t *caps, int n, hipStream_t s);
int launch_tile_depth_sort_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, const BinView *b, int n,
                                 int longest_expected, int queue_expected, hipStream_t s);
int launch_group_count(const lvdgs_args &a, const GeomView &g, const ImageView &im, const RenderScratch &w, hipStream_t s);
// prefixes over the chunks, tile ranges, the pair count (*total_out, may be null), the queue of over-long segments
int launch_group_scan(const lvdgs_args &a, const ImageView &im, const RenderScratch &w, int64_t capacity, uint32_t *total_out, hipStream_t s,
                      uint32_t *host_out = nullptr, uint32_t host_seq = 0);   // host_out: pinned words the tile scan writes the pair count + hints + host_seq to
// slot_scan: also makes slot_base from tiles_touched and w.chunk_sums (launch_preprocess_count's leftovers)
// total_out / host_out / host_seq: launch_group_scan's, for the tile scan that rides in this launch (binning.hip: LVDGS_SCAN_IN_SCATTER)
int launch_group_scatter(const lvdgs_args &a, const GeomView &g, const ImageView &im, const RenderScratch &w, unsigned long long *keys64,
                         int64_t capacity, bool slot_scan, uint8_t *pair_valid, hipStream_t s, uint32_t *total_out = nullptr, uint32_t *host_out = nullptr,
                         uint32_t host_seq = 0);
int launch_emit_pairs(const lvdgs_args &a, const GeomView &g, uint32_t *tile_keys, uint32_t *ids, int64_t capacity, hipStream_t s);
// two-level grouping (binning.hip): is it in use for this call; super_count + scans + scatter of the super grid; the expansion of the
// sorted super lists into the tiles' segments of point_list
bool super_tiles_in_use(const lvdgs_args &a);
int super_tiles_of(int W, int H);
int launch_super_count(const lvdgs_args &a, const GeomView &g, const SuperView &sv, hipStream_t s);   // before launch_group_scan, which then scans both grids
int launch_super_scatter(const lvdgs_args &a, const GeomView &g, const SuperView &sv, const RenderScratch &w, unsigned long long *keys64, int64_t capacity,
                         bool slot_scan, uint8_t *pair_valid, hipStream_t s);
int launch_super_expand_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, const BinView *b, int n, hipStream_t s);
int launch_super_expand(const lvdgs_args &a, const GeomView &g, const SuperView &sv, const ImageView &im, const uint32_t *super_list, uint32_t *point_list,
                        hipStream_t s);
// Sorts the segment of every tile in [t_lo, t_hi) by (view-depth bits, id) and leaves the ids in point_list.  keys64 holds
// D 64-bit keys: already filled per segment (counting path, keys_ready; the queue of segments longer than
// tile_sort_wave_limit() is then filled too), or scratch for over-long segments whose keys are gathered from the ids in
// point_list (radix path).  longest_expected / queue_expected: the longest queued segment and the queue's length of the
// previous frame on this device (0: none; longest < 0: unknown) -- which of the kernels for long segments get launched
// behind the tile sort (a hint; without it such segments are sorted by the tile sort's own last workgroups).
int tile_sort_wave_limit();
int launch_tile_depth_sort(const ImageView &im, int num_tiles, int t_lo, int t_hi, const float *rec, uint32_t *point_list, void *keys64,
                           bool keys_ready, int longest_expected, int queue_expected, int dbg, hipStream_t s);
int launch_tile_ranges(const uint32_t *tile_keys, int64_t D, const uint32_t *D_dev, const ImageView &im, int num_tiles, int dbg,
                       hipStream_t s);

int launch_blend_fwd(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, bool deep_lists, hipStream_t s);   // deep_lists: a hint (which build of the kernel), never a result
struct LossParams;
int launch_blend_fwd_batch(const lvdgs_args *const *a, const GeomView *g, const BinView *b, const ImageView *im, int n, bool deep_lists, hipStream_t s);
int launch_blend_fwd_bwd_fused_loss(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, const BwdScratch &w,
                                    const LossParams &loss, int propagate_opacity, bool deep_lists, const uint32_t *pair_total, uint32_t pair_capacity,
                                    hipStream_t s);
// The static-mask mapping loss of one view as the backward blend pass reads it (lvdgs_masked_loss_args, checked by api.hip).
struct MaskedLossView {
    const float *d_image;        // 3*P: d loss / d colour, written by lvdgs_masked_loss_batch
    const float *depth, *gt_depth; const uint8_t *static_mask;   // depth term (gt_depth null: none)
    float depth_lambda;
    const float *out;            // lvdgs_masked_loss_args::out: [4] = |M|
};
int launch_blend_bwd_masked_loss(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, const BwdScratch &w,
                                 const MaskedLossView &m, hipStream_t s);
// masked: null, or per view null (the view is scored by loss[v]) / its static-mask loss
int launch_blend_bwd_fused_loss_batch(const lvdgs_args *const *a, const GeomView *g, const BinView *b, const ImageView *im, const BwdScratch *w,
                                      const LossParams *loss, const MaskedLossView *const *masked, int n, int propagate_opacity, hipStream_t s);
int launch_blend_bwd(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, const BwdScratch &w,
                     hipStream_t s);

}  // namespace lvdgs
