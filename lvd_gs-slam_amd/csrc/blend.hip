// Per-tile alpha compositing (forward) and its gradient (backward).
//
// Layout of one workgroup: 256 threads = 4 waves = one 16x16 tile; each wave owns an 8x8 pixel
// quadrant (lane -> pixel).  The tile's depth-ordered Gaussian list is consumed in rounds of 256
// entries: every thread gathers one entry's 64-byte record (its first 48 bytes in the forward pass) from HBM/L2 into LDS, then each wave
//   1. tests 64 staged Gaussians at once (lane -> Gaussian) against its own quadrant with an
//      exact ellipse-vs-rectangle minimum of the exponent, giving a 64-bit survivor mask, and
//   2. walks only the survivors (scalar loop over mask bits), broadcasting one Gaussian from LDS
//      to all 64 pixels of the wave.
// Skipped Gaussians are exactly those whose alpha is below 1/255 on every pixel of the quadrant,
// so results are unchanged; a typical scene discards well over half of the (pixel, Gaussian) work.
//
// Backward walks the same list back to front (as published: T is recovered by dividing out
// 1-alpha).  The per-Gaussian sums over pixels are done without atomics: in the current kernel (blend_bwd3) a
// "pixel pass" leaves the two state-dependent numbers of every (pixel, Gaussian) in an LDS matrix and a "splat pass"
// (lane -> Gaussian of a batch of eight) accumulates the ten sums in registers, one fold per batch; then per-wave
// LDS slots, a fixed-order 4-wave sum, and one 40-byte record per (Gaussian, tile) pair
// stored at slot_base[id] + (rank of the tile among the Gaussian's kept tiles): a Gaussian's records form one contiguous
// run, and the runs follow each other in id order, so the per-Gaussian kernel (one lane per id) reads
// whole cache lines.  Gradients are therefore bitwise reproducible.
//
// Kernels in this file: blend_fwd2_kernel (its survivor step is the hand-scheduled composite_one) and
// blend_bwd3_kernel<LOSS, DEPTH_GRAD, POSE_ONLY> (LOSS: where the pixels' gradients come from -- gradient images, the photometric
// loss evaluated in the prologue for lvdgs_backward_fused_loss, or the static-mask mapping loss of lvdgs_backward_masked_loss).  Their predecessors (the compiler's form of the forward step, the single-pass and the
// first two-pass backward) left the library in round 3: profiles/experiments/r02_superseded_blend_kernels.hip.txt.
// What bounds the kernels and what was tried: DESIGN.md section 2, profiles/experiments/README.md.
#include <type_traits>
#if defined(LVDGS_DIAG_GRID) || defined(LVDGS_DIAG_REPEAT)
#include <cstdlib>
#endif

#include "common.hpp"
#include "device_utils.hpp"
#include "photometric.hpp"

namespace lvdgs {

namespace {

constexpr float LOG2E = 1.4426950408889634f;

struct BlendParams {
    int W, H, gx, gy;
    int num_tiles, tile_base;    // tiles rendered by this launch (a band of whole tile rows, lvdgs_args.tile_row_*) and the first of them
    const uint2 *ranges;
    const uint32_t *tile_order;  // tiles by descending list length, or null (tile_of_workgroup)
    const uint32_t *order_valid; // 1 once this frame's grouping has written tile_order
    const uint32_t *point_list;
    const float *rec;
    const uint32_t *slot_base;
    const uint4 *rect;           // tile rectangle + kept-tile mask of every Gaussian (backward: pair_slot)
    const float *bg;
    // forward
    float *out_color, *out_depth, *out_opacity, *final_T;
    uint32_t *n_contrib;
    int32_t *n_touched;
    // backward
    const float *dL_dcolor, *dL_ddepth, *dL_dopacity;
    float *pair_grads;
    uint8_t *pair_valid;         // 1 where a record of pair_grads has been written by this backward pass (BinView::pair_valid)
    // backward with the photometric loss evaluated in place of reading dL_d* (lvdgs_backward_fused_loss)
    LossParams loss;             // loss.partial: 4 sums per TILE
    int loss_propagate_opacity;  // dL/d(opacity image) feeds the blend (rasterizer.PROPAGATE_OPACITY_GRAD)
    // LOSS_MASKED (lvdgs_backward_masked_loss: a keyframe with a static mask, reference utils/slam_backend.py:196-261): the colour
    // gradient is the image lvdgs_masked_loss_batch wrote (dL_dcolor: the SSIM term is not local), the depth term's gradient is
    // evaluated per pixel from loss.depth (rendered), loss.gt_depth (mono depth), loss.grad_mask (static mask), loss.w_d
    // (depth_lambda) and loss.grad_out[4] (|M|, finished by that call)
    int loss_mode;               // LOSS_FUSED or LOSS_MASKED: which of the two this VIEW takes in a launch compiled for LOSS_PER_VIEW
    // blend_fwd_bwd_kernel (forward and backward in one launch, enqueued before the host knows the pair count): the frame's count as the
    // tile scan left it and the capacity the record buffer was sized for -- beyond it the backward half is left out (its slots
    // would lie outside the buffer; the caller re-runs both passes with room)
    const uint32_t *pair_total;
    uint32_t pair_capacity;
};
#ifndef LVDGS_HEAVY_PRIO
#define LVDGS_HEAVY_PRIO 0        // A/B builds: 1..3 = s_setprio level of the workgroups at the head of the tile order (blend_fwd_bwd_kernel); measured in round 5: nothing (KITTI pose-only 5090-5130 it/s with level 3 on the first 1/16 or 1/64 of the tiles, level 1 on the first 1/8, or none -- the heaviest tile's chain is its own latency, not lost issue slots)
#endif
#ifndef LVDGS_HEAVY_PRIO_SHIFT
#define LVDGS_HEAVY_PRIO_SHIFT 4  // ... the first num_tiles >> SHIFT of them
#endif
// where the backward blend pass takes dL/d(colour, depth, opacity) of its pixels from
constexpr int LOSS_IMAGES = 0;    // gradient images (lvdgs_backward)
constexpr int LOSS_FUSED = 1;     // the photometric loss evaluated in the prologue (lvdgs_backward_fused_loss)
constexpr int LOSS_MASKED = 2;    // the static-mask mapping loss (above)
constexpr int LOSS_PER_VIEW = 3;  // batch launches whose views differ: BlendParams::loss_mode decides

// Where the partial gradient of the pair (Gaussian id, tile (tx, ty)) goes: the Gaussian's slots follow its kept tiles in
// row-major order of its rectangle (common.hpp: rect_rank), the order the pairs were counted in.
__device__ __forceinline__ uint32_t pair_slot(const BlendParams &p, uint32_t id, int tx, int ty) {
    uint4 r;
    if constexpr (REC_FLOATS >= 16) {   // the copy in the record: the line the staging loads have just brought in
        const float4 q = reinterpret_cast<const float4 *>(p.rec + (size_t)id * REC_FLOATS)[3];
        r = make_uint4(__float_as_uint(q.x), __float_as_uint(q.y), __float_as_uint(q.z), __float_as_uint(q.w));
    } else {
        r = p.rect[id];
    }
    const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
    const int w = x1 - x0;
    return p.slot_base[id] + rect_rank(r, (ty - y0) * w + (tx - x0), w * (y1 - y0));
}

// Workgroup -> tile.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one) and an XCD takes its
// workgroups in order.  Runs of TILE_RUN row-major tiles go to the XCDs in turn: neighbouring tiles share most of their
// Gaussians and find each other's record lines in the same 4 MiB L2, while every XCD sees every part of the image (one
// contiguous run per XCD gave the XCD holding the densest image rows 1.5 x its share of the work at KITTI's 1226x370;
// tools/tile_schedule_model.py).  Bijective for any count: whole groups of 8 runs are permuted, the rest is identity.
#ifndef LVDGS_TILE_RUN
#define LVDGS_TILE_RUN 16
#endif
constexpr int TILE_RUN = LVDGS_TILE_RUN;
__device__ __forceinline__ int tile_of_workgroup(int b, int n) {
    const int grouped = n - n % (8 * TILE_RUN);
    if (b >= grouped) return b;
    const int xcd = b & 7, k = b >> 3;
    return ((k / TILE_RUN) * 8 + xcd) * TILE_RUN + k % TILE_RUN;
}

// Small grids (every workgroup resident at once: tile_order, longest list first): which entry of the order workgroup b
// takes.  Workgroups b and b + 256 land on the same CU (dealt round-robin over 8 XCDs x 32 CUs), so taking the order
// as it comes gives CU 0 the longest list of every stratum of 256 and CU 255 the shortest of every one; walking the odd
// strata backwards evens the sums out.  Bijective for any count (the last, partial stratum is mirrored on itself).
#ifndef LVDGS_TILE_SNAKE
#define LVDGS_TILE_SNAKE 1
#endif
__device__ __forceinline__ int order_slot_of_workgroup(int b, int n) {
    if (!LVDGS_TILE_SNAKE) return b;
    const int k = b >> 8, j = b & 255;
    if ((k & 1) == 0) return b;
    const int m = min(256, n - (k << 8));   // entries in this stratum
    return (k << 8) + (m - 1 - j);
}

// ------------------------------------------------------------------------------------------
// One survivor of the quadrant test against the wave's 64 pixels, hand-scheduled: the hit lanes run the compositing
// under EXEC instead of through selects (the compiler's form of the same statements is 31 vector + 17 scalar
// instructions per survivor; this one 25 + 8, and both pipes are what blend_fwd is bound by).  Arithmetic, operand for
// operand, as the plain C++ statements.  Returns the number of pixels with transmittance still above 1/2 after this Gaussian.
#ifndef LVDGS_FWD_SKIP_COUNT
#define LVDGS_FWD_SKIP_COUNT 2   // A/B builds: 0 counts touched pixels for every survivor, 1 decides once per chunk of 64 staged entries
#endif
// The survivor step up to and including the update of T and the last contributor.  The two hit tests narrow EXEC
// themselves (v_cmpx, gfx9: writes EXEC and VCC) and the hit block runs without a branch around it: a survivor of the
// exact quadrant test nearly always has a hit pixel, and with EXEC = 0 the block is ten empty issue slots whose last
// compare leaves VCC = 0, i.e. a count of 0 by itself.  Three scalar instructions per survivor in this block instead
// of seven, on a scalar unit four SIMDs share (blend_fwd 141.1 -> 139.0 us at config 3, same box).
#define LVDGS_COMPOSITE_HEAD                                                                                            \
    "v_sub_f32 %[dx], %[ax], %[pxe]\n\t"                                                                                \
    "v_sub_f32 %[dy], %[ay], %[pyf]\n\t"                                                                                \
    "v_mul_f32 %[t], %[ab], %[dy]\n\t"   /* kb * dy */                                                                  \
    "v_mul_f32 %[u], %[bc], %[dy]\n\t"   /* kc * dy */                                                                  \
    "v_fmac_f32 %[t], %[aa], %[dx]\n\t"  /* ka * dx + kb * dy */                                                        \
    "v_mul_f32 %[u], %[u], %[dy]\n\t"    /* (kc * dy) * dy */                                                           \
    "v_fmac_f32 %[u], %[dx], %[t]\n\t"   /* log2 of the falloff */                                                      \
    "v_exp_f32 %[t], %[u]\n\t"                                                                                          \
    "v_cmpx_ge_f32 vcc, 0, %[u]\n\t"     /* EXEC: exponent not positive ... */                                          \
    "v_mul_f32 %[a], %[op], %[t]\n\t"                                                                                   \
    "v_min_f32 %[a], %[amax], %[a]\n\t"                                                                                 \
    "v_cmpx_le_f32 vcc, %[amin], %[a]\n\t" /* ... and alpha >= 1/255: the lanes that hit */                             \
    "v_sub_f32 %[t], 1.0, %[a]\n\t"                                                                                     \
    "v_mul_f32 %[tt], %[T], %[t]\n\t"    /* transmittance behind this Gaussian */                                       \
    "v_cmp_gt_f32 vcc, %[tstop], %[tt]\n\t" /* ... below 1e-4: the pixel is finished, this Gaussian not composited */   \
    "v_cndmask_b32 %[pxe], %[pxe], %[far], vcc\n\t"                                                                     \
    "s_andn2_b64 exec, exec, vcc\n\t"                                                                                   \
    "v_mul_f32 %[t], %[a], %[T]\n\t"     /* weight */                                                                   \
    "v_fmac_f32 %[C0], %[cr], %[t]\n\t"                                                                                 \
    "v_fmac_f32 %[C1], %[cg], %[t]\n\t"                                                                                 \
    "v_fmac_f32 %[C2], %[cb], %[t]\n\t"                                                                                 \
    "v_fmac_f32 %[Dp], %[cd], %[t]\n\t"                                                                                 \
    "v_mov_b32 %[T], %[tt]\n\t"                                                                                         \
    "v_mov_b32 %[last], %[index]\n\t"
#define LVDGS_COMPOSITE_OUT                                                                                             \
    [dx] "=&v"(dx), [dy] "=&v"(dy), [t] "=&v"(t), [u] "=&v"(u), [a] "=&v"(a), [tt] "=&v"(tt), [pxe] "+v"(pxe), [T] "+v"(T),      \
        [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [Dp] "+v"(Dp), [last] "+v"(last)
#define LVDGS_COMPOSITE_IN                                                                                              \
    [ax] "v"(A.x), [ay] "v"(A.y), [aa] "v"(A.z), [ab] "v"(A.w), [bc] "v"(B.x), [op] "v"(B.y), [cr] "v"(Cc.x), [cg] "v"(Cc.y),    \
        [cb] "v"(Cc.z), [cd] "v"(Cc.w), [pyf] "v"(pyf), [far] "v"(far_away), [index] "s"(index), [amax] "s"(ALPHA_MAX),          \
        [amin] "s"(ALPHA_MIN), [tstop] "s"(T_STOP), [full] "s"(full)
// `full`: the EXEC mask outside (all of the wave's lanes; read once per chunk, restored after every survivor).
// COUNT = false: no pixel of the quadrant has transmittance above 1/2 any more (it only falls), the count is 0 and the
// compare, the population count and the caller's v_writelane are left out.
template <bool COUNT = true>
__device__ __forceinline__ int composite_one(const float4 A, const float2 B, const float4 Cc, float &pxe, const float pyf, float &T,
                                             float &C0, float &C1, float &C2, float &Dp, uint32_t &last, const uint32_t index,
                                             const float far_away, const unsigned long long full) {
    float dx, dy, t, u, a, tt;
    int n = 0;
    if constexpr (COUNT)
        asm(LVDGS_COMPOSITE_HEAD
            "v_cmp_lt_f32 vcc, 0.5, %[tt]\n\t"
            "s_bcnt1_i32_b64 %[n], vcc\n\t"
            "s_mov_b64 exec, %[full]"
            : LVDGS_COMPOSITE_OUT, [n] "=&s"(n) : LVDGS_COMPOSITE_IN : "vcc", "scc");
    else
        asm(LVDGS_COMPOSITE_HEAD "s_mov_b64 exec, %[full]" : LVDGS_COMPOSITE_OUT : LVDGS_COMPOSITE_IN : "vcc", "scc");
    return n;
}

// ------------------------------------------------------------------------------------------
#ifndef LVDGS_FWD_BATCH
#define LVDGS_FWD_BATCH 8
#endif
#ifndef LVDGS_FWD_QUADRANT_EXIT
#define LVDGS_FWD_QUADRANT_EXIT 1   // A/B builds: 0 = a finished quadrant walks to the end of the round of 256 staged entries
#endif
// DEEP_LISTS: a build for frames whose tile lists run to a thousand entries and more, of which the pixels composite the first
// tenth (opaque surfaces): a quadrant looks after every batch of eight whether any of its pixels is still open.  Same results;
// which of the two kernels runs is decided by the previous frame's longest list (a hint, api.hip).
// The forward pass's LDS: one object, so that the three reads of a survivor share one address register; laid out so that they are
// a 16-, an 8- and a 16-byte read (4 + 2 + 4 LDS cycles per wave; a 12-byte read alone costs 8).  (A type of its own since round 5:
// the kernel that runs the forward and the backward pass of a tile in one launch overlays it with the backward pass's.)
struct FwdShared {
    float4 a[256], b[256], c[256];
    float d[256];
    int touch[256];   // pixels of this tile each staged Gaussian "touched" (T after it > 0.5)
};
template <bool DEEP_LISTS>
__device__ __forceinline__ void blend_fwd2_body(const BlendParams &p, FwdShared &s_recs) {
    float4 *const s_a = s_recs.a;  // x, y, -a/2*log2e, -b*log2e
    float4 *const s_b = s_recs.b;  // -c/2*log2e, opacity | raw conic a, b (for the quadrant test; survivors read the first half only)
    float4 *const s_c = s_recs.c;  // r, g, b, depth
    float *const s_d = s_recs.d;   // raw conic c
    int *const s_touch = s_recs.touch;

#ifdef LVDGS_DIAG_REPEAT   // diagnostic build: the grid repeated $LVDGS_DIAG_REPEAT times (what a launch over several views of this size would cost)
    const int diag_b = (int)blockIdx.x % p.num_tiles;
#define blockIdx_x_ diag_b
#else
#define blockIdx_x_ ((int)blockIdx.x)
#endif
    const int tile = (p.tile_order && *p.order_valid) ? (int)p.tile_order[order_slot_of_workgroup(blockIdx_x_, p.num_tiles)] : p.tile_base + tile_of_workgroup(blockIdx_x_, p.num_tiles);
#undef blockIdx_x_
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qx0 = tx * TILE + (wave & 1) * 8, qy0 = ty * TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < p.W && py < p.H;
    const float pxf = (float)px, pyf = (float)py;
    const float rx0 = (float)qx0, ry0 = (float)qy0, rx1 = (float)(qx0 + 7), ry1 = (float)(qy0 + 7);

    const uint2 range = p.ranges[tile];
    const int todo = (int)(range.y - range.x);
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0;
    // A finished pixel (transmittance below 1e-4, or outside the image) is moved to x = 1e30: every later Gaussian
    // then evaluates to alpha = 0 there and fails the 1/255 test by itself, so the per-Gaussian code carries no
    // "done" mask (that mask cost five scalar instructions per Gaussian, on a scalar unit shared by four SIMDs).
    constexpr float FAR_AWAY = 1e30f;
    float pxe = inside ? pxf : FAR_AWAY;
    const float far_away = FAR_AWAY;

    for (int base = 0; base < todo; base += 256) {
        const bool done = pxe == FAR_AWAY;
        if (__syncthreads_and(done)) break;
        const int cnt = min(256, todo - base);
        uint32_t my_id = 0;
        if (tid < cnt) {
            my_id = p.point_list[range.x + base + tid];
            const float4 *r4 = reinterpret_cast<const float4 *>(p.rec + (size_t)my_id * REC_FLOATS);
            const float4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
            s_a[tid] = make_float4(r0.x, r0.y, -0.5f * LOG2E * r0.z, -LOG2E * r0.w);
            s_b[tid] = make_float4(-0.5f * LOG2E * r1.x, r1.y, r0.z, r0.w);
            s_c[tid] = make_float4(r1.z, r1.w, r2.x, r2.y);
            s_d[tid] = r1.x;
        }
        s_touch[tid] = 0;
        __syncthreads();
        if (__ballot(!done) != 0ull) {
            for (int c0 = 0; c0 < cnt; c0 += 64) {
                // A quadrant whose 64 pixels are all finished (opaque surfaces: after a tenth of the list) sits the rest of the
                // round out: no test, no walk.  (Until round 4 it went on testing and walking -- every step with EXEC = 0 -- to
                // the end of the 256 staged entries.)
                if (LVDGS_FWD_QUADRANT_EXIT && __ballot(pxe != FAR_AWAY) == 0ull) break;
                // ---- lane -> Gaussian: which of these 64 can touch this wave's quadrant? ----
                const int jl = c0 + lane;
                bool keep = false;
                if (jl < cnt) {
                    const float4 A = s_a[jl];
                    const float4 B = s_b[jl];
                    keep = reaches_rect(A.x, A.y, B.z, B.w, s_d[jl], B.y, rx0, ry0, rx1, ry1);
                }
                uint64_t live = __ballot(keep);
                // ---- lane -> pixel: composite the survivors in list order ----
                int vcnt = 0;  // lane -> staged Gaussian c0 + lane: pixels of this quadrant it "touched"
                const unsigned long long full = __ballot(true);
                // Transmittance only falls: once no pixel of the quadrant is above 1/2, nothing is "touched" any more
                // and the survivors take the step without the count (two vector and two scalar instructions less each).
                auto any_above_half = [&]() { return !LVDGS_FWD_SKIP_COUNT || __ballot(T > T_TOUCH && pxe != FAR_AWAY) != 0ull; };
                bool counting = any_above_half();
                auto one = [&](auto count) {
                    const int jb = __builtin_ctzll(live);
                    const int jj = c0 + jb;
                    live = mask_clear_bit(live, jb);
                    const float4 A = s_a[jj];
                    const float2 B = *reinterpret_cast<const float2 *>(&s_b[jj]);
                    const float4 Cc = s_c[jj];
                    const int n = composite_one<decltype(count)::value>(A, B, Cc, pxe, pyf, T, C0, C1, C2, Dp, last, (uint32_t)(base + jj + 1), far_away, full);
                    if constexpr (decltype(count)::value) vcnt = write_lane(vcnt, n, jb);  // 0 by itself once no pixel of the quadrant has transmittance above 1/2
                };
                // eight survivors per trip while there are that many (batches of 1 / 4 / 8: 149.7 / 147.7 / 146.5 us at config 3): the compiler is free to issue the next survivors' reads
                // above the current one's (register-only) compositing block, and the loop test is paid once in eight
                while (counting && __popcll(live) >= LVDGS_FWD_BATCH) {   // (looked at again after every batch)
#pragma unroll
                    for (int u = 0; u < LVDGS_FWD_BATCH; u++) one(std::true_type{});
                    if (LVDGS_FWD_SKIP_COUNT > 1) counting = any_above_half();
                }
                if (counting)
                    while (live) one(std::true_type{});
                while (__popcll(live) >= LVDGS_FWD_BATCH) {
#pragma unroll
                    for (int u = 0; u < LVDGS_FWD_BATCH; u++) one(std::false_type{});
                    // (at config 3, whose pixels rarely finish, this look was +4 us)
                    if (DEEP_LISTS && __ballot(pxe != FAR_AWAY) == 0ull) live = 0ull;   // ... nor the rest of this chunk
                }
                while (live) one(std::false_type{});
                if (vcnt) atomicAdd(&s_touch[c0 + lane], vcnt);
            }
        }
        __syncthreads();
        // one integer atomic per (tile, Gaussian) that touched anything
        if (tid < cnt) {
            const int n = s_touch[tid];
            if (n) atomicAdd(&p.n_touched[my_id], n);
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * p.W + px, P = (size_t)p.W * p.H;
        p.final_T[pix] = T;
        p.n_contrib[pix] = last;
        p.out_color[pix] = fmaf(T, p.bg[0], C0);
        p.out_color[P + pix] = fmaf(T, p.bg[1], C1);
        p.out_color[2 * P + pix] = fmaf(T, p.bg[2], C2);
        p.out_depth[pix] = Dp;
        p.out_opacity[pix] = 1.f - T;
    }
}

__global__ void __launch_bounds__(256) blend_fwd2_kernel(BlendParams p) { __shared__ FwdShared s; blend_fwd2_body<false>(p, s); }
__global__ void __launch_bounds__(256, 8) blend_fwd2_deep_kernel(BlendParams p) { __shared__ FwdShared s; blend_fwd2_body<true>(p, s); }
// Several views of one size in ONE launch (lvdgs_blend_forward_batch: the views of a mapping window): blockIdx.y picks the view.
// A KITTI-size frame leaves the chip's wave slots half empty and ends in a tail of its heaviest tiles; ten of them fill it
// (per view 54 -> 34 us, blend_bwd 105 -> 77: LVDGS_DIAG_REPEAT builds).  Up to BATCH_VIEWS argument blocks travel as the kernel's
// argument (344 bytes each, 4 KB of kernel arguments).
constexpr int BATCH_VIEWS = 11;
struct BlendBatch { BlendParams v[BATCH_VIEWS]; };
static_assert(sizeof(BlendBatch) <= 4096, "kernel arguments");
__global__ void __launch_bounds__(256) blend_fwd2_batch_kernel(BlendBatch b) { __shared__ FwdShared s; blend_fwd2_body<false>(b.v[blockIdx.y], s); }
__global__ void __launch_bounds__(256, 8) blend_fwd2_deep_batch_kernel(BlendBatch b) { __shared__ FwdShared s; blend_fwd2_body<true>(b.v[blockIdx.y], s); }

// ------------------------------------------------------------------------------------------
constexpr int ACC_STRIDE = 10;  // floats per (wave, entry) accumulator slot
constexpr int BR = 64;          // list entries staged per round in the backward pass

// ------------------------------------------------------------------------------------------
// Third form of the backward pass: the two passes above with the per-entry control flow taken out.
//
// At 4-5 waves per SIMD the pixel pass is bound by the LENGTH of one wave's instruction stream per entry, not by any
// pipe (tools/scalar_rec.hip: a 32-instruction dependent stream with its record from LDS runs at 80 cycles per entry
// per SIMD; the pixel pass of blend_bwd2 needed 137): per entry it spent ~21 scalar instructions, four branches and an
// EXEC save/restore on `if (hit)`, "any lane hit?", the batch counter and the two loop exits.  Here a batch is the next
// NB SURVIVORS of the quadrant test (hit or not), so its size is known before it starts: full batches are straight-line
// code (no branch, no counter, slot offsets immediate), the lanes that miss go through the same arithmetic with
// alpha = G = 0 (T, R unchanged, (u, w) = 0 -- same values as before for every lane), and the compiler is free to
// start an entry's reads and exponent while the previous entry's T/R chain finishes.
// The splat pass takes the pixel's coordinates and image gradients from the lane that owns the pixel inside the
// instruction that uses them (DPP row_ror:s on the subtract / multiply-add), instead of rotating two registers per step
// and reading the four gradients from LDS: no per-step moves, half the LDS reads, 4 KB less LDS (5 workgroups per CU).
template <int S>
__device__ __forceinline__ float sub_rotated(float a, float p) {  // a - (p of lane (i - S) mod 16 of the row)
    if constexpr (S == 0) return a - p;
    float d;
    asm("v_subrev_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(p), "v"(a), "n"(S));
    return d;
}
template <int S>
__device__ __forceinline__ float fma_rotated(float g, float w, float acc) {  // acc + (g of lane (i - S) mod 16) * w
    if constexpr (S == 0) return fmaf(g, w, acc);
    asm("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g), "v"(w), "n"(S));
    return acc;
}

#ifndef LVDGS_BWD_NB
#define LVDGS_BWD_NB 8   // survivors per batch of the backward pass: 8 or 4 (A/B builds)
#endif
#ifdef LVDGS_DIAG_PHASES
// diagnostic build: where the workgroup at the head of the order (the longest list) spends its time, in shader clock ticks of wave 0:
// [0] prologue, [1] staging up to its barrier, [2] test + survivor batches, [3] wait for the other waves, [4] flush, [5] rounds, [6] survivors of wave 0, [7] launches
__device__ unsigned long long g_phase_diag[8];
// ... and of EVERY workgroup of the last launch: start, end (s_memrealtime: 100 MHz, one clock for the chip), survivors of its four waves,
// list length, HW_ID, XCC_ID
__device__ unsigned long long g_wg_diag[8192][4];
__device__ unsigned g_wg_surv[8192];
#endif
#ifdef LVDGS_DIAG_FILL
// diagnostic build (tools/fill_diag.py): survivors, splat batches, full batches, (wave, round) pairs with survivors, summed over launches
__device__ unsigned long long g_fill_diag[4];
#endif
// POSE_ONLY (LVDGS_FLAG_POSE_ONLY: the tracking loop, which consumes only dL/dtau and the exposure gradients): the sums that feed
// nothing but the colour and opacity gradients (Su, C0..C2) are left out -- six accumulators per (wave, entry) instead of
// ten, and without a depth gradient the (u, w) matrix holds u alone.
template <bool POSE_ONLY, bool DEPTH_GRAD>
struct Bwd3Shared {
    static constexpr int NB = LVDGS_BWD_NB;
    static constexpr int ACC = POSE_ONLY ? 6 : ACC_STRIDE;   // POSE_ONLY: Sx Sy Sxx Sxy Syy CD
    using MT = std::conditional_t<POSE_ONLY && !DEPTH_GRAD, float, float2>;
    float4 a[BR];                        // x, y, a, b
    float4 b[BR];                        // -c/2*log2e, opacity, depth, -a/2*log2e
    float4 c[BR];                        // r, g, b, -b*log2e
    float craw[BR];                      // c
    uint32_t slot[BR];
    float acc[4][BR * ACC];              // per wave: Sx Sy Sxx Sxy Syy Su C0 C1 C2 CD of every entry it accumulated
    unsigned long long mask[4];
    uint32_t wmax[4], wall[4];           // wall: per wave, the deepest list position its pixels reached (wmax: unused since round 4, kept for the layout)
    float loss_sum[4][4];                // fused loss: [sum][wave]
    MT M[4][NB][64];                     // per wave: (u, w) of [batch slot][pixel]
    uint32_t bj[4][NB];                  // per wave: entry (position in the round) of every batch slot
};

// DEPTH_GRAD = false: the loss has no depth term / the caller passed no gradient of the depth image (monocular tracking):
// the depth image's gradient is identically zero and its terms (one multiply-add per survivor in each pass) are left out.
#ifndef LVDGS_BWD_WGS
#define LVDGS_BWD_WGS 5   // workgroups per CU the backward blend is compiled for (A/B builds)
#endif
#ifndef LVDGS_BWD_WGS_POSE
#define LVDGS_BWD_WGS_POSE 7   // ... and its pose-only form (18.2 KB of LDS, 74 VGPRs; 26.4 KB with a depth gradient: six). Same box, config 3 / KITTI geometry: 5: 240.9 / 92.0 us, 6: 240.4 / 92.1, 7: 233.7 / 92.6, 8: 233.6 / 95.8
#endif
template <int LOSS, bool DEPTH_GRAD, bool POSE_ONLY>
__device__ __forceinline__ void blend_bwd3_body(const BlendParams &p, Bwd3Shared<POSE_ONLY, DEPTH_GRAD> &sh) {
    using Shared = Bwd3Shared<POSE_ONLY, DEPTH_GRAD>;
    constexpr int NB = Shared::NB, ACC = Shared::ACC;
    constexpr bool U_ONLY = POSE_ONLY && !DEPTH_GRAD;   // the matrix holds u alone
    using MT = typename Shared::MT;

#ifdef LVDGS_DIAG_REPEAT   // diagnostic build: the grid repeated $LVDGS_DIAG_REPEAT times (what a launch over several views of this size would cost)
    const int diag_b = (int)blockIdx.x % p.num_tiles;
#define blockIdx_x_ diag_b
#else
#define blockIdx_x_ ((int)blockIdx.x)
#endif
    const int tile = (p.tile_order && *p.order_valid) ? (int)p.tile_order[order_slot_of_workgroup(blockIdx_x_, p.num_tiles)] : p.tile_base + tile_of_workgroup(blockIdx_x_, p.num_tiles);
#undef blockIdx_x_
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qx0 = tx * TILE + (wave & 1) * 8, qy0 = ty * TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < p.W && py < p.H;
    const float pxf = (float)px, pyf = (float)py;
    const float rx0 = (float)qx0, ry0 = (float)qy0, rx1 = (float)(qx0 + 7), ry1 = (float)(qy0 + 7);
    const size_t pix = (size_t)py * p.W + px, P = (size_t)p.W * p.H;

#ifdef LVDGS_DIAG_PHASES
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && blockIdx.x < 8192) g_wg_surv[blockIdx.x] = 0u;
    unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph[7] = {0, 0, 0, 0, 0, 0, 0};
    auto ph_mark = [&](int k) { const unsigned long long n = __builtin_amdgcn_s_memtime(); ph[k] += n - ph_t; ph_t = n; };
#endif
    const uint2 range = p.ranges[tile];
    const float T_final = inside ? p.final_T[pix] : 0.f;
    const uint32_t my_last = inside ? p.n_contrib[pix] : 0u;
    float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gO = 0.f;
    const bool fused_loss = LOSS == LOSS_FUSED || (LOSS == LOSS_PER_VIEW && p.loss_mode == LOSS_FUSED);   // (uniform over the launch / the view)
    if (LOSS == LOSS_MASKED || (LOSS == LOSS_PER_VIEW && !fused_loss)) {
        if (inside) {
            gC0 = p.dL_dcolor[pix]; gC1 = p.dL_dcolor[P + pix]; gC2 = p.dL_dcolor[2 * P + pix];
            if (DEPTH_GRAD && p.loss.gt_depth) {
                // d(depth_lambda * mean over M of |D - Z|) / dD: the statements of masked_depth_kernel<true> (loss.hip)
                const float Dv = p.loss.depth[pix], Z = p.loss.gt_depth[pix];
                const bool in = (!p.loss.grad_mask || p.loss.grad_mask[pix]) && Z > 0.f && Dv > 0.f;
                const float n = p.loss.grad_out[4];
                const float scale = n > 0.f ? p.loss.w_d / n : 0.f;
                const float r = Dv - Z;
                gD = in ? (r > 0.f ? scale : (r < 0.f ? -scale : 0.f)) : 0.f;
            }
        }
    } else if (LOSS == LOSS_FUSED || LOSS == LOSS_PER_VIEW) {
        // the loss's gradient w.r.t. this pixel's colour / depth / opacity, from the rendered images and the targets:
        // what photometric_kernel<2> would have written into three to five gradient images for this pass to read back
        const LossParams &lp = p.loss;
        const LossConsts lc(lp, true);
        float G[3] = {0.f, 0.f, 0.f}, I[3] = {0.f, 0.f, 0.f}, op = 1.f, Z = 0.f, Dv = 0.f, gm = 1.f;
        if (inside) {
#pragma unroll
            for (int c = 0; c < 3; c++) { G[c] = lp.gt_image[c * P + pix]; I[c] = lp.image[c * P + pix]; }
            if (lp.opacity) op = lp.opacity[pix];
            if (lc.has_d) { Z = lp.gt_depth[pix]; Dv = lp.depth[pix]; }
            if (lp.grad_mask) gm = lp.grad_mask[pix] ? 1.f : 0.f;
        }
        const PixelLoss px_loss = photometric_pixel<true, true>(lp, lc, G, I, op, Z, Dv, gm, inside);
        gC0 = px_loss.dI[0]; gC1 = px_loss.dI[1]; gC2 = px_loss.dI[2]; gD = px_loss.dD;
        gO = p.loss_propagate_opacity ? px_loss.dO : 0.f;
        // the tile's four partial sums (loss value: colour, depth; exposure gradients: a, b), wave by wave
        const float sums[4] = {px_loss.v_rgb, px_loss.v_d, px_loss.s_a, px_loss.s_b};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float v = wave_sum_to_lane63(sums[q]);
            if (lane == 63) sh.loss_sum[q][wave] = v;
        }
    } else if (inside) {
        gC0 = p.dL_dcolor[pix]; gC1 = p.dL_dcolor[P + pix]; gC2 = p.dL_dcolor[2 * P + pix];
        if (p.dL_ddepth) gD = p.dL_ddepth[pix];
        if (p.dL_dopacity) gO = p.dL_dopacity[pix];
    }
    if constexpr (!DEPTH_GRAD) gD = 0.f;
    const float tail = T_final * (p.bg[0] * gC0 + p.bg[1] * gC1 + p.bg[2] * gC2 - gO);

    uint32_t m = my_last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    // A quadrant whose 64 pixels all receive a zero gradient (masked out of the loss: the tracking loss's edge and
    // brightness masks, a static mask) adds exactly zero to every sum: its wave sits the lists out.
    const uint32_t m_all = m;
    if (__ballot(gC0 != 0.f || gC1 != 0.f || gC2 != 0.f || gD != 0.f || gO != 0.f) == 0ull) m = 0u;
    if (lane == 0) sh.wall[wave] = m_all;
    const int wave_last = (int)m;
    __syncthreads();
    if (fused_loss) {
        if (tid < 4) p.loss.partial[4 * (size_t)tile + tid] = ((sh.loss_sum[tid][0] + sh.loss_sum[tid][1]) + sh.loss_sum[tid][2]) + sh.loss_sum[tid][3];
    }
    // Only the entries some pixel of the tile composited can receive anything: the list is walked from the deepest of them.
    // The pairs behind it -- on opaque surfaces nine tenths of a list -- are neither staged nor written; their slots keep
    // pair_valid = 0 and preprocess_bwd passes over them.  (The depth is the forward pass's alone, not this call's
    // gradients': every backward pass after one forward writes the same set of records.)
    const int todo = min((int)(range.y - range.x), (int)max(max(sh.wall[0], sh.wall[1]), max(sh.wall[2], sh.wall[3])));

    float T = T_final, R = tail;  // see the single-pass kernel for the scalar recurrence
    // splat pass: this lane's slot in the batch and its 16-lane row; at step s it looks at the pixel of lane (col - s) mod 16
    const int row = lane >> 4, col = lane & 15;
    const int my_slot = col & (NB - 1);
    const MT *Mrow = &sh.M[wave][my_slot][row * 16];
    MT *const Mmine = &sh.M[wave][0][lane];

    const int rounds = (todo + BR - 1) / BR;
#ifdef LVDGS_DIAG_FILL
    unsigned diag_surv = 0, diag_batches = 0, diag_full = 0, diag_rounds = 0;
#endif
#ifdef LVDGS_DIAG_PHASES
    ph_mark(0);
#endif
    // The id a staging thread gathers its record with is requested a round AHEAD (round 5): a round's staging is two dependent trips to
    // memory -- the id, then the record -- in front of a barrier the whole workgroup waits at, and on a small grid the kernel lasts as
    // long as its longest list's chain of rounds (a KITTI frame's heaviest tile: 14 rounds), with nobody else on the CU to hide them.
#ifndef LVDGS_BWD_ID_AHEAD
#define LVDGS_BWD_ID_AHEAD 1   // A/B builds: 0 = the id is loaded in the round that stages it
#endif
    uint32_t id_ahead = 0u;
    if (LVDGS_BWD_ID_AHEAD && rounds > 0 && tid < min(BR, todo - (rounds - 1) * BR)) id_ahead = p.point_list[range.x + (rounds - 1) * BR + tid];
    for (int r = rounds - 1; r >= 0; r--) {
        const int base = r * BR;
        const int cnt = min(BR, todo - base);
        if (tid < cnt) {
            const uint32_t id = LVDGS_BWD_ID_AHEAD ? id_ahead : p.point_list[range.x + base + tid];
            const float4 *r4 = reinterpret_cast<const float4 *>(p.rec + (size_t)id * REC_FLOATS);
            const float4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
            sh.a[tid] = r0;
            sh.b[tid] = make_float4(-0.5f * LOG2E * r1.x, r1.y, r2.y, -0.5f * LOG2E * r0.z);
            sh.c[tid] = make_float4(r1.z, r1.w, r2.x, -LOG2E * r0.w);
            sh.craw[tid] = r1.x;
            sh.slot[tid] = pair_slot(p, id, tx, ty);
        }
        __syncthreads();
        if (LVDGS_BWD_ID_AHEAD && r > 0 && tid < BR) id_ahead = p.point_list[range.x + base - BR + tid];   // (every round but the last listed is full)
#ifdef LVDGS_DIAG_PHASES
        ph_mark(1); ph[5]++;
#endif
        uint64_t wrote = 0ull;
        if (base < wave_last) {   // (a wave none of whose pixels got this far has nothing to test: on opaque surfaces two thirds of the (wave, round) pairs)
            bool keep = false;
            if (lane < cnt && base + lane < wave_last) {
                const float4 A = sh.a[lane];
                keep = reaches_rect(A.x, A.y, A.z, A.w, sh.craw[lane], sh.b[lane].y, rx0, ry0, rx1, ry1);
            }
            uint64_t live = __ballot(keep);
            const uint32_t rel_last = my_last > (uint32_t)base ? my_last - (uint32_t)base : 0u;  // entries below this position composited
            wrote = live;
#ifdef LVDGS_DIAG_PHASES
            ph[6] += (unsigned long long)__popcll(live);
#endif
#ifdef LVDGS_DIAG_FILL
            diag_surv += (unsigned)__popcll(live); diag_rounds += live ? 1u : 0u;
#endif
            while (live) {
                // ---------------- pixel pass: the next NB survivors, back to front ----------------
                const uint64_t before = live;
                const int nb = min(NB, (int)__popcll(live));
#ifdef LVDGS_DIAG_FILL
                diag_batches++; diag_full += nb == NB ? 1u : 0u;
#endif
                struct Rec { float4 A, B, Cc; };
                auto next_entry = [&]() {
                    const int j = 63 - __builtin_clzll(live);
                    live = mask_clear_bit(live, j);
                    return j;
                };
                auto fetch = [&](int j) { return Rec{sh.a[j], sh.b[j], sh.c[j]}; };
                auto entry = [&](auto slot, int j, const Rec &rc) {
                    const float4 A = rc.A, B = rc.B, Cc = rc.Cc;
                    const float dx = A.x - pxf, dy = A.y - pyf;
                    // the same expression, operand for operand, as the forward pass: identical hit set
                    const float pw2 = fmaf(dx, fmaf(B.w, dx, Cc.w * dy), B.x * dy * dy);
                    const float G = __builtin_amdgcn_exp2f(pw2);
                    const float alpha = fminf(ALPHA_MAX, B.y * G);
                    const bool hit = ((uint32_t)j < rel_last) && (pw2 <= 0.f) && (alpha >= ALPHA_MIN);
                    // lanes that miss run the same arithmetic with alpha = G = 0: 1 / (1 - 0) is exactly 1, so T and R
                    // keep their values and (u, w) = (0, 0)
                    const float alpha_h = hit ? alpha : 0.f, G_h = hit ? G : 0.f;
                    const float k_rgb = fmaf(Cc.z, gC2, fmaf(Cc.y, gC1, Cc.x * gC0));
                    const float k = DEPTH_GRAD ? fmaf(B.z, gD, k_rgb) : k_rgb;
                    const float inv = __builtin_amdgcn_rcpf(1.f - alpha_h);
                    T *= inv;
                    const float w = alpha_h * T;
                    const float dL_dalpha = fmaf(k, T, -(R * inv));
                    R = fmaf(k, w, R);
                    if constexpr (U_ONLY) Mmine[(int)slot * 64] = G_h * dL_dalpha;
                    else Mmine[(int)slot * 64] = make_float2(G_h * dL_dalpha, w);
                };
                if (nb == NB) {
                    // straight-line; the records of entries s + 1 and s + 2 are on their way while entry s is evaluated
                    int js[NB];
#pragma unroll
                    for (int s = 0; s < NB; s++) js[s] = next_entry();
                    Rec r0 = fetch(js[0]), r1 = fetch(js[1]), r2 = fetch(js[2]);
                    entry(std::integral_constant<int, 0>{}, js[0], r0); r0 = fetch(js[3]);
                    if constexpr (NB == 8) {
                        entry(std::integral_constant<int, 1>{}, js[1], r1); r1 = fetch(js[4]);
                        entry(std::integral_constant<int, 2>{}, js[2], r2); r2 = fetch(js[5]);
                        entry(std::integral_constant<int, 3>{}, js[3], r0); r0 = fetch(js[6]);
                        entry(std::integral_constant<int, 4>{}, js[4], r1); r1 = fetch(js[NB - 1]);
                        entry(std::integral_constant<int, 5>{}, js[5], r2);
                        entry(std::integral_constant<int, 6>{}, js[6], r0);
                        entry(std::integral_constant<int, NB - 1>{}, js[NB - 1], r1);
                    } else {
                        entry(std::integral_constant<int, 1>{}, js[1], r1);
                        entry(std::integral_constant<int, 2>{}, js[2], r2);
                        entry(std::integral_constant<int, 3>{}, js[3], r0);
                    }
                } else {
                    for (int s = 0; s < nb; s++) {
                        const int j = next_entry();
                        entry(s, j, fetch(j));
                    }
                }
                const uint64_t batch = before ^ live;
                // slot table: the entry at bit position `lane` of the batch was given slot = number of batch bits above it
                if ((batch >> lane) & 1ull) sh.bj[wave][__popcll(batch >> lane) - 1] = (uint32_t)lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // ---------------- splat pass: lane -> (slot, 8 pixels of its row) ----------------
                const bool valid = my_slot < nb;
                const int jj = valid ? (int)sh.bj[wave][my_slot] : 0;
                const float2 Aj = *reinterpret_cast<const float2 *>(&sh.a[jj]);
                float Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, Su = 0.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, CD = 0.f;
                // all eight (u, w) reads go out first: their addresses are known, and at 5 waves per SIMD a read issued
                // one step ahead is still on its way when it is needed
                float2 uw8[NB];
#pragma unroll
                for (int s8 = 0; s8 < NB; s8++) {
                    if constexpr (U_ONLY) uw8[s8] = make_float2(Mrow[(col - s8) & 15], 0.f);
                    else uw8[s8] = Mrow[(col - s8) & 15];
                }
#define LVDGS_SPLAT_STEP(S)                                                                          \
                {                                                                                    \
                    const float2 uw = uw8[S];                                                        \
                    const float dx = sub_rotated<S>(Aj.x, pxf), dy = sub_rotated<S>(Aj.y, pyf);      \
                    const float t1 = uw.x * dx, t2 = uw.x * dy;                                      \
                    Sx += t1; Sy += t2;                                                              \
                    Sxx = fmaf(t1, dx, Sxx); Sxy = fmaf(t1, dy, Sxy); Syy = fmaf(t2, dy, Syy);       \
                    if constexpr (!POSE_ONLY) {                                                      \
                        Su += uw.x;                                                                  \
                        C0 = fma_rotated<S>(gC0, uw.y, C0); C1 = fma_rotated<S>(gC1, uw.y, C1);      \
                        C2 = fma_rotated<S>(gC2, uw.y, C2);                                          \
                    }                                                                                \
                    if constexpr (DEPTH_GRAD) CD = fma_rotated<S>(gD, uw.y, CD);                     \
                }
                LVDGS_SPLAT_STEP(0) LVDGS_SPLAT_STEP(1) LVDGS_SPLAT_STEP(2) LVDGS_SPLAT_STEP(3)
                if constexpr (NB == 8) { LVDGS_SPLAT_STEP(NB - 4) LVDGS_SPLAT_STEP(NB - 3) LVDGS_SPLAT_STEP(NB - 2) LVDGS_SPLAT_STEP(NB - 1) }
#undef LVDGS_SPLAT_STEP
                // fold the 8 partial sums of every slot: rows first (two pairwise folds, ten registers -> three) ...
                float q0 = fold16(fold32(Sx, Sy), fold32(Sxx, Sxy));   // rows: Sx Sxx Sy Sxy
                // (POSE_ONLY: Syy and CD take the places -- and so the summation trees -- they have in the full form: the
                // pose gradient comes out bit for bit the same)
                float q1 = POSE_ONLY ? fold16(fold32(Syy, CD), Syy) : fold16(fold32(Syy, Su), fold32(C0, C1));    // rows: Syy C0 Su C1 | Syy x CD x
                float q2 = POSE_ONLY ? 0.f : fold16(fold32(C2, CD), C1);                                             // rows: C2 x CD x
                q0 += row_rotate<8>(q0);  // ... then the two half-rows (NB = 4: four quarter-rows) that share a slot
                q1 += row_rotate<8>(q1);
                if constexpr (!POSE_ONLY) q2 += row_rotate<8>(q2);
                if constexpr (NB == 4) { q0 += row_rotate<4>(q0); q1 += row_rotate<4>(q1); q2 += row_rotate<4>(q2); }
                if (valid && col < NB) {
                    // value index held by this row: q0 -> {Sx, Sxx, Sy, Sxy}, q1 -> {Syy, C0, Su, C1}, q2 -> {C2, -, CD, -}
                    const int i0 = row == 0 ? 0 : (row == 1 ? 2 : (row == 2 ? 1 : 3));
                    float *o = &sh.acc[wave][jj * ACC];
                    o[i0] = q0;
                    if constexpr (POSE_ONLY) {
                        if ((row & 1) == 0) o[row == 0 ? 4 : 5] = q1;   // Syy, CD
                    } else {
                        const int i1 = row == 0 ? 4 : (row == 1 ? 6 : (row == 2 ? 5 : 7));
                        o[i1] = q1;
                        if ((row & 1) == 0) o[row == 0 ? 8 : 9] = q2;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) sh.mask[wave] = wrote;
#ifdef LVDGS_DIAG_PHASES
        ph_mark(2);
#endif
        __syncthreads();
#ifdef LVDGS_DIAG_PHASES
        ph_mark(3);
#endif
        if (tid < cnt) {
            float acc[ACC];
#pragma unroll
            for (int k = 0; k < ACC; k++) acc[k] = 0.f;
            const unsigned long long bit = 1ull << tid;
#pragma unroll
            for (int w = 0; w < 4; w++)
                if (sh.mask[w] & bit) {
                    const float2 *o = reinterpret_cast<const float2 *>(&sh.acc[w][tid * ACC]);
#pragma unroll
                    for (int k = 0; k < ACC / 2; k++) { const float2 t = o[k]; acc[2 * k] += t.x; acc[2 * k + 1] += t.y; }
                }
            // acc: Sx Sy Sxx Sxy Syy Su C0 C1 C2 CD (sums of u, not yet of h = opacity * u); POSE_ONLY: Sx Sy Sxx Sxy Syy CD
            const float4 A = sh.a[tid];
            const float op = sh.b[tid].y;
            const float sx = op * acc[0], sy = op * acc[1];
            // the pair's record: ten floats, 40 bytes (8-byte aligned: five 8-byte stores), and its "written" flag;
            // POSE_ONLY: d/d(mean, conic, view depth) alone, six floats at the same 8-byte alignment
            p.pair_valid[sh.slot[tid]] = 1;
            float2 *dst = reinterpret_cast<float2 *>(p.pair_grads + (size_t)sh.slot[tid] * (POSE_ONLY ? PAIR_FLOATS_POSE : PAIR_FLOATS));
            dst[0] = make_float2(-fmaf(A.z, sx, A.w * sy), -fmaf(sh.craw[tid], sy, A.w * sx));
            dst[1] = make_float2(-0.5f * (op * acc[2]), -(op * acc[3]));
            dst[2] = make_float2(-0.5f * (op * acc[4]), acc[5]);
            if constexpr (!POSE_ONLY) {
                dst[3] = make_float2(acc[6], acc[7]);
                dst[4] = make_float2(acc[8], acc[9]);
            }
        }
        __syncthreads();
#ifdef LVDGS_DIAG_PHASES
        ph_mark(4);
#endif
    }
#ifdef LVDGS_DIAG_PHASES
    if (blockIdx.x == 0 && tid == 0) {
        for (int k = 0; k < 7; k++) atomicAdd(&g_phase_diag[k], ph[k]);
        atomicAdd(&g_phase_diag[7], 1ull);
    }
    if (blockIdx.x < 8192) {
        if (lane == 0) atomicAdd(&g_wg_surv[blockIdx.x], (unsigned)ph[6]);
        if (tid == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
            g_wg_diag[blockIdx.x][0] = wg_t0; g_wg_diag[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
            g_wg_diag[blockIdx.x][2] = ((unsigned long long)(range.y - range.x) << 32) | (unsigned)tile;
            g_wg_diag[blockIdx.x][3] = ((unsigned long long)xcc << 32) | hw;
        }
    }
#endif
#ifdef LVDGS_DIAG_FILL
    if (lane == 0) {
        atomicAdd(&g_fill_diag[0], (unsigned long long)diag_surv); atomicAdd(&g_fill_diag[1], (unsigned long long)diag_batches);
        atomicAdd(&g_fill_diag[2], (unsigned long long)diag_full); atomicAdd(&g_fill_diag[3], (unsigned long long)diag_rounds);
    }
#endif
}

template <int LOSS, bool DEPTH_GRAD = true, bool POSE_ONLY = false>
__global__ void __launch_bounds__(256, POSE_ONLY ? (DEPTH_GRAD ? 6 : LVDGS_BWD_WGS_POSE) : LVDGS_BWD_WGS) blend_bwd3_kernel(BlendParams p) {
    __shared__ Bwd3Shared<POSE_ONLY, DEPTH_GRAD> sh;
    blend_bwd3_body<LOSS, DEPTH_GRAD, POSE_ONLY>(p, sh);
}
// several views in one launch (blockIdx.y: the view), the loss in the prologue
template <int LOSS, bool DEPTH_GRAD, bool POSE_ONLY>
__global__ void __launch_bounds__(256, POSE_ONLY ? (DEPTH_GRAD ? 6 : LVDGS_BWD_WGS_POSE) : LVDGS_BWD_WGS) blend_bwd3_batch_kernel(BlendBatch b) {
    __shared__ Bwd3Shared<POSE_ONLY, DEPTH_GRAD> sh;
    blend_bwd3_body<LOSS, DEPTH_GRAD, POSE_ONLY>(b.v[blockIdx.y], sh);
}

// ------------------------------------------------------------------------------------------
// The forward AND the backward blend pass of a tile in ONE launch (lvdgs_forward_backward_fused_loss: a view whose loss is the
// photometric one, evaluated per pixel -- the tracking iteration, a mapping view without a static mask).  Everything the backward
// pass of a tile reads of the forward pass is that tile's own: its pixels' colour / depth / opacity, final transmittance and last
// contributor, and the loss's gradient at a pixel depends on that pixel alone.  On a small grid (a KITTI frame: 1848 tiles for
// 1280-2048 workgroup places) each of the two kernels lasts as long as its heaviest tiles -- 41 and 75 us of the 54 and 105 the
// whole frame takes -- while most of the chip has run dry: one launch pays that tail once, and the light tiles' workgroups are
// through both passes while the heavy ones are still in their first.  The two passes are the bodies above, one after the other;
// their LDS is overlaid; a thread reads back only what it wrote itself (the same thread -> pixel mapping in both passes).
template <bool DEPTH_GRAD, bool POSE_ONLY, bool DEEP_LISTS>
__global__ void __launch_bounds__(256, POSE_ONLY ? (DEPTH_GRAD ? 6 : LVDGS_BWD_WGS_POSE) : LVDGS_BWD_WGS) blend_fwd_bwd_kernel(BlendParams p) {
    __shared__ union U { FwdShared f; Bwd3Shared<POSE_ONLY, DEPTH_GRAD> b; __device__ U() {} } u;
#if LVDGS_HEAVY_PRIO
    // The launch lasts as long as its heaviest tile's own chain of rounds (a KITTI frame: 116 us of ~136 when that workgroup has its
    // CU to itself, more beside four others): the waves of the heaviest tiles -- the head of the tile order -- issue first.
    if (p.tile_order && *p.order_valid) {
        const int slot = order_slot_of_workgroup((int)blockIdx.x, p.num_tiles);
        if (slot < (p.num_tiles >> LVDGS_HEAVY_PRIO_SHIFT)) __builtin_amdgcn_s_setprio(LVDGS_HEAVY_PRIO);
    }
#endif
    blend_fwd2_body<DEEP_LISTS>(p, u.f);
    if (p.pair_total && *p.pair_total > p.pair_capacity) return;   // (uniform: the forward's outputs are invalid too, the caller knows)
    __threadfence_block();
    __syncthreads();   // (every wave is through with the forward pass's LDS)
    blend_bwd3_body<LOSS_FUSED, DEPTH_GRAD, POSE_ONLY>(p, u.b);
}

BlendParams make_params(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im) {
    BlendParams p{};
    p.W = a.image_width; p.H = a.image_height;
    p.gx = (p.W + TILE - 1) / TILE; p.gy = (p.H + TILE - 1) / TILE;
    int row0, row1;
    tile_row_band(a, &row0, &row1);
    p.num_tiles = p.gx * (row1 - row0); p.tile_base = p.gx * row0;
    const int all_tiles = p.gx * p.gy;
    p.ranges = im.ranges; p.tile_order = tile_order_in_use(all_tiles) ? im.long_tiles + all_tiles : nullptr; p.order_valid = im.long_count + 1; p.point_list = b.point_list; p.rec = g.rec; p.slot_base = g.slot_base; p.rect = g.rect; p.bg = a.bg;
    p.out_color = a.out_color; p.out_depth = a.out_depth; p.out_opacity = a.out_opacity;
    p.final_T = im.final_T; p.n_contrib = im.n_contrib; p.n_touched = a.n_touched;
    p.dL_dcolor = a.dL_dout_color; p.dL_ddepth = a.dL_dout_depth; p.dL_dopacity = a.dL_dout_opacity;
    return p;
}

}  // namespace

int launch_blend_fwd(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, bool deep_lists, hipStream_t s) {
    BlendParams p = make_params(a, g, b, im);
    if (p.num_tiles == 0) return LVDGS_OK;
#ifdef LVDGS_DIAG_GRID   // diagnostic build: only the first $LVDGS_DIAG_GRID workgroups (small grids: the heaviest tiles) -- timings, not results
    if (const char *e = getenv("LVDGS_DIAG_GRID")) p.num_tiles = min(p.num_tiles, atoi(e));
#endif
    ProfScope ps("blend_fwd", s);
    int grid = p.num_tiles;
#ifdef LVDGS_DIAG_REPEAT
    if (const char *e = getenv("LVDGS_DIAG_REPEAT")) grid *= atoi(e);
#endif
    if (deep_lists) hipLaunchKernelGGL(blend_fwd2_deep_kernel, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(blend_fwd2_kernel, dim3(grid), dim3(256), 0, s, p);
    LVDGS_LAUNCH_CHECK("blend_fwd", a.debug, s);
    return LVDGS_OK;
}

#ifndef LVDGS_BWD_DEPTH_ALWAYS
#define LVDGS_BWD_DEPTH_ALWAYS 0   // A/B builds: 1 keeps the depth-gradient terms whatever the loss
#endif
int launch_blend_bwd(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, const BwdScratch &w,
                     hipStream_t s) {
    BlendParams p = make_params(a, g, b, im);
    p.pair_grads = w.pair_grads; p.pair_valid = b.pair_valid;
    if (p.num_tiles == 0) return LVDGS_OK;
#ifdef LVDGS_DIAG_GRID
    if (const char *e = getenv("LVDGS_DIAG_GRID")) p.num_tiles = min(p.num_tiles, atoi(e));
#endif
    ProfScope ps("blend_bwd", s);
    int grid = p.num_tiles;
#ifdef LVDGS_DIAG_REPEAT
    if (const char *e = getenv("LVDGS_DIAG_REPEAT")) grid *= atoi(e);
#endif
    const bool depth = LVDGS_BWD_DEPTH_ALWAYS || p.dL_ddepth, pose_only = (a.flags & LVDGS_FLAG_POSE_ONLY) != 0;
    if (pose_only && depth) hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_IMAGES, true, true>), dim3(grid), dim3(256), 0, s, p);
    else if (pose_only) hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_IMAGES, false, true>), dim3(grid), dim3(256), 0, s, p);
    else if (depth) hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_IMAGES, true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_IMAGES, false>), dim3(grid), dim3(256), 0, s, p);
    LVDGS_LAUNCH_CHECK("blend_bwd", a.debug, s);
    return LVDGS_OK;
}

int launch_blend_bwd_fused_loss(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, const BwdScratch &w,
                                const LossParams &loss, int propagate_opacity, hipStream_t s) {
    BlendParams p = make_params(a, g, b, im);
    p.pair_grads = w.pair_grads; p.pair_valid = b.pair_valid;
    p.loss = loss;
    p.loss_propagate_opacity = propagate_opacity;
    if (p.num_tiles == 0) return LVDGS_OK;
#ifdef LVDGS_DIAG_GRID
    if (const char *e = getenv("LVDGS_DIAG_GRID")) p.num_tiles = min(p.num_tiles, atoi(e));
#endif
    ProfScope ps("blend_bwd", s);
    int grid = p.num_tiles;
#ifdef LVDGS_DIAG_REPEAT
    if (const char *e = getenv("LVDGS_DIAG_REPEAT")) grid *= atoi(e);
#endif
    const bool depth = LVDGS_BWD_DEPTH_ALWAYS || (loss.depth && loss.gt_depth && loss.w_d != 0.f), pose_only = (a.flags & LVDGS_FLAG_POSE_ONLY) != 0;
    if (pose_only && depth) hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_FUSED, true, true>), dim3(grid), dim3(256), 0, s, p);
    else if (pose_only) hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_FUSED, false, true>), dim3(grid), dim3(256), 0, s, p);
    else if (depth) hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_FUSED, true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_FUSED, false>), dim3(grid), dim3(256), 0, s, p);
    LVDGS_LAUNCH_CHECK("blend_bwd", a.debug, s);
    return LVDGS_OK;
}

// The static-mask mapping loss's fields where the kernel looks for them (BlendParams::loss_mode).
static void set_masked_loss(BlendParams &p, const MaskedLossView &m) {
    p.loss = LossParams{};
    p.dL_dcolor = m.d_image; p.dL_ddepth = nullptr; p.dL_dopacity = nullptr;
    p.loss.depth = m.gt_depth ? m.depth : nullptr; p.loss.gt_depth = m.gt_depth; p.loss.grad_mask = m.static_mask;
    p.loss.w_d = m.depth_lambda; p.loss.grad_out = m.out;
    p.loss_mode = LOSS_MASKED; p.loss_propagate_opacity = 0;
}

int launch_blend_bwd_masked_loss(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, const BwdScratch &w,
                                 const MaskedLossView &m, hipStream_t s) {
    BlendParams p = make_params(a, g, b, im);
    p.pair_grads = w.pair_grads; p.pair_valid = b.pair_valid;
    set_masked_loss(p, m);
    if (p.num_tiles == 0) return LVDGS_OK;
    ProfScope ps("blend_bwd", s);
    if (LVDGS_BWD_DEPTH_ALWAYS || m.gt_depth) hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_MASKED, true>), dim3(p.num_tiles), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((blend_bwd3_kernel<LOSS_MASKED, false>), dim3(p.num_tiles), dim3(256), 0, s, p);
    LVDGS_LAUNCH_CHECK("blend_bwd (masked loss)", a.debug, s);
    return LVDGS_OK;
}

// One launch for both blend passes of a view (api.hip: lvdgs_forward_backward_fused_loss decides when).
int launch_blend_fwd_bwd_fused_loss(const lvdgs_args &a, const GeomView &g, const BinView &b, const ImageView &im, const BwdScratch &w,
                                    const LossParams &loss, int propagate_opacity, bool deep_lists, const uint32_t *pair_total, uint32_t pair_capacity,
                                    hipStream_t s) {
    BlendParams p = make_params(a, g, b, im);
    p.pair_grads = w.pair_grads; p.pair_valid = b.pair_valid;
    p.pair_total = pair_total; p.pair_capacity = pair_capacity;
    p.loss = loss; p.loss_mode = LOSS_FUSED;
    p.loss_propagate_opacity = propagate_opacity;
    if (p.num_tiles == 0) return LVDGS_OK;
    ProfScope ps("blend_fwd_bwd", s);
    const dim3 grid(p.num_tiles), block(256);
    const bool depth = LVDGS_BWD_DEPTH_ALWAYS || (loss.depth && loss.gt_depth && loss.w_d != 0.f), pose_only = (a.flags & LVDGS_FLAG_POSE_ONLY) != 0;
#define LVDGS_FUSED_LAUNCH(D, PO)                                                                                        \
    do {                                                                                                                \
        if (deep_lists) hipLaunchKernelGGL((blend_fwd_bwd_kernel<D, PO, true>), grid, block, 0, s, p);                  \
        else hipLaunchKernelGGL((blend_fwd_bwd_kernel<D, PO, false>), grid, block, 0, s, p);                            \
    } while (0)
    if (pose_only && depth) LVDGS_FUSED_LAUNCH(true, true);
    else if (pose_only) LVDGS_FUSED_LAUNCH(false, true);
    else if (depth) LVDGS_FUSED_LAUNCH(true, false);
    else LVDGS_FUSED_LAUNCH(false, false);
#undef LVDGS_FUSED_LAUNCH
    LVDGS_LAUNCH_CHECK("blend_fwd_bwd", a.debug, s);
    return LVDGS_OK;
}

// The blend passes of n views of one size in one launch each (api.hip: lvdgs_blend_forward_batch / lvdgs_blend_backward_fused_loss_batch).
int launch_blend_fwd_batch(const lvdgs_args *const *a, const GeomView *g, const BinView *b, const ImageView *im, int n, bool deep_lists, hipStream_t s) {
    for (int first = 0; first < n; first += BATCH_VIEWS) {
        const int m = min(BATCH_VIEWS, n - first);
        BlendBatch batch{};
        for (int k = 0; k < m; k++) batch.v[k] = make_params(*a[first + k], g[first + k], b[first + k], im[first + k]);
        if (batch.v[0].num_tiles == 0) continue;
        ProfScope ps("blend_fwd", s);
        if (deep_lists) hipLaunchKernelGGL(blend_fwd2_deep_batch_kernel, dim3(batch.v[0].num_tiles, m), dim3(256), 0, s, batch);
        else hipLaunchKernelGGL(blend_fwd2_batch_kernel, dim3(batch.v[0].num_tiles, m), dim3(256), 0, s, batch);
        LVDGS_LAUNCH_CHECK("blend_fwd (batch)", a[first]->debug, s);
    }
    return LVDGS_OK;
}

// masked[v] (may be null as a whole): the view is scored by the static-mask mapping loss instead of loss[v].
int launch_blend_bwd_fused_loss_batch(const lvdgs_args *const *a, const GeomView *g, const BinView *b, const ImageView *im, const BwdScratch *w,
                                      const LossParams *loss, const MaskedLossView *const *masked, int n, int propagate_opacity, hipStream_t s) {
    for (int first = 0; first < n; first += BATCH_VIEWS) {
        const int m = min(BATCH_VIEWS, n - first);
        BlendBatch batch{};
        bool depth = false;
        int n_masked = 0;
        for (int k = 0; k < m; k++) {
            const int v = first + k;
            BlendParams &p = batch.v[k];
            p = make_params(*a[v], g[v], b[v], im[v]);
            p.pair_grads = w[v].pair_grads; p.pair_valid = b[v].pair_valid;
            if (masked && masked[v]) {
                set_masked_loss(p, *masked[v]);
                depth = depth || LVDGS_BWD_DEPTH_ALWAYS || masked[v]->gt_depth;
                n_masked++;
            } else {
                p.loss = loss[v];
                p.loss_mode = LOSS_FUSED;
                p.loss_propagate_opacity = propagate_opacity;
                depth = depth || LVDGS_BWD_DEPTH_ALWAYS || (loss[v].depth && loss[v].gt_depth && loss[v].w_d != 0.f);
            }
        }
        if (batch.v[0].num_tiles == 0) continue;
        ProfScope ps("blend_bwd", s);
        const bool pose_only = (a[first]->flags & LVDGS_FLAG_POSE_ONLY) != 0;
        const dim3 grid(batch.v[0].num_tiles, m);
        if (n_masked == m) {
            if (depth) hipLaunchKernelGGL((blend_bwd3_batch_kernel<LOSS_MASKED, true, false>), grid, dim3(256), 0, s, batch);
            else hipLaunchKernelGGL((blend_bwd3_batch_kernel<LOSS_MASKED, false, false>), grid, dim3(256), 0, s, batch);
        } else if (n_masked) {
            if (depth) hipLaunchKernelGGL((blend_bwd3_batch_kernel<LOSS_PER_VIEW, true, false>), grid, dim3(256), 0, s, batch);
            else hipLaunchKernelGGL((blend_bwd3_batch_kernel<LOSS_PER_VIEW, false, false>), grid, dim3(256), 0, s, batch);
        } else if (pose_only && depth) hipLaunchKernelGGL((blend_bwd3_batch_kernel<LOSS_FUSED, true, true>), grid, dim3(256), 0, s, batch);
        else if (pose_only) hipLaunchKernelGGL((blend_bwd3_batch_kernel<LOSS_FUSED, false, true>), grid, dim3(256), 0, s, batch);
        else if (depth) hipLaunchKernelGGL((blend_bwd3_batch_kernel<LOSS_FUSED, true, false>), grid, dim3(256), 0, s, batch);
        else hipLaunchKernelGGL((blend_bwd3_batch_kernel<LOSS_FUSED, false, false>), grid, dim3(256), 0, s, batch);
        LVDGS_LAUNCH_CHECK("blend_bwd (batch)", a[first]->debug, s);
    }
    return LVDGS_OK;
}

}  // namespace lvdgs

#ifdef LVDGS_DIAG_PHASES
extern "C" int lvdgs_diag_phases(unsigned long long *out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(lvdgs::g_phase_diag), 8 * sizeof(unsigned long long)) != hipSuccess) return LVDGS_E_HIP;
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lvdgs::g_phase_diag), z, sizeof(z)) != hipSuccess) return LVDGS_E_HIP;
    }
    return LVDGS_OK;
}
#endif
#ifdef LVDGS_DIAG_PHASES
extern "C" int lvdgs_diag_workgroups(unsigned long long *out_8192x4, unsigned *surv_8192) {
    if (hipMemcpyFromSymbol(out_8192x4, HIP_SYMBOL(lvdgs::g_wg_diag), 8192 * 4 * sizeof(unsigned long long)) != hipSuccess) return LVDGS_E_HIP;
    if (hipMemcpyFromSymbol(surv_8192, HIP_SYMBOL(lvdgs::g_wg_surv), 8192 * sizeof(unsigned)) != hipSuccess) return LVDGS_E_HIP;
    return LVDGS_OK;
}
#endif
#ifdef LVDGS_DIAG_FILL
extern "C" int lvdgs_diag_fill(unsigned long long *out4, int reset) {
    if (out4 && hipMemcpyFromSymbol(out4, HIP_SYMBOL(lvdgs::g_fill_diag), 4 * sizeof(unsigned long long)) != hipSuccess) return LVDGS_E_HIP;
    if (reset) {
        const unsigned long long z[4] = {0ull, 0ull, 0ull, 0ull};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lvdgs::g_fill_diag), z, sizeof(z)) != hipSuccess) return LVDGS_E_HIP;
    }
    return LVDGS_OK;
}
#endif
