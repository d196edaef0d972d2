/*
 * lvdgs.h -- C ABI of the MI355X (gfx950) differentiable 3D-Gaussian tile rasterizer.
 *
 * This is the drop-in boundary for the path LVD-GS reaches through
 *     gaussian_splatting.gaussian_renderer.render(viewpoint, gaussians, pipe, bg)
 * (call sites: reference utils/slam_frontend.py:1493, utils/slam_backend.py:98,184,277,407,
 *  utils/eval_utils_0806.py:215, utils/init_pose.py:145).  Upstream that facade calls the
 * `diff_gaussian_rasterization` extension (reference README.md:43; sources absent from the
 * reference checkout), whose Python-visible entry points are
 *     _C.rasterize_gaussians(...)            -> lvdgs_forward_prepare + lvdgs_forward_render
 *     _C.rasterize_gaussians_backward(...)   -> lvdgs_backward
 *     _C.mark_visible(...)                   -> lvdgs_mark_visible
 * Other absent native dependencies the north star names:
 *     simple_knn._C.distCUDA2  (README.md:42) -> lvdgs_dist2_knn3
 *     curope rope_2d           (README.md:49) -> lvdgs_rope2d
 *
 * Conventions
 *  - plain C: raw device pointers, sizes, a hipStream_t passed as void*; no C++ exceptions
 *    cross the boundary; every entry point returns an lvdgs_status (0 = ok) and
 *    lvdgs_last_error() returns a thread-local message for the last failure;
 *  - all work is enqueued on the given stream and is ordered with it; the host waits for the
 *    device in two places only, both for the number of (Gaussian, tile) pairs the caller sizes
 *    the binning buffer with (the device->host read upstream performs): lvdgs_forward_prepare
 *    synchronises the stream, lvdgs_forward waits -- with everything else of the frame already
 *    enqueued -- until the tile-scan kernel has stored the count into pinned host memory;
 *  - the library owns no device memory: the caller allocates the three state buffers
 *    (geometry, binning, image -- upstream's geomBuffer / binningBuffer / imgBuffer) and the
 *    scratch buffers, with the sizes the lvdgs_*_bytes functions report, and keeps the state
 *    buffers alive until backward (PyTorch: ctx.save_for_backward);
 *  - no global mutable state that results depend on: besides the optional profiling counters, lvdgs_forward keeps a
 *    per-thread, per-device block of pinned, device-visible words (the tile-scan kernel writes the pair count lvdgs_forward
 *    returns there, and the frame's longest tile segment, which only selects WHICH kernels the next frame launches for its long tile
 *    lists -- the sort kernels, and the deep-lists build of the forward blend kernel: a hint, never a result); the environment is looked at once per
 *    process for the test hook LVDGS_FORCE_RADIX_GROUPING;
 *  - all float tensors are float32, contiguous, row-major; matrices are 4x4 in the
 *    row-vector layout the reference's Camera produces (world_view_transform =
 *    getWorld2View2(R,T).transpose(0,1), utils/camera_utils.py:106-116).
 */
#ifndef LVDGS_H
#define LVDGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LVDGS_OK = 0,
    LVDGS_E_INVALID = 1, /* bad argument (null pointer, size mismatch, buffer too small) */
    LVDGS_E_HIP = 2,     /* a HIP runtime call or kernel launch failed                   */
    LVDGS_E_RANGE = 3,   /* problem exceeds a built-in limit (e.g. > 2^31 pairs)         */
    LVDGS_E_CAPACITY = 4 /* lvdgs_forward: pair_capacity was too small; see lvdgs_forward */
} lvdgs_status;

/* One argument block serves the three rasterizer calls; each call reads the fields it needs
 * (marked P = prepare, R = render, B = backward).  Zero-initialise, then fill. */
typedef struct lvdgs_args {
    /* ---- GaussianRasterizationSettings (P R B) ---- */
    int32_t image_height, image_width;
    float tanfovx, tanfovy;
    float scale_modifier;
    int32_t sh_degree;   /* 0..3 */
    int32_t prefiltered; /* accepted for signature parity; unused */
    int32_t debug;       /* 1: synchronise and check after every launch */
    const float *bg;             /* device, 3  */
    const float *viewmatrix;     /* device, 16 */
    const float *projmatrix;     /* device, 16: viewmatrix * projmatrix_raw */
    const float *projmatrix_raw; /* device, 16 (B: pose gradient) */
    const float *campos;         /* device, 3  */

    /* ---- Gaussians (P B) ---- */
    int32_t num_gaussians; /* N */
    int32_t sh_coeffs;     /* M: coefficients per Gaussian in `shs` (>= (sh_degree+1)^2) */
    const float *means3D;        /* N*3 */
    const float *opacities;      /* N   (activated, in [0,1]) */
    const float *scales;         /* N*3 (activated) or NULL with cov3D_precomp */
    const float *rotations;      /* N*4 (r,x,y,z; used as given) or NULL */
    const float *cov3D_precomp;  /* N*6 (xx,xy,xz,yy,yz,zz) or NULL */
    const float *shs;            /* N*M*3 or NULL with colors_precomp */
    const float *colors_precomp; /* N*3 or NULL */

    /* ---- state + scratch buffers (caller-allocated, 256-byte aligned) ---- */
    void *geom_state;     size_t geom_bytes;    /* P w, R rw, B r : lvdgs_geom_bytes(N)           */
    void *binning_state;  size_t binning_bytes; /*      R w,  B r : lvdgs_binning_bytes(D)        */
    void *image_state;    size_t image_bytes;   /*      R w,  B r : lvdgs_image_bytes(W,H)        */
    void *scratch;        size_t scratch_bytes; /* P: lvdgs_prepare_scratch_bytes(N)
                                                   R: lvdgs_render_scratch_bytes(N,D,W,H)
                                                   B: lvdgs_backward_scratch_bytes(N,D)           */
    int64_t num_rendered; /* D, as returned by prepare (R B) */

    /* ---- forward outputs ---- */
    int32_t *radii;     /* N       (P w, B r) */
    float *out_color;   /* 3*H*W   (R w, B r) */
    float *out_depth;   /* H*W     (R w)      */
    float *out_opacity; /* H*W     (R w)      */
    int32_t *n_touched; /* N       (R w)      */

    /* ---- backward inputs ---- */
    const float *dL_dout_color;   /* 3*H*W         */
    const float *dL_dout_depth;   /* H*W or NULL   */
    const float *dL_dout_opacity; /* H*W or NULL   */

    /* ---- backward outputs (every element is written) ---- */
    float *dL_dmeans3D;   /* N*3 */
    float *dL_dmeans2D;   /* N*3: d/d NDC x,y of the projected mean (viewspace_points.grad), z = 0 */
    float *dL_dopacities; /* N   */
    float *dL_dscales;    /* N*3 or NULL with cov3D_precomp */
    float *dL_drotations; /* N*4 or NULL with cov3D_precomp */
    float *dL_dcov3D;     /* N*6 or NULL (only written when cov3D_precomp != NULL) */
    float *dL_dshs;       /* N*M*3 or NULL with colors_precomp */
    float *dL_dcolors;    /* N*3 or NULL with shs */
    float *dL_dtau;       /* 6: [d/d rho (3), d/d theta (3)] of T_w2c <- Exp(tau) T_w2c
                             (reference utils/pose_utils.py:70-87).  NULL: the per-workgroup partial sums stay in
                             `scratch` and lvdgs_tracking_tail reduces them (one launch less) */

    /* ---- single-call forward only ---- */
    int64_t pair_capacity; /* pairs binning_state / scratch were sized for (lvdgs_forward) */

    /* ---- optional: activations fused into the projection kernel (P B) ----
     * bit 0: `scales` holds log-scales (exp applied), bit 1: `rotations` is un-normalised (divided by its
     * norm), bit 2: `opacities` holds logits (sigmoid applied).  The matching gradient outputs are then
     * w.r.t. the raw values.  0 = inputs are already activated, as upstream's render() passes them. */
    int32_t activations;

    /* ---- optional switches (P R B; the same value in all calls of one frame) ---- */
    int32_t flags;          /* LVDGS_FLAG_* below; 0 = default behaviour */
    /* ---- optional: render a horizontal band of the image only (P R B) ----
     * Tile rows [tile_row_begin, tile_row_end) of the 16-pixel tile grid; both 0 = the whole image.  Pixels outside the
     * band are not written (out_color / out_depth / out_opacity / image_state keep what they held), only the band's
     * (Gaussian, tile) pairs are listed, n_touched counts the band's pixels, and the backward returns the band's share
     * of every gradient: bands that partition the image add up to the whole frame's gradients (the loss's partial sums
     * of lvdgs_backward_fused_loss are left for the band's tiles, zeros elsewhere).  This is what lets one view of the
     * mapping window (reference utils/slam_backend.py:180-306: losses summed over views before one backward) be split
     * between GPUs. */
    int32_t tile_row_begin, tile_row_end;
} lvdgs_args;

/* lvdgs_args.flags */
enum {
    LVDGS_FLAG_LIST_ALL_TILES = 1, /* list every tile of a Gaussian's 3-sigma rectangle -- the reference's pair list, bit
                                      for bit (num_rendered, point_list, ranges, n_contrib) -- instead of only the tiles
                                      on which it can reach alpha >= 1/255 (outputs are the same either way) */
    LVDGS_FLAG_NO_BLEND = 8,       /* R, B: the call leaves its blend pass out -- lvdgs_forward / lvdgs_forward_render stop behind the
                                      per-tile depth sort (out_color / out_depth / out_opacity / n_touched and the image
                                      state's final_T / n_contrib are not written), lvdgs_backward_fused_loss starts behind the
                                      backward blend pass -- because the caller runs that pass for several views of one size in
                                      ONE launch: lvdgs_blend_forward_batch / lvdgs_blend_backward_fused_loss_batch (the views
                                      of a mapping window; the results are the single calls', bit for bit) */
    LVDGS_FLAG_SUPER_TILES = 16,   /* R: a HINT for frames whose Gaussians are listed on many tiles each (opaque surfaces of large flat Gaussians: 70-80
                                      tiles per Gaussian): the (Gaussian, tile) pairs are grouped and depth-sorted per 64 x 64-pixel super-tile -- a tenth of
                                      the keys to scatter and sort there -- and every tile's list is read off its super-tile's sorted list.  point_list,
                                      ranges and every output are the same bits with and without it; worth setting when num_rendered exceeds ~16 pairs per
                                      Gaussian (lvdgs.rasterizer decides from the previous frame's count).  Ignored with LVDGS_FLAG_LIST_ALL_TILES, with a
                                      band of tile rows and on frames of fewer than 64 tiles; the views of lvdgs_forward_batch must agree on it.
                                      B: the per-Gaussian pass of lvdgs_backward*, lvdgs_forward_backward_fused_loss and lvdgs_gaussian_backward_batch runs with helper waves (workgroups of eight waves: the second half of a
                                      large-footprint wave's pair records is summed beside the first) -- the same sums in the same order, sooner. */
    LVDGS_FLAG_POSE_ONLY = 4,      /* B: only the camera-pose gradient (dL_dtau, or its partial sums for lvdgs_tracking_tail) and --
                                      lvdgs_backward_fused_loss -- the loss value and exposure gradients are produced.  The
                                      reference's tracking optimiser holds the pose and the exposure alone
                                      (utils/slam_frontend.py:1468-1490, stepped at :1520); the Gaussian gradients autograd
                                      computes beside them are dropped.  With this flag they are not computed: the dL_d*
                                      outputs other than dL_dtau are ignored (may be NULL) and NOT written, the backward blend
                                      pass leaves out the sums that feed only colours and opacities, the per-Gaussian pass
                                      neither reads opacities / SH coefficients nor writes N x 14 gradients.  dL_dtau is bit for
                                      bit the full backward's.  Needs sh_degree 0 or colors_precomp (a view-dependent colour
                                      feeds the pose gradient through the colour gradient): LVDGS_E_INVALID otherwise */
    LVDGS_FLAG_ACCUMULATE_PARAM_GRADS = 2 /* B: the gradients w.r.t. the Gaussian parameters (dL_dmeans3D, dL_dopacities,
                                      dL_dscales, dL_drotations, dL_dcov3D, dL_dshs / dL_dcolors) are ADDED to what their
                                      buffers hold -- a later view of a mapping iteration, whose losses are summed before one
                                      backward (reference utils/slam_backend.py:180-306) -- instead of written; dL_dmeans2D
                                      and dL_dtau (per view) are written as always */
};

/* ---- sizes ---- */
size_t lvdgs_geom_bytes(int32_t num_gaussians);
size_t lvdgs_prepare_scratch_bytes(int32_t num_gaussians);
size_t lvdgs_binning_bytes(int64_t num_rendered);
size_t lvdgs_image_bytes(int32_t width, int32_t height);
size_t lvdgs_render_scratch_bytes(int32_t num_gaussians, int64_t num_rendered, int32_t width, int32_t height);
size_t lvdgs_backward_scratch_bytes(int32_t num_gaussians, int64_t num_rendered);

/* ---- rasterizer ---- */
/* Projects the Gaussians and counts (Gaussian, tile) pairs.  Writes radii and
 * geom_state; returns the pair count D in *num_rendered (synchronises the stream once).
 * D counts the (Gaussian, tile) pairs that are listed: the tiles of the reference's 3-sigma rectangle
 * (forward.cu's getRect) on which the Gaussian can reach alpha >= 1/255.  Pairs that cannot are dropped
 * (they contribute to no pixel; outputs are unchanged), so D <= the reference's num_rendered.  With
 * LVDGS_FLAG_LIST_ALL_TILES in a->flags every tile of the rectangle is listed and D equals it. */
int lvdgs_forward_prepare(const lvdgs_args *a, int64_t *num_rendered, void *stream);
/* Groups the pairs by tile, orders each tile's list by (depth, id) and composites front to back.
 * Writes out_color / out_depth / out_opacity / n_touched, binning_state and image_state. */
int lvdgs_forward_render(const lvdgs_args *a, void *stream);
/* Single-call forward without a pipeline bubble.  The caller sizes binning_state and scratch for
 * `pair_capacity` pairs (scratch >= max(lvdgs_prepare_scratch_bytes(N), lvdgs_render_scratch_bytes(N,cap,W,H)));
 * all kernels are enqueued with the pair count left on the device, and only then does the host wait
 * for the count (the GPU is busy with the tile sort and the blend meanwhile).  Returns LVDGS_OK and
 * the count in *num_rendered, or LVDGS_E_CAPACITY when the count exceeds pair_capacity: outputs are
 * then invalid, geom_state is valid, and the caller re-runs lvdgs_forward_render with
 * num_rendered = *num_rendered and buffers of that size.  Results are identical to the two-call form. */
int lvdgs_forward(const lvdgs_args *a, int64_t *num_rendered, void *stream);
/* Gradient of the three images w.r.t. every Gaussian parameter and the camera pose. */
int lvdgs_backward(const lvdgs_args *a, void *stream);
/* present[i] = Gaussian i is in front of the near plane of the view (GaussianRasterizer.markVisible). */
int lvdgs_mark_visible(int32_t num_gaussians, const float *means3D, const float *viewmatrix,
                       const float *projmatrix, uint8_t *present, void *stream);

/* ---- views into the state buffers (parity tests read intermediates through these) ---- */
typedef struct lvdgs_state_layout {
    /* byte offsets into geom_state */
    size_t geom_rec;           /* N x geom_rec_floats float: x, y, conic a, b, c, opacity, r, g, b, view depth,
                                  (unused), i32 radius, then (16-float records) the tile rectangle x0 | x1 << 16,
                                  y0 | y1 << 16 and the 64-bit mask of its tiles that are listed */
    size_t geom_tiles_touched; /* N x u32 */
    size_t geom_slot_base;     /* N x u32: exclusive scan of tiles_touched in id order (first pair of a Gaussian) */
    /* byte offsets into binning_state */
    /* (binning_state is laid out by its SIZE: these offsets hold for a buffer of exactly lvdgs_binning_bytes(num_rendered) bytes;
     *  point_list always starts the buffer) */
    size_t bin_point_list;     /* D x u32: Gaussian ids, (tile, depth, id) ordered */
    size_t bin_tile_keys;      /* D x u32: tile id of each entry of point_list -- written only on the radix
                                  path (images above 16384 tiles); otherwise derive it from img_ranges */
    /* byte offsets into image_state */
    size_t img_ranges;         /* T x 2 u32: [begin, end) of each tile in point_list */
    size_t img_final_T;        /* P x float */
    size_t img_n_contrib;      /* P x u32 */
    size_t geom_rec_floats;    /* floats per record of geom_rec (16) */
} lvdgs_state_layout;
int lvdgs_state_layout_query(int32_t num_gaussians, int64_t num_rendered, int32_t width, int32_t height,
                             lvdgs_state_layout *out);

/* ---- other native dependencies of the SLAM loop ---- */
/* mean squared distance of every point to its 3 nearest neighbours (simple_knn.distCUDA2). */
size_t lvdgs_knn_scratch_bytes(int32_t num_points);
int lvdgs_dist2_knn3(int32_t num_points, const float *points /* P*3 */, float *mean_dist2 /* P */,
                     void *scratch, size_t scratch_bytes, void *stream);
/* in-place 2-D rotary embedding of tokens (B, N, H, D) with integer positions (B, N, 2)
 * (croco curope.rope_2d): first half of D rotates by y, second half by x; fwd = +1 or -1. */
int lvdgs_rope2d(float *tokens, const int64_t *positions, int32_t B, int32_t N, int32_t H, int32_t D,
                 float base, float fwd, void *stream);
/* The same rotation for f32 / f16 / bf16 tokens addressed through element strides (D contiguous): element (b, n, h, d)
 * lives at tokens + b*stride_b + n*stride_n + h*stride_h + d.  croco calls curope on q / k in their (B, H, N, D)
 * attention layout, i.e. stride_h = N*D, stride_n = D, under autocast in half precision; arithmetic is f32 and each
 * element is rounded once when stored.  `dtype` is one of the LVDGS_F32 / LVDGS_F16 / LVDGS_BF16 codes. */
enum { LVDGS_F32 = 0, LVDGS_F16 = 1, LVDGS_BF16 = 2 };
int lvdgs_rope2d_strided(void *tokens, int32_t dtype, const int64_t *positions, int32_t B, int32_t N, int32_t H, int32_t D,
                         int64_t stride_b, int64_t stride_n, int64_t stride_h, float base, float fwd, void *stream);

/* ---- fused photometric losses (reference utils/slam_utils.py:42-121) ---- */
/*   loss = weight_rgb   * mean_{c,p} [ omega_p * |(e^a I_cp + b) m_p - G_cp m_p| ]
 *        + weight_depth * mean_p     [ |D_p k_p - Z_p k_p| ]
 *   m_p = (sum_c G_cp > rgb_boundary_threshold) [* grad_mask_p],  omega_p = opacity_p if weight_by_opacity else 1,
 *   k_p = (Z_p > 0.01) [* (opacity_p > 0.95) if depth_needs_opaque].
 * get_loss_tracking_rgb  (slam_utils.py:53-62):  weight_rgb 1, weight_by_opacity 1, grad_mask, no depth term;
 * get_loss_tracking_rgbd (:65-79):  weight_rgb alpha, weight_depth 1-alpha, depth_needs_opaque 1;
 * get_loss_mapping_rgb   (:95-104): weight_rgb 1;   get_loss_mapping_rgbd (:107-121): alpha, 1-alpha. */
typedef struct lvdgs_loss_args {
    int32_t width, height;
    const float *image;        /* 3*H*W rendered colour                       */
    const float *depth;        /* H*W rendered depth or NULL                  */
    const float *opacity;      /* H*W rendered opacity or NULL                */
    const float *gt_image;     /* 3*H*W                                       */
    const float *gt_depth;     /* H*W or NULL (no depth term)                 */
    const uint8_t *grad_mask;  /* H*W bytes (0 / non-0) or NULL               */
    const float *exposure_a;   /* 1 or NULL (no exposure correction)          */
    const float *exposure_b;   /* 1 or NULL                                   */
    float rgb_boundary_threshold, weight_rgb, weight_depth;
    int32_t weight_by_opacity, depth_needs_opaque;
    void *scratch; size_t scratch_bytes;   /* lvdgs_loss_scratch_bytes(W,H)   */
    float *loss;               /* forward out: 1                              */
    const float *grad_loss;    /* backward in: 1 (d objective / d loss)       */
    float *d_image;            /* backward out: 3*H*W                         */
    float *d_depth;            /* H*W or NULL                                 */
    float *d_opacity;          /* H*W or NULL                                 */
    float *d_exposure_a;       /* 1 or NULL                                   */
    float *d_exposure_b;       /* 1 or NULL                                   */
} lvdgs_loss_args;
size_t lvdgs_loss_scratch_bytes(int32_t width, int32_t height);
int lvdgs_photometric_loss_forward(const lvdgs_loss_args *a, void *stream);
int lvdgs_photometric_loss_backward(const lvdgs_loss_args *a, void *stream);
/* Value and every gradient in ONE pass over the images (for callers whose objective is this loss, so that
 * d objective / d loss is known up front): grad_loss may be NULL (= 1) or a device scalar.  Same results as the two
 * calls above. */
int lvdgs_photometric_loss_value_and_grad(const lvdgs_loss_args *a, void *stream);
/* The same pass WITHOUT the final reduction: d_image / d_depth / d_opacity are written, the per-workgroup partial sums
 * of the loss value and of the exposure gradients stay in `scratch` for lvdgs_tracking_tail (`loss`, `d_exposure_a/b`
 * are not written yet).  One launch. */
int lvdgs_photometric_loss_partials(const lvdgs_loss_args *a, void *stream);

/* ---- per-frame pose optimiser step (reference utils/slam_frontend.py:1518-1521, utils/pose_utils.py:70-87) ----
 * One launch = `pose_optimizer.step()` (torch.optim.Adam: betas, eps, one learning rate per group) on the frame's
 * cam_rot_delta, cam_trans_delta, exposure_a, exposure_b, then `update_pose`: T_w2c <- SE3_exp([trans, rot]) @ [R T],
 * deltas zeroed, and the matrices the next render reads (Camera.world_view_transform / full_proj_transform /
 * camera_center, utils/camera_utils.py:106-120).  `state` = 24 floats owned by the caller and zeroed before the first
 * step of a frame: [0..15] Adam's (exp_avg, exp_avg_sq) of the 8 scalars (rot xyz, trans xyz, exposure a, b), [16] calls
 * applied, [17] a STICKY converged flag (||tau|| < converged_threshold at some step), [18] the number of steps applied,
 * [19..22] the Adam step count of the rot / trans / exposure_a / exposure_b group (a group whose gradient pointer is NULL in
 * a call is skipped, moments and count, as torch.optim.Adam skips parameters without .grad).  Once the flag is set further
 * calls change nothing, so the host may enqueue iterations ahead and read the flag late. */
typedef struct lvdgs_pose_step_args {
    float *R;                      /* 9, row-major world-to-camera rotation (in / out)                */
    float *T;                      /* 3 (in / out)                                                    */
    float *cam_rot_delta;          /* 3 parameter values (in / out: stepped, consumed, zeroed)        */
    float *cam_trans_delta;        /* 3                                                               */
    float *exposure_a;             /* 1 or NULL                                                       */
    float *exposure_b;             /* 1 or NULL                                                       */
    const float *grad_tau;         /* 6: dL/d(trans, rot), as lvdgs_backward writes dL_dtau; NULL = 0 */
    const float *grad_exposure_a;  /* 1 or NULL                                                       */
    const float *grad_exposure_b;  /* 1 or NULL                                                       */
    float *state;                  /* 24 floats, see above                                            */
    double lr_rot, lr_trans, lr_exposure, beta1, beta2, eps;  /* doubles, like the Python floats torch.optim.Adam computes with */
    float converged_threshold;
    const float *projmatrix_raw;   /* 16: the camera's projection_matrix (row-vector layout) or NULL  */
    float *viewmatrix;             /* out 16 or NULL                                                  */
    float *projmatrix;             /* out 16 or NULL: viewmatrix @ projmatrix_raw                     */
    float *campos;                 /* out 3 or NULL                                                   */
    /* The mapping loop's keyframes (utils/slam_backend.py:381-389) hold their gradients in separate tensors and may
     * lack some: when grad_tau is NULL these are read instead, and a parameter whose gradient pointer is NULL is left
     * untouched, moments included -- torch.optim.Adam skips parameters without a gradient.  R == NULL (with T NULL)
     * steps the exposure only (keyframes outside the pose window).  converged_threshold < 0 never raises the flag. */
    const float *grad_rot;         /* 3 or NULL */
    const float *grad_trans;       /* 3 or NULL */
    /* Optional: 2 floats of pinned host memory as the DEVICE addresses them (lvdgs_host_device_pointer), zeroed by the caller
     * before a frame's first step.  Every step stores state[18] (steps applied) to [1] and, when it raises the converged flag,
     * 1 to [0]: the host watches the optimisation without enqueuing a device-to-host copy per iteration (on this stack such a
     * copy is a blit kernel of its own). */
    float *host_flags;
} lvdgs_pose_step_args;
int lvdgs_pose_step(const lvdgs_pose_step_args *a, void *stream);
/* The address under which the device sees `host` (page-locked, mapped host memory, e.g. a pinned PyTorch tensor); fails with
 * LVDGS_E_HIP when the memory is not mapped into the device's address space. */
int lvdgs_host_device_pointer(void *host, void **device);
/* `count` independent steps (distinct cameras: the keyframes of a mapping window) in one launch instead of `count`. */
int lvdgs_pose_step_batch(const lvdgs_pose_step_args *steps, int32_t count, void *stream);

/* lvdgs_backward with the photometric loss evaluated INSIDE the backward blend pass: every pixel's dL/d(colour, depth,
 * opacity) is computed from `loss` (the formulas and parameters of lvdgs_photometric_loss_value_and_grad; d objective /
 * d loss = *grad_loss, or 1 when NULL) as the pass reads its pixels -- no gradient images are written or read, and there
 * is no separate pass over the frame (at 1080p: one launch and ~80 MB of traffic less per iteration).  a->dL_dout_* and
 * loss->d_image / d_depth / d_opacity are ignored; the opacity image's gradient feeds the blend iff
 * propagate_opacity_grad.  The loss's four partial sums are left per TILE in loss->scratch
 * (lvdgs_loss_scratch_bytes covers that layout too) for lvdgs_tracking_tail(..., partials_per_tile = 1), which writes
 * loss->loss and loss->d_exposure_a / _b.  An empty map or a view that lists no pair (num_gaussians or num_rendered == 0) is
 * fine: the loss of the background image is evaluated, every Gaussian / pose gradient is zero. */
int lvdgs_backward_fused_loss(const lvdgs_args *a, const lvdgs_loss_args *loss, int32_t propagate_opacity_grad, void *stream);

/* lvdgs_forward followed by lvdgs_backward_fused_loss as ONE call (a view whose loss is the photometric one: the tracking iteration,
 * utils/slam_frontend.py:1492-1517; a mapping view without a static mask).  Same arguments as the two calls (a->pair_capacity sizes
 * binning_state; scratch >= max of the three scratch sizes AT the capacity: lvdgs_backward_scratch_bytes(N, pair_capacity)), same results,
 * bit for bit.  What it buys: on frames of up to 4096 tiles the forward and the backward blend pass of a tile run in one launch -- the
 * loss's gradient at a pixel depends on that pixel alone, so a tile's backward needs nothing but the tile's own forward -- and a frame
 * of KITTI's size, on which each blend kernel lasts as long as its few heaviest tiles while the chip runs dry, pays that tail once
 * (KITTI geometry, pose-only: 0.209 -> 0.19 ms per tracking iteration).  Larger frames: the two calls in turn.  Returns LVDGS_OK and
 * the pair count, or LVDGS_E_CAPACITY (geom_state valid, everything else invalid): the caller grows binning_state / scratch and calls
 * lvdgs_forward_render, then lvdgs_backward_fused_loss.  The per-tile partial sums of the loss and the pose-gradient partials are left
 * for lvdgs_tracking_tail / lvdgs_map_view_tail as by lvdgs_backward_fused_loss (with a->dL_dtau == NULL). */
int lvdgs_forward_backward_fused_loss(const lvdgs_args *a, const lvdgs_loss_args *loss, int32_t propagate_opacity_grad, int64_t *num_rendered,
                                      void *stream);

/* The blend passes of `count` views in one launch each (no counterpart upstream, which renders a mapping window's keyframes one
 * after the other: utils/slam_backend.py:175-266).  A frame of KITTI's size (1848 tiles) leaves a 256-CU chip's wave slots half
 * empty and ends in a tail of its heaviest tiles; the ten views of a mapping window together fill it (per view: forward blend
 * 54 -> 34 us, backward 105 -> 77).  views[k]: the argument block of a view that has been through lvdgs_forward /
 * lvdgs_forward_render (resp. is about to go through lvdgs_backward_fused_loss) with LVDGS_FLAG_NO_BLEND -- the same
 * block, every buffer its own; all views the same image size and band of tile rows, losses[k] as for
 * lvdgs_backward_fused_loss.  Each view's outputs are what its single call without the flag writes, bit for bit. */
int lvdgs_blend_forward_batch(const lvdgs_args *const *views, int32_t count, void *stream);
/* lvdgs_forward for `count` views of ONE map (the same Gaussian tensors, the same num_gaussians / activations) and ONE image size,
 * every stage of all views in one launch: projection + counting, the two scans, the scatter, the per-tile depth sort and (unless
 * LVDGS_FLAG_NO_BLEND is set in the views' flags) the forward blend -- 6 launches for a mapping window's ten views instead of 50
 * (upstream renders them one after the other, utils/slam_backend.py:180-184; at KITTI's frame size the five stages before the blend are
 * latency-bound launches of 6-16 us each, half a millisecond per window).  Every view brings its own argument block with buffers of its
 * own, sized for its own pair_capacity as for lvdgs_forward; images of at most 16384 tiles.  The host waits once, with everything
 * enqueued, for all the pair counts: num_rendered[k] = view k's.  Returns LVDGS_OK, or LVDGS_E_CAPACITY when some view's count exceeds
 * its capacity -- THAT view's outputs are invalid (its geom_state is valid) and the caller re-runs lvdgs_forward_render for it with
 * buffers of num_rendered[k] pairs; the other views are complete.  Each view's state and outputs are lvdgs_forward's, bit for bit. */
int lvdgs_forward_batch(const lvdgs_args *const *views, int32_t count, int64_t *num_rendered, void *stream);
int lvdgs_blend_backward_fused_loss_batch(const lvdgs_args *const *views, const lvdgs_loss_args *const *losses, int32_t count,
                                          int32_t propagate_opacity_grad, void *stream);

/* The end of a tracking iteration in ONE launch (instead of three at ~6 us each on the iteration's critical path):
 *   - finishes the loss: sums the partial sums lvdgs_photometric_loss_partials(loss) left per 1024 pixels
 *     (partials_per_tile = 0) or lvdgs_backward_fused_loss left per tile (1), writes loss->loss, loss->d_exposure_a / _b;
 *   - reduces the pose-gradient partials lvdgs_backward(bwd) / lvdgs_backward_fused_loss(bwd, ...) left in bwd->scratch
 *     when called with bwd->dL_dtau == NULL (same `bwd` block, untouched in between: num_gaussians, num_rendered,
 *     scratch) and writes dL_dtau (6 floats);
 *   - applies lvdgs_pose_step(pose) with those gradients (pose->grad_* are ignored: the pose deltas take dL_dtau, the
 *     exposure parameters that `pose` names take loss->d_exposure_a / _b).  pose == NULL: the two reductions only (a
 *     view of the mapping iteration, whose keyframe is stepped after all views).
 * Same additions in the same order as the separate launches: bit-identical results. */
int lvdgs_tracking_tail(const lvdgs_loss_args *loss, const lvdgs_args *bwd, const lvdgs_pose_step_args *pose, float *dL_dtau,
                        int32_t partials_per_tile, void *stream);

/* The end of one VIEW of the mapping iteration in one launch: lvdgs_tracking_tail(loss, bwd, NULL, dL_dtau, 1, ...) -- the
 * view's loss value, exposure gradients and pose gradient from the partial sums lvdgs_backward_fused_loss left -- and
 * lvdgs_view_stats on the view's outputs (bwd->radii, bwd->n_touched, bwd->dL_dmeans2D; see lvdgs_view_stats below for the
 * meaning of the fields, any of vis_count / touched_row / split_xy may be NULL).  loss == NULL: a view scored by
 * lvdgs_masked_loss_batch, whose loss value is finished already -- the pose gradient and the statistics only. */
typedef struct lvdgs_view_stats_args {
    int32_t *radii_max;   /* N   */
    float *norm_sum;      /* N   */
    float *vis_count;     /* N or NULL */
    uint8_t *touched_row; /* N or NULL */
    float *split_xy;      /* N*2 or NULL */
} lvdgs_view_stats_args;
int lvdgs_map_view_tail(const lvdgs_loss_args *loss, const lvdgs_args *bwd, float *dL_dtau, const lvdgs_view_stats_args *stats, void *stream);
/* lvdgs_map_view_tail for `count` views in ONE launch (a mapping window whose views were rendered and differentiated together:
 * lvdgs_forward_batch ... ): losses[k] may be NULL as above (losses itself too: no view has a loss block), stats[k] is required.  The
 * statistics are taken per Gaussian over the views IN ORDER: radii_max / norm_sum / vis_count receive what `count` single calls in
 * that order give them, bit for bit. */
int lvdgs_map_view_tail_batch(const lvdgs_loss_args *const *losses, const lvdgs_args *const *bwds, float *const *dL_dtau,
                              const lvdgs_view_stats_args *const *stats, int32_t count, void *stream);

/* ---- Adam step of the Gaussian map (reference utils/slam_backend.py:144, :378, :458: gaussians.optimizer.step()) ----
 * All parameter tensors in one launch, one pass over (grad, exp_avg, exp_avg_sq, param); torch.optim.Adam's arithmetic
 * (no weight decay, no amsgrad), `step` = that tensor's step count INCLUDING this step (bias corrections). */
#define LVDGS_ADAM_MAX_TENSORS 8
typedef struct lvdgs_adam_tensor {
    float *param; const float *grad; float *exp_avg; float *exp_avg_sq;
    int64_t numel; int64_t step; double lr;
} lvdgs_adam_tensor;
int lvdgs_adam_step(const lvdgs_adam_tensor *tensors, int32_t count, double beta1, double beta2, double eps, void *stream);

/* Isotropic regulariser of the mapping loss (reference utils/slam_backend.py:303-305):
 *   loss = weight * mean_{i,k} | s_ik - mean_k s_ik |,  s = exp(raw_scales)   (the model's scaling activation)
 * writes the value to *loss and ADDS its gradient w.r.t. the raw (log) scales to grad_raw_scales (NULL: value only). */
size_t lvdgs_isotropic_scratch_bytes(int32_t num_gaussians);
int lvdgs_isotropic_reg(int32_t num_gaussians, const float *raw_scales /* N*3 */, float *grad_raw_scales /* N*3 or NULL */,
                        float weight, void *scratch, size_t scratch_bytes, float *loss, void *stream);
/* What the back end derives from one view's render package (utils/slam_backend.py:311-315, :350-357), accumulated over
 * the views rendered so far: radii_max = max(radii_max, radii); for visible Gaussians (radii > 0) norm_sum += |viewspace
 * gradient xy| and vis_count += 1 (NULL: not counted); touched_row[i] = n_touched[i] > 0 (NULL for views outside the window).
 * A view rendered in bands by several GPUs (lvdgs_args.tile_row_*) holds only a share of the gradient on each: pass
 * split_xy (N x 2) and the band's xy is written THERE instead of its norm being added to norm_sum -- the caller sums the
 * bands (all-reduce) and lvdgs_map_stats_apply takes the norm of the sum; vis_count then goes with ONE of the bands. */
int lvdgs_view_stats(int32_t num_gaussians, const int32_t *radii, const int32_t *n_touched, const float *viewspace_grad /* N*3 or NULL */,
                     int32_t *radii_max, float *norm_sum, float *vis_count, uint8_t *touched_row, float *split_xy, void *stream);
/* ... and their way into the model, one launch (:350-357 on the reduced values):
 *   max_radii2D = max(max_radii2D, radii_max); xyz_gradient_accum += norm_sum + sum_k |split_xy[k]|; denom += vis_count
 * split_xy: n_split planes of N x 2 floats (the summed band gradients of the views that were split), or NULL with 0. */
int lvdgs_map_stats_apply(int32_t num_gaussians, const int32_t *radii_max, const float *norm_sum, const float *vis_count,
                          const float *split_xy, int32_t n_split, float *max_radii2D, float *xyz_gradient_accum, float *denom, void *stream);

/* ---- depth term of the static-mask mapping loss (reference utils/slam_backend.py:216-261) ----
 *   M    = static_mask & (mono_depth > 0) & (rendered depth > 0)          (static_mask NULL = every pixel)
 *   loss = depth_lambda-free mean over M of |D_p - Z_p|;  0 when M is empty (the reference then adds nothing)
 * The mean is over |M|, not over the image (that is what distinguishes it from lvdgs_loss_args' depth term).
 * forward writes out[0] = loss, out[1] = |M|; backward reads out[1] back and writes
 *   d_depth_p = grad_loss * sign(D_p - Z_p) / |M| on M, 0 elsewhere. */
typedef struct lvdgs_masked_depth_args {
    int32_t width, height;
    const float *depth;          /* H*W rendered depth                           */
    const float *gt_depth;       /* H*W mono depth                               */
    const uint8_t *static_mask;  /* H*W bytes (non-0 = static) or NULL           */
    void *scratch; size_t scratch_bytes;   /* lvdgs_masked_depth_scratch_bytes(W,H); forward only */
    float *out;                  /* 2 floats: loss, |M|  (forward writes, backward reads [1]) */
    const float *grad_loss;      /* backward in: 1                               */
    float *d_depth;              /* backward out: H*W                            */
} lvdgs_masked_depth_args;
size_t lvdgs_masked_depth_scratch_bytes(int32_t width, int32_t height);
int lvdgs_masked_depth_l1_forward(const lvdgs_masked_depth_args *a, void *stream);
int lvdgs_masked_depth_l1_backward(const lvdgs_masked_depth_args *a, void *stream);

/* ---- fused L1 + SSIM image loss (reference utils/slam_backend.py:199-215, 438-454) ----
 * The mapping and colour-refinement losses `(1 - lambda) * l1_loss(a, b) + lambda * (1 - ssim(a, b))` call
 * gaussian_splatting.utils.loss_utils.l1_loss / ssim (package absent from the checkout; 11-tap Gaussian window,
 * sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over pixels and channels).  One launch computes both
 * means and, when d_img1 is given, the combined gradient
 *     d_img1 = weight_l1 * d mean|a - b| / d a + weight_ssim * d mean SSIM(a, b) / d a.
 * With keep_mask, pixels whose mask byte is 0 are first replaced by bg[plane % channels] in BOTH images
 * (slam_backend.py:205-209); they receive zero gradient. */
typedef struct lvdgs_ssim_args {
    int32_t width, height;
    int32_t planes;            /* batch * channels image planes of H*W floats  */
    int32_t channels;          /* planes per image (bg index = plane % channels) */
    const float *img1;         /* planes*H*W (the differentiated image)        */
    const float *img2;         /* planes*H*W                                   */
    const uint8_t *keep_mask;  /* H*W bytes or NULL                            */
    const float *bg;           /* channels floats or NULL (0)                  */
    float weight_l1, weight_ssim;
    void *scratch; size_t scratch_bytes;   /* lvdgs_ssim_scratch_bytes(W,H,planes) */
    float *out;                /* 2: mean |a - b|, mean SSIM                   */
    float *d_img1;             /* planes*H*W or NULL (values only)             */
} lvdgs_ssim_args;
size_t lvdgs_ssim_scratch_bytes(int32_t width, int32_t height, int32_t planes);
int lvdgs_ssim_l1(const lvdgs_ssim_args *a, void *stream);

/* ---- the static-mask mapping loss as ONE object (reference utils/slam_backend.py:196-261; colour refinement :420-454) ----
 * LVD-GS's front end attaches a static_mask to every tracked frame and keyframe (utils/slam_frontend.py:1218,1309-1329,1429-1433:
 * dynamic_filtering.enabled defaults to True), so this -- not get_loss_mapping -- is the loss of every window keyframe of
 * BackEnd.map:
 *     loss = (1 - lambda) * mean|a - b| + lambda * (1 - mean SSIM(a, b)) + depth_lambda * mean_{p in M} |D_p - Z_p|
 *     a, b = rendered / target colour with bg[c] written under the dynamic pixels (static_mask byte 0) of both,
 *     M    = static_mask & (Z > 0) & (D > 0)     (no depth term when gt_depth is NULL, or M is empty)
 * lvdgs_masked_loss_batch evaluates it for `count` views of one size in TWO launches whatever the count: the fused L1 + SSIM
 * kernel over every plane of every view (it also takes the depth term's sum and exact pixel count per 32x32 tile), and a
 * finish (one workgroup per view).  Per view it writes d_image = d loss / d a (the SSIM gradient is not local, so it is an
 * image) and out[0..4] = loss, mean|a - b|, mean SSIM, depth term, |M|.  The depth term's gradient is NOT written anywhere:
 * lvdgs_backward_masked_loss / lvdgs_blend_backward_window_batch evaluate depth_lambda * sign(D - Z) / |M| per pixel from
 * depth / gt_depth / static_mask and out[4] as the backward blend pass reads its pixels.  Same numbers as lvdgs_ssim_l1 +
 * lvdgs_masked_depth_l1_forward / _backward + lvdgs_backward on their gradient images (d_image, d_depth: the same bits). */
typedef struct lvdgs_masked_loss_args {
    int32_t width, height;
    const float *image;          /* 3*H*W rendered colour                               */
    const float *gt_image;       /* 3*H*W                                               */
    const uint8_t *static_mask;  /* H*W bytes (non-0 = static) or NULL (every pixel)    */
    const float *bg;             /* 3 or NULL (0): colour written under the mask        */
    const float *depth;          /* H*W rendered depth, or NULL                         */
    const float *gt_depth;       /* H*W mono depth, or NULL: no depth term              */
    float lambda_dssim, depth_lambda;
    void *scratch; size_t scratch_bytes;   /* lvdgs_masked_loss_scratch_bytes(W,H)      */
    float *d_image;              /* out: 3*H*W                                          */
    float *out;                  /* out: 8 floats, [0..4] as above                      */
} lvdgs_masked_loss_args;
size_t lvdgs_masked_loss_scratch_bytes(int32_t width, int32_t height);
int lvdgs_masked_loss_batch(const lvdgs_masked_loss_args *const *views, int32_t count, void *stream);
/* lvdgs_backward for a view scored by lvdgs_masked_loss_batch: the backward blend pass reads loss->d_image and evaluates the
 * depth term's gradient itself (a->dL_dout_* are ignored, the opacity image gets no gradient), then the per-Gaussian pass.
 * With LVDGS_FLAG_NO_BLEND it starts behind the blend pass (lvdgs_blend_backward_window_batch has run it). */
int lvdgs_backward_masked_loss(const lvdgs_args *a, const lvdgs_masked_loss_args *loss, void *stream);
/* The backward blend passes of a mapping window in one launch, every view with the loss it is scored by: view k takes
 * masked[k] when that is non-NULL (a keyframe with a static mask, as lvdgs_backward_masked_loss) and losses[k] otherwise (the
 * two random older views, keyframes without a mask: as lvdgs_blend_backward_fused_loss_batch).  `masked` may be NULL (no view
 * is masked); `losses` may be NULL when every view is masked.  Each view's records are its single call's, bit for bit. */
int lvdgs_blend_backward_window_batch(const lvdgs_args *const *views, const lvdgs_loss_args *const *losses,
                                      const lvdgs_masked_loss_args *const *masked, int32_t count, int32_t propagate_opacity_grad,
                                      void *stream);

/* The per-Gaussian passes (what lvdgs_backward_fused_loss / lvdgs_backward_masked_loss do with LVDGS_FLAG_NO_BLEND) of `count`
 * views of one map in ONE launch, behind lvdgs_blend_backward_window_batch.  The views share the map and the gradient buffers
 * (views[0]'s LVDGS_FLAG_ACCUMULATE_PARAM_GRADS says whether the sums are added to what those hold; the later views carry the
 * flag); a thread walks its Gaussian through the views in order with the parameter gradients in registers and writes them once,
 * where the view-after-view calls read and write them per view -- the same additions in the same order, the same bits.
 * Every view keeps its own dL_dmeans2D and pose-gradient partials (dL_dtau NULL: left in its scratch for lvdgs_map_view_tail*).
 * For SH colours of one coefficient (sh_degree 0, sh_coeffs 1) with scales + rotations; LVDGS_E_INVALID otherwise.
 * Replaces: the reference's per-view loss.backward() accumulation into .grad (utils/slam_backend.py:262-306). */
int lvdgs_gaussian_backward_batch(const lvdgs_args *const *views, int32_t count, void *stream);

/* ---- diagnostics ---- */
const char *lvdgs_last_error(void);
const char *lvdgs_version(void);
/* Optional per-kernel timing with HIP events recorded on the caller's stream around each
 * launch.  Off by default; enabling it does not change results. */
void lvdgs_profile_enable(int on);
void lvdgs_profile_reset(void);
/* Synchronises outstanding events, then fills up to `cap` entries; returns the entry count. */
typedef struct lvdgs_kernel_time {
    char name[48];
    int64_t launches;
    double total_ms;
} lvdgs_kernel_time;
int lvdgs_profile_read(lvdgs_kernel_time *out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* LVDGS_H */
