// Per-pixel arithmetic of the photometric tracking / mapping loss (reference utils/slam_utils.py:42-121), shared by the
// loss kernels (loss.hip) and by the backward blend pass when it evaluates the loss itself (blend.hip,
// lvdgs_backward_fused_loss): one statement of the formulas, the same float operations in the same order in both places.
#pragma once
#include "common.hpp"

namespace lvdgs {

struct LossParams {
    int P;                       // pixels
    const float *image;          // 3*P
    const float *depth;          // P or null
    const float *opacity;        // P or null
    const float *gt_image;       // 3*P
    const float *gt_depth;       // P or null
    const uint8_t *grad_mask;    // P or null
    const float *exposure_a, *exposure_b;  // 1 each or null (identity)
    float rgb_thr, w_rgb, w_d;
    int weight_by_opacity, depth_needs_opaque;
    // forward
    float *partial;              // 4 per block: rgb sum, depth sum (value), d_a sum, d_b sum (gradients)
    float *loss;                 // 1
    // backward
    const float *grad_out;       // 1
    float *d_image, *d_depth, *d_opacity;  // 3*P, P or null, P or null
    float *d_a, *d_b;            // 1 each or null
};

struct LossConsts {   // per-launch values every pixel uses
    float ea, eb, Wr, Wd;
    bool has_d;
    __device__ __forceinline__ LossConsts(const LossParams &p, bool backward) {
        ea = p.exposure_a ? expf(p.exposure_a[0]) : 1.f;   // (expf: what torch.exp(exposure_a) gives the autograd path)
        eb = p.exposure_b ? p.exposure_b[0] : 0.f;
        const float g = backward ? (p.grad_out ? p.grad_out[0] : 1.f) : 0.f;
        Wr = p.w_rgb / (3.f * (float)p.P) * g;
        Wd = p.w_d / (float)p.P * g;
        has_d = p.depth && p.gt_depth;
    }
};

struct PixelLoss {
    float v_rgb, v_d;            // this pixel's terms of the two value sums
    float dI[3], dO, dD;         // gradients w.r.t. the rendered colour, opacity, depth
    float s_a, s_b;              // this pixel's terms of the exposure gradients
};

// G: target colour, I: rendered colour, op: rendered opacity (1 when absent), Z / Dv: target / rendered depth, gm: edge
// mask (1 when absent), live: the pixel exists.
template <bool VALUE, bool BACKWARD>
__device__ __forceinline__ PixelLoss photometric_pixel(const LossParams &p, const LossConsts &c, const float G[3], const float I[3], float op,
                                                       float Z, float Dv, float gm, bool live) {
    PixelLoss o{};
    const float m = ((G[0] + G[1] + G[2] > p.rgb_thr) ? 1.f : 0.f) * gm * (live ? 1.f : 0.f);
    const float om = p.weight_by_opacity ? op : 1.f;
    const float r0 = (c.ea * I[0] + c.eb) * m - G[0] * m, r1 = (c.ea * I[1] + c.eb) * m - G[1] * m, r2 = (c.ea * I[2] + c.eb) * m - G[2] * m;
    float kd = 0.f, rd = 0.f;
    if (c.has_d) {
        kd = (Z > 0.01f && live) ? 1.f : 0.f;
        if (p.depth_needs_opaque) kd *= op > 0.95f ? 1.f : 0.f;
        rd = Dv * kd - Z * kd;
    }
    if (VALUE) {
        o.v_rgb = om * (fabsf(r0) + fabsf(r1) + fabsf(r2));
        o.v_d = fabsf(rd);
    }
    if (BACKWARD) {
        auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
        const float q0 = c.Wr * om * sgn(r0) * m, q1 = c.Wr * om * sgn(r1) * m, q2 = c.Wr * om * sgn(r2) * m;
        o.dI[0] = c.ea * q0; o.dI[1] = c.ea * q1; o.dI[2] = c.ea * q2;
        o.dO = p.weight_by_opacity ? c.Wr * (fabsf(r0) + fabsf(r1) + fabsf(r2)) : 0.f;
        o.dD = c.Wd * sgn(rd) * kd;
        o.s_a = c.ea * (q0 * I[0] + q1 * I[1] + q2 * I[2]);
        o.s_b = q0 + q1 + q2;
    }
    return o;
}

}  // namespace lvdgs
