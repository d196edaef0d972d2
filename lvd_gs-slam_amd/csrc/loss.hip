// Fused photometric losses of the tracking / mapping loops (reference utils/slam_utils.py:42-121).
//
// The reference builds each loss from ~10 full-frame elementwise PyTorch kernels and autograd adds
// as many again in backward; at 1920x1080 that is more GPU time than the forward blend.  Here one
// pass over the pixels computes the loss, and one pass computes every gradient (image, depth,
// opacity, exposure a / b).  Reductions are per-workgroup partial sums finished by one small
// workgroup in a fixed order (deterministic, no atomics).
//
//   loss = w_rgb * mean_{c,p} [ omega_p * | (e^a I_cp + b) m_p - G_cp m_p | ]
//        + w_d   * mean_p     [ | D_p k_p - Z_p k_p | ]
//   m_p = (sum_c G_cp > rgb_thr) [* grad_mask_p]          omega_p = opacity_p or 1
//   k_p = (Z_p > 0.01) [* (opacity_p > 0.95)]
#include "common.hpp"
#include "device_utils.hpp"

namespace lvdgs {
namespace {

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
    float *partial;              // nblk * 2 (rgb sum, depth sum)
    float *loss;                 // 1
    // backward
    const float *grad_out;       // 1
    float *d_image, *d_depth, *d_opacity;  // 3*P, P or null, P or null
    float *d_a, *d_b;            // 1 each or null
};

constexpr int LOSS_THREADS = 256;
constexpr int LOSS_PIX_PER_THREAD = 4;

__device__ __forceinline__ float block_sum(float v, float *s /* [4] */) {
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) s[wave] = v;
    __syncthreads();
    return ((s[0] + s[1]) + s[2]) + s[3];
}

// 4 consecutive pixels per lane: 16-byte loads / stores on every plane when P % 4 == 0 (VEC), scalar otherwise.
template <bool BACKWARD, bool VEC>
__global__ void __launch_bounds__(LOSS_THREADS) photometric_kernel(LossParams p) {
    __shared__ float s_red[4];
    const float ea = p.exposure_a ? __expf(p.exposure_a[0]) : 1.f;
    const float eb = p.exposure_b ? p.exposure_b[0] : 0.f;
    const float g = BACKWARD ? p.grad_out[0] : 0.f;
    const float Wr = p.w_rgb / (3.f * (float)p.P) * g, Wd = p.w_d / (float)p.P * g;
    float acc0 = 0.f, acc1 = 0.f;  // fwd: rgb sum, depth sum ; bwd: d_a sum, d_b sum
    const int base = (blockIdx.x * LOSS_THREADS + threadIdx.x) * LOSS_PIX_PER_THREAD;
    const size_t P = (size_t)p.P;
    const bool has_d = p.depth && p.gt_depth;
    if (base < p.P) {
        float G[3][4], I[3][4], op[4], Z[4], Dv[4], gm[4];
        const int n = VEC ? 4 : min(4, p.P - base);
        auto load4 = [&](const float *src, float out[4], float fill) {
            if (VEC) {
                const float4 v = *reinterpret_cast<const float4 *>(src + base);
                out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) out[k] = k < n ? src[base + k] : fill;
            }
        };
#pragma unroll
        for (int c = 0; c < 3; c++) { load4(p.gt_image + c * P, G[c], 0.f); load4(p.image + c * P, I[c], 0.f); }
        if (p.opacity) load4(p.opacity, op, 1.f);
        else { op[0] = op[1] = op[2] = op[3] = 1.f; }
        if (has_d) { load4(p.gt_depth, Z, 0.f); load4(p.depth, Dv, 0.f); }
        if (p.grad_mask) {
            if (VEC) {
                const uint32_t m4 = *reinterpret_cast<const uint32_t *>(p.grad_mask + base);
#pragma unroll
                for (int k = 0; k < 4; k++) gm[k] = ((m4 >> (8 * k)) & 0xffu) ? 1.f : 0.f;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) gm[k] = (k < n && p.grad_mask[base + k]) ? 1.f : 0.f;
            }
        } else { gm[0] = gm[1] = gm[2] = gm[3] = 1.f; }
        float dI[3][4], dO[4], dD[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool live = VEC || k < n;
            const float m = ((G[0][k] + G[1][k] + G[2][k] > p.rgb_thr) ? 1.f : 0.f) * gm[k] * (live ? 1.f : 0.f);
            const float om = p.weight_by_opacity ? op[k] : 1.f;
            const float r0 = (ea * I[0][k] + eb) * m - G[0][k] * m, r1 = (ea * I[1][k] + eb) * m - G[1][k] * m,
                        r2 = (ea * I[2][k] + eb) * m - G[2][k] * m;
            float kd = 0.f, rd = 0.f;
            if (has_d) {
                kd = (Z[k] > 0.01f && live) ? 1.f : 0.f;
                if (p.depth_needs_opaque) kd *= op[k] > 0.95f ? 1.f : 0.f;
                rd = Dv[k] * kd - Z[k] * kd;
            }
            if (!BACKWARD) {
                acc0 += om * (fabsf(r0) + fabsf(r1) + fabsf(r2));
                acc1 += fabsf(rd);
            } else {
                auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
                const float q0 = Wr * om * sgn(r0) * m, q1 = Wr * om * sgn(r1) * m, q2 = Wr * om * sgn(r2) * m;
                dI[0][k] = ea * q0; dI[1][k] = ea * q1; dI[2][k] = ea * q2;
                dO[k] = p.weight_by_opacity ? Wr * (fabsf(r0) + fabsf(r1) + fabsf(r2)) : 0.f;
                dD[k] = Wd * sgn(rd) * kd;
                acc0 += ea * (q0 * I[0][k] + q1 * I[1][k] + q2 * I[2][k]);
                acc1 += q0 + q1 + q2;
            }
        }
        if (BACKWARD) {
            auto store4 = [&](float *dst, const float v[4]) {
                if (VEC) *reinterpret_cast<float4 *>(dst + base) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (k < n) dst[base + k] = v[k];
                }
            };
#pragma unroll
            for (int c = 0; c < 3; c++) store4(p.d_image + c * P, dI[c]);
            if (p.d_opacity) store4(p.d_opacity, dO);
            if (p.d_depth) store4(p.d_depth, dD);
        }
    }
    const float s0 = block_sum(acc0, s_red);
    const float s1 = block_sum(acc1, s_red);
    if (threadIdx.x == 0) { p.partial[2 * blockIdx.x] = s0; p.partial[2 * blockIdx.x + 1] = s1; }
}

template <bool BACKWARD>
__global__ void __launch_bounds__(256) photometric_finish_kernel(LossParams p, int nblk) {
    __shared__ float s[2][256];
    float a0 = 0.f, a1 = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) { a0 += p.partial[2 * b]; a1 += p.partial[2 * b + 1]; }
    s[0][threadIdx.x] = a0; s[1][threadIdx.x] = a1;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { s[0][threadIdx.x] += s[0][threadIdx.x + st]; s[1][threadIdx.x] += s[1][threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (!BACKWARD) {
            p.loss[0] = p.w_rgb * (s[0][0] / (3.f * (float)p.P)) + p.w_d * (s[1][0] / (float)p.P);
        } else {
            if (p.d_a) p.d_a[0] = s[0][0];
            if (p.d_b) p.d_b[0] = s[1][0];
        }
    }
}

}  // namespace
}  // namespace lvdgs

using namespace lvdgs;

extern "C" {

size_t lvdgs_loss_scratch_bytes(int32_t width, int32_t height) {
    const int64_t P = (int64_t)width * height;
    return align256((size_t)cdiv(P, LOSS_THREADS * LOSS_PIX_PER_THREAD) * 2 * sizeof(float) + 256);
}

static int loss_common(const lvdgs_loss_args *a, LossParams &p, int &nblk) {
    if (!a || a->width <= 0 || a->height <= 0) { set_error("loss: bad image size"); return LVDGS_E_INVALID; }
    if (!a->image || !a->gt_image || !a->scratch) { set_error("loss: image / gt_image / scratch is NULL"); return LVDGS_E_INVALID; }
    if (a->scratch_bytes < lvdgs_loss_scratch_bytes(a->width, a->height)) { set_error("loss: scratch too small"); return LVDGS_E_INVALID; }
    if ((a->weight_by_opacity || a->depth_needs_opaque) && !a->opacity) { set_error("loss: opacity is NULL"); return LVDGS_E_INVALID; }
    p = LossParams{};
    p.P = a->width * a->height;
    p.image = a->image; p.depth = a->depth; p.opacity = a->opacity; p.gt_image = a->gt_image; p.gt_depth = a->gt_depth;
    p.grad_mask = a->grad_mask; p.exposure_a = a->exposure_a; p.exposure_b = a->exposure_b;
    p.rgb_thr = a->rgb_boundary_threshold; p.w_rgb = a->weight_rgb; p.w_d = (a->depth && a->gt_depth) ? a->weight_depth : 0.f;
    p.weight_by_opacity = a->weight_by_opacity; p.depth_needs_opaque = a->depth_needs_opaque;
    p.partial = (float *)a->scratch;
    nblk = cdiv(p.P, LOSS_THREADS * LOSS_PIX_PER_THREAD);
    return LVDGS_OK;
}

int lvdgs_photometric_loss_forward(const lvdgs_loss_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    LossParams p; int nblk;
    if (int e = loss_common(a, p, nblk)) return e;
    if (!a->loss) { set_error("loss: output is NULL"); return LVDGS_E_INVALID; }
    p.loss = a->loss;
    { ProfScope ps("loss_fwd", s); if (p.P % 4 == 0) hipLaunchKernelGGL((photometric_kernel<false, true>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p); else hipLaunchKernelGGL((photometric_kernel<false, false>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p); LVDGS_LAUNCH_CHECK("loss_fwd", 0, s); }
    { ProfScope ps("loss_fwd_finish", s); hipLaunchKernelGGL(photometric_finish_kernel<false>, dim3(1), dim3(256), 0, s, p, nblk); LVDGS_LAUNCH_CHECK("loss_fwd_finish", 0, s); }
    return LVDGS_OK;
}

int lvdgs_photometric_loss_backward(const lvdgs_loss_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    LossParams p; int nblk;
    if (int e = loss_common(a, p, nblk)) return e;
    if (!a->grad_loss || !a->d_image) { set_error("loss backward: grad_loss / d_image is NULL"); return LVDGS_E_INVALID; }
    p.grad_out = a->grad_loss; p.d_image = a->d_image; p.d_depth = a->d_depth; p.d_opacity = a->d_opacity;
    p.d_a = a->d_exposure_a; p.d_b = a->d_exposure_b;
    { ProfScope ps("loss_bwd", s); if (p.P % 4 == 0) hipLaunchKernelGGL((photometric_kernel<true, true>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p); else hipLaunchKernelGGL((photometric_kernel<true, false>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p); LVDGS_LAUNCH_CHECK("loss_bwd", 0, s); }
    { ProfScope ps("loss_bwd_finish", s); hipLaunchKernelGGL(photometric_finish_kernel<true>, dim3(1), dim3(256), 0, s, p, nblk); LVDGS_LAUNCH_CHECK("loss_bwd_finish", 0, s); }
    return LVDGS_OK;
}

}  // extern "C"
