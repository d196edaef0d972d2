// Fused Adam step over all parameter tensors of the Gaussian map in ONE launch.
//
// The reference steps the map with torch.optim.Adam every mapping iteration (utils/slam_backend.py:144, :378, :458).
// PyTorch's multi-tensor path takes about nine launches per moment / parameter update across the six groups (0.6 ms of
// GPU time per step at 500k Gaussians, each pass re-reading the 28 MB of parameters or the 56 MB of moments).  Here a
// thread owns four consecutive elements of one tensor: grad, exp_avg, exp_avg_sq and the parameter are read once
// and written once -- 28 bytes per element, one pass.
//
// Arithmetic: torch.optim.Adam's single-tensor statements in float32, scalars as doubles rounded where they meet the
// tensors (see pose.hip: adam_update):
//     exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
//     denom = (exp_avg_sq.sqrt() / sqrt(1 - beta2^t)).add_(eps); param.addcdiv_(exp_avg, denom, value=-lr / (1 - beta1^t))
#include <math.h>

#include "common.hpp"
#include "map_stats.hpp"

namespace lvdgs {
namespace {

struct AdamTensor {
    float *param;
    const float *grad;
    float *m, *v;
    long long numel;
    float step_size, bc2_sqrt;       // lr / (1 - beta1^t), sqrt(1 - beta2^t)
};
struct AdamParams {
    AdamTensor t[LVDGS_ADAM_MAX_TENSORS];
    float w1, b2, w2, eps;           // 1 - beta1, beta2, 1 - beta2, eps
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamParams &P, const AdamTensor &t) {
    m = m + P.w1 * (g - m);
    v = v * P.b2 + P.w2 * (g * g);
    const float denom = sqrtf(v) / t.bc2_sqrt + P.eps;
    p = p - t.step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adam_kernel(AdamParams P) {
    const AdamTensor t = P.t[blockIdx.y];
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (base >= t.numel) return;
    const bool vec = (base + 4 <= t.numel) && ((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0);
    if (vec) {
        float4 p = *reinterpret_cast<const float4 *>(t.param + base), g = *reinterpret_cast<const float4 *>(t.grad + base);
        float4 m = *reinterpret_cast<const float4 *>(t.m + base), v = *reinterpret_cast<const float4 *>(t.v + base);
        adam_one(p.x, g.x, m.x, v.x, P, t); adam_one(p.y, g.y, m.y, v.y, P, t);
        adam_one(p.z, g.z, m.z, v.z, P, t); adam_one(p.w, g.w, m.w, v.w, P, t);
        *reinterpret_cast<float4 *>(t.m + base) = m; *reinterpret_cast<float4 *>(t.v + base) = v;
        *reinterpret_cast<float4 *>(t.param + base) = p;
    } else {
        for (long long i = base; i < base + 4 && i < t.numel; i++) {
            float p = t.param[i], m = t.m[i], v = t.v[i];
            adam_one(p, t.grad[i], m, v, P, t);
            t.m[i] = m; t.v[i] = v; t.param[i] = p;
        }
    }
}

}  // namespace
}  // namespace lvdgs

using namespace lvdgs;

extern "C" int lvdgs_adam_step(const lvdgs_adam_tensor *tensors, int32_t count, double beta1, double beta2, double eps, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (count < 0 || count > LVDGS_ADAM_MAX_TENSORS || (count > 0 && !tensors)) { set_error("adam: between 0 and %d tensors per call", LVDGS_ADAM_MAX_TENSORS); return LVDGS_E_INVALID; }
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0)) { set_error("adam: betas must lie in [0, 1)"); return LVDGS_E_INVALID; }
    AdamParams P{};
    P.w1 = (float)(1.0 - beta1); P.b2 = (float)beta2; P.w2 = (float)(1.0 - beta2); P.eps = (float)eps;
    long long longest = 0;
    int n = 0;
    for (int i = 0; i < count; i++) {
        const lvdgs_adam_tensor &t = tensors[i];
        if (t.numel < 0 || t.step < 1) { set_error("adam: negative size or step count below 1"); return LVDGS_E_INVALID; }
        if (t.numel == 0) continue;
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) { set_error("adam: NULL tensor"); return LVDGS_E_INVALID; }
        AdamTensor &o = P.t[n++];
        o.param = t.param; o.grad = t.grad; o.m = t.exp_avg; o.v = t.exp_avg_sq; o.numel = t.numel;
        o.step_size = (float)(t.lr / (1.0 - pow(beta1, (double)t.step)));
        o.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)t.step));
        if (t.numel > longest) longest = t.numel;
    }
    if (n == 0) return LVDGS_OK;
    ProfScope ps("adam_step", s);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((longest + 1023) / 1024), (unsigned)n), dim3(256), 0, s, P);
    LVDGS_LAUNCH_CHECK("adam_step", 0, s);
    return LVDGS_OK;
}

// ------------------------------------------------------------------------------------------
// Two small fusions for the mapping iteration (reference utils/slam_backend.py:303-305, :350-357).
namespace lvdgs {
namespace {

// Isotropic regulariser  L = weight * mean_{i,k} | s_ik - mean_k s_ik |,  s = exp(raw log-scale)  (slam_backend.py:303-305):
// the value (per-workgroup partial sums, finished in a fixed order by the last launch below) and its gradient w.r.t. the
// RAW scales, ADDED to the gradient the render's backward left there.  PyTorch: exp, mean, sub, abs, mean, mul and
// their six backward kernels over N x 3.
__global__ void __launch_bounds__(256) isotropic_kernel(int N, const float *__restrict__ raw, float *__restrict__ grad, float scale,
                                                        float *__restrict__ partial) {
    __shared__ float s_red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < N) {
        const float s0 = expf(raw[3 * (size_t)i]), s1 = expf(raw[3 * (size_t)i + 1]), s2 = expf(raw[3 * (size_t)i + 2]);   // (expf, like torch.exp and the projection kernel)
        const float m = (s0 + s1 + s2) / 3.f;
        const float d0 = s0 - m, d1 = s1 - m, d2 = s2 - m;
        acc = fabsf(d0) + fabsf(d1) + fabsf(d2);
        auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
        const float g0 = sgn(d0), g1 = sgn(d1), g2 = sgn(d2), gm = (g0 + g1 + g2) / 3.f;
        if (grad) {
            grad[3 * (size_t)i] += scale * (g0 - gm) * s0;
            grad[3 * (size_t)i + 1] += scale * (g1 - gm) * s1;
            grad[3 * (size_t)i + 2] += scale * (g2 - gm) * s2;
        }
    }
    // block sum in a fixed order
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

__global__ void __launch_bounds__(256) isotropic_finish_kernel(int nblk, const float *__restrict__ partial, float scale, float *__restrict__ loss) {
    __shared__ float s[256];
    float a = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) a += partial[b];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = scale * s[0];
}

// What the back end derives from one view's render package (slam_backend.py:311-315, :350-357), accumulated over the
// views a rank rendered: element-wise max of the radii, sum of the screen-space gradient norms and count of the views that
// saw each Gaussian, and the view's own (n_touched > 0) row.  One launch per view instead of ~10.
// A view rendered in bands by several ranks (lvdgs_args.tile_row_*) has only a share of its screen-space gradient here:
// the norm is of the SUM over the bands, so the band's xy goes to split_xy (N x 2, summed over the ranks by the caller's
// all-reduce; lvdgs_map_stats_apply then takes the norm) instead of into norm_sum; vis_count is passed for one band only.
__global__ void __launch_bounds__(256) view_stats_kernel(ViewStats v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < v.N) view_stats_one(v, i);
}

// The statistics' way into the model, one launch (slam_backend.py:350-357 on the reduced values):
//   max_radii2D = max(max_radii2D, radii_max);  xyz_gradient_accum += norm_sum + sum over split views |split_xy|;  denom += vis_count
__global__ void __launch_bounds__(256) map_stats_apply_kernel(int N, const int32_t *__restrict__ radii_max, const float *__restrict__ norm_sum,
                                                              const float *__restrict__ vis_count, const float *__restrict__ split_xy, int n_split,
                                                              float *__restrict__ max_radii2D, float *__restrict__ grad_accum, float *__restrict__ denom) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float r = (float)radii_max[i];
    if (r > max_radii2D[i]) max_radii2D[i] = r;
    float g = norm_sum[i];
    for (int k = 0; k < n_split; k++) {
        const float2 v = *reinterpret_cast<const float2 *>(split_xy + 2 * ((size_t)k * N + i));
        g += sqrtf(v.x * v.x + v.y * v.y);
    }
    grad_accum[i] += g;
    denom[i] += vis_count[i];
}

}  // namespace
}  // namespace lvdgs

extern "C" size_t lvdgs_isotropic_scratch_bytes(int32_t num_gaussians) { return align256((size_t)cdiv(num_gaussians > 0 ? num_gaussians : 1, 256) * sizeof(float) + 256); }

extern "C" int lvdgs_isotropic_reg(int32_t N, const float *raw_scales, float *grad_raw_scales, float weight, void *scratch, size_t scratch_bytes,
                                   float *loss, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (N < 0 || !loss || (N > 0 && (!raw_scales || !scratch))) { set_error("isotropic reg: bad arguments"); return LVDGS_E_INVALID; }
    if (N == 0) return check_hip(hipMemsetAsync(loss, 0, sizeof(float), s), "memset loss");
    if (scratch_bytes < lvdgs_isotropic_scratch_bytes(N)) { set_error("isotropic reg: scratch too small"); return LVDGS_E_INVALID; }
    const int nblk = cdiv(N, 256);
    const float scale = weight / (3.f * (float)N);
    { ProfScope ps("isotropic_reg", s); hipLaunchKernelGGL(isotropic_kernel, dim3(nblk), dim3(256), 0, s, N, raw_scales, grad_raw_scales, scale, (float *)scratch); LVDGS_LAUNCH_CHECK("isotropic_reg", 0, s); }
    { ProfScope ps("isotropic_finish", s); hipLaunchKernelGGL(isotropic_finish_kernel, dim3(1), dim3(256), 0, s, nblk, (const float *)scratch, scale, loss); LVDGS_LAUNCH_CHECK("isotropic_finish", 0, s); }
    return LVDGS_OK;
}

extern "C" int lvdgs_view_stats(int32_t N, const int32_t *radii, const int32_t *n_touched, const float *viewspace_grad, int32_t *radii_max,
                                float *norm_sum, float *vis_count, uint8_t *touched_row, float *split_xy, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (N < 0 || (N > 0 && (!radii || !radii_max || (!norm_sum && !split_xy) || (touched_row && !n_touched)))) {
        set_error("view stats: a required pointer is NULL"); return LVDGS_E_INVALID;
    }
    if (N == 0) return LVDGS_OK;
    ProfScope ps("view_stats", s);
    hipLaunchKernelGGL(view_stats_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, ViewStats{N, radii, n_touched, viewspace_grad, radii_max, norm_sum, vis_count, touched_row, split_xy});
    LVDGS_LAUNCH_CHECK("view_stats", 0, s);
    return LVDGS_OK;
}

extern "C" int lvdgs_map_stats_apply(int32_t N, const int32_t *radii_max, const float *norm_sum, const float *vis_count, const float *split_xy,
                                     int32_t n_split, float *max_radii2D, float *xyz_gradient_accum, float *denom, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (N < 0 || n_split < 0 || (N > 0 && (!radii_max || !norm_sum || !vis_count || !max_radii2D || !xyz_gradient_accum || !denom || (n_split > 0 && !split_xy)))) {
        set_error("map stats apply: a required pointer is NULL"); return LVDGS_E_INVALID;
    }
    if (N == 0) return LVDGS_OK;
    ProfScope ps("map_stats_apply", s);
    hipLaunchKernelGGL(map_stats_apply_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, N, radii_max, norm_sum, vis_count, split_xy, n_split, max_radii2D,
                       xyz_gradient_accum, denom);
    LVDGS_LAUNCH_CHECK("map_stats_apply", 0, s);
    return LVDGS_OK;
}
