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
