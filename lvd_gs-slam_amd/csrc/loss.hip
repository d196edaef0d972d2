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
#include "photometric.hpp"

namespace lvdgs {
namespace {

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
// MODE 0: the loss value; 1: every gradient; 2: both in ONE pass over the images (the tracking session's case: the
// objective IS the loss, d objective / d loss is known -- 1, or a device scalar -- before the pixels are read).
template <int MODE, bool VEC>
__global__ void __launch_bounds__(LOSS_THREADS) photometric_kernel(LossParams p) {
    constexpr bool BACKWARD = MODE != 0, VALUE = MODE != 1;
    __shared__ float s_red[4];
    const LossConsts lc(p, BACKWARD);
    float acc0 = 0.f, acc1 = 0.f;  // rgb sum, depth sum (value)
    float acc2 = 0.f, acc3 = 0.f;  // d_a sum, d_b sum (gradients)
    const int base = (blockIdx.x * LOSS_THREADS + threadIdx.x) * LOSS_PIX_PER_THREAD;
    const size_t P = (size_t)p.P;
    const bool has_d = lc.has_d;
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
            const float Gk[3] = {G[0][k], G[1][k], G[2][k]}, Ik[3] = {I[0][k], I[1][k], I[2][k]};
            const PixelLoss px = photometric_pixel<VALUE, BACKWARD>(p, lc, Gk, Ik, op[k], has_d ? Z[k] : 0.f, has_d ? Dv[k] : 0.f, gm[k], VEC || k < n);
            if (VALUE) { acc0 += px.v_rgb; acc1 += px.v_d; }
            if (BACKWARD) {
                dI[0][k] = px.dI[0]; dI[1][k] = px.dI[1]; dI[2][k] = px.dI[2];
                dO[k] = px.dO; dD[k] = px.dD;
                acc2 += px.s_a; acc3 += px.s_b;
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
    // partial sums of a block: [0], [1] value (rgb, depth), [2], [3] gradients (d_a, d_b)
    if (VALUE) {
        const float s0 = block_sum(acc0, s_red);
        const float s1 = block_sum(acc1, s_red);
        if (threadIdx.x == 0) { p.partial[4 * blockIdx.x] = s0; p.partial[4 * blockIdx.x + 1] = s1; }
    }
    if (BACKWARD) {
        const float s2 = block_sum(acc2, s_red);
        const float s3 = block_sum(acc3, s_red);
        if (threadIdx.x == 0) { p.partial[4 * blockIdx.x + 2] = s2; p.partial[4 * blockIdx.x + 3] = s3; }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) photometric_finish_kernel(LossParams p, int nblk) {
    constexpr bool BACKWARD = MODE != 0, VALUE = MODE != 1;
    __shared__ float s[4][4];   // [value][wave]
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblk; b += 256) {
        if (VALUE) { a[0] += p.partial[4 * b]; a[1] += p.partial[4 * b + 1]; }
        if (BACKWARD) { a[2] += p.partial[4 * b + 2]; a[3] += p.partial[4 * b + 3]; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float v = wave_sum_to_lane63(a[k]);
        if (lane == 63) s[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 4; k++) s[k][0] = ((s[k][0] + s[k][1]) + s[k][2]) + s[k][3];
    if (threadIdx.x == 0) {
        if (VALUE) p.loss[0] = p.w_rgb * (s[0][0] / (3.f * (float)p.P)) + p.w_d * (s[1][0] / (float)p.P);
        if (BACKWARD) {
            if (p.d_a) p.d_a[0] = s[2][0];
            if (p.d_b) p.d_b[0] = s[3][0];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Depth term of the static-mask mapping loss (reference utils/slam_backend.py:216-261):
//     M = static_mask & (mono_depth > 0) & (rendered depth > 0)
//     loss = mean over M of |D - Z|        (nothing is added when M is empty)
// Unlike the photometric depth term above the mean runs over the pixels of M, not over the whole image, so the
// pass keeps an exact integer count next to the sum; the finish kernel divides.  Backward: sign(D - Z) / |M| on M.
struct MaskedDepthParams {
    int P;
    const float *depth, *gt_depth;
    const uint8_t *mask;      // P bytes or null (every pixel static)
    float *partial_sum;       // nblk
    uint32_t *partial_cnt;    // nblk
    float *out;               // [0] loss, [1] |M| as a float
    const float *grad_out;    // 1
    float *d_depth;           // P
};

template <bool BACKWARD, bool VEC>
__global__ void __launch_bounds__(LOSS_THREADS) masked_depth_kernel(MaskedDepthParams p) {
    __shared__ float s_red[4];
    __shared__ uint32_t s_cnt[4];
    const int base = (blockIdx.x * LOSS_THREADS + threadIdx.x) * LOSS_PIX_PER_THREAD;
    float acc = 0.f;
    uint32_t cnt = 0;
    float scale = 0.f;
    if (BACKWARD) { const float n = p.out[1]; scale = n > 0.f ? p.grad_out[0] / n : 0.f; }
    if (base < p.P) {
        const int n = VEC ? 4 : min(4, p.P - base);
        float D[4], Z[4], dD[4];
        bool m[4];
        if (VEC) {
            const float4 d4 = *reinterpret_cast<const float4 *>(p.depth + base), z4 = *reinterpret_cast<const float4 *>(p.gt_depth + base);
            D[0] = d4.x; D[1] = d4.y; D[2] = d4.z; D[3] = d4.w; Z[0] = z4.x; Z[1] = z4.y; Z[2] = z4.z; Z[3] = z4.w;
            const uint32_t m4 = p.mask ? *reinterpret_cast<const uint32_t *>(p.mask + base) : 0x01010101u;
#pragma unroll
            for (int k = 0; k < 4; k++) m[k] = ((m4 >> (8 * k)) & 0xffu) != 0u;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                D[k] = k < n ? p.depth[base + k] : 0.f; Z[k] = k < n ? p.gt_depth[base + k] : 0.f;
                m[k] = k < n && (!p.mask || p.mask[base + k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool in = m[k] && Z[k] > 0.f && D[k] > 0.f;
            const float r = D[k] - Z[k];
            if (!BACKWARD) { acc += in ? fabsf(r) : 0.f; cnt += in ? 1u : 0u; }
            else dD[k] = in ? (r > 0.f ? scale : (r < 0.f ? -scale : 0.f)) : 0.f;
        }
        if (BACKWARD) {
            if (VEC) *reinterpret_cast<float4 *>(p.d_depth + base) = make_float4(dD[0], dD[1], dD[2], dD[3]);
            else {
#pragma unroll
                for (int k = 0; k < 4; k++) if (k < n) p.d_depth[base + k] = dD[k];
            }
        }
    }
    if (!BACKWARD) {
        const float s0 = block_sum(acc, s_red);
        // exact integer count: wave popcount-free sum through the same lane-63 reduction on floats would round above 2^24
        uint32_t c = cnt;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, 64);
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) { p.partial_sum[blockIdx.x] = s0; p.partial_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]; }
    }
}

__global__ void __launch_bounds__(256) masked_depth_finish_kernel(MaskedDepthParams p, int nblk) {
    __shared__ float s[256];
    __shared__ unsigned long long c[256];
    float a = 0.f;
    unsigned long long n = 0ull;
    for (int b = threadIdx.x; b < nblk; b += 256) { a += p.partial_sum[b]; n += p.partial_cnt[b]; }
    s[threadIdx.x] = a; c[threadIdx.x] = n;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { s[threadIdx.x] += s[threadIdx.x + st]; c[threadIdx.x] += c[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p.out[0] = c[0] ? s[0] / (float)c[0] : 0.f;
        p.out[1] = (float)c[0];
    }
}

}  // namespace
}  // namespace lvdgs

using namespace lvdgs;

extern "C" {

size_t lvdgs_loss_scratch_bytes(int32_t width, int32_t height) {
    // 4 partial sums per workgroup of the loss kernels (1024 pixels each) or, when the backward blend pass evaluates the
    // loss (lvdgs_backward_fused_loss), per 16x16 tile: room for the larger of the two
    const int64_t P = (int64_t)width * height;
    const int64_t tiles = (int64_t)cdiv(width, TILE) * cdiv(height, TILE);
    const int64_t blocks = cdiv(P, LOSS_THREADS * LOSS_PIX_PER_THREAD);
    return align256((size_t)(blocks > tiles ? blocks : tiles) * 4 * sizeof(float) + 256);
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
    { ProfScope ps("loss_fwd", s); if (p.P % 4 == 0) hipLaunchKernelGGL((photometric_kernel<0, true>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p); else hipLaunchKernelGGL((photometric_kernel<0, false>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p); LVDGS_LAUNCH_CHECK("loss_fwd", 0, s); }
    { ProfScope ps("loss_fwd_finish", s); hipLaunchKernelGGL(photometric_finish_kernel<0>, dim3(1), dim3(256), 0, s, p, nblk); LVDGS_LAUNCH_CHECK("loss_fwd_finish", 0, s); }
    return LVDGS_OK;
}

int lvdgs_photometric_loss_backward(const lvdgs_loss_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    LossParams p; int nblk;
    if (int e = loss_common(a, p, nblk)) return e;
    if (!a->grad_loss || !a->d_image) { set_error("loss backward: grad_loss / d_image is NULL"); return LVDGS_E_INVALID; }
    p.grad_out = a->grad_loss; p.d_image = a->d_image; p.d_depth = a->d_depth; p.d_opacity = a->d_opacity;
    p.d_a = a->d_exposure_a; p.d_b = a->d_exposure_b;
    { ProfScope ps("loss_bwd", s); if (p.P % 4 == 0) hipLaunchKernelGGL((photometric_kernel<1, true>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p); else hipLaunchKernelGGL((photometric_kernel<1, false>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p); LVDGS_LAUNCH_CHECK("loss_bwd", 0, s); }
    { ProfScope ps("loss_bwd_finish", s); hipLaunchKernelGGL(photometric_finish_kernel<1>, dim3(1), dim3(256), 0, s, p, nblk); LVDGS_LAUNCH_CHECK("loss_bwd_finish", 0, s); }
    return LVDGS_OK;
}

static int value_and_grad_params(const lvdgs_loss_args *a, LossParams &p, int &nblk, bool need_loss, bool need_images = true) {
    if (int e = loss_common(a, p, nblk)) return e;
    if ((need_loss && !a->loss) || (need_images && !a->d_image)) { set_error("loss value_and_grad: loss / d_image is NULL"); return LVDGS_E_INVALID; }
    p.loss = a->loss;
    p.grad_out = a->grad_loss;  // NULL: d objective / d loss = 1
    p.d_image = a->d_image; p.d_depth = a->d_depth; p.d_opacity = a->d_opacity; p.d_a = a->d_exposure_a; p.d_b = a->d_exposure_b;
    return LVDGS_OK;
}

static void launch_value_and_grad_pass(const LossParams &p, int nblk, hipStream_t s) {
    ProfScope ps("loss_both", s);
    if (p.P % 4 == 0) hipLaunchKernelGGL((photometric_kernel<2, true>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p);
    else hipLaunchKernelGGL((photometric_kernel<2, false>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p);
}

int lvdgs_photometric_loss_value_and_grad(const lvdgs_loss_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    LossParams p; int nblk;
    if (int e = value_and_grad_params(a, p, nblk, true)) return e;
    launch_value_and_grad_pass(p, nblk, s);
    LVDGS_LAUNCH_CHECK("loss_both", 0, s);
    { ProfScope ps("loss_both_finish", s); hipLaunchKernelGGL(photometric_finish_kernel<2>, dim3(1), dim3(256), 0, s, p, nblk); LVDGS_LAUNCH_CHECK("loss_both_finish", 0, s); }
    return LVDGS_OK;
}

int lvdgs_photometric_loss_partials(const lvdgs_loss_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    LossParams p; int nblk;
    if (int e = value_and_grad_params(a, p, nblk, false)) return e;
    launch_value_and_grad_pass(p, nblk, s);
    LVDGS_LAUNCH_CHECK("loss_both", 0, s);
    return LVDGS_OK;
}

}  // extern "C"

namespace lvdgs {
int loss_tail_params(const lvdgs_loss_args *a, bool partials_per_tile, LossTail *out) {
    LossParams p; int nblk;
    if (int e = value_and_grad_params(a, p, nblk, true, !partials_per_tile)) return e;
    if (partials_per_tile) nblk = cdiv(a->width, TILE) * cdiv(a->height, TILE);
    *out = LossTail{p.partial, nblk, p.P, p.w_rgb, p.w_d, p.loss, p.d_a, p.d_b};
    return LVDGS_OK;
}

int loss_fused_params(const lvdgs_loss_args *a, LossParams *out) {
    int nblk;
    return value_and_grad_params(a, *out, nblk, false, false);
}
}  // namespace lvdgs

extern "C" {

size_t lvdgs_masked_depth_scratch_bytes(int32_t width, int32_t height) {
    const int64_t P = (int64_t)width * height;
    return align256((size_t)cdiv(P, LOSS_THREADS * LOSS_PIX_PER_THREAD) * (sizeof(float) + sizeof(uint32_t)) + 256);
}

static int masked_depth_common(const lvdgs_masked_depth_args *a, MaskedDepthParams &p, int &nblk, bool &vec) {
    if (!a || a->width <= 0 || a->height <= 0) { set_error("masked depth loss: bad image size"); return LVDGS_E_INVALID; }
    if (!a->depth || !a->gt_depth || !a->out) { set_error("masked depth loss: depth / gt_depth / out is NULL"); return LVDGS_E_INVALID; }
    p = MaskedDepthParams{};
    p.P = a->width * a->height;
    p.depth = a->depth; p.gt_depth = a->gt_depth; p.mask = a->static_mask; p.out = a->out;
    nblk = cdiv(p.P, LOSS_THREADS * LOSS_PIX_PER_THREAD);
    vec = p.P % 4 == 0 && ((uintptr_t)a->depth | (uintptr_t)a->gt_depth | (uintptr_t)a->d_depth) % 16 == 0 && (uintptr_t)a->static_mask % 4 == 0;
    return LVDGS_OK;
}

int lvdgs_masked_depth_l1_forward(const lvdgs_masked_depth_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    MaskedDepthParams p; int nblk; bool vec;
    if (int e = masked_depth_common(a, p, nblk, vec)) return e;
    if (!a->scratch || a->scratch_bytes < lvdgs_masked_depth_scratch_bytes(a->width, a->height)) { set_error("masked depth loss: scratch missing or too small"); return LVDGS_E_INVALID; }
    p.partial_sum = (float *)a->scratch;
    p.partial_cnt = (uint32_t *)(p.partial_sum + nblk);
    {
        ProfScope ps("masked_depth_fwd", s);
        if (vec) hipLaunchKernelGGL((masked_depth_kernel<false, true>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p);
        else hipLaunchKernelGGL((masked_depth_kernel<false, false>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p);
        LVDGS_LAUNCH_CHECK("masked_depth_fwd", 0, s);
    }
    { ProfScope ps("masked_depth_finish", s); hipLaunchKernelGGL(masked_depth_finish_kernel, dim3(1), dim3(256), 0, s, p, nblk); LVDGS_LAUNCH_CHECK("masked_depth_finish", 0, s); }
    return LVDGS_OK;
}

int lvdgs_masked_depth_l1_backward(const lvdgs_masked_depth_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    MaskedDepthParams p; int nblk; bool vec;
    if (int e = masked_depth_common(a, p, nblk, vec)) return e;
    if (!a->grad_loss || !a->d_depth) { set_error("masked depth loss backward: grad_loss / d_depth is NULL"); return LVDGS_E_INVALID; }
    p.grad_out = a->grad_loss; p.d_depth = a->d_depth;
    ProfScope ps("masked_depth_bwd", s);
    if (vec) hipLaunchKernelGGL((masked_depth_kernel<true, true>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p);
    else hipLaunchKernelGGL((masked_depth_kernel<true, false>), dim3(nblk), dim3(LOSS_THREADS), 0, s, p);
    LVDGS_LAUNCH_CHECK("masked_depth_bwd", 0, s);
    return LVDGS_OK;
}

}  // extern "C"
