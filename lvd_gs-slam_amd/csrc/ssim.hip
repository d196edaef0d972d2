// Fused L1 + SSIM image loss with its gradient (reference utils/slam_backend.py:199-215, 438-454:
// `(1 - lambda) * l1_loss(a, b) + lambda * (1 - ssim(a, b))` on the rendered and the ground-truth image,
// optionally after the dynamic pixels of both were overwritten with the background colour).
//
// SSIM as published with 3DGS (gaussian_splatting.utils.loss_utils.ssim; the package is absent from the
// reference checkout): 11-tap Gaussian window, sigma 1.5, zero padding, per channel,
//   A = w*x, B = w*y, S = w*(x^2 + y^2), Z = w*(xy)
//   m = (2AB + C1)(2(Z - AB) + C2) / ((A^2 + B^2 + C1)(S - A^2 - B^2 + C2)),   C1 = 0.01^2, C2 = 0.03^2
// and the mean of m over every pixel and channel.
//
// PyTorch evaluates this with five 11x11 grouped convolutions and ~20 full-frame elementwise kernels,
// and autograd doubles it.  Here one workgroup owns a 32x32 output tile of one plane and keeps every
// intermediate in LDS:
//   1. load the 52x52 input window of both images (zero outside the image, bg under the mask);
//   2. horizontal then vertical 11-tap pass over (x, y, x^2 + y^2, xy) -> A, B, S, Z on 42x42 pixels;
//   3. m and its partial derivatives dm/dA, dm/dS, dm/dZ on those 42x42 pixels (zero outside the image);
//   4. horizontal then vertical pass over the three derivative maps -> on the 32x32 tile
//        d mean(m) / d x_p = [ w*dm/dA + 2 x_p (w*dm/dS) + y_p (w*dm/dZ) ]_p / count.
// Every thread computes 4-8 neighbouring outputs per pass so one LDS read feeds several FMAs.
// The two means are per-workgroup partial sums finished in a fixed order (deterministic, no atomics).
#include "common.hpp"
#include "device_utils.hpp"

namespace lvdgs {
namespace {

constexpr int ST = 32;           // output tile edge
constexpr int SR = 5;            // window radius
constexpr int SW = 2 * SR + 1;   // taps
constexpr int S1 = ST + 4 * SR;  // 52: input window edge
constexpr int S2 = ST + 2 * SR;  // 42: edge of the region where m is needed
constexpr int P1 = S1 + 1;       // odd LDS pitches: row-strided accesses fall in different banks
constexpr int P2 = S2 + 1;
constexpr int P3 = ST + 1;
constexpr int SSIM_THREADS = 512;  // 8 waves per workgroup, 2 workgroups per CU (LDS): 4 waves per SIMD
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

// One launch serves up to SSIM_BATCH images of one size (lvdgs_masked_loss_batch: the masked keyframes of a mapping window;
// blockIdx.z = view * planes + plane), each with its own pointers, mask, background and weights.
constexpr int SSIM_BATCH = 10;
struct SsimView {
    const float *x, *y;          // planes*H*W
    const uint8_t *keep;         // H*W or null
    const float *bg;             // channels or null
    float w_l1, w_ssim;          // gradient weights, already divided by the element count
    float *partial;              // (workgroups of the view) * 2
    float *d_x;                  // planes*H*W or null
    // depth term of the static-mask mapping loss (reference utils/slam_backend.py:216-261), taken by the workgroups of the
    // view's plane 0 over their own 32x32 pixels: sum of |D - Z| and the exact count of M = keep & (Z > 0) & (D > 0)
    const float *depth, *gt_depth;   // H*W each, or null: no depth term
    float *dpart;                // per tile: sum (float), count (uint32 bits)
};
struct SsimParams {
    int W, H, planes, channels;  // planes: per view
    float win[SW];
    SsimView v[SSIM_BATCH];
};

__device__ __forceinline__ float block_sum8(float v, float *s) {
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) s[wave] = v;
    __syncthreads();
    return (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])));
}

__device__ __forceinline__ float block_sum4(float v, float *s) {
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) s[wave] = v;
    __syncthreads();
    return ((s[0] + s[1]) + s[2]) + s[3];
}

template <bool GRAD>
__global__ void __launch_bounds__(SSIM_THREADS) ssim_l1_kernel(SsimParams p) {
    // sxy: the two input windows; later the three derivative maps (3 * S2 * P2 <= 2 * S1 * P1)
    __shared__ float s_in[2 * S1 * P1];
    // horizontal pass results, TWO maps at a time ((x, y), then (x^2 + y^2, xy)): 2 x S1 x P2; later the horizontal pass of the
    // derivative maps (3 x S2 x P3).  With all four maps resident the kernel held 57.8 KB of LDS -- two workgroups per CU, four
    // waves per SIMD, and it is latency-bound (a dozen barriers, dependent LDS round trips): 39.9 KB lets a third and fourth in.
    __shared__ float s_h[2 * S1 * P2];
    __shared__ float s_red[8];
    static_assert(3 * S2 * P2 <= 2 * S1 * P1, "derivative maps must fit the input windows");
    static_assert(3 * S2 * P3 <= 2 * S1 * P2, "second horizontal pass must fit the first");

    const int tid = threadIdx.x;
    const int view = (int)blockIdx.z / p.planes, plane = (int)blockIdx.z - view * p.planes;
    const SsimView &v = p.v[view];
    const int tx0 = blockIdx.x * ST, ty0 = blockIdx.y * ST;
    const size_t plane_off = (size_t)plane * p.W * p.H;
    const float *__restrict__ X = v.x + plane_off;
    const float *__restrict__ Y = v.y + plane_off;
    const float bgc = v.bg ? v.bg[plane % p.channels] : 0.f;
    // depth term: this workgroup's own pixels, two per thread, requested with the input windows
    const bool depth_term = plane == 0 && v.gt_depth != nullptr;
    float dD[2] = {0.f, 0.f}, dZ[2] = {0.f, 0.f};
    bool dM[2] = {false, false};
    if (depth_term) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = tid + k * SSIM_THREADS, r = i / ST, c = i - r * ST;
            const int gy = ty0 + r, gx = tx0 + c;
            if (gy < p.H && gx < p.W) {
                const size_t o = (size_t)gy * p.W + gx;
                dD[k] = v.depth[o]; dZ[k] = v.gt_depth[o];
                dM[k] = !v.keep || v.keep[o];
            }
        }
    }
    float w[SW];
#pragma unroll
    for (int k = 0; k < SW; k++) w[k] = p.win[k];

    // ---- 1. input windows (+ the L1 sum over this tile's own pixels) ----
    float l1 = 0.f;
    float *sx = s_in, *sy = s_in + S1 * P1;
    {
        // A wave takes 64 columns of one window row (52 of them exist), the workgroup's eight waves eight rows: seven rounds, row and
        // column without a division (the round-4 form -- thread i takes window element i + 512 k, row = element / 52 -- spent a
        // third of the kernel's vector instructions on index arithmetic: SQ_INSTS_VALU 1162 per wave for 561 multiply-adds).
        // All global loads of the window are issued before the first use.
        constexpr int NLD = (S1 + 7) / 8;
        const int lc = tid & 63, lr = tid >> 6;
        const int gx = tx0 - 2 * SR + lc;
        const bool col_in = lc < S1 && gx >= 0 && gx < p.W;
        const int gy0 = ty0 - 2 * SR + lr;
        float la[NLD], lb[NLD];
        uint8_t lk[NLD];
#pragma unroll
        for (int it = 0; it < NLD; it++) {
            const int r = lr + 8 * it, gy = gy0 + 8 * it;
            const bool in = col_in && r < S1 && gy >= 0 && gy < p.H;
            const size_t o = in ? (size_t)gy * p.W + gx : 0;
            la[it] = in ? X[o] : 0.f;
            lb[it] = in ? Y[o] : 0.f;
            lk[it] = (in && v.keep) ? v.keep[o] : (uint8_t)1;
        }
        const bool col_own = lc >= 2 * SR && lc < 2 * SR + ST;
#pragma unroll
        for (int it = 0; it < NLD; it++) {
            const int r = lr + 8 * it;
            if (lc < S1 && r < S1) {
                float a = la[it], b = lb[it];   // (outside the image: zeros, mask byte 1)
                if (!lk[it]) a = b = bgc;
                if (col_own && r >= 2 * SR && r < 2 * SR + ST) l1 += fabsf(a - b);   // (own pixels outside the image add |0 - 0|)
                sx[r * P1 + lc] = a;
                sy[r * P1 + lc] = b;
            }
        }
    }
    __syncthreads();

    // ---- 2. separable 11-tap passes, two maps at a time: (x, y) -> A, B, then (x^2 + y^2, xy) -> S, Z.
    //      horizontal: rows 0..S1, output columns 0..S2; one item = 5 outputs of one row, so the 52 x 9 items fill one round of the
    //      workgroup (the last column group overlaps its neighbour); vertical: one item = 4 outputs of one column (42 x 11 items,
    //      the last row group overlaps), kept in registers for step 3 ----
    constexpr int VOUT = 4, VNIN = VOUT + SW - 1, VGROUPS = (S2 + VOUT - 1) / VOUT;  // 42 columns x 11 row groups = 462 items
    static_assert(S2 * VGROUPS <= SSIM_THREADS, "one round");
    const bool v_item = tid < S2 * VGROUPS;
    const int vc = tid % S2, v_rfirst = (tid / S2) * VOUT, v_r0 = min(v_rfirst, S2 - VOUT);  // the last group overlaps
    float vA[VOUT], vB[VOUT], vS[VOUT], vZ[VOUT];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        {
            constexpr int OUT = 5, NIN = OUT + SW - 1, GROUPS = (S2 + OUT - 1) / OUT;  // 52 rows x 9 column groups = 468 items
            static_assert(S1 * GROUPS <= SSIM_THREADS, "one round");
            if (tid < S1 * GROUPS) {
                const int r = tid % S1, g = tid / S1;
                const int c0 = min(g * OUT, S2 - OUT);
                float p0[NIN], p1[NIN];
#pragma unroll
                for (int i = 0; i < NIN; i++) {
                    const float xv = sx[r * P1 + c0 + i], yv = sy[r * P1 + c0 + i];
                    p0[i] = pass == 0 ? xv : fmaf(xv, xv, yv * yv);
                    p1[i] = pass == 0 ? yv : xv * yv;
                }
#pragma unroll
                for (int o = 0; o < OUT; o++) {
                    float a = w[0] * p0[o], b = w[0] * p1[o];
#pragma unroll
                    for (int k = 1; k < SW; k++) {
                        a = fmaf(w[k], p0[o + k], a);
                        b = fmaf(w[k], p1[o + k], b);
                    }
                    const int d = r * P2 + c0 + o;
                    s_h[d] = a;
                    s_h[S1 * P2 + d] = b;
                }
            }
        }
        __syncthreads();
        if (v_item) {
            float v0[VNIN], v1[VNIN];
#pragma unroll
            for (int i = 0; i < VNIN; i++) {
                const int d = (v_r0 + i) * P2 + vc;
                v0[i] = s_h[d];
                v1[i] = s_h[S1 * P2 + d];
            }
#pragma unroll
            for (int o = 0; o < VOUT; o++) {
                float a = w[0] * v0[o], b = w[0] * v1[o];
#pragma unroll
                for (int k = 1; k < SW; k++) {
                    a = fmaf(w[k], v0[o + k], a);
                    b = fmaf(w[k], v1[o + k], b);
                }
                if (pass == 0) { vA[o] = a; vB[o] = b; } else { vS[o] = a; vZ[o] = b; }
            }
        }
        __syncthreads();   // (pass 0: s_h is written again; pass 1: the input windows make room for the derivative maps)
    }

    // ---- 3. m and its derivatives on the S2 x S2 region ----
    float msum = 0.f;
    float *sdA = s_in, *sdS = s_in + S2 * P2, *sdZ = s_in + 2 * S2 * P2;
    if (v_item) {
        const int gx = tx0 - SR + vc;
#pragma unroll
        for (int o = 0; o < VOUT; o++) {
            const float A = vA[o], B = vB[o], S = vS[o], Z = vZ[o];
            const int r = v_r0 + o;
            const int gy = ty0 - SR + r;
            const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const float AB = A * B, AA_BB = fmaf(A, A, B * B);
            const float num1 = fmaf(2.f, AB, SSIM_C1), num2 = fmaf(2.f, Z - AB, SSIM_C2);
            const float den1 = AA_BB + SSIM_C1, den2 = (S - AA_BB) + SSIM_C2;
            // both denominators are >= C1, C2 > 0: the hardware reciprocal (1 ulp) is well inside the tolerance
            const float inv1 = __builtin_amdgcn_rcpf(den1), inv2 = __builtin_amdgcn_rcpf(den2);
            const float inv12 = inv1 * inv2;
            const float m = (num1 * num2) * inv12;
            if (inside && r >= v_rfirst && r >= SR && r < SR + ST && vc >= SR && vc < SR + ST) msum += m;  // overlapped rows count once
            if (GRAD) {
                // dm/dA = 2B (num2 - num1) / (den1 den2) - 2A m (1/den1 - 1/den2); dm/dS = -m / den2; dm/dZ = 2 num1 / (den1 den2)
                const float dA = 2.f * B * (num2 - num1) * inv12 - 2.f * A * m * (inv1 - inv2);
                const float dS = -m * inv2;
                const float dZ = 2.f * num1 * inv12;
                const int d = r * P2 + vc;
                sdA[d] = inside ? dA : 0.f;
                sdS[d] = inside ? dS : 0.f;
                sdZ[d] = inside ? dZ : 0.f;
            }
        }
    }

    // ---- the two sums of this workgroup ----
    {
        const float t1 = block_sum8(l1, s_red);
        const float t2 = block_sum8(msum, s_red);
        if (tid == 0) {
            const size_t wg = ((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            v.partial[2 * wg] = t1;
            v.partial[2 * wg + 1] = t2;
        }
    }
    if (depth_term) {   // (uniform over the workgroup)
        float acc = 0.f;
        uint32_t cnt = 0u;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const bool in = dM[k] && dZ[k] > 0.f && dD[k] > 0.f;
            acc += in ? fabsf(dD[k] - dZ[k]) : 0.f;
            cnt += in ? 1u : 0u;
        }
        const float t3 = block_sum8(acc, s_red);
        // an exact integer count (a float sum would round above 2^24 pixels)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = __uint_as_float(cnt);
        __syncthreads();
        if (tid == 0) {
            uint32_t n = 0u;
#pragma unroll
            for (int w8 = 0; w8 < 8; w8++) n += __float_as_uint(s_red[w8]);
            const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            v.dpart[2 * tile] = t3;
            v.dpart[2 * tile + 1] = __uint_as_float(n);
        }
    }
    if (!GRAD) return;
    __syncthreads();

    // ---- 4a. horizontal pass over the derivative maps: rows 0..S2, output columns 0..ST, 3 per item
    //      (42 x 11 = 462 items, the last column group overlaps its neighbour) ----
    {
        constexpr int OUT = 3, NIN = OUT + SW - 1, GROUPS = (ST + OUT - 1) / OUT;  // 42 rows x 11 column groups = 462 items
        static_assert(S2 * GROUPS <= SSIM_THREADS, "one round");
        if (tid < S2 * GROUPS) {
            const int r = tid % S2, c0 = min((tid / S2) * OUT, ST - OUT);
            float va[NIN], vs[NIN], vz[NIN];
#pragma unroll
            for (int i = 0; i < NIN; i++) {
                const int d = r * P2 + c0 + i;
                va[i] = sdA[d];
                vs[i] = sdS[d];
                vz[i] = sdZ[d];
            }
#pragma unroll
            for (int o = 0; o < OUT; o++) {
                float a = w[0] * va[o], sq = w[0] * vs[o], z = w[0] * vz[o];
#pragma unroll
                for (int k = 1; k < SW; k++) {
                    a = fmaf(w[k], va[o + k], a);
                    sq = fmaf(w[k], vs[o + k], sq);
                    z = fmaf(w[k], vz[o + k], z);
                }
                const int d = r * P3 + c0 + o;
                s_h[d] = a;
                s_h[S2 * P3 + d] = sq;
                s_h[2 * S2 * P3 + d] = z;
            }
        }
    }
    __syncthreads();

    // ---- 4b. vertical pass, 2 outputs per item (32 columns x 16 row pairs = 512 items), and the gradient of this tile ----
    float *__restrict__ G = v.d_x + plane_off;
    for (int item = tid; item < ST * (ST / 2); item += SSIM_THREADS) {
        const int c = item % ST, r0 = (item / ST) * 2;
        // the pixels' own values (for 2 x (w * dm/dS) + y (w * dm/dZ) and the L1 sign): requested before the passes' LDS reads
        float own_x[2] = {0.f, 0.f}, own_y[2] = {0.f, 0.f};
        bool own_keep[2] = {false, false};
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const int gy = ty0 + r0 + o, gx = tx0 + c;
            if (gy < p.H && gx < p.W) {
                const size_t off = (size_t)gy * p.W + gx;
                own_keep[o] = !(v.keep && !v.keep[off]);
                own_x[o] = X[off]; own_y[o] = Y[off];
            }
        }
        float va[12], vs[12], vz[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int d = (r0 + i) * P3 + c;
            va[i] = s_h[d];
            vs[i] = s_h[S2 * P3 + d];
            vz[i] = s_h[2 * S2 * P3 + d];
        }
        const int gx = tx0 + c;
#pragma unroll
        for (int o = 0; o < 2; o++) {
            float a = w[0] * va[o], s = w[0] * vs[o], z = w[0] * vz[o];
#pragma unroll
            for (int k = 1; k < SW; k++) {
                a = fmaf(w[k], va[o + k], a);
                s = fmaf(w[k], vs[o + k], s);
                z = fmaf(w[k], vz[o + k], z);
            }
            const int gy = ty0 + r0 + o;
            if (gy < p.H && gx < p.W) {
                const size_t off = (size_t)gy * p.W + gx;
                float g = 0.f;
                if (own_keep[o]) {  // overwritten pixels do not depend on the input
                    const float xv = own_x[o], yv = own_y[o];
                    const float diff = xv - yv;
                    const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                    g = v.w_ssim * (a + 2.f * xv * s + yv * z) + v.w_l1 * sgn;
                }
                G[off] = g;
            }
        }
    }
}

__global__ void __launch_bounds__(512) ssim_finish_kernel(const float2 *__restrict__ partial, int nwg, float inv_count,
                                                          float *__restrict__ out) {
    __shared__ float s_red[8];
    float a = 0.f, b = 0.f;
    for (int i0 = threadIdx.x; i0 < nwg; i0 += 4 * 512) {  // four independent 8-byte loads in flight per thread
        float2 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i0 + k * 512 < nwg ? partial[i0 + k * 512] : make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; k++) { a += v[k].x; b += v[k].y; }
    }
    const float ta = block_sum8(a, s_red);
    const float tb = block_sum8(b, s_red);
    if (threadIdx.x == 0) {
        out[0] = ta * inv_count;
        out[1] = tb * inv_count;
    }
}

// One workgroup per view of lvdgs_masked_loss_batch: the two means in ssim_finish_kernel's order, the depth term's sum and
// exact count over the view's tiles, and the loss value.
struct MaskedFinishView { const float2 *partial; const float2 *dpart; float *out; float lambda, depth_lambda; };
struct MaskedFinishParams { int nwg, ntiles; float inv_count; MaskedFinishView v[SSIM_BATCH]; };

__global__ void __launch_bounds__(512) masked_loss_finish_kernel(MaskedFinishParams p) {
    __shared__ float s_red[8];
    __shared__ unsigned long long s_cnt[8];
    const MaskedFinishView &v = p.v[blockIdx.x];
    float a = 0.f, b = 0.f;
    for (int i0 = threadIdx.x; i0 < p.nwg; i0 += 4 * 512) {
        float2 q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = i0 + k * 512 < p.nwg ? v.partial[i0 + k * 512] : make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; k++) { a += q[k].x; b += q[k].y; }
    }
    const float ta = block_sum8(a, s_red);
    const float tb = block_sum8(b, s_red);
    float d = 0.f;
    unsigned long long n = 0ull;
    if (v.dpart)
        for (int i = threadIdx.x; i < p.ntiles; i += 512) { const float2 q = v.dpart[i]; d += q.x; n += (unsigned long long)__float_as_uint(q.y); }
    const float td = block_sum8(d, s_red);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += (unsigned long long)__shfl_xor((long long)n, off, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long c = 0ull;
#pragma unroll
        for (int w8 = 0; w8 < 8; w8++) c += s_cnt[w8];
        const float l1 = ta * p.inv_count, ss = tb * p.inv_count;
        const float dterm = c ? td / (float)c : 0.f;   // nothing is added when M is empty (reference :250)
        v.out[0] = ((1.f - v.lambda) * l1 - v.lambda * ss + v.lambda) + v.depth_lambda * dterm;
        v.out[1] = l1;
        v.out[2] = ss;
        v.out[3] = dterm;
        v.out[4] = (float)c;
    }
}

void fill_window(float (&win)[SW]) {
    // the window upstream builds in float32: exp(-(k - 5)^2 / (2 * 1.5^2)), normalised
    float g[SW], sum = 0.f;
    for (int k = 0; k < SW; k++) { g[k] = expf(-(float)((k - SR) * (k - SR)) / (2.f * 1.5f * 1.5f)); sum += g[k]; }
    for (int k = 0; k < SW; k++) win[k] = g[k] / sum;
}

}  // namespace
}  // namespace lvdgs

using namespace lvdgs;

extern "C" {

size_t lvdgs_ssim_scratch_bytes(int32_t width, int32_t height, int32_t planes) {
    const size_t nwg = (size_t)cdiv(width, ST) * cdiv(height, ST) * (size_t)(planes > 0 ? planes : 0);
    return align256(nwg * 2 * sizeof(float) + 256);
}

int lvdgs_ssim_l1(const lvdgs_ssim_args *a, void *stream) {
    if (!a || a->width <= 0 || a->height <= 0 || a->planes <= 0 || a->channels <= 0) { set_error("ssim: bad image size"); return LVDGS_E_INVALID; }
    if (a->planes % a->channels) { set_error("ssim: planes must be a multiple of channels"); return LVDGS_E_INVALID; }
    if (!a->img1 || !a->img2 || !a->out || !a->scratch) { set_error("ssim: img1 / img2 / out / scratch is NULL"); return LVDGS_E_INVALID; }
    if (a->scratch_bytes < lvdgs_ssim_scratch_bytes(a->width, a->height, a->planes)) { set_error("ssim: scratch too small"); return LVDGS_E_INVALID; }
    if (a->planes > 65535) { set_error("ssim: more than 65535 planes"); return LVDGS_E_RANGE; }
    hipStream_t s = (hipStream_t)stream;
    SsimParams p{};
    p.W = a->width; p.H = a->height; p.planes = a->planes; p.channels = a->channels;
    SsimView &v = p.v[0];
    v.x = a->img1; v.y = a->img2; v.keep = a->keep_mask; v.bg = a->bg;
    const double count = (double)a->width * a->height * a->planes;
    v.w_l1 = (float)(a->weight_l1 / count);
    v.w_ssim = (float)(a->weight_ssim / count);
    v.partial = (float *)a->scratch;
    v.d_x = a->d_img1;
    fill_window(p.win);
    const dim3 grid(cdiv(a->width, ST), cdiv(a->height, ST), a->planes);
    const int nwg = (int)(grid.x * grid.y * grid.z);
    {
        ProfScope ps(a->d_img1 ? "ssim_l1_grad" : "ssim_l1", s);
        if (a->d_img1) hipLaunchKernelGGL(ssim_l1_kernel<true>, grid, dim3(SSIM_THREADS), 0, s, p);
        else hipLaunchKernelGGL(ssim_l1_kernel<false>, grid, dim3(SSIM_THREADS), 0, s, p);
        LVDGS_LAUNCH_CHECK("ssim_l1", 0, s);
    }
    {
        ProfScope ps("ssim_finish", s);
        hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(512), 0, s, (const float2 *)v.partial, nwg, (float)(1.0 / count), a->out);
        LVDGS_LAUNCH_CHECK("ssim_finish", 0, s);
    }
    return LVDGS_OK;
}

// scratch of one view: the L1 / SSIM partial sums of its 3 planes, then (sum, count) per 32x32 tile for the depth term
size_t lvdgs_masked_loss_scratch_bytes(int32_t width, int32_t height) {
    const size_t tiles = (size_t)cdiv(width, ST) * cdiv(height, ST);
    return align256(3 * tiles * 2 * sizeof(float)) + align256(tiles * 2 * sizeof(float)) + 256;
}

int lvdgs_masked_loss_batch(const lvdgs_masked_loss_args *const *views, int32_t count, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (count < 0 || (count > 0 && !views)) { set_error("masked loss: bad view list"); return LVDGS_E_INVALID; }
    for (int32_t first = 0; first < count; first += SSIM_BATCH) {
        const int m = count - first < SSIM_BATCH ? count - first : SSIM_BATCH;
        SsimParams p{};
        MaskedFinishParams f{};
        const lvdgs_masked_loss_args *a0 = views[first];
        if (!a0 || a0->width <= 0 || a0->height <= 0) { set_error("masked loss: view %d is NULL or has a bad image size", first); return LVDGS_E_INVALID; }
        p.W = a0->width; p.H = a0->height; p.planes = 3; p.channels = 3;
        fill_window(p.win);
        const size_t tiles = (size_t)cdiv(p.W, ST) * cdiv(p.H, ST);
        const double n = (double)p.W * p.H * 3;
        for (int k = 0; k < m; k++) {
            const lvdgs_masked_loss_args *a = views[first + k];
            if (!a || a->width != p.W || a->height != p.H) { set_error("masked loss: view %d is NULL or differs in image size", first + k); return LVDGS_E_INVALID; }
            if (!a->image || !a->gt_image || !a->d_image || !a->out || !a->scratch) { set_error("masked loss: image / gt_image / d_image / out / scratch is NULL"); return LVDGS_E_INVALID; }
            if (a->scratch_bytes < lvdgs_masked_loss_scratch_bytes(p.W, p.H)) { set_error("masked loss: scratch too small"); return LVDGS_E_INVALID; }
            if (a->gt_depth && !a->depth) { set_error("masked loss: gt_depth without the rendered depth"); return LVDGS_E_INVALID; }
            SsimView &v = p.v[k];
            v.x = a->image; v.y = a->gt_image; v.keep = a->static_mask; v.bg = a->static_mask ? a->bg : nullptr;
            // d loss / d a = (1 - lambda) d mean|a - b| - lambda d mean SSIM
            v.w_l1 = (float)((1.0 - (double)a->lambda_dssim) / n);
            v.w_ssim = (float)(-(double)a->lambda_dssim / n);
            v.partial = (float *)a->scratch;
            v.d_x = a->d_image;
            v.depth = a->gt_depth ? a->depth : nullptr; v.gt_depth = a->gt_depth;
            v.dpart = a->gt_depth ? (float *)((char *)a->scratch + align256(3 * tiles * 2 * sizeof(float))) : nullptr;
            f.v[k] = MaskedFinishView{(const float2 *)v.partial, (const float2 *)v.dpart, a->out, a->lambda_dssim, a->gt_depth ? a->depth_lambda : 0.f};
        }
        f.nwg = (int)(3 * tiles); f.ntiles = (int)tiles; f.inv_count = (float)(1.0 / n);
        const dim3 grid(cdiv(p.W, ST), cdiv(p.H, ST), 3 * m);
        {
            ProfScope ps("masked_loss", s);
            hipLaunchKernelGGL(ssim_l1_kernel<true>, grid, dim3(SSIM_THREADS), 0, s, p);
            LVDGS_LAUNCH_CHECK("masked_loss", 0, s);
        }
        {
            ProfScope ps("masked_loss_finish", s);
            hipLaunchKernelGGL(masked_loss_finish_kernel, dim3(m), dim3(512), 0, s, f);
            LVDGS_LAUNCH_CHECK("masked_loss_finish", 0, s);
        }
    }
    return LVDGS_OK;
}

}  // extern "C"
