// Per-Gaussian kernels: projection (forward), parameter / pose gradients (backward), frustum test.
// One lane per Gaussian, wave64, 256-thread workgroups.  These are HBM-streaming kernels:
// forward reads 56 B and writes 60 B per Gaussian, backward reads 56 + 48*(tiles touched) B and
// writes the 14-float parameter gradient.  Compiled with -ffp-contract=off: the float expressions
// that decide integers (radius, tile rectangle, depth sort key) are evaluated as written.
#include "common.hpp"
#include "binning.hpp"
#include "device_utils.hpp"

namespace lvdgs {

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

struct Cam {
    const float *view, *proj, *proj_raw, *campos;
    float tanx, tany, fx, fy, scale_mod;
    int W, H, gx, gy, sh_degree, M;
};

__device__ __forceinline__ void xform3(const float p[3], const float *__restrict__ m, float o[3]) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ float xform_w(const float p[3], const float *__restrict__ m) {
    return m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

__device__ __forceinline__ void quat_rot(const float q[4], float R[3][3]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = (R diag(mod*s)) (R diag(mod*s))^T as xx,xy,xz,yy,yz,zz
__device__ __forceinline__ void cov3d_of(const float s[3], float mod, const float q[4], float c6[6]) {
    float R[3][3], M[3][3];
    quat_rot(q, R);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[i][j] = R[i][j] * (mod * s[j]);
    c6[0] = M[0][0] * M[0][0] + M[0][1] * M[0][1] + M[0][2] * M[0][2];
    c6[1] = M[0][0] * M[1][0] + M[0][1] * M[1][1] + M[0][2] * M[1][2];
    c6[2] = M[0][0] * M[2][0] + M[0][1] * M[2][1] + M[0][2] * M[2][2];
    c6[3] = M[1][0] * M[1][0] + M[1][1] * M[1][1] + M[1][2] * M[1][2];
    c6[4] = M[1][0] * M[2][0] + M[1][1] * M[2][1] + M[1][2] * M[2][2];
    c6[5] = M[2][0] * M[2][0] + M[2][1] * M[2][1] + M[2][2] * M[2][2];
}

struct Ewa {
    float T[2][3];
    float t[3];
    bool clx, cly;
};

__device__ __forceinline__ void ewa_setup(const float pv[3], const float *__restrict__ V, const Cam &c, Ewa &e) {
    const float limx = FOV_GUARD * c.tanx, limy = FOV_GUARD * c.tany;
    const float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    e.clx = (txtz < -limx) || (txtz > limx);
    e.cly = (tytz < -limy) || (tytz > limy);
    const float cx = txtz < -limx ? -limx : (txtz > limx ? limx : txtz);
    const float cy = tytz < -limy ? -limy : (tytz > limy ? limy : tytz);
    e.t[0] = cx * pv[2]; e.t[1] = cy * pv[2]; e.t[2] = pv[2];
    const float j00 = c.fx / e.t[2], j02 = -(c.fx * e.t[0]) / (e.t[2] * e.t[2]);
    const float j11 = c.fy / e.t[2], j12 = -(c.fy * e.t[1]) / (e.t[2] * e.t[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float w0 = V[4 * k + 0], w1 = V[4 * k + 1], w2 = V[4 * k + 2];
        e.T[0][k] = j00 * w0 + j02 * w2;
        e.T[1][k] = j11 * w1 + j12 * w2;
    }
}

__device__ __forceinline__ void sym6(const float c6[6], float S[3][3]) {
    S[0][0] = c6[0]; S[0][1] = S[1][0] = c6[1]; S[0][2] = S[2][0] = c6[2];
    S[1][1] = c6[3]; S[1][2] = S[2][1] = c6[4]; S[2][2] = c6[5];
}

__device__ __forceinline__ void cov2d_of(const Ewa &e, const float c6[6], float &a, float &b, float &c) {
    float S[3][3], TS[2][3];
    sym6(c6, S);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) TS[i][j] = e.T[i][0] * S[0][j] + e.T[i][1] * S[1][j] + e.T[i][2] * S[2][j];
    a = TS[0][0] * e.T[0][0] + TS[0][1] * e.T[0][1] + TS[0][2] * e.T[0][2] + LOWPASS;
    b = TS[0][0] * e.T[1][0] + TS[0][1] * e.T[1][1] + TS[0][2] * e.T[1][2];
    c = TS[1][0] * e.T[1][0] + TS[1][1] * e.T[1][1] + TS[1][2] * e.T[1][2] + LOWPASS;
}

__device__ __forceinline__ void sh_basis(int deg, const float d[3], float B[16]) {
    const float x = d[0], y = d[1], z = d[2];
#pragma unroll
    for (int k = 0; k < 16; k++) B[k] = 0.f;
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.f * zz - xx - yy);
            B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = SH_C3[0] * y * (3.f * xx - yy); B[10] = SH_C3[1] * xy * z;
                B[11] = SH_C3[2] * y * (4.f * zz - xx - yy); B[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                B[13] = SH_C3[4] * x * (4.f * zz - xx - yy); B[14] = SH_C3[5] * z * (xx - yy);
                B[15] = SH_C3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}

// gradient of basis k w.r.t. the unit direction (x,y,z)
__device__ __forceinline__ void sh_basis_grad(int deg, const float d[3], float G[16][3]) {
    const float x = d[0], y = d[1], z = d[2];
#pragma unroll
    for (int k = 0; k < 16; k++) G[k][0] = G[k][1] = G[k][2] = 0.f;
    if (deg > 0) { G[1][1] = -SH_C1; G[2][2] = SH_C1; G[3][0] = -SH_C1; }
    if (deg > 1) {
        G[4][0] = SH_C2[0] * y; G[4][1] = SH_C2[0] * x;
        G[5][1] = SH_C2[1] * z; G[5][2] = SH_C2[1] * y;
        G[6][0] = SH_C2[2] * -2.f * x; G[6][1] = SH_C2[2] * -2.f * y; G[6][2] = SH_C2[2] * 4.f * z;
        G[7][0] = SH_C2[3] * z; G[7][2] = SH_C2[3] * x;
        G[8][0] = SH_C2[4] * 2.f * x; G[8][1] = SH_C2[4] * -2.f * y;
    }
    if (deg > 2) {
        const float xx = x * x, yy = y * y, zz = z * z;
        G[9][0] = SH_C3[0] * 6.f * x * y; G[9][1] = SH_C3[0] * (3.f * xx - 3.f * yy);
        G[10][0] = SH_C3[1] * y * z; G[10][1] = SH_C3[1] * x * z; G[10][2] = SH_C3[1] * x * y;
        G[11][0] = SH_C3[2] * -2.f * x * y; G[11][1] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); G[11][2] = SH_C3[2] * 8.f * y * z;
        G[12][0] = SH_C3[3] * -6.f * x * z; G[12][1] = SH_C3[3] * -6.f * y * z; G[12][2] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
        G[13][0] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); G[13][1] = SH_C3[4] * -2.f * x * y; G[13][2] = SH_C3[4] * 8.f * x * z;
        G[14][0] = SH_C3[5] * 2.f * x * z; G[14][1] = SH_C3[5] * -2.f * y * z; G[14][2] = SH_C3[5] * (xx - yy);
        G[15][0] = SH_C3[6] * (3.f * xx - 3.f * yy); G[15][1] = SH_C3[6] * -6.f * x * y;
    }
}

// Optional activations fused into the projection (lvdgs_args.activations): the model's raw parameters are
// read and activated here, and preprocess_bwd applies the chain rule, instead of separate elementwise
// kernels (exp / sigmoid / normalise and their backward) before and after the rasterizer.
constexpr int ACT_EXP_SCALES = 1, ACT_NORMALIZE_ROT = 2, ACT_SIGMOID_OPACITY = 4;

// the activations of raw scales / rotations (s, q hold the raw values on entry)
__device__ __forceinline__ void activate_scale_rot(int act, float s[3], float q[4], float &qnorm) {
    if (act & ACT_EXP_SCALES) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
    qnorm = 1.f;
    if (act & ACT_NORMALIZE_ROT) {
        qnorm = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
        q[0] /= qnorm; q[1] /= qnorm; q[2] /= qnorm; q[3] /= qnorm;
    }
}

struct FwdParams {
    Cam cam;
    int N, act;
    int tile_cull;         // 0: list every tile of the rectangle (the reference's pair list)
    int row_begin, row_end; // tile rows that are rendered (lvdgs_args.tile_row_begin / _end): pairs outside are not listed
    const float *means3D, *opacities, *scales, *rotations, *cov3D_precomp, *shs, *colors_precomp;
    float *rec;
    uint32_t *tiles_touched, *depth_bits;
    uint4 *rect;
    int32_t *radii;
    uint32_t *blocksums;   // per workgroup: sum of tiles_touched (first level of the slot scan, sortscan.hip)
    // two-level grouping (LVDGS_FLAG_SUPER_TILES; preprocess_count_kernel<..., true>): the super-tile grid's rectangles, count matrix, queue counters
    uint4 *super_rect; uint32_t *super_hist, *super_queue_counts; int super_gx, super_T;
};

// What the projection reads of one Gaussian whatever becomes of it, requested in ONE round of loads (position, then --
// only if in front of the camera -- scale and rotation, then -- only if on the image -- opacity and colour were three
// dependent round trips, each of them exposed: the kernels below run as a single resident round of workgroups, every
// wave in the same phase).  Gaussians that turn out culled have 44 bytes read for nothing.
struct RawGaussian {
    float pos[3], sc[3], q[4], opac, col[3];   // raw (not activated) values; col: colors_precomp or the SH DC coefficients
};
__device__ __forceinline__ RawGaussian load_raw(const FwdParams &p, int i) {
    RawGaussian r;
#pragma unroll
    for (int k = 0; k < 3; k++) r.pos[k] = p.means3D[3 * (size_t)i + k];
    if (!p.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 3; k++) r.sc[k] = p.scales[3 * (size_t)i + k];
#pragma unroll
        for (int k = 0; k < 4; k++) r.q[k] = p.rotations[4 * (size_t)i + k];
    } else {
        r.sc[0] = r.sc[1] = r.sc[2] = 0.f; r.q[0] = 1.f; r.q[1] = r.q[2] = r.q[3] = 0.f;
    }
    r.opac = p.opacities[i];
    const float *col = p.colors_precomp ? p.colors_precomp + 3 * (size_t)i : p.shs + (size_t)i * p.cam.M * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) r.col[k] = col[k];
    return r;
}
__device__ __forceinline__ void preprocess_one(const FwdParams &p, int i, const RawGaussian &raw, uint32_t &tiles_out, uint4 &rect_out);

__global__ void __launch_bounds__(256) preprocess_fwd_kernel(FwdParams p) {
    __shared__ uint32_t s_sum[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t tiles = 0;
    uint4 rect;
    if (i < p.N) preprocess_one(p, i, load_raw(p, i), tiles, rect);
    // the workgroup's pair count: the slot scan starts from these sums instead of re-reading tiles_touched in a launch of its own
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tiles += (uint32_t)__shfl_xor((int)tiles, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = tiles;
    __syncthreads();
    if (threadIdx.x == 0) p.blocksums[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// The projection that also counts: a workgroup owns the chunk of GROUP_THREADS x PER consecutive Gaussians the grouping kernels
// (binning.hip) work in, keeps the per-tile pair counters of that chunk in LDS while it projects, and leaves the chunk's
// row of the [chunk][tile] count matrix and the chunk's pair total -- the grouping's first kernel and its re-read of
// every rectangle are gone from the single-call forward (lvdgs_forward).  Also clears what the later kernels of the
// frame accumulate into (n_touched, the tile-sort queue counters).
// SUPER_COUNT (two-level grouping, binning.hip): the chunk's row of the SUPER-TILE grid's count matrix and the Gaussians' super
// rectangles are made here too -- a second set of LDS counters behind the tiles', a second walk over (far fewer) cells -- instead of
// by a kernel of their own that re-reads every rectangle (17 us on the opaque-surface workload).  A template parameter: the default
// instantiation is the kernel it was.
template <int GROUP_THREADS, int OWNERS, int PER, bool SUPER_COUNT = false>
__device__ __forceinline__ void preprocess_count_body(const FwdParams &p, int T, uint32_t *__restrict__ hist,
                                                      uint32_t *__restrict__ chunk_sums, int32_t *__restrict__ n_touched,
                                                      uint32_t *__restrict__ queue_counts) {
    constexpr bool HELPERS = GROUP_THREADS > OWNERS;   // waves without a Gaussian of their own: they help with the large rectangles (binning.hpp)
    static_assert(!HELPERS || PER == 1, "helper waves: one Gaussian per owner thread");
    extern __shared__ uint32_t s_tile[];
    __shared__ uint32_t s_sum[GROUP_THREADS / 64];
    __shared__ BigRectQueue s_big;
    const bool owner = (int)threadIdx.x < OWNERS;
    // the first Gaussian's values are on their way while the counters are cleared
    const int i_first = blockIdx.x * (OWNERS * PER) + (int)threadIdx.x;
    RawGaussian raw{};
    if (owner && i_first < p.N) raw = load_raw(p, i_first);
    const int T_all = SUPER_COUNT ? T + p.super_T : T;   // (the super-tile counters lie behind the tiles')
    uint32_t *s_super = s_tile + T;
    for (int t = threadIdx.x; t < T_all; t += GROUP_THREADS) s_tile[t] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        queue_counts[threadIdx.x] = 0u;
        if constexpr (SUPER_COUNT) p.super_queue_counts[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) s_big.count = 0u;
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll 1
    for (int k = 0; k < PER; k++) {
        const int i = blockIdx.x * (OWNERS * PER) + k * OWNERS + (int)threadIdx.x;
        uint32_t tiles = 0;
        uint4 rect = make_uint4(0u, 0u, 0u, 0u);
        if (owner && i < p.N) {
            if (k > 0) raw = load_raw(p, i);
            preprocess_one(p, i, raw, tiles, rect);
            n_touched[i] = 0;
        }
        mine += tiles;
        if constexpr (HELPERS) for_each_pair_of_rect_wg(rect, i, p.cam.gx, 0u, s_big, [&](int tile, uint32_t, uint32_t) { atomicAdd(&s_tile[tile], 1u); });
        else for_each_pair_of_rect(rect, i, p.cam.gx, 0u, [&](int tile, uint32_t, uint32_t) { atomicAdd(&s_tile[tile], 1u); });
        if constexpr (SUPER_COUNT) {
            const uint4 rs = super_rect_of(rect);
            if (owner && i < p.N) p.super_rect[i] = rs;
            if constexpr (HELPERS) {
                __syncthreads();                         // (every wave is done with the queue of the tile walk)
                if (threadIdx.x == 0) s_big.count = 0u;
                __syncthreads();
                for_each_pair_of_rect_wg(rs, i, p.super_gx, 0u, s_big, [&](int cell, uint32_t, uint32_t) { atomicAdd(&s_super[cell], 1u); });
            } else {
                for_each_pair_of_rect(rs, i, p.super_gx, 0u, [&](int cell, uint32_t, uint32_t) { atomicAdd(&s_super[cell], 1u); });
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < GROUP_THREADS / 64; w++) total += s_sum[w];
        chunk_sums[blockIdx.x] = total;
    }
    uint32_t *row = hist + (size_t)blockIdx.x * T;
    for (int t = threadIdx.x; t < T; t += GROUP_THREADS) row[t] = s_tile[t];
    if constexpr (SUPER_COUNT) {
        uint32_t *row_s = p.super_hist + (size_t)blockIdx.x * p.super_T;
        for (int t = threadIdx.x; t < p.super_T; t += GROUP_THREADS) row_s[t] = s_super[t];
    }
}

template <int GROUP_THREADS, int OWNERS, int PER, bool SUPER_COUNT = false>
__global__ void __launch_bounds__(GROUP_THREADS, 8) preprocess_count_kernel(FwdParams p, int T, uint32_t *__restrict__ hist,
                                                                        uint32_t *__restrict__ chunk_sums, int32_t *__restrict__ n_touched,
                                                                        uint32_t *__restrict__ queue_counts) {
    preprocess_count_body<GROUP_THREADS, OWNERS, PER, SUPER_COUNT>(p, T, hist, chunk_sums, n_touched, queue_counts);
}
// The views of a mapping window in ONE launch (lvdgs_forward_batch; blockIdx.y: the view -- same map, same image size, a camera,
// state buffers and count matrix each).  A KITTI-size frame's projection is 391 workgroups of latency-bound work on 256 CUs: ten
// of them side by side cost little more than one.
struct PrepCountView { FwdParams p; uint32_t *hist, *chunk_sums; int32_t *n_touched; uint32_t *queue_counts; };
struct PrepCountBatch { PrepCountView v[FWD_BATCH_VIEWS]; };
static_assert(sizeof(PrepCountBatch) <= 4064, "kernel arguments");
template <int GROUP_THREADS, int OWNERS, int PER, bool SUPER_COUNT = false>
__global__ void __launch_bounds__(GROUP_THREADS, 8) preprocess_count_batch_kernel(PrepCountBatch b, int T) {
    const PrepCountView &v = b.v[blockIdx.y];
    preprocess_count_body<GROUP_THREADS, OWNERS, PER, SUPER_COUNT>(v.p, T, v.hist, v.chunk_sums, v.n_touched, v.queue_counts);
}

__device__ __forceinline__ void preprocess_one(const FwdParams &p, int i, const RawGaussian &raw, uint32_t &tiles_out, uint4 &rect_out) {
    const Cam &c = p.cam;
    // culled unless proven visible
    int radius = 0;
    uint32_t tiles = 0;
    uint4 rect = make_uint4(0u, 0u, 0u, 0u);
    uint32_t depth_bits = 0u;
    const float pos[3] = {raw.pos[0], raw.pos[1], raw.pos[2]};
    float pv[3];
    xform3(pos, c.view, pv);
    if (pv[2] > NEAR_CULL) {
        float ph[3];
        xform3(pos, c.proj, ph);
        const float phw = xform_w(pos, c.proj);
        const float pw = 1.f / (phw + HOMOG_EPS);
        const float ndcx = ph[0] * pw, ndcy = ph[1] * pw;
        float c6[6];
        if (p.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = p.cov3D_precomp[6 * (size_t)i + k];
        } else {
            float s[3] = {raw.sc[0], raw.sc[1], raw.sc[2]}, q[4] = {raw.q[0], raw.q[1], raw.q[2], raw.q[3]}, qn;
            activate_scale_rot(p.act, s, q, qn);
            cov3d_of(s, c.scale_mod, q, c6);
        }
        Ewa e;
        ewa_setup(pv, c.view, c, e);
        float ca, cb, cc;
        cov2d_of(e, c6, ca, cb, cc);
        const float det = ca * cc - cb * cb;
        if (det != 0.f) {
            const float det_inv = 1.f / det;
            const float k0 = cc * det_inv, k1 = -cb * det_inv, k2 = ca * det_inv;
            const float mid = 0.5f * (ca + cc);
            float disc = mid * mid - det;
            if (disc < LAMBDA_FLOOR) disc = LAMBDA_FLOOR;
            const float l1 = mid + sqrtf(disc), l2 = mid - sqrtf(disc);
            const float lmax = l1 > l2 ? l1 : l2;
            const int rad = (int)ceilf(3.f * sqrtf(lmax));
            const float px = ((ndcx + 1.f) * (float)c.W - 1.f) * 0.5f;
            const float py = ((ndcy + 1.f) * (float)c.H - 1.f) * 0.5f;
            int x0 = (int)((px - (float)rad) / (float)TILE), y0 = (int)((py - (float)rad) / (float)TILE);
            int x1 = (int)((px + (float)rad + (float)(TILE - 1)) / (float)TILE);
            int y1 = (int)((py + (float)rad + (float)(TILE - 1)) / (float)TILE);
            x0 = min(c.gx, max(0, x0)); x1 = min(c.gx, max(0, x1));
            y0 = min(c.gy, max(0, y0)); y1 = min(c.gy, max(0, y1));
            // (a Gaussian is visible -- radius > 0, record written -- when its rectangle meets the IMAGE; which of its tiles
            // are listed is then a matter of the band being rendered and of the tile culling)
            const bool on_image = (x1 - x0) * (y1 - y0) > 0;
            y0 = min(p.row_end, max(p.row_begin, y0)); y1 = min(p.row_end, max(p.row_begin, y1));
            const int area = (x1 - x0) * (y1 - y0);
            if (on_image) {
                float rgb[3];
                if (p.colors_precomp) {
                    rgb[0] = raw.col[0]; rgb[1] = raw.col[1]; rgb[2] = raw.col[2];
                } else {
                    float d[3] = {pos[0] - c.campos[0], pos[1] - c.campos[1], pos[2] - c.campos[2]};
                    const float inv = 1.f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    d[0] *= inv; d[1] *= inv; d[2] *= inv;
                    float B[16];
                    sh_basis(c.sh_degree, d, B);
                    const int nb = (c.sh_degree + 1) * (c.sh_degree + 1);
                    const float *sh = p.shs + (size_t)i * c.M * 3;
                    rgb[0] = 0.f + B[0] * raw.col[0]; rgb[1] = 0.f + B[0] * raw.col[1]; rgb[2] = 0.f + B[0] * raw.col[2];
#pragma unroll
                    for (int k = 1; k < 16; k++)
                        if (k < nb) {
                            rgb[0] += B[k] * sh[3 * k]; rgb[1] += B[k] * sh[3 * k + 1]; rgb[2] += B[k] * sh[3 * k + 2];
                        }
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) { rgb[ch] += 0.5f; rgb[ch] = rgb[ch] < 0.f ? 0.f : rgb[ch]; }
                }
                float opac = raw.opac;
                if (p.act & ACT_SIGMOID_OPACITY) opac = 1.f / (1.f + expf(-opac));
                // the tiles of the rectangle the Gaussian can reach with alpha >= 1/255 (common.hpp: rect_keeps)
                uint64_t mask = ~0ull;
                radius = rad; tiles = (uint32_t)area;
                if (area == 0) { x0 = x1 = y0 = y1 = 0; mask = 0ull; }   // not in this band
                else if (p.tile_cull) {
                    const float far_x = fmaxf(fabsf(px - (float)(x0 * TILE)), fabsf(px - (float)(x1 * TILE - 1)));
                    const float far_y = fmaxf(fabsf(py - (float)(y0 * TILE)), fabsf(py - (float)(y1 * TILE - 1)));
                    const TileReach reach(px, py, k0, k1, k2, opac, far_x, far_y);
                    mask = 0ull;
                    if (area <= RECT_MASK_TILES) {
                        uint64_t bit = 1ull;
                        for (int y = y0; y < y1; y++)
                            for (int x = x0; x < x1; x++) {
                                if (reach.tile((float)(x * TILE), (float)(y * TILE))) mask |= bit;
                                bit <<= 1;
                            }
                        tiles = (uint32_t)__popcll(mask);
                    } else {
                        // one bit per block of tiles (common.hpp: RectBlocks): the same test on the block's pixel rectangle
                        const RectBlocks g(x1 - x0, y1 - y0);
                        tiles = 0;
                        for (int b = 0; b < 64; b++) {
                            const int bwid = g.width(b), bhei = g.height(b);
                            if (bwid <= 0 || bhei <= 0) continue;
                            const int tx = x0 + (b & 7) * g.bw, ty = y0 + (b >> 3) * g.bh;
                            if (reach.rect((float)(tx * TILE), (float)(ty * TILE), (float)(bwid * TILE - 1), (float)(bhei * TILE - 1))) {
                                mask |= 1ull << b;
                                tiles += (uint32_t)(bwid * bhei);
                            }
                        }
                    }
                } else if (area > RECT_MASK_TILES) {
                    // every tile listed: all blocks that exist are kept
                    const RectBlocks g(x1 - x0, y1 - y0);
                    mask = 0ull;
                    for (int b = 0; b < 64; b++)
                        if (g.width(b) > 0 && g.height(b) > 0) mask |= 1ull << b;
                }
                rect = make_uint4((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16), (uint32_t)mask, (uint32_t)(mask >> 32));
                depth_bits = __float_as_uint(pv[2]);
                float4 *r4 = reinterpret_cast<float4 *>(p.rec + (size_t)i * REC_FLOATS);
                r4[0] = make_float4(px, py, k0, k1);
                if constexpr (REC_FLOATS >= 16) r4[3] = make_float4(__uint_as_float(rect.x), __uint_as_float(rect.y), __uint_as_float(rect.z), __uint_as_float(rect.w));
                r4[1] = make_float4(k2, opac, rgb[0], rgb[1]);
                r4[2] = make_float4(rgb[2], pv[2], 0.f, __int_as_float(rad));
            }
        }
    }
    p.radii[i] = radius;
    p.tiles_touched[i] = tiles;
    p.rect[i] = rect;
    p.depth_bits[i] = depth_bits;
    tiles_out = tiles;
    rect_out = rect;
}

// ------------------------------------------------------------------------------------------
struct BwdParams {
    Cam cam;
    int N, act;
    const float *means3D, *opacities, *scales, *rotations, *cov3D_precomp, *shs, *colors_precomp;
    const int32_t *radii;
    const float *rec;
    const uint32_t *tiles_touched, *slot_base;
    const float *pair_grads;
    const uint8_t *pair_valid;   // 1 where blend_bwd wrote the pair's record (pairs behind their tile's last contributor have none)
    float *dmeans3D, *dmeans2D, *dopac, *dscales, *drot, *dcov3D, *dshs, *dcolors;
    float *tau_part;
    int accumulate;   // LVDGS_FLAG_ACCUMULATE_PARAM_GRADS: the parameter gradients are added to what their buffers hold
    // lvdgs_forward_backward_fused_loss enqueues this pass before the host knows the frame's pair count: when the count (left on the
    // device by the tile scan) exceeds the capacity the buffers were sized for, the pass does NOTHING -- its slots would lie beyond
    // the record buffer -- and the caller runs the backward again behind a forward with room.  Null: no such check.
    const uint32_t *pair_total;
    uint32_t pair_capacity;
};

#ifndef LVDGS_WAVE_CHUNK
#define LVDGS_WAVE_CHUNK 192
#endif
constexpr int WAVE_CHUNK = LVDGS_WAVE_CHUNK;  // pair records a wave stages per round (a multiple of 4): 7.5 KB of LDS per wave at 192, five workgroups per CU
constexpr int BIG_RUN = 64;      // a Gaussian with more pairs than this is summed by its whole wave
#ifndef LVDGS_PBWD_BIG_UNROLL
#define LVDGS_PBWD_BIG_UNROLL 2
#endif
constexpr int BIG_UNROLL = LVDGS_PBWD_BIG_UNROLL;   // ... LVDGS_PBWD_BIG_UNROLL records per lane and trip
#ifndef LVDGS_PBWD_TAKE
#define LVDGS_PBWD_TAKE 2   // records a lane of the compacted sweep requests from LDS before it adds the first (sum_region_compacted, step 3)
#endif
#ifndef LVDGS_PBWD_SEG
#define LVDGS_PBWD_SEG 512   // slots of a large-footprint wave's region swept per round (sum_region_compacted): 512 or 1024
#endif
// a wave's LDS staging area: WAVE_CHUNK records of the streaming path, or the compacted sweep's pass of records + its lists
template <int PF>
constexpr int STAGE_BYTES = (WAVE_CHUNK * PF * 4 > 128 * PF * 4 + LVDGS_PBWD_SEG * 2 + 320 ? WAVE_CHUNK * PF * 4 : (128 * PF * 4 + LVDGS_PBWD_SEG * 2 + 320 + 15) / 16 * 16);

#ifndef LVDGS_PBWD_WGS
#define LVDGS_PBWD_WGS 5
#endif
#ifndef LVDGS_PBWD_ABLATE
#define LVDGS_PBWD_ABLATE 0   // diagnostic builds: 1 = no pair sums, 2 = no per-Gaussian chain
#endif
// The values are in their registers -- their loads waited for -- at this point of the program.
__device__ __forceinline__ void wait_for_vector_memory() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void loads_complete_here(int32_t &radius, uint32_t &slot, uint32_t &tiles, float (&pos)[3], float &opac, float (&sc)[3],
                                                    float (&q)[4], float (&c6)[6]) {
    asm volatile("" : "+v"(radius), "+v"(slot), "+v"(tiles), "+v"(pos[0]), "+v"(pos[1]), "+v"(pos[2]), "+v"(opac));
    asm volatile("" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    asm volatile("" : "+v"(c6[0]), "+v"(c6[1]), "+v"(c6[2]), "+v"(c6[3]), "+v"(c6[4]), "+v"(c6[5]));
}

// The pair sums of a wave that holds large-footprint Gaussians (a function of its own so that it can be built as a call,
// LVDGS_PBWD_INLINE_BIG = 0: measured, slower).
// s_mem: the wave's LDS staging area.
#ifdef LVDGS_DIAG_PBWD
// diagnostic build (tools/pbwd_diag.py): clocks of the compacted sweep per wave (registers; one row of eight values per wave and launch,
// no atomics -- thousands of waves adding to the same eight words cost more than the kernel).
// [0] 1, [1] whole sweep, [2] flags -> list (wait, masks, scan, list, barrier), [3] gather (requests -> LDS, barrier), [4] sums,
// [5] segments, [6] passes, [7] records
constexpr int PBWD_DIAG_WAVES = 8192;
constexpr int PBWD_DIAG_VALUES = 12;   // [8] trips of the lanes' own loop (the longest lane's), [9] Gaussians summed by the whole wave, [10] clocks of those
__device__ unsigned long long g_pbwd_diag[PBWD_DIAG_WAVES * PBWD_DIAG_VALUES];
#define PBWD_CLK(x) const unsigned long long x = __builtin_readcyclecounter()
#define PBWD_ADD(k, v) (diag[k] += (unsigned long long)(v))
#else
#define PBWD_CLK(x)
#define PBWD_ADD(k, v)
#endif
struct PairSums { float A[10]; };
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool POSE_ONLY>
__device__ __forceinline__ void unpack_sums(const v2f (&P)[POSE_ONLY ? PAIR_FLOATS_POSE / 2 : PAIR_FLOATS / 2], float (&A)[10]) {
    A[0] = P[0].x; A[1] = P[0].y; A[2] = P[1].x; A[3] = P[1].y; A[4] = P[2].x;
    if constexpr (POSE_ONLY) { A[5] = 0.f; A[6] = 0.f; A[7] = 0.f; A[8] = 0.f; A[9] = P[2].y; }
    else { A[5] = P[2].y; A[6] = P[3].x; A[7] = P[3].y; A[8] = P[4].x; A[9] = P[4].y; }
}
#ifndef LVDGS_PBWD_INLINE_BIG
#define LVDGS_PBWD_INLINE_BIG 1   // A/B builds: 0 = a function call (same box, config 3 / opaque surfaces: 50.5 / 96 us against 44.5 / 74.3 inlined: the spills around the call cost more than the separate register allocation returns)
#endif
#if LVDGS_PBWD_INLINE_BIG
#define LVDGS_BIG_PATH_ATTR __forceinline__
#else
#define LVDGS_BIG_PATH_ATTR __attribute__((noinline))
#endif
// A2: the ten sums as packed pairs, in the record's own layout ([0,1] [2,3] [4,5] [6,7] [8,9]; pose-only [0,1] [2,3] [4,9]): a record is
// added with PF / 2 v_pk_add_f32 -- the same IEEE additions, two per instruction.  The records of [w_first, w_hi) are added to what A2 holds.
template <bool POSE_ONLY>
__device__ LVDGS_BIG_PATH_ATTR void sum_region_compacted(const float *__restrict__ pair_grads, const uint8_t *__restrict__ pair_valid, char *s_mem,
                                                              uint32_t first, uint32_t last, uint32_t w_first, uint32_t w_hi,
                                                              v2f (&A2)[POSE_ONLY ? PAIR_FLOATS_POSE / 2 : PAIR_FLOATS / 2]) {
    constexpr int PF = POSE_ONLY ? PAIR_FLOATS_POSE : PAIR_FLOATS;
    const int lane = threadIdx.x & 63;
    {
        // A wave that holds large-footprint Gaussians (hundreds of pairs each: the stuff opaque surfaces are made of).  Of their
        // records only those in front of their tiles' last contributors exist -- a tenth on opaque surfaces -- and a lane that
        // walks its own run (flag, record, flag, record ...: two dependent round trips per slot) while 63 wait was 1.5 ms of this
        // kernel at 100 k such Gaussians; round 3's "the whole wave sums one large Gaussian after the other" still was a chain of
        // two round trips per Gaussian and 64 slots (76 us of 88 on that workload).  Now the wave sweeps its whole region -- the
        // runs of its 64 Gaussians follow each other in memory -- 512 slots at a time:
        //   1. the segment's FLAGS, eight per lane in one load (the next segment's are requested before this one is worked on),
        //      compacted into a list of the slots that hold a record (wave scan of the per-lane counts);
        //   2. just those records, gathered densely into the wave's LDS, BIG_UNROLL per lane in flight;
        //   3. every lane adds up ITS Gaussian's records of the pass from LDS, in slot order (the order of the streaming path);
        //      a Gaussian with more than 64 records in the pass is summed by the whole wave (lane l: records l, l + 64, ...) and
        //      folded in a fixed order.
        constexpr uint32_t SEG = LVDGS_PBWD_SEG;
        constexpr uint32_t FL = SEG / 64u;   // flags (slots) per lane
        static_assert(FL == 8u || FL == 16u, "one 8- or 16-byte load of flags per lane");
        constexpr int AUX_BYTES = (int)SEG * 2 + 64 * 2 + 64 * 2 + PF * 4;                      // list, per-lane prefix, per-lane flag bits, a record of zeros
        constexpr uint32_t CAP = 128u;   // records per pass: the same number in both forms of the kernel, so that both add in the same order
        static_assert(CAP * PF * 4 + AUX_BYTES <= STAGE_BYTES<PF>, "fits the wave's staging area");
        float2 *const s_rec = reinterpret_cast<float2 *>(s_mem);
        uint16_t *const s_list = reinterpret_cast<uint16_t *>(s_mem + CAP * PF * 4);   // offsets (in the segment) of the slots with a record
        uint16_t *const s_before = s_list + SEG;                                                  // records of the segment in front of lane l's eight slots
        uint16_t *const s_bits = s_before + 64;                                                   // lane l's FL flags
        // a record of zeros behind them (8-byte aligned): what the lanes of step 3 read where their run has ended -- adding +0 to a sum
        // that started at +0 leaves its bits as they are -- so that TAKE records can be requested from LDS before the first is added
        float2 *const s_zero = reinterpret_cast<float2 *>(s_bits + 64);
        constexpr uint32_t ZERO_AT = (uint32_t)((CAP * PF * 4 + SEG * 2 + 64 * 2 + 64 * 2) / 8);   // s_zero as an index of s_rec's float2
        static_assert((CAP * PF * 4 + SEG * 2 + 64 * 2 + 64 * 2) % 8 == 0, "aligned");
        if (lane < PF / 2) s_zero[lane] = make_float2(0.f, 0.f);
        const float2 *pg = reinterpret_cast<const float2 *>(pair_grads);
        const uint32_t w_lo = w_first & ~(FL - 1u);   // w_first rounded down to the flags' 8- / 16-byte loads
        struct Flags { uint32_t w[FL / 4]; };
        auto flags_of = [&](uint32_t seg) {   // (pair_valid is padded by 16 bytes)
            const uint32_t s0 = seg + FL * (uint32_t)lane;
            Flags f{};
            if (s0 < w_hi) {
                if constexpr (FL == 8u) { const uint2 v = *reinterpret_cast<const uint2 *>(pair_valid + s0); f.w[0] = v.x; f.w[1] = v.y; }
                else { const uint4 v = *reinterpret_cast<const uint4 *>(pair_valid + s0); f.w[0] = v.x; f.w[1] = v.y; f.w[2] = v.z; f.w[3] = v.w; }
            }
            return f;
        };
        Flags fl_next = w_lo < w_hi ? flags_of(w_lo) : Flags{};
#ifdef LVDGS_DIAG_PBWD
        unsigned long long diag[PBWD_DIAG_VALUES] = {};
#endif
        PBWD_CLK(t_begin);
        PBWD_ADD(0, 1);
        for (uint32_t seg = w_lo; seg < w_hi; seg += SEG) {
            PBWD_CLK(t_seg);
            PBWD_ADD(5, 1);
            const Flags fl = fl_next;
            if (seg + SEG < w_hi) fl_next = flags_of(seg + SEG);
            // ---- 1. which of the segment's slots hold a record ----
            uint32_t mine = 0;   // bit b: slot seg + FL lane + b
#pragma unroll
            for (int w = 0; w < (int)(FL / 4); w++)
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if ((fl.w[w] >> (8 * b)) & 0xffu) mine |= 1u << (4 * w + b);
            {   // slots outside the wave's region are other waves' (or, behind the frame's last pair, nobody's: their flags are stale)
                constexpr uint32_t ALL = (1u << FL) - 1u;
                const uint32_t s0 = seg + FL * (uint32_t)lane;
                const uint32_t keep_hi = s0 >= w_hi ? 0u : (w_hi - s0 >= FL ? ALL : (1u << (w_hi - s0)) - 1u);
                const uint32_t keep_lo = s0 >= w_first ? ALL : (w_first - s0 >= FL ? 0u : (ALL << (w_first - s0)) & ALL);
                mine &= keep_hi & keep_lo;
            }
            const uint32_t cnt = (uint32_t)__popc(mine);
            // inclusive prefix over the wave: row_shr:1,2,4,8 inside the 16-lane rows, row_bcast:15 / :31 chain the rows (six DPP adds
            // instead of six ds_bpermute round trips)
            uint32_t inc = cnt;
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            {
                uint32_t at = inc - cnt;
                s_before[lane] = (uint16_t)at;
                s_bits[lane] = (uint16_t)mine;
                for (uint32_t m = mine; m; m &= m - 1u) s_list[at++] = (uint16_t)(FL * (uint32_t)lane + (uint32_t)__builtin_ctz(m));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // this lane's Gaussian: its records are entries [lo, hi) of the segment's list
            auto records_before = [&](uint32_t slot) {   // slot in [seg, seg + SEG]
                const uint32_t o = slot - seg;
                if (o >= SEG) return total;
                return (uint32_t)s_before[o / FL] + (uint32_t)__popc((uint32_t)s_bits[o / FL] & ((1u << (o % FL)) - 1u));
            };
            uint32_t lo = 0u, hi = 0u;
            if (first < last && first < seg + SEG && last > seg) { lo = records_before(max(first, seg)); hi = records_before(min(last, seg + SEG)); }
            PBWD_CLK(t_list);
            PBWD_ADD(2, t_list - t_seg);
            PBWD_ADD(7, total);
            for (uint32_t p0 = 0; p0 < total; p0 += CAP) {
                PBWD_CLK(t_pass);
                PBWD_ADD(6, 1);
                const uint32_t n = min(CAP, total - p0);
                // ---- 2. the pass's records, densely into LDS ----
                for (uint32_t j0 = (uint32_t)lane; j0 < n; j0 += 64u * BIG_UNROLL) {
                    float2 v[BIG_UNROLL][PF / 2];
#pragma unroll
                    for (int u = 0; u < BIG_UNROLL; u++) {
                        const uint32_t j = j0 + 64u * (uint32_t)u;
                        const float2 *r = pg + (size_t)(PF / 2) * (seg + (uint32_t)s_list[p0 + min(j, n - 1u)]);
#pragma unroll
                        for (int k = 0; k < PF / 2; k++) v[u][k] = r[k];
                    }
#pragma unroll
                    for (int u = 0; u < BIG_UNROLL; u++) {
                        const uint32_t j = j0 + 64u * (uint32_t)u;
                        if (j < n) {
#pragma unroll
                            for (int k = 0; k < PF / 2; k++) s_rec[(PF / 2) * j + k] = v[u][k];
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // ---- 3. every Gaussian's share of the pass ----
                PBWD_CLK(t_gathered);
                PBWD_ADD(3, t_gathered - t_pass);
                const uint32_t a = max(lo, p0), b = min(hi, p0 + n);   // (empty when a >= b)
                const bool wide = b > a && b - a > 64u;
                auto take = [&](v2f (&S)[PF / 2], uint32_t t) {
                    const v2f *r = reinterpret_cast<const v2f *>(s_rec) + (PF / 2) * (t - p0);
#pragma unroll
                    for (int k = 0; k < PF / 2; k++) S[k] += r[k];
                };
                if (!wide) {
                    // TAKE records requested before the first is added (a lane's run is a chain of LDS round trips otherwise: the
                    // lane with the longest run of the pass -- tens of records where a near surface fills its tiles -- sets the
                    // wave's time); the additions are the same ones in the same order
                    constexpr int TAKE = LVDGS_PBWD_TAKE;
                    for (uint32_t t = a; t < b; t += TAKE) {
                        v2f v[TAKE][PF / 2];
#pragma unroll
                        for (int u = 0; u < TAKE; u++) {
                            const v2f *r = reinterpret_cast<const v2f *>(s_rec) + (t + u < b ? (PF / 2) * (t + u - p0) : ZERO_AT);
#pragma unroll
                            for (int k = 0; k < PF / 2; k++) v[u][k] = r[k];
                        }
#pragma unroll
                        for (int u = 0; u < TAKE; u++) {
#pragma unroll
                            for (int k = 0; k < PF / 2; k++) A2[k] += v[u][k];
                        }
                    }
                }
#ifdef LVDGS_DIAG_PBWD
                {
                    int trips = (!wide && b > a) ? (int)((b - a + LVDGS_PBWD_TAKE - 1) / LVDGS_PBWD_TAKE) : 0;
                    for (int off = 32; off; off >>= 1) trips = max(trips, __shfl_xor(trips, off, 64));
                    PBWD_ADD(8, trips);
                    PBWD_ADD(9, __popcll(__ballot(wide)));
                }
#endif
                PBWD_CLK(t_wide);
                for (uint64_t todo = __ballot(wide); todo; todo &= todo - 1) {
                    const int src = __builtin_ctzll(todo);
                    const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane((int)a, src), wb = (uint32_t)__builtin_amdgcn_readlane((int)b, src);   // (src is wave-uniform)
                    v2f S2[PF / 2];
#pragma unroll
                    for (int k = 0; k < PF / 2; k++) S2[k] = v2f{0.f, 0.f};
                    for (uint32_t t = wa + (uint32_t)lane; t < wb; t += 64u) take(S2, t);
                    float S[10];
                    unpack_sums<POSE_ONLY>(S2, S);
                    // the 64 partial sums of every value, folded in a fixed order: halves of the wave, pairs of rows, then inside the rows
                    float b0 = fold16(fold32(S[0], S[1]), fold32(S[2], S[3]));   // rows: S0 S2 S1 S3
                    float b1 = fold16(fold32(S[4], S[5]), fold32(S[6], S[7]));   // rows: S4 S6 S5 S7
                    const float e89 = fold32(S[8], S[9]);
                    float b2 = fold16(e89, e89);                                 // rows: S8 S8 S9 S9
                    row_sums3(b0, b1, b2);                                       // lane 15 of a row: the row's total
                    auto at_lane = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
                    const float tot[10] = {at_lane(b0, 15), at_lane(b0, 47), at_lane(b0, 31), at_lane(b0, 63), at_lane(b1, 15), at_lane(b1, 47),
                                           at_lane(b1, 31), at_lane(b1, 63), at_lane(b2, 15), at_lane(b2, 47)};
                    if (lane == src) {   // (tot[5..8] are zero in the pose-only form)
                        A2[0] += v2f{tot[0], tot[1]}; A2[1] += v2f{tot[2], tot[3]};
                        if constexpr (POSE_ONLY) A2[2] += v2f{tot[4], tot[9]};
                        else { A2[2] += v2f{tot[4], tot[5]}; A2[3] += v2f{tot[6], tot[7]}; A2[4] += v2f{tot[8], tot[9]}; }
                    }
                }
                PBWD_CLK(t_wide_done);
                PBWD_ADD(10, t_wide_done - t_wide);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                PBWD_CLK(t_summed);
                PBWD_ADD(4, t_summed - t_gathered);
            }
        }
        PBWD_CLK(t_end);
        PBWD_ADD(1, t_end - t_begin);
#ifdef LVDGS_DIAG_PBWD
        {
            const int w = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);   // (helper waves have rows of their own)
            if (lane == 0 && w < PBWD_DIAG_WAVES)
                for (int k = 0; k < PBWD_DIAG_VALUES; k++) g_pbwd_diag[PBWD_DIAG_VALUES * w + k] += diag[k];
        }
#endif
    }
}

// POSE_ONLY (LVDGS_FLAG_POSE_ONLY): the pose gradient alone -- six-float pair records (d/d 2-D mean, conic, view depth), no
// opacity / colour reads, no parameter-gradient stores, no scale / quaternion chain.  The statements that make dL/dtau are the
// same ones in the same order: the partial sums are bit for bit those of the full form.
// IN_REGS (preprocess_bwd_views_kernel: the per-Gaussian passes of several views of a mapping window in one launch): the parameter
// gradients are not added to memory view after view -- 56 bytes read and 56 written per visible Gaussian and view -- but kept in the
// thread's registers over the views (acc) and written once by the caller.  The additions are the ones the view-after-view launches
// make, in the same order: `assign` (the launch's first view when its gradients are not added to what the buffers hold) assigns.
struct GradAcc { float opac, m3[3], sc[3], rot[4], sh[3]; bool touched; };
// HELPERS (preprocess_bwd_helpers_kernel: frames whose Gaussians have large footprints, LVDGS_FLAG_SUPER_TILES): workgroups of EIGHT
// waves -- wave 4 + w owns nothing and sweeps the second part of wave w's region of pair records while wave w sweeps the first.  The sums
// of a large-footprint wave are DEFINED in two parts, (records in front of the split) + (records behind it), the split a function of
// the region alone: without helpers the wave sweeps the parts one after the other, so both forms add the same numbers in the same order.
template <bool POSE_ONLY, bool IN_REGS = false, bool HELPERS = false>
__device__ __forceinline__ void preprocess_bwd_body(const BwdParams &p, GradAcc *acc = nullptr, bool assign = false) {
    static_assert(!(POSE_ONLY && IN_REGS), "the pose-only pass has no parameter gradients to keep");
    constexpr int AREAS = HELPERS ? 8 : 4;   // LDS staging areas: one per wave
    const bool helper = HELPERS && threadIdx.x >= 256;
    constexpr int PF = POSE_ONLY ? PAIR_FLOATS_POSE : PAIR_FLOATS;   // floats per pair record
    __shared__ float s_tau[4][6];
    if (p.pair_total && *p.pair_total > p.pair_capacity) return;   // (uniform over the launch)
    const int i = blockIdx.x * 256 + (threadIdx.x & 255);   // (a helper thread: its owner's Gaussian)
    const Cam &c = p.cam;
    // The camera's matrices, read once into scalar registers (the compiler reads them with vector loads where they are
    // used -- the pointers are not known to be invariant -- and such a load's first use would end the overlap below).
    float V[16];
    {
        auto uniform = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
#pragma unroll
        for (int k = 0; k < 16; k++) V[k] = uniform(c.view[k]);
    }
    const float *Vg = c.view, *PMg = c.proj, *PRg = c.proj_raw;   // (after the sums: read where they are used, as before)
    float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // One round of ordinary loads, all of them waited for before the records are requested: the compiler waits for
    // EVERYTHING outstanding (vmcnt(0)) at the first use of an ordinary load's result while LDS-DMA loads are in flight,
    // so nothing loaded the ordinary way may be used for the first time between the request and the sums.  (Parameters of
    // Gaussians that turn out invisible are read for nothing: 44 bytes each.)
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, area = threadIdx.x >> 6;
    const bool in_map = i < p.N;
    int32_t radius_i = 0;
    uint32_t slot_i = 0u, tiles_i = 0u;
    float pos[3] = {0.f, 0.f, 0.f}, opac_raw = 0.f, c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sc[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f};
    if (in_map) {
        radius_i = p.radii[i]; slot_i = p.slot_base[i]; tiles_i = p.tiles_touched[i];
    }
    if (in_map && !helper) {
        pos[0] = p.means3D[3 * i]; pos[1] = p.means3D[3 * i + 1]; pos[2] = p.means3D[3 * i + 2];
        if constexpr (!POSE_ONLY) opac_raw = p.opacities[i];
        if (p.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = p.cov3D_precomp[6 * (size_t)i + k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) sc[k] = p.scales[3 * (size_t)i + k];
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = p.rotations[4 * (size_t)i + k];
        }
    }
    loads_complete_here(radius_i, slot_i, tiles_i, pos, opac_raw, sc, q, c6);
    // (a visible Gaussian without a listed pair -- none of its tiles in the band being rendered, or every tile ruled out by the
    // reach test -- has all-zero sums, and every output is linear in them: zeros are written and the arithmetic left out)
    const bool has_run = in_map && radius_i > 0 && tiles_i > 0u;   // (its records: slots slot_i ... slot_i + tiles_i - 1)
    const bool live = has_run && !helper;
    // gradients w.r.t. the parameters: written, or (a later view of a mapping iteration) added to what is there
    const bool accumulate = p.accumulate != 0;
    // (r: the value's place in the registers of an IN_REGS pass)
    auto put = [=](float *dst, float v, float *r) {
        if constexpr (IN_REGS) *r = assign ? v : *r + v;
        else *dst = accumulate ? *dst + v : v;
    };
    // a Gaussian's three / four values as ONE 12- / 16-byte access per lane: the wave's stores are whole runs of memory
    // instead of three or four passes of every-third-word stores over the same sectors
    struct f3 { float x, y, z; };
    struct f4 { float x, y, z, w; };
    auto put3 = [=](float *dst, float a, float b, float c, float *r) {
        if constexpr (IN_REGS) {
            r[0] = assign ? a : r[0] + a; r[1] = assign ? b : r[1] + b; r[2] = assign ? c : r[2] + c;
        } else {
            f3 *d = reinterpret_cast<f3 *>(dst);
            if (accumulate) { const f3 o = *d; a += o.x; b += o.y; c += o.z; }
            *d = f3{a, b, c};
        }
    };
    auto put4 = [=](float *dst, float a, float b, float c, float e, float *r) {
        if constexpr (IN_REGS) {
            r[0] = assign ? a : r[0] + a; r[1] = assign ? b : r[1] + b; r[2] = assign ? c : r[2] + c; r[3] = assign ? e : r[3] + e;
        } else {
            f4 *d = reinterpret_cast<f4 *>(dst);
            if (accumulate) { const f4 o = *d; a += o.x; b += o.y; c += o.z; e += o.w; }
            *d = f4{a, b, c, e};
        }
    };
    if constexpr (IN_REGS) { if (live) acc->touched = true; }
    if (!POSE_ONLY && i < p.N && !live && !helper) {
#pragma unroll
        for (int k = 0; k < 3; k++) p.dmeans2D[3 * (size_t)i + k] = 0.f;
        if (!IN_REGS && !accumulate) {   // (adding zero: nothing to do; IN_REGS: the caller writes what the registers hold)
#pragma unroll
            for (int k = 0; k < 3; k++) p.dmeans3D[3 * (size_t)i + k] = 0.f;
            p.dopac[i] = 0.f;
            if (p.dscales) { for (int k = 0; k < 3; k++) p.dscales[3 * (size_t)i + k] = 0.f; }
            if (p.drot) { for (int k = 0; k < 4; k++) p.drot[4 * (size_t)i + k] = 0.f; }
            if (p.dcov3D) { for (int k = 0; k < 6; k++) p.dcov3D[6 * (size_t)i + k] = 0.f; }
            if (p.dcolors) { for (int k = 0; k < 3; k++) p.dcolors[3 * (size_t)i + k] = 0.f; }
            if (p.dshs) { for (int k = 0; k < 3 * c.M; k++) p.dshs[(size_t)i * 3 * c.M + k] = 0.f; }
        }
    }
    // ---- sum every Gaussian's per-tile partial gradients (a contiguous run of 40-byte records, fixed order) ----
    // Records exist where blend_bwd wrote them (pair_valid): pairs behind their tile's last contributor have none.
    //
    // The runs of a wave's 64 Gaussians follow each other in memory (slots are in id order), so the wave streams that
    // region through its own piece of LDS with coalesced 16-byte loads and every lane then picks its own records out of
    // it -- instead of 64 lanes walking 64 different runs with one gather each per step, which moved 2.2x the bytes
    // (r01 / r02_a counters).  The loads of the first chunk (at config 3 the only one for most waves) are issued BEFORE
    // the part of the per-Gaussian arithmetic that does not depend on the sums -- projection, covariance, the EWA matrices,
    // the rotation matrix -- and land while it runs: with the workgroup-wide staging of before (two workgroup barriers per
    // chunk) the kernel was the sum of a memory phase and an arithmetic phase, every resident workgroup in the same one
    // (ablation builds: 26.7 us without the arithmetic, 28.9 without the sums, 49.5 together).
    __shared__ float4 s_pg4[AREAS][STAGE_BYTES<PF> / 16];   // (the last load instruction of a chunk is masked to the lanes inside it)
    __shared__ uint32_t s_valid4[4][WAVE_CHUNK / 4];
    float A[10];
#pragma unroll
    for (int k = 0; k < 10; k++) A[k] = 0.f;
    const uint32_t first = has_run ? slot_i : 0u, npairs = has_run ? tiles_i : 0u, last = first + npairs;
    const bool big = npairs > BIG_RUN;
    const bool stream = __ballot(big) == 0ull;   // (wave-uniform)
    // the wave's region: from the first slot of its first Gaussian to the end of its last one's (slot_base is the running
    // sum of tiles_touched over ALL Gaussians, visible or not)
    uint32_t r_lo = 0u, r_hi = 0u;
    if (stream && !helper && i - lane < p.N) {
        const int last_lane = min(63, p.N - 1 - (i - lane));
        r_lo = (uint32_t)__shfl((int)slot_i, 0, 64) & ~3u;   // a chunk starts at a multiple of 4 records: on a 16-byte boundary of the records and a word of flags
        r_hi = (uint32_t)__shfl((int)(slot_i + tiles_i), last_lane, 64);
    }
    const float4 *pg_all = reinterpret_cast<const float4 *>(p.pair_grads);
    const uint32_t *valid_all = reinterpret_cast<const uint32_t *>(p.pair_valid);
    constexpr int QUADS_PER_LANE = (WAVE_CHUNK * PF / 4 + 63) / 64;
    // global -> LDS directly (global_load_lds: the wave's lanes land side by side, 1 KiB per instruction; no registers held
    // while the chunk is on its way): one chunk of records and their flags
    auto request = [&](uint32_t c0) {
        const uint32_t n = min((uint32_t)WAVE_CHUNK, r_hi - c0);
        const uint32_t quads = (n * PF + 3) / 4, q0 = c0 / 4 * PF;   // c0 is a multiple of 4: record c0 starts at float4 c0 * PF / 4
#pragma unroll
        for (int u = 0; u < QUADS_PER_LANE; u++) {
            const uint32_t k = (uint32_t)lane + 64u * (uint32_t)u;
            if (k < quads) __builtin_amdgcn_global_load_lds(pg_all + (size_t)q0 + k, &s_pg4[wave][64 * u], 16, 0, 0);
        }
        if ((uint32_t)lane * 4u < n) __builtin_amdgcn_global_load_lds(valid_all + c0 / 4 + (uint32_t)lane, &s_valid4[wave][0], 4, 0, 0);
    };
    auto consume = [&](uint32_t c0) {   // LDS -> every lane's own records, in slot order
        const uint32_t n = min((uint32_t)WAVE_CHUNK, r_hi - c0);
        wait_for_vector_memory();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float2 *s_pg = reinterpret_cast<const float2 *>(s_pg4[wave]);
        const uint8_t *s_valid = reinterpret_cast<const uint8_t *>(s_valid4[wave]);
        const uint32_t lo = max(first, c0), hi = min(last, c0 + n);
        for (uint32_t t = lo; t < hi; t++) {
            if (!s_valid[t - c0]) continue;   // (a record blend_bwd did not write: a pair behind its tile's last contributor)
            const float2 *r = s_pg + (PF / 2) * (t - c0);
            const float2 a0 = r[0], a1 = r[1], a2 = r[2];
            A[0] += a0.x; A[1] += a0.y; A[2] += a1.x; A[3] += a1.y; A[4] += a2.x;
            if constexpr (POSE_ONLY) A[9] += a2.y;
            else {
                const float2 a3 = r[3], a4 = r[4];
                A[5] += a2.y; A[6] += a3.x; A[7] += a3.y; A[8] += a4.x; A[9] += a4.y;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
#if LVDGS_PBWD_ABLATE != 1
    if (r_lo < r_hi) request(r_lo);
#endif

    // ---- the part of the per-Gaussian arithmetic that needs no sums (the first chunk is on its way) ----
    float pv[3] = {0.f, 0.f, 1.f}, qnorm = 1.f;
    Ewa e{};
    float Q00 = 0.f, Q01 = 0.f, Q11 = 0.f;
#if LVDGS_PBWD_ABLATE != 2
    if (live) {
        xform3(pos, V, pv);
        if (!p.cov3D_precomp) {
            activate_scale_rot(p.act, sc, q, qnorm);
            cov3d_of(sc, c.scale_mod, q, c6);
        }
        ewa_setup(pv, V, c, e);
        float ca, cb, cc;
        cov2d_of(e, c6, ca, cb, cc);
        const float det = ca * cc - cb * cb, di = 1.f / det;
        Q00 = cc * di; Q01 = -cb * di; Q11 = ca * di;
    }
#endif

#if LVDGS_PBWD_ABLATE != 1
    if (stream) {
        for (uint32_t c0 = r_lo; c0 < r_hi; c0 += WAVE_CHUNK) {
            if (c0 != r_lo) request(c0);
            consume(c0);
        }
    } else {
        // A wave that holds large-footprint Gaussians: sum_region_compacted (above)
        uint32_t w_first = 0u, w_hi = 0u;   // the wave's region
        if (i - lane < p.N) {
            const int last_lane = min(63, p.N - 1 - (i - lane));
            w_first = (uint32_t)__shfl((int)slot_i, 0, 64);
            w_hi = (uint32_t)__shfl((int)(slot_i + tiles_i), last_lane, 64);
        }
        // the two parts of the region (the template's comment): the split half-way, at a whole number of sweep segments from the start
        const uint32_t split = min(w_hi, w_first + ((w_hi - w_first) / 2u + (uint32_t)LVDGS_PBWD_SEG - 1u) / (uint32_t)LVDGS_PBWD_SEG * (uint32_t)LVDGS_PBWD_SEG);
        char *const stage = reinterpret_cast<char *>(s_pg4[area]);
        v2f A2[PF / 2];
#pragma unroll
        for (int k = 0; k < PF / 2; k++) A2[k] = v2f{0.f, 0.f};
        if constexpr (HELPERS) {
            sum_region_compacted<POSE_ONLY>(p.pair_grads, p.pair_valid, stage, first, last, helper ? split : w_first, helper ? w_hi : split, A2);
            unpack_sums<POSE_ONLY>(A2, A);
            if (helper) {   // the second part's sums, to the owner: lane l's ten values side by side in this wave's staging area
                float *out = reinterpret_cast<float *>(stage) + 10 * lane;
#pragma unroll
                for (int k = 0; k < 10; k++) out[k] = A[k];
            }
        } else {
            // one part after the other.  A run lies in one part -- the other part's sum is +0, and x + 0 is x (no sum is ever -0: they
            // start at +0) -- except the ONE run that holds the split: that lane's first-part sums wait in scalar registers while its
            // registers start the second part at zero, and are added in front afterwards.
            sum_region_compacted<POSE_ONLY>(p.pair_grads, p.pair_valid, stage, first, last, w_first, split, A2);
            const uint64_t across = __ballot(first < split && last > split);
            const int owner_lane = across ? __builtin_ctzll(across) : 0;
            float kept[PF];
#pragma unroll
            for (int k = 0; k < PF / 2; k++) {
                kept[2 * k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A2[k].x), owner_lane));
                kept[2 * k + 1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A2[k].y), owner_lane));
                if (across && lane == owner_lane) A2[k] = v2f{0.f, 0.f};
            }
            sum_region_compacted<POSE_ONLY>(p.pair_grads, p.pair_valid, stage, first, last, split, w_hi, A2);
#pragma unroll
            for (int k = 0; k < PF / 2; k++)
                if (across && lane == owner_lane) A2[k] = v2f{kept[2 * k], kept[2 * k + 1]} + A2[k];
            unpack_sums<POSE_ONLY>(A2, A);
        }
    }
    if constexpr (HELPERS) {
        __syncthreads();   // (every wave of the workgroup, whichever path it took)
        if (!stream && !helper) {
            const float *in = reinterpret_cast<const float *>(s_pg4[area + 4]) + 10 * lane;
#pragma unroll
            for (int k = 0; k < 10; k++) A[k] = A[k] + in[k];
        }
    }
#endif
#if LVDGS_PBWD_ABLATE == 2
    if (live) { p.dmeans2D[3 * (size_t)i] = ((A[0] + A[1]) + (A[2] + A[3])) + ((A[4] + A[5]) + (A[6] + A[7])) + (A[8] + A[9]); }
    if (false) {
#else
    if (live) {
#endif
        // A: [0,1] d/d pixel mean, [2..4] d/d conic a,b,c, [5] d/d opacity, [6..8] d/d rgb, [9] d/d view depth
        if constexpr (!POSE_ONLY) {
            // d/d(logit) = d/d(opacity) * o (1 - o) when the sigmoid is fused (o re-evaluated with the forward's expression: the
            // record holds it too, but reading 4 bytes of a 64-byte record per Gaussian moved 20 MB for 2)
            float o = opac_raw;
            if (p.act & ACT_SIGMOID_OPACITY) { o = 1.f / (1.f + expf(-o)); put(&p.dopac[i], A[5] * o * (1.f - o), IN_REGS ? &acc->opac : nullptr); }
            else put(&p.dopac[i], A[5], IN_REGS ? &acc->opac : nullptr);
        }
        const float g_ndc[2] = {A[0] * 0.5f * (float)c.W, A[1] * 0.5f * (float)c.H};
        if constexpr (!POSE_ONLY) *reinterpret_cast<f3 *>(&p.dmeans2D[3 * (size_t)i]) = f3{g_ndc[0], g_ndc[1], 0.f};

        float g_rgb[3] = {A[6], A[7], A[8]};
        float g_pview[3] = {0.f, 0.f, A[9]};
        float g_world[3] = {0.f, 0.f, 0.f};

        // ---- colour ---- (POSE_ONLY: colours without view dependence only -- api.hip -- which give the pose nothing)
        if constexpr (POSE_ONLY) {
        } else if (p.colors_precomp) {
            put3(&p.dcolors[3 * (size_t)i], g_rgb[0], g_rgb[1], g_rgb[2], nullptr);   // (IN_REGS: SH colours of one coefficient only, launch_preprocess_bwd_views)
        } else {
            float d[3] = {pos[0] - c.campos[0], pos[1] - c.campos[1], pos[2] - c.campos[2]};
            const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const float u[3] = {d[0] / len, d[1] / len, d[2] / len};
            float B[16], G[16][3];
            sh_basis(c.sh_degree, u, B);
            sh_basis_grad(c.sh_degree, u, G);
            const int nb = (c.sh_degree + 1) * (c.sh_degree + 1);
            const float *sh = p.shs + (size_t)i * c.M * 3;
            float *dsh = p.dshs + (size_t)i * c.M * 3;
            // recompute the clamp mask
            float val[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < nb) { val[0] += B[k] * sh[3 * k]; val[1] += B[k] * sh[3 * k + 1]; val[2] += B[k] * sh[3 * k + 2]; }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) if (val[ch] + 0.5f < 0.f) g_rgb[ch] = 0.f;
            float g_u[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < nb) {
                    put3(&dsh[3 * k], B[k] * g_rgb[0], B[k] * g_rgb[1], B[k] * g_rgb[2], IN_REGS ? acc->sh : nullptr);   // (IN_REGS: k = 0 is the only one)
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const float sg = sh[3 * k + ch] * g_rgb[ch];
                        g_u[0] += G[k][0] * sg; g_u[1] += G[k][1] * sg; g_u[2] += G[k][2] * sg;
                    }
                }
            if (!IN_REGS && !accumulate)
                for (int k = nb; k < c.M; k++) { dsh[3 * k] = 0.f; dsh[3 * k + 1] = 0.f; dsh[3 * k + 2] = 0.f; }
            // The view direction d = mean - camera centre moves with the mean and with the camera: C = -R^T T, and under
            // T_w2c <- Exp(tau) T_w2c dC/drho = -R^T, dC/dtheta = 0 at tau = 0, so dL/drho += R g_d (oracle: same statement,
            // pinned against the dense autograd formulation in float64).  Zero at SH degree 0.
            const float dot = u[0] * g_u[0] + u[1] * g_u[1] + u[2] * g_u[2];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float g_d = (g_u[a] - u[a] * dot) / len;
                g_world[a] += g_d;
#pragma unroll
                for (int j = 0; j < 3; j++) tau[j] += Vg[4 * a + j] * g_d;
            }
        }

        // ---- conic -> cov2D -> (cov3D, T) ----
        const float G00 = A[2], G01 = 0.5f * A[3], G11 = A[4];
        const float QG00 = Q00 * G00 + Q01 * G01, QG01 = Q00 * G01 + Q01 * G11;
        const float QG10 = Q01 * G00 + Q11 * G01, QG11 = Q01 * G01 + Q11 * G11;
        const float Gs[2][2] = {{-(QG00 * Q00 + QG01 * Q01), -(QG00 * Q01 + QG01 * Q11)},
                                {-(QG00 * Q01 + QG01 * Q11), -(QG10 * Q01 + QG11 * Q11)}};
        float Sg[3][3], g_S[3][3];
        sym6(c6, Sg);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int s = 0; s < 2; s++) v += e.T[r][a] * Gs[r][s] * e.T[s][b];
                g_S[a][b] = v;
            }
        if (!POSE_ONLY && p.dcov3D) {
            float *o = p.dcov3D + 6 * (size_t)i;
            put(o + 0, g_S[0][0], nullptr); put(o + 1, 2.f * g_S[0][1], nullptr); put(o + 2, 2.f * g_S[0][2], nullptr);   // (never IN_REGS)
            put(o + 3, g_S[1][1], nullptr); put(o + 4, 2.f * g_S[1][2], nullptr); put(o + 5, g_S[2][2], nullptr);
        }
        float TS[2][3], g_T[2][3];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int b = 0; b < 3; b++) TS[r][b] = e.T[r][0] * Sg[0][b] + e.T[r][1] * Sg[1][b] + e.T[r][2] * Sg[2][b];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int b = 0; b < 3; b++) g_T[r][b] = 2.f * (Gs[r][0] * TS[0][b] + Gs[r][1] * TS[1][b]);
        float g_J00 = 0.f, g_J02 = 0.f, g_J11 = 0.f, g_J12 = 0.f;
        const float tz = e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float j00 = c.fx / tz, j02 = -(c.fx * e.t[0]) / tz2, j11 = c.fy / tz, j12 = -(c.fy * e.t[1]) / tz2;
        float g_W[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float w0 = Vg[4 * k + 0], w1 = Vg[4 * k + 1], w2 = Vg[4 * k + 2];
            g_J00 += g_T[0][k] * w0; g_J02 += g_T[0][k] * w2;
            g_J11 += g_T[1][k] * w1; g_J12 += g_T[1][k] * w2;
            g_W[0][k] = j00 * g_T[0][k];
            g_W[1][k] = j11 * g_T[1][k];
            g_W[2][k] = j02 * g_T[0][k] + j12 * g_T[1][k];
        }
        g_pview[0] += e.clx ? 0.f : -(c.fx / tz2) * g_J02;
        g_pview[1] += e.cly ? 0.f : -(c.fy / tz2) * g_J12;
        g_pview[2] += -(c.fx / tz2) * g_J00 - (c.fy / tz2) * g_J11 + (2.f * c.fx * e.t[0] / tz3) * g_J02 +
                      (2.f * c.fy * e.t[1] / tz3) * g_J12;

        // ---- projected mean through the full projection (world) and the raw projection (pose) ----
        float ph[3];
        xform3(pos, PMg, ph);
        const float phw = xform_w(pos, PMg);
        const float pw = 1.f / (phw + HOMOG_EPS);
        const float gh0 = g_ndc[0] * pw, gh1 = g_ndc[1] * pw, gh3 = -(g_ndc[0] * ph[0] + g_ndc[1] * ph[1]) * pw * pw;
        float g_pview_proj[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            g_world[a] += PMg[4 * a + 0] * gh0 + PMg[4 * a + 1] * gh1 + PMg[4 * a + 3] * gh3;
            g_pview_proj[a] = PRg[4 * a + 0] * gh0 + PRg[4 * a + 1] * gh1 + PRg[4 * a + 3] * gh3;
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            g_world[a] += Vg[4 * a + 0] * g_pview[0] + Vg[4 * a + 1] * g_pview[1] + Vg[4 * a + 2] * g_pview[2];
        }
        if constexpr (!POSE_ONLY) put3(&p.dmeans3D[3 * (size_t)i], g_world[0], g_world[1], g_world[2], IN_REGS ? acc->m3 : nullptr);

        // ---- camera pose: T' = Exp(tau) T ----
        const float gv[3] = {g_pview[0] + g_pview_proj[0], g_pview[1] + g_pview_proj[1], g_pview[2] + g_pview_proj[2]};
        tau[0] += gv[0]; tau[1] += gv[1]; tau[2] += gv[2];
        tau[3] = pv[1] * gv[2] - pv[2] * gv[1];
        tau[4] = pv[2] * gv[0] - pv[0] * gv[2];
        tau[5] = pv[0] * gv[1] - pv[1] * gv[0];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float w0 = Vg[4 * k + 0], w1 = Vg[4 * k + 1], w2 = Vg[4 * k + 2];
            tau[3] += w1 * g_W[2][k] - w2 * g_W[1][k];
            tau[4] += w2 * g_W[0][k] - w0 * g_W[2][k];
            tau[5] += w0 * g_W[1][k] - w1 * g_W[0][k];
        }

        // ---- Sigma3 -> scale, quaternion ----
        if (!POSE_ONLY && !p.cov3D_precomp) {
            float R[3][3];
            quat_rot(q, R);
            const float sm[3] = {c.scale_mod * sc[0], c.scale_mod * sc[1], c.scale_mod * sc[2]};
            float g_R[3][3], g_sc[3];
#pragma unroll
            for (int b = 0; b < 3; b++) {
                float v = 0.f;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const float gm = 2.f * (g_S[a][0] * R[0][b] * sm[b] + g_S[a][1] * R[1][b] * sm[b] + g_S[a][2] * R[2][b] * sm[b]);
                    v += gm * R[a][b];
                    g_R[a][b] = gm * sm[b];
                }
                // fused exp: d/d(log s) = d/ds * s
                g_sc[b] = v * c.scale_mod * ((p.act & ACT_EXP_SCALES) ? sc[b] : 1.f);
            }
            put3(&p.dscales[3 * (size_t)i], g_sc[0], g_sc[1], g_sc[2], IN_REGS ? acc->sc : nullptr);
            const float r = q[0], x = q[1], y = q[2], z = q[3];
            float dq[4];
            dq[0] = 2.f * (-z * g_R[0][1] + y * g_R[0][2] + z * g_R[1][0] - x * g_R[1][2] - y * g_R[2][0] + x * g_R[2][1]);
            dq[1] = 2.f * (y * g_R[0][1] + z * g_R[0][2] + y * g_R[1][0] - 2.f * x * g_R[1][1] - r * g_R[1][2] + z * g_R[2][0] + r * g_R[2][1] - 2.f * x * g_R[2][2]);
            dq[2] = 2.f * (-2.f * y * g_R[0][0] + x * g_R[0][1] + r * g_R[0][2] + x * g_R[1][0] + z * g_R[1][2] - r * g_R[2][0] + z * g_R[2][1] - 2.f * y * g_R[2][2]);
            dq[3] = 2.f * (-2.f * z * g_R[0][0] - r * g_R[0][1] + x * g_R[0][2] + r * g_R[1][0] - 2.f * z * g_R[1][1] + y * g_R[1][2] + x * g_R[2][0] + y * g_R[2][1]);
            if (p.act & ACT_NORMALIZE_ROT) {
                // q = raw / |raw|: d/d raw = (g - q (q . g)) / |raw|
                const float dot = q[0] * dq[0] + q[1] * dq[1] + q[2] * dq[2] + q[3] * dq[3];
#pragma unroll
                for (int k = 0; k < 4; k++) dq[k] = (dq[k] - q[k] * dot) / qnorm;
            }
            put4(&p.drot[4 * (size_t)i], dq[0], dq[1], dq[2], dq[3], IN_REGS ? acc->rot : nullptr);
        }
    }
    // ---- workgroup sum of the pose gradient -> one partial per workgroup (no atomics) ----
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const float s = wave_sum_to_lane63(tau[k]);
        if (lane == 63 && !helper) s_tau[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6)
        p.tau_part[(size_t)blockIdx.x * 6 + threadIdx.x] =
            ((s_tau[0][threadIdx.x] + s_tau[1][threadIdx.x]) + s_tau[2][threadIdx.x]) + s_tau[3][threadIdx.x];
}

// (two plain kernels around the body: a kernel TEMPLATE with this body loses its host stub -- "substitution failure" -- on this
// toolchain)
__global__ void __launch_bounds__(256, LVDGS_PBWD_WGS) preprocess_bwd_kernel(BwdParams p) { preprocess_bwd_body<false>(p); }
#ifndef LVDGS_PBWD_WGS_POSE
#define LVDGS_PBWD_WGS_POSE 5   // (6: 80 VGPRs with 7 spilled; same box 25.1 against 24.1 us at config 3)
#endif
__global__ void __launch_bounds__(256, LVDGS_PBWD_WGS_POSE) preprocess_bwd_pose_kernel(BwdParams p) { preprocess_bwd_body<true>(p); }
// ... with helper waves (512 threads: the body's comment)
__global__ void __launch_bounds__(512, 2) preprocess_bwd_helpers_kernel(BwdParams p) { preprocess_bwd_body<false, false, true>(p); }
__global__ void __launch_bounds__(512, 2) preprocess_bwd_pose_helpers_kernel(BwdParams p) { preprocess_bwd_body<true, false, true>(p); }

// The per-Gaussian passes of up to PBWD_VIEWS views of one map in ONE launch (lvdgs_gaussian_backward_batch: the views of a mapping
// window behind their batched blend pass).  A thread walks its Gaussian through the views in order with the parameter gradients in
// registers (GradAcc above) and writes them once: the view-after-view launches read and wrote 56 bytes per visible Gaussian and view.
// Every view keeps what is its own: its records, its dL/d(2-D mean) array, its pose-gradient partials.
constexpr int PBWD_VIEWS = 12;
struct BwdView {
    Cam cam; const int32_t *radii; const float *rec; const uint32_t *tiles_touched, *slot_base; const float *pair_grads; const uint8_t *pair_valid;
    float *dmeans2D, *tau_part;
};
struct BwdViews { BwdParams common; int n; BwdView v[PBWD_VIEWS]; };
static_assert(sizeof(BwdViews) <= 4000, "kernel arguments");
#ifndef LVDGS_PBWD_VIEWS_WGS
#define LVDGS_PBWD_VIEWS_WGS 4
#endif
template <bool HELPERS>
__device__ __forceinline__ void preprocess_bwd_views_body(const BwdViews &b) {
    const bool owner = !HELPERS || threadIdx.x < 256;   // (helper waves: the body's comment; they own no Gaussian and hold no gradients)
    const int i = owner ? (int)blockIdx.x * 256 + (int)threadIdx.x : b.common.N;
    const bool add_to_memory = b.common.accumulate != 0;   // (the launch's sums are added to what the buffers hold)
    GradAcc acc{};
    if (add_to_memory && i < b.common.N) {
        acc.opac = b.common.dopac[i];
#pragma unroll
        for (int k = 0; k < 3; k++) { acc.m3[k] = b.common.dmeans3D[3 * (size_t)i + k]; acc.sc[k] = b.common.dscales[3 * (size_t)i + k]; acc.sh[k] = b.common.dshs[3 * (size_t)i + k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) acc.rot[k] = b.common.drot[4 * (size_t)i + k];
    }
    for (int k = 0; k < b.n; k++) {
        BwdParams p = b.common;
        const BwdView &v = b.v[k];
        p.cam = v.cam; p.radii = v.radii; p.rec = v.rec; p.tiles_touched = v.tiles_touched; p.slot_base = v.slot_base;
        p.pair_grads = v.pair_grads; p.pair_valid = v.pair_valid; p.dmeans2D = v.dmeans2D; p.tau_part = v.tau_part;
        preprocess_bwd_body<false, true, HELPERS>(p, &acc, k == 0 && !add_to_memory);
        __syncthreads();   // (the next view's pass uses the workgroup's LDS again)
    }
    if (i < b.common.N && (acc.touched || !add_to_memory)) {   // (a Gaussian no view of the launch saw: zeros, or what was there)
        struct f3 { float x, y, z; };
        struct f4 { float x, y, z, w; };
        b.common.dopac[i] = acc.opac;
        *reinterpret_cast<f3 *>(&b.common.dmeans3D[3 * (size_t)i]) = f3{acc.m3[0], acc.m3[1], acc.m3[2]};
        *reinterpret_cast<f3 *>(&b.common.dscales[3 * (size_t)i]) = f3{acc.sc[0], acc.sc[1], acc.sc[2]};
        *reinterpret_cast<f3 *>(&b.common.dshs[3 * (size_t)i]) = f3{acc.sh[0], acc.sh[1], acc.sh[2]};
        *reinterpret_cast<f4 *>(&b.common.drot[4 * (size_t)i]) = f4{acc.rot[0], acc.rot[1], acc.rot[2], acc.rot[3]};
    }
}
__global__ void __launch_bounds__(256, LVDGS_PBWD_VIEWS_WGS) preprocess_bwd_views_kernel(BwdViews b) { preprocess_bwd_views_body<false>(b); }
__global__ void __launch_bounds__(512, 4) preprocess_bwd_views_helpers_kernel(BwdViews b) { preprocess_bwd_views_body<true>(b); }

// fixed-order reduction of the per-workgroup pose partials (strided per-thread sums, then a wave fold and a four-term sum:
// two barriers fewer than an LDS tree, the order of the additions fixed by the code either way)
__global__ void __launch_bounds__(256) tau_reduce_kernel(const float *part, int nblk, float *out) {
    __shared__ float s[4][6];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
        for (int k = 0; k < 6; k++) acc[k] += part[(size_t)b * 6 + k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const float v = wave_sum_to_lane63(acc[k]);
        if (lane == 63) s[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) out[threadIdx.x] = ((s[0][threadIdx.x] + s[1][threadIdx.x]) + s[2][threadIdx.x]) + s[3][threadIdx.x];
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int N, const float *means3D, const float *view, uint8_t *present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float pos[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float pv[3];
    xform3(pos, view, pv);
    present[i] = pv[2] > NEAR_CULL;
}

Cam make_cam(const lvdgs_args &a) {
    Cam c;
    c.view = a.viewmatrix; c.proj = a.projmatrix; c.proj_raw = a.projmatrix_raw; c.campos = a.campos;
    c.tanx = a.tanfovx; c.tany = a.tanfovy;
    c.W = a.image_width; c.H = a.image_height;
    c.fx = (float)c.W / (2.0f * c.tanx); c.fy = (float)c.H / (2.0f * c.tany);
    c.scale_mod = a.scale_modifier;
    c.gx = (c.W + TILE - 1) / TILE; c.gy = (c.H + TILE - 1) / TILE;
    c.sh_degree = a.sh_degree; c.M = a.sh_coeffs;
    return c;
}

}  // namespace

static FwdParams make_fwd_params(const lvdgs_args &a, const GeomView &g) {
    FwdParams p;
    p.blocksums = nullptr;
    p.super_rect = nullptr; p.super_hist = p.super_queue_counts = nullptr; p.super_gx = p.super_T = 0;
    p.cam = make_cam(a); p.N = a.num_gaussians; p.act = a.activations;
    p.tile_cull = !(a.flags & LVDGS_FLAG_LIST_ALL_TILES);
    tile_row_band(a, &p.row_begin, &p.row_end);
    p.means3D = a.means3D; p.opacities = a.opacities; p.scales = a.scales; p.rotations = a.rotations;
    p.cov3D_precomp = a.cov3D_precomp; p.shs = a.shs; p.colors_precomp = a.colors_precomp;
    p.rec = g.rec; p.tiles_touched = g.tiles_touched; p.depth_bits = g.depth_bits; p.rect = g.rect; p.radii = a.radii;
    return p;
}

int launch_preprocess_fwd(const lvdgs_args &a, const GeomView &g, uint32_t *blocksums, hipStream_t s) {
    if (a.num_gaussians == 0) return LVDGS_OK;
    FwdParams p = make_fwd_params(a, g);
    p.blocksums = blocksums;
    ProfScope ps("preprocess_fwd", s);
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(cdiv(p.N, 256)), dim3(256), 0, s, p);
    LVDGS_LAUNCH_CHECK("preprocess_fwd", a.debug, s);
    return LVDGS_OK;
}

int launch_preprocess_count(const lvdgs_args &a, const GeomView &g, const ImageView &im, const RenderScratch &w, hipStream_t s) {
    const int N = a.num_gaussians;
    if (N == 0) return LVDGS_OK;
    FwdParams p = make_fwd_params(a, g);
    const int T = p.cam.gx * p.cam.gy;
    const int nchunks = (int)group_chunks(N);
    const bool super = super_tiles_in_use(a);   // two-level grouping: the super-tile grid is counted here too (launch_super_count is for the two-call API)
    if (super) {
        p.super_rect = w.super.rect; p.super_hist = w.super.hist; p.super_queue_counts = w.super.long_count;
        p.super_gx = cdiv(p.cam.gx, SUPER); p.super_T = super_tiles_of(a.image_width, a.image_height);
    }
    const size_t lds = (size_t)(T + (super ? p.super_T : 0)) * sizeof(uint32_t);
    static unsigned char done[2 * GROUP_SHAPES][16];
    ProfScope ps("preprocess_fwd", s);
    if (int e = group_dispatch(group_shape_for(N), [&](auto threads_, auto owners_, auto per_, int d) {
            constexpr int THREADS = decltype(threads_)::value, OWNERS = decltype(owners_)::value, PER = decltype(per_)::value;
            if (super) {
                if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&preprocess_count_kernel<THREADS, OWNERS, PER, true>), (GROUP_MAX_TILES + GROUP_MAX_TILES / (SUPER * SUPER) + 64) * 4, done[GROUP_SHAPES + d])) return e;
                hipLaunchKernelGGL((preprocess_count_kernel<THREADS, OWNERS, PER, true>), dim3(nchunks), dim3(THREADS), lds, s, p, T, w.group_hist, w.chunk_sums, a.n_touched,
                                   im.long_count);
                return (int)LVDGS_OK;
            }
            if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&preprocess_count_kernel<THREADS, OWNERS, PER>), GROUP_MAX_TILES * 4, done[d])) return e;
            hipLaunchKernelGGL((preprocess_count_kernel<THREADS, OWNERS, PER>), dim3(nchunks), dim3(THREADS), lds, s, p, T, w.group_hist, w.chunk_sums, a.n_touched,
                               im.long_count);
            return (int)LVDGS_OK;
        })) return e;
    LVDGS_LAUNCH_CHECK("preprocess_count", a.debug, s);
    return LVDGS_OK;
}

// n views of one map and one image size (lvdgs_forward_batch)
int launch_preprocess_count_batch(const lvdgs_args *const *a, const GeomView *g, const ImageView *im, const RenderScratch *w, int n, hipStream_t s) {
    const int N = a[0]->num_gaussians;
    if (N == 0 || n == 0) return LVDGS_OK;
    if (n > FWD_BATCH_VIEWS) { set_error("internal: more than %d views in one forward batch", FWD_BATCH_VIEWS); return LVDGS_E_INVALID; }
    PrepCountBatch batch{};
    const bool super = super_tiles_in_use(*a[0]);   // two-level grouping: the super-tile grid is counted here too
    for (int k = 0; k < n; k++) {
        batch.v[k] = PrepCountView{make_fwd_params(*a[k], g[k]), w[k].group_hist, w[k].chunk_sums, a[k]->n_touched, im[k].long_count};
        if (super) {
            FwdParams &p = batch.v[k].p;
            p.super_rect = w[k].super.rect; p.super_hist = w[k].super.hist; p.super_queue_counts = w[k].super.long_count;
            p.super_gx = cdiv(p.cam.gx, SUPER); p.super_T = super_tiles_of(a[k]->image_width, a[k]->image_height);
        }
    }
    const int T = batch.v[0].p.cam.gx * batch.v[0].p.cam.gy;
    const int nchunks = (int)group_chunks(N);
    const size_t lds = (size_t)(T + (super ? batch.v[0].p.super_T : 0)) * sizeof(uint32_t);
    static unsigned char done[2 * GROUP_SHAPES][16];
    ProfScope ps("preprocess_fwd", s);
    if (int e = group_dispatch(group_shape_for(N), [&](auto threads_, auto owners_, auto per_, int d) {
            constexpr int THREADS = decltype(threads_)::value, OWNERS = decltype(owners_)::value, PER = decltype(per_)::value;
            if (super) {
                if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&preprocess_count_batch_kernel<THREADS, OWNERS, PER, true>), (GROUP_MAX_TILES + GROUP_MAX_TILES / (SUPER * SUPER) + 64) * 4, done[GROUP_SHAPES + d])) return e;
                hipLaunchKernelGGL((preprocess_count_batch_kernel<THREADS, OWNERS, PER, true>), dim3(nchunks, n), dim3(THREADS), lds, s, batch, T);
                return (int)LVDGS_OK;
            }
            if (int e = allow_dynamic_lds(reinterpret_cast<const void *>(&preprocess_count_batch_kernel<THREADS, OWNERS, PER>), GROUP_MAX_TILES * 4, done[d])) return e;
            hipLaunchKernelGGL((preprocess_count_batch_kernel<THREADS, OWNERS, PER>), dim3(nchunks, n), dim3(THREADS), lds, s, batch, T);
            return (int)LVDGS_OK;
        })) return e;
    LVDGS_LAUNCH_CHECK("preprocess_count (batch)", a[0]->debug, s);
    return LVDGS_OK;
}

int launch_preprocess_bwd(const lvdgs_args &a, const GeomView &g, const BwdScratch &b, const uint8_t *pair_valid, hipStream_t s,
                          const uint32_t *pair_total, uint32_t pair_capacity) {
    const int N = a.num_gaussians;
    const int nblk = cdiv(N, 256);
    if (N > 0) {
        BwdParams p;
        p.cam = make_cam(a); p.N = N; p.act = a.activations;
        p.means3D = a.means3D; p.opacities = a.opacities; p.scales = a.scales; p.rotations = a.rotations; p.cov3D_precomp = a.cov3D_precomp;
        p.shs = a.shs; p.colors_precomp = a.colors_precomp; p.radii = a.radii;
        p.rec = g.rec; p.tiles_touched = g.tiles_touched; p.slot_base = g.slot_base; p.pair_grads = b.pair_grads; p.pair_valid = pair_valid;
        p.dmeans3D = a.dL_dmeans3D; p.dmeans2D = a.dL_dmeans2D; p.dopac = a.dL_dopacities; p.dscales = a.dL_dscales;
        p.drot = a.dL_drotations; p.dcov3D = a.cov3D_precomp ? a.dL_dcov3D : nullptr; p.dshs = a.dL_dshs;
        p.dcolors = a.dL_dcolors; p.tau_part = b.tau_part; p.accumulate = (a.flags & LVDGS_FLAG_ACCUMULATE_PARAM_GRADS) ? 1 : 0;
        p.pair_total = pair_total; p.pair_capacity = pair_capacity;
        ProfScope ps("preprocess_bwd", s);
        // helper waves where the caller expects large footprints (the two-level grouping's hint): the same sums, bit for bit, sooner
        const bool helpers = (a.flags & LVDGS_FLAG_SUPER_TILES) != 0;
        if (a.flags & LVDGS_FLAG_POSE_ONLY) {
            if (helpers) hipLaunchKernelGGL(preprocess_bwd_pose_helpers_kernel, dim3(nblk), dim3(512), 0, s, p);
            else hipLaunchKernelGGL(preprocess_bwd_pose_kernel, dim3(nblk), dim3(256), 0, s, p);
        } else {
            if (helpers) hipLaunchKernelGGL(preprocess_bwd_helpers_kernel, dim3(nblk), dim3(512), 0, s, p);
            else hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(nblk), dim3(256), 0, s, p);
        }
        LVDGS_LAUNCH_CHECK("preprocess_bwd", a.debug, s);
    }
    if (a.dL_dtau) {   // NULL: the partials stay in the scratch for lvdgs_tracking_tail
        ProfScope ps("tau_reduce", s);
        hipLaunchKernelGGL(tau_reduce_kernel, dim3(1), dim3(256), 0, s, b.tau_part, nblk, a.dL_dtau);
        LVDGS_LAUNCH_CHECK("tau_reduce", a.debug, s);
    }
    return LVDGS_OK;
}

// The per-Gaussian passes of n views of one map in one launch (preprocess_bwd_views_kernel); the caller has checked that the views
// share the map and the gradient buffers, colour by SH of ONE coefficient, scales + rotations (no precomputed covariance).
int launch_preprocess_bwd_views(const lvdgs_args *const *a, const GeomView *g, const BwdScratch *w, const BinView *b, int n, hipStream_t s) {
    const lvdgs_args &a0 = *a[0];
    const int N = a0.num_gaussians;
    const int nblk = cdiv(N, 256);
    for (int first = 0; first < n && N > 0; first += PBWD_VIEWS) {
        const int m = n - first < PBWD_VIEWS ? n - first : PBWD_VIEWS;
        BwdViews bv{};
        BwdParams &p = bv.common;
        p.N = N; p.act = a0.activations;
        p.means3D = a0.means3D; p.opacities = a0.opacities; p.scales = a0.scales; p.rotations = a0.rotations; p.cov3D_precomp = nullptr;
        p.shs = a0.shs; p.colors_precomp = nullptr;
        p.dmeans3D = a0.dL_dmeans3D; p.dopac = a0.dL_dopacities; p.dscales = a0.dL_dscales; p.drot = a0.dL_drotations; p.dcov3D = nullptr; p.dshs = a0.dL_dshs;
        p.dcolors = nullptr;
        // (a later group of the same call adds to what the first group wrote)
        p.accumulate = (first > 0 || (a0.flags & LVDGS_FLAG_ACCUMULATE_PARAM_GRADS)) ? 1 : 0;
        p.pair_total = nullptr; p.pair_capacity = 0;
        bv.n = m;
        for (int k = 0; k < m; k++) {
            const int v = first + k;
            bv.v[k] = BwdView{make_cam(*a[v]), a[v]->radii, g[v].rec, g[v].tiles_touched, g[v].slot_base, w[v].pair_grads, b[v].pair_valid, a[v]->dL_dmeans2D, w[v].tau_part};
        }
        ProfScope ps("preprocess_bwd", s);
        if (a0.flags & LVDGS_FLAG_SUPER_TILES) hipLaunchKernelGGL(preprocess_bwd_views_helpers_kernel, dim3(nblk), dim3(512), 0, s, bv);   // (as in launch_preprocess_bwd)
        else hipLaunchKernelGGL(preprocess_bwd_views_kernel, dim3(nblk), dim3(256), 0, s, bv);
        LVDGS_LAUNCH_CHECK("preprocess_bwd (views)", a0.debug, s);
    }
    for (int v = 0; v < n; v++)
        if (a[v]->dL_dtau) {
            if (N == 0) { if (int e = check_hip(hipMemsetAsync(a[v]->dL_dtau, 0, 6 * sizeof(float), s), "memset tau")) return e; continue; }
            ProfScope ps("tau_reduce", s);
            hipLaunchKernelGGL(tau_reduce_kernel, dim3(1), dim3(256), 0, s, w[v].tau_part, nblk, a[v]->dL_dtau);
            LVDGS_LAUNCH_CHECK("tau_reduce", a[v]->debug, s);
        }
    return LVDGS_OK;
}

int launch_mark_visible(int N, const float *means3D, const float *view, uint8_t *present, hipStream_t s) {
    if (N == 0) return LVDGS_OK;
    ProfScope ps("mark_visible", s);
    hipLaunchKernelGGL(mark_visible_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, N, means3D, view, present);
    LVDGS_LAUNCH_CHECK("mark_visible", 0, s);
    return LVDGS_OK;
}

}  // namespace lvdgs

#ifdef LVDGS_DIAG_PBWD
extern "C" int lvdgs_diag_pbwd(unsigned long long *out_8192x12, int reset) {
    constexpr size_t BYTES = (size_t)lvdgs::PBWD_DIAG_WAVES * lvdgs::PBWD_DIAG_VALUES * sizeof(unsigned long long);
    if (out_8192x12 && hipMemcpyFromSymbol(out_8192x12, HIP_SYMBOL(lvdgs::g_pbwd_diag), BYTES) != hipSuccess) return LVDGS_E_HIP;
    if (reset) {
        void *dptr = nullptr;
        if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(lvdgs::g_pbwd_diag)) != hipSuccess || hipMemset(dptr, 0, BYTES) != hipSuccess) return LVDGS_E_HIP;
    }
    return LVDGS_OK;
}
#endif
