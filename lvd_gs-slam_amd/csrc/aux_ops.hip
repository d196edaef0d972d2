// The two other native dependencies of the SLAM loop the north star names:
//   lvdgs_dist2_knn3 : simple_knn.distCUDA2 -- mean squared distance to the 3 nearest neighbours,
//                      used to seed the scale of new Gaussians (reference README.md:42; keyframe rate);
//   lvdgs_rope2d     : croco curope rope_2d -- in-place 2-D rotary embedding of ViT tokens inside
//                      MASt3R (reference README.md:49-50; two MASt3R passes per tracked frame).
// Both sources are absent from the reference checkout; the algorithms are the published ones.
#include <float.h>

#include "common.hpp"
#include "device_utils.hpp"

namespace lvdgs {
namespace {

// ------------------------------------------------------------------------------------------
// exact 3-NN: Morton order -> boxes of 64 or 256 consecutive points -> every workgroup owns one box, scans itself and
// its eight Morton neighbours, then tests all other boxes in parallel (AABB-to-AABB gap against the workgroup's
// current worst "third best") and stages only the survivors through LDS.
constexpr int KNN_BOUNDS_BLOCKS = 128;
constexpr int KNN_BOX_SMALL = 64, KNN_BOX_LARGE = 256;  // points per box: small boxes prune better and give small inputs
                                                        // enough workgroups; large inputs want fewer boxes to test
constexpr int KNN_SMALL_LIMIT = 200000;
inline int knn_box_size(int P) { return P <= KNN_SMALL_LIMIT ? KNN_BOX_SMALL : KNN_BOX_LARGE; }

struct KnnScratch {
    uint32_t *keys[2], *vals[2], *hist, *totals;
    float *bounds;   // partial bounding boxes: KNN_BOUNDS_BLOCKS x (min xyz, max xyz)
    float *box_lo;   // nbox * 3
    float *box_hi;   // nbox * 3
};

size_t knn_layout(int P, KnnScratch *v, void *base) {
    KnnScratch tmp;
    if (!v) v = &tmp;
    size_t off = 0;
    char *b = (char *)base;
    auto carve = [&](auto *&ptr, size_t count) {
        using T = std::remove_reference_t<decltype(*ptr)>;
        ptr = b ? reinterpret_cast<T *>(b + off) : nullptr;
        off += align256(count * sizeof(T));
    };
    const size_t n = (size_t)(P > 0 ? P : 1), nbox = (n + KNN_BOX_SMALL - 1) / KNN_BOX_SMALL;
    carve(v->keys[0], n); carve(v->keys[1], n); carve(v->vals[0], n); carve(v->vals[1], n);
    carve(v->hist, radix_hist_entries(P)); carve(v->totals, (size_t)1 << SORT_MAX_BITS);
    carve(v->bounds, KNN_BOUNDS_BLOCKS * 6); carve(v->box_lo, nbox * 3); carve(v->box_hi, nbox * 3);
    return off;
}

// per-workgroup bounding boxes of a grid-stride share of the points: partial[block][6] (min xyz, max xyz)
__global__ void __launch_bounds__(256) knn_bounds_kernel(int P, const float *__restrict__ pts, float *__restrict__ partial) {
    __shared__ float s[6][256];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * (size_t)i + a];
            lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; a++) { s[a][threadIdx.x] = lo[a]; s[3 + a][threadIdx.x] = hi[a]; }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + st]);
                s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + st]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256) knn_morton_kernel(int P, const float *__restrict__ pts, const float *__restrict__ partial,
                                                         int nparts, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    // every workgroup folds the partial bounding boxes itself (a few KB from L2): no separate reduction launch
    __shared__ float s_b[6];
    if (threadIdx.x < 6) {
        const bool is_min = threadIdx.x < 3;
        float v = is_min ? FLT_MAX : -FLT_MAX;
        for (int k = 0; k < nparts; k++) {
            const float x = partial[k * 6 + threadIdx.x];
            v = is_min ? fminf(v, x) : fmaxf(v, x);
        }
        s_b[threadIdx.x] = v;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = s_b[a], ext = s_b[3 + a] - lo;
        const float t = ext > 0.f ? (pts[3 * (size_t)i + a] - lo) / ext : 0.f;
        const uint32_t q = (uint32_t)fminf(fmaxf(t * 1023.f, 0.f), 1023.f);
        code |= spread10(q) << a;
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

template <int KNN_BOX>
__global__ void __launch_bounds__(KNN_BOX) knn_boxes_kernel(int P, const float *__restrict__ pts, const uint32_t *__restrict__ order,
                                                            float *__restrict__ box_lo, float *__restrict__ box_hi) {
    __shared__ float s[6][KNN_BOX];
    const int i = blockIdx.x * KNN_BOX + threadIdx.x;
    float p[3] = {0.f, 0.f, 0.f};
    const bool ok = i < P;
    if (ok) { const uint32_t id = order[i]; p[0] = pts[3 * (size_t)id]; p[1] = pts[3 * (size_t)id + 1]; p[2] = pts[3 * (size_t)id + 2]; }
#pragma unroll
    for (int a = 0; a < 3; a++) { s[a][threadIdx.x] = ok ? p[a] : FLT_MAX; s[3 + a][threadIdx.x] = ok ? p[a] : -FLT_MAX; }
    __syncthreads();
    for (int st = KNN_BOX / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + st]);
                s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + st]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 3) { box_lo[3 * blockIdx.x + threadIdx.x] = s[threadIdx.x][0]; box_hi[3 * blockIdx.x + threadIdx.x] = s[3 + threadIdx.x][0]; }
}

__device__ __forceinline__ void push3(float d, float &b0, float &b1, float &b2) {
    if (d < b2) {
        if (d < b1) {
            b2 = b1;
            if (d < b0) { b1 = b0; b0 = d; } else b1 = d;
        } else b2 = d;
    }
}

// One workgroup per box.  Order of work: the own box (gives every point a first bound), the eight Morton neighbours,
// then ALL other boxes are tested 256 at a time -- one box per thread, AABB-to-AABB gap against the workgroup's worst
// "third best" -- and only the survivors are staged and scanned.  Inside a staged box a point whose own third best is
// already closer than the box's AABB skips the scan.  (The first version walked the boxes one by one: two dependent
// global loads per box per workgroup, 54 ms at 0.5 M points.)
template <int KNN_BOX>
__global__ void __launch_bounds__(KNN_BOX) knn_search_kernel(int P, int nbox, const float *__restrict__ pts,
                                                             const uint32_t *__restrict__ order, const float *__restrict__ box_lo,
                                                             const float *__restrict__ box_hi, float *__restrict__ out) {
    __shared__ float s_pts[KNN_BOX][3];
    __shared__ float s_red[KNN_BOX / 64];
    __shared__ int s_list[KNN_BOX];
    __shared__ int s_count;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = b * KNN_BOX + tid;
    const bool ok = i < P;
    uint32_t id = 0;
    float p[3] = {0.f, 0.f, 0.f};
    if (ok) { id = order[i]; p[0] = pts[3 * (size_t)id]; p[1] = pts[3 * (size_t)id + 1]; p[2] = pts[3 * (size_t)id + 2]; }
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    const float mylo[3] = {box_lo[3 * b], box_lo[3 * b + 1], box_lo[3 * b + 2]};
    const float myhi[3] = {box_hi[3 * b], box_hi[3 * b + 1], box_hi[3 * b + 2]};

    auto scan_box = [&](int cc) {  // workgroup-uniform cc
        __syncthreads();
        const int j = cc * KNN_BOX + tid;
        if (j < P) {
            const uint32_t jd = order[j];
            s_pts[tid][0] = pts[3 * (size_t)jd]; s_pts[tid][1] = pts[3 * (size_t)jd + 1]; s_pts[tid][2] = pts[3 * (size_t)jd + 2];
        }
        __syncthreads();
        const int cnt = min(KNN_BOX, P - cc * KNN_BOX);
        // this point against the candidate box: nothing in the box can beat a third best that is already closer
        float gap2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float g = fmaxf(0.f, fmaxf(box_lo[3 * cc + a] - p[a], p[a] - box_hi[3 * cc + a]));
            gap2 += g * g;
        }
        if (ok && gap2 < b2) {
            for (int k = 0; k < cnt; k++) {
                if (cc == b && k == tid) continue;
                const float dx = s_pts[k][0] - p[0], dy = s_pts[k][1] - p[1], dz = s_pts[k][2] - p[2];
                push3(dx * dx + dy * dy + dz * dz, b0, b1, b2);
            }
        }
    };
    auto workgroup_bound = [&]() {  // max over the workgroup's points of their current third-best distance
        float m = ok ? b2 : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (KNN_BOX == 64) return m;
        __syncthreads();
        if (lane == 0) s_red[wave] = m;
        __syncthreads();
        float r = s_red[0];
        for (int w = 1; w < KNN_BOX / 64; w++) r = fmaxf(r, s_red[w]);
        return r;
    };

    constexpr int NEAR = 4;
    scan_box(b);
    for (int off = 1; off <= NEAR; off++) {
        if (b - off >= 0) scan_box(b - off);
        if (b + off < nbox) scan_box(b + off);
    }
    float bound = workgroup_bound();
    for (int c0 = 0; c0 < nbox; c0 += KNN_BOX) {
        // one candidate box per thread
        const int cc = c0 + tid;
        bool take = false;
        if (cc < nbox && (cc < b - NEAR || cc > b + NEAR)) {
            float gap2 = 0.f;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float g = fmaxf(0.f, fmaxf(box_lo[3 * cc + a] - myhi[a], mylo[a] - box_hi[3 * cc + a]));
                gap2 += g * g;
            }
            take = gap2 <= bound;
        }
        if (tid == 0) s_count = 0;
        __syncthreads();
        if (take) s_list[atomicAdd(&s_count, 1)] = cc;  // any order: the result is a minimum
        __syncthreads();
        const int n = s_count;
        for (int q = 0; q < n; q++) {
            const int cq = s_list[q];
            // the bound may have tightened since the test
            float gap2 = 0.f;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float g = fmaxf(0.f, fmaxf(box_lo[3 * cq + a] - myhi[a], mylo[a] - box_hi[3 * cq + a]));
                gap2 += g * g;
            }
            if (gap2 > bound) continue;
            scan_box(cq);
            bound = workgroup_bound();
        }
        __syncthreads();
    }
    if (ok) {
        // fewer than 4 points: average what exists
        float sum = 0.f; int n = 0;
        if (b0 < FLT_MAX) { sum += b0; n++; }
        if (b1 < FLT_MAX) { sum += b1; n++; }
        if (b2 < FLT_MAX) { sum += b2; n++; }
        out[id] = n ? sum / (float)n : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// RoPE 2D: one workgroup per token (b, n).  The 2 * D/4 (cos, sin) pairs of the token are computed once into LDS
// and reused by all H heads.  Tokens are addressed through (batch, token, head) strides with the D axis contiguous, so
// both the (B, N, H, D) layout and the (B, H, N, D) layout croco's attention keeps are rotated where they lie (no
// transposed copy); elements are f32, f16 or bf16, the arithmetic is f32 and every element is rounded once at the store.
// A work item rotates V neighbouring pairs (i .. i+V-1, i+Q .. i+Q+V-1) of one half of one head: two V-wide loads and
// two V-wide stores (16 bytes for f32, 8 for the 16-bit types at V = 4).
template <typename T> struct RopeIO;
template <> struct RopeIO<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct RopeIO<_Float16> {
    static __device__ __forceinline__ float ld(const _Float16 *p) { return (float)*p; }
    static __device__ __forceinline__ void st(_Float16 *p, float v) { *p = (_Float16)v; }
};
template <> struct RopeIO<__bf16> {
    static __device__ __forceinline__ float ld(const __bf16 *p) { return (float)*p; }
    static __device__ __forceinline__ void st(__bf16 *p, float v) { *p = (__bf16)v; }  // round to nearest even; NaN stays NaN
};

template <typename T, int V>
__global__ void __launch_bounds__(256) rope2d_kernel(T *__restrict__ tokens, const int64_t *__restrict__ pos, int N, int H, int D,
                                                     int64_t stride_b, int64_t stride_n, int64_t stride_h, float base, float fwd) {
    extern __shared__ float s_cs[];  // [2][Q][2]
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int Q = D / 4, Dh = D / 2;
    const size_t tok = blockIdx.x;
    for (int k = threadIdx.x; k < 2 * Q; k += blockDim.x) {
        const int half = k / Q, i = k - half * Q;
        const float inv_freq = fwd / powf(base, (float)i / (float)Q);
        const float ang = (float)pos[tok * 2 + half] * inv_freq;
        float sn, cs;
        sincosf(ang, &sn, &cs);
        s_cs[2 * k] = cs; s_cs[2 * k + 1] = sn;
    }
    __syncthreads();
    const int64_t b = (int64_t)(tok / (size_t)N), n = (int64_t)(tok % (size_t)N);
    T *t = tokens + b * stride_b + n * stride_n;
    const int QV = Q / V;  // V divides Q (checked by the launcher)
    for (int w = threadIdx.x; w < H * 2 * QV; w += blockDim.x) {
        const int h = w / (2 * QV), r = w - h * 2 * QV, half = r / QV, i = (r - half * QV) * V;
        T *x = t + (int64_t)h * stride_h + half * Dh + i;
        const float *cs = s_cs + 2 * (half * Q + i);
        if constexpr (V == 1) {
            const float u = RopeIO<T>::ld(x), v = RopeIO<T>::ld(x + Q);
            RopeIO<T>::st(x, u * cs[0] - v * cs[1]);
            RopeIO<T>::st(x + Q, v * cs[0] + u * cs[1]);
        } else {
            const vec_t U = *reinterpret_cast<const vec_t *>(x), W = *reinterpret_cast<const vec_t *>(x + Q);
            vec_t A, Bv;
#pragma unroll
            for (int k = 0; k < V; k++) {
                const float u = (float)U[k], v = (float)W[k];
                A[k] = (T)(u * cs[2 * k] - v * cs[2 * k + 1]);
                Bv[k] = (T)(v * cs[2 * k] + u * cs[2 * k + 1]);
            }
            *reinterpret_cast<vec_t *>(x) = A;
            *reinterpret_cast<vec_t *>(x + Q) = Bv;
        }
    }
}

template <typename T>
int launch_rope2d(void *tokens, const int64_t *positions, int B, int N, int H, int D, int64_t sb, int64_t sn, int64_t sh,
                  float base, float fwd, hipStream_t s) {
    const int Q = D / 4;
    // 4-wide accesses need every (token, head, half, i) address aligned to 4 elements
    const bool wide = (Q % 4 == 0) && (sb % 4 == 0) && (sn % 4 == 0) && (sh % 4 == 0) && ((uintptr_t)tokens % (4 * sizeof(T)) == 0);
    ProfScope ps("rope2d", s);
    const dim3 grid((unsigned)((int64_t)B * N)), block(256);
    const size_t lds = (size_t)D * 2 * sizeof(float);
    if (wide) hipLaunchKernelGGL((rope2d_kernel<T, 4>), grid, block, lds, s, (T *)tokens, positions, N, H, D, sb, sn, sh, base, fwd);
    else hipLaunchKernelGGL((rope2d_kernel<T, 1>), grid, block, lds, s, (T *)tokens, positions, N, H, D, sb, sn, sh, base, fwd);
    LVDGS_LAUNCH_CHECK("rope2d", 0, s);
    return LVDGS_OK;
}

}  // namespace
}  // namespace lvdgs

using namespace lvdgs;

extern "C" {

size_t lvdgs_knn_scratch_bytes(int32_t num_points) { return knn_layout(num_points, nullptr, nullptr); }

int lvdgs_dist2_knn3(int32_t P, const float *points, float *mean_dist2, void *scratch, size_t scratch_bytes, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || (P > 0 && (!points || !mean_dist2 || !scratch))) { set_error("bad dist2_knn3 arguments"); return LVDGS_E_INVALID; }
    if (P == 0) return LVDGS_OK;
    if (scratch_bytes < lvdgs_knn_scratch_bytes(P)) { set_error("knn scratch too small"); return LVDGS_E_INVALID; }
    KnnScratch w;
    knn_layout(P, &w, scratch);
    const int box = knn_box_size(P), nbox = cdiv(P, box);
    { ProfScope ps("knn_bounds", s); hipLaunchKernelGGL(knn_bounds_kernel, dim3(KNN_BOUNDS_BLOCKS), dim3(256), 0, s, P, points, w.bounds); LVDGS_LAUNCH_CHECK("knn_bounds", 0, s); }
    { ProfScope ps("knn_morton", s); hipLaunchKernelGGL(knn_morton_kernel, dim3(cdiv(P, 256)), dim3(256), 0, s, P, points, w.bounds, KNN_BOUNDS_BLOCKS, w.keys[0], w.vals[0]); LVDGS_LAUNCH_CHECK("knn_morton", 0, s); }
    bool in_a = true;
    if (int e = radix_sort_pairs(w.keys[0], w.vals[0], w.keys[1], w.vals[1], P, 30, w.hist, w.totals, &in_a, 0, s)) return e;
    const uint32_t *order = in_a ? w.vals[0] : w.vals[1];
    {
        ProfScope ps("knn_boxes", s);
        if (box == KNN_BOX_SMALL) hipLaunchKernelGGL(knn_boxes_kernel<KNN_BOX_SMALL>, dim3(nbox), dim3(KNN_BOX_SMALL), 0, s, P, points, order, w.box_lo, w.box_hi);
        else hipLaunchKernelGGL(knn_boxes_kernel<KNN_BOX_LARGE>, dim3(nbox), dim3(KNN_BOX_LARGE), 0, s, P, points, order, w.box_lo, w.box_hi);
        LVDGS_LAUNCH_CHECK("knn_boxes", 0, s);
    }
    {
        ProfScope ps("knn_search", s);
        if (box == KNN_BOX_SMALL) hipLaunchKernelGGL(knn_search_kernel<KNN_BOX_SMALL>, dim3(nbox), dim3(KNN_BOX_SMALL), 0, s, P, nbox, points, order, w.box_lo, w.box_hi, mean_dist2);
        else hipLaunchKernelGGL(knn_search_kernel<KNN_BOX_LARGE>, dim3(nbox), dim3(KNN_BOX_LARGE), 0, s, P, nbox, points, order, w.box_lo, w.box_hi, mean_dist2);
        LVDGS_LAUNCH_CHECK("knn_search", 0, s);
    }
    return LVDGS_OK;
}

int lvdgs_rope2d_strided(void *tokens, int32_t dtype, const int64_t *positions, int32_t B, int32_t N, int32_t H, int32_t D,
                         int64_t stride_b, int64_t stride_n, int64_t stride_h, float base, float fwd, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B < 0 || N < 0 || H <= 0 || D <= 0 || (D % 4) != 0) { set_error("rope2d: D must be a positive multiple of 4"); return LVDGS_E_INVALID; }
    if (dtype < LVDGS_F32 || dtype > LVDGS_BF16) { set_error("rope2d: dtype must be LVDGS_F32, LVDGS_F16 or LVDGS_BF16"); return LVDGS_E_INVALID; }
    if ((int64_t)B * N == 0) return LVDGS_OK;
    if ((int64_t)B * N > 0x7fffffffLL) { set_error("rope2d: more than 2^31 - 1 tokens"); return LVDGS_E_RANGE; }
    if (!tokens || !positions) { set_error("rope2d: NULL tensor"); return LVDGS_E_INVALID; }
    if (stride_b < 0 || stride_n < 0 || stride_h < 0 || (H > 1 && stride_h < D) || (N > 1 && stride_n < D)) {
        set_error("rope2d: strides must be non-negative, heads and tokens at least D elements apart (rotated in place)");
        return LVDGS_E_INVALID;
    }
    switch (dtype) {
        case LVDGS_F16: return launch_rope2d<_Float16>(tokens, positions, B, N, H, D, stride_b, stride_n, stride_h, base, fwd, s);
        case LVDGS_BF16: return launch_rope2d<__bf16>(tokens, positions, B, N, H, D, stride_b, stride_n, stride_h, base, fwd, s);
        default: return launch_rope2d<float>(tokens, positions, B, N, H, D, stride_b, stride_n, stride_h, base, fwd, s);
    }
}

int lvdgs_rope2d(float *tokens, const int64_t *positions, int32_t B, int32_t N, int32_t H, int32_t D, float base, float fwd,
                 void *stream) {
    return lvdgs_rope2d_strided(tokens, LVDGS_F32, positions, B, N, H, D, (int64_t)N * H * D, (int64_t)H * D, D, base, fwd, stream);
}

}  // extern "C"
