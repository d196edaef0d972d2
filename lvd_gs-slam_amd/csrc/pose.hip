// Per-frame pose optimiser step on the device: Adam on (cam_rot_delta, cam_trans_delta, exposure_a, exposure_b), the
// SE(3) retraction of the pose, and the view / projection matrices the next render reads -- one tiny launch.
//
// What the tracking loop does on the host between two renders (reference utils/slam_frontend.py:1518-1521):
//     pose_optimizer.step()                 torch.optim.Adam, four parameter groups
//     converged = update_pose(viewpoint)    utils/pose_utils.py:70-87: T_w2c <- SE3_exp([rho, theta]) @ T_w2c,
//                                           deltas zeroed, converged = ||tau|| < 1e-4
// followed, at the next render, by Camera.world_view_transform / full_proj_transform / camera_center
// (utils/camera_utils.py:106-120).  In PyTorch that is ~60 small launches and three host synchronisations
// (`if angle < 1e-5` twice in SO3_exp / V, `if converged`), each of which drains the GPU.  Here nothing returns to
// the host: the flag stays on the device, and once it is set further steps leave the pose alone, so a host that
// runs a few iterations ahead ends with exactly the state of the loop that broke out at the converged iteration.
//
// Arithmetic mirrors the PyTorch statements in float32 (Adam's bias corrections in double, like Python floats).
#include "common.hpp"
#include "device_utils.hpp"
#include "map_stats.hpp"

namespace lvdgs {
namespace {

struct PoseStepParams {
    lvdgs_pose_step_args a;
};

__device__ void so3_coeffs(float angle, float &A, float &B, float &C) {
    // utils/pose_utils.py:30,46: truncated series below 1e-5 rad
    if (angle < 1e-5f) { A = 1.f; B = 0.5f; C = 1.f / 6.f; return; }
    const float a2 = angle * angle;
    const float s = sinf(angle), c = cosf(angle);
    A = s / angle; B = (1.f - c) / a2; C = (angle - s) / (a2 * angle);
}

__device__ void mat3_mul(const float *X, const float *Y, float *Z) {  // row-major 3x3
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Z[3 * i + j] = X[3 * i] * Y[j] + X[3 * i + 1] * Y[3 + j] + X[3 * i + 2] * Y[6 + j];
}

__device__ double int_pow(double x, unsigned n) {
    double r = 1.0;
    for (; n; n >>= 1, x *= x)
        if (n & 1u) r *= x;
    return r;
}

__device__ float adam_update(float p, float g, float *m, float *v, double lr, double beta1, double beta2, double eps, double bc1, double bc2_sqrt) {
    // torch.optim.Adam, single-tensor path: exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2);
    // denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps); param.addcdiv_(exp_avg, denom, value=-lr / bias_correction1).
    // The scalars are Python floats (doubles) rounded to float32 where they meet the tensors.
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2);
    const float m1 = *m + w1 * (g - *m);
    const float v1 = *v * b2 + w2 * (g * g);
    *m = m1; *v = v1;
    const float step_size = (float)(lr / bc1);
    const float denom = sqrtf(v1) / (float)bc2_sqrt + (float)eps;
    return p - step_size * (m1 / denom);
}

// One wave.  Lanes 0..7 each take one of the eight scalars through Adam (rot xyz, trans xyz, exposure a, b) -- the bias
// corrections are powers in double precision, a few hundred dependent instructions that used to run eight times in a row
// on one lane -- then lane 0 does the retraction and the matrices with everything it needs already in registers (every
// load of the step is issued before the first dependent instruction: the one-lane version paid a memory round trip per
// statement, a third of the launch).
// state: [0..15] (m, v) of the eight scalars; [16] calls applied; [17] converged (sticky); [18] iterations applied;
// [19..22] Adam step count of the rot / trans / exposure_a / exposure_b group -- a group without a gradient in a call is
// skipped, moments AND step count, as torch.optim.Adam skips parameters without .grad (bias corrections follow the group's
// own count, not the number of calls).
__device__ void pose_step_body(const lvdgs_pose_step_args &a) {   // the first wave of the workgroup, all 64 lanes
    const int lane = threadIdx.x & 63;
    float *st = a.state;
    if (st[17] != 0.f) return;  // already converged: the host may have run ahead (uniform: every lane reads the same word)
    // ---- everything the step reads, up front ----
    const float calls = st[16];
    const float *g_rot = a.grad_tau ? a.grad_tau + 3 : a.grad_rot, *g_trans = a.grad_tau ? a.grad_tau : a.grad_trans;
    const int group = lane < 3 ? 0 : (lane < 6 ? 1 : (lane == 6 ? 2 : 3));
    float *param = nullptr;
    const float *grad = nullptr;
    double lr = 0.0;
    if (lane < 3) { if (a.R) { param = a.cam_rot_delta + lane; grad = g_rot ? g_rot + lane : nullptr; } lr = a.lr_rot; }
    else if (lane < 6) { if (a.R) { param = a.cam_trans_delta + (lane - 3); grad = g_trans ? g_trans + (lane - 3) : nullptr; } lr = a.lr_trans; }
    else if (lane == 6) { param = a.exposure_a; grad = a.exposure_a ? a.grad_exposure_a : nullptr; lr = a.lr_exposure; }
    else if (lane == 7) { param = a.exposure_b; grad = a.exposure_b ? a.grad_exposure_b : nullptr; lr = a.lr_exposure; }
    const bool mine = lane < 8 && param != nullptr;
    const float p0 = mine ? *param : 0.f, g0 = (mine && grad) ? *grad : 0.f;
    const float m0 = lane < 8 ? st[2 * lane] : 0.f, v0 = lane < 8 ? st[2 * lane + 1] : 0.f;
    const float steps0 = lane < 8 ? st[19 + group] : 0.f;
    float R0[9], T0[3], PR[16];
    const bool lead = lane == 0 && a.R != nullptr;
    if (lead) {
#pragma unroll
        for (int i = 0; i < 9; i++) R0[i] = a.R[i];
#pragma unroll
        for (int i = 0; i < 3; i++) T0[i] = a.T[i];
        if (a.projmatrix && a.projmatrix_raw) {
#pragma unroll
            for (int i = 0; i < 16; i++) PR[i] = a.projmatrix_raw[i];
        }
    }
    // ---- Adam, one scalar per lane ----
    float p1 = p0;
    if (mine && grad) {
        const float step = steps0 + 1.f;
        // beta^step by squaring (the step count is a small integer).  It can differ from Python's beta ** step in the last
        // bits of the double, which the float32 step size and denominator never see.
        const double bc1 = 1.0 - int_pow(a.beta1, (unsigned)step);
        const double bc2_sqrt = sqrt(1.0 - int_pow(a.beta2, (unsigned)step));
        float m = m0, v = v0;
        p1 = adam_update(p0, g0, &m, &v, lr, a.beta1, a.beta2, a.eps, bc1, bc2_sqrt);
        st[2 * lane] = m; st[2 * lane + 1] = v;
        if (lane == 0 || lane == 3 || lane >= 6) st[19 + group] = step;
        if (lane >= 6) *param = p1;
    }
    if (lane == 0) { st[16] = calls + 1.f; st[18] = calls + 1.f; if (a.host_flags) a.host_flags[1] = calls + 1.f; }
    float rot[3], trans[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { rot[k] = __shfl(p1, k, 64); trans[k] = __shfl(p1, 3 + k, 64); }
    if (!lead) return;  // exposure only, or not the lane that holds the pose
    // ---- update_pose: T_w2c <- SE3_exp([trans, rot]) @ [R T] ----
    const float angle = sqrtf(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
    float A, B, C;
    so3_coeffs(angle, A, B, C);
    const float K[9] = {0.f, -rot[2], rot[1], rot[2], 0.f, -rot[0], -rot[1], rot[0], 0.f};
    float K2[9];
    mat3_mul(K, K, K2);
    float dR[9], Vm[9];
    for (int i = 0; i < 9; i++) {
        const float eye = (i % 4 == 0) ? 1.f : 0.f;
        dR[i] = eye + A * K[i] + B * K2[i];
        Vm[i] = eye + B * K[i] + C * K2[i];
    }
    float dt[3];
    for (int i = 0; i < 3; i++) dt[i] = Vm[3 * i] * trans[0] + Vm[3 * i + 1] * trans[1] + Vm[3 * i + 2] * trans[2];
    float R1[9], T1[3];
    mat3_mul(dR, R0, R1);
    for (int i = 0; i < 3; i++) T1[i] = dR[3 * i] * T0[0] + dR[3 * i + 1] * T0[1] + dR[3 * i + 2] * T0[2] + dt[i];
    for (int i = 0; i < 9; i++) a.R[i] = R1[i];
    for (int i = 0; i < 3; i++) a.T[i] = T1[i];
    const float tau_norm = sqrtf(trans[0] * trans[0] + trans[1] * trans[1] + trans[2] * trans[2] + rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
    if (tau_norm < a.converged_threshold) { st[17] = 1.f; if (a.host_flags) a.host_flags[0] = 1.f; }
    for (int k = 0; k < 3; k++) { a.cam_rot_delta[k] = 0.f; a.cam_trans_delta[k] = 0.f; }
    // ---- derived matrices (row-vector layout, utils/camera_utils.py:106-120) ----
    // world_view_transform = [[R, T], [0, 1]]^T ; full_proj_transform = world_view_transform @ projection_matrix ;
    // camera_center = inverse(world_view_transform)[3, :3] = -R^T T for a rigid transform
    float view[16];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) view[4 * i + j] = R1[3 * j + i];
        view[4 * i + 3] = 0.f;
    }
    view[12] = T1[0]; view[13] = T1[1]; view[14] = T1[2]; view[15] = 1.f;
    if (a.viewmatrix) for (int i = 0; i < 16; i++) a.viewmatrix[i] = view[i];
    if (a.projmatrix && a.projmatrix_raw) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                float s = 0.f;
                for (int k = 0; k < 4; k++) s += view[4 * i + k] * PR[4 * k + j];
                a.projmatrix[4 * i + j] = s;
            }
    }
    if (a.campos)
        for (int i = 0; i < 3; i++) a.campos[i] = -(R1[i] * T1[0] + R1[3 + i] * T1[1] + R1[6 + i] * T1[2]);
}

__global__ void pose_step_kernel(PoseStepParams pp) {
    if (blockIdx.x != 0) return;
    pose_step_body(pp.a);
}

// Several independent pose steps (the window's keyframes after a mapping iteration) in one launch: one workgroup each.
constexpr int POSE_BATCH_MAX = 16;
struct PoseBatchParams {
    lvdgs_pose_step_args a[POSE_BATCH_MAX];
};
__global__ void pose_step_batch_kernel(PoseBatchParams pp) {
    pose_step_body(pp.a[blockIdx.x]);
}

// The end of a tracking iteration in one launch (lvdgs_tracking_tail): what photometric_finish_kernel<2>, tau_reduce_kernel
// and pose_step_kernel do one after the other -- the same additions in the same order, so the same bits -- without two
// launches of ~6 us each on the iteration's critical path.
struct TailParams {
    LossTail loss;
    const float *tau_part; int tau_blocks; float *dL_dtau;
    lvdgs_pose_step_args pose;   // grad_tau / grad_exposure_* already point at dL_dtau and loss.d_a / d_b
    int has_pose;                // 0: the two reductions only (a mapping view: its keyframe is stepped later)
};

// THREADS = 256: the order of photometric_finish_kernel / tau_reduce_kernel (bit-identical to them); 1024: for the four
// times as many per-tile partial sums lvdgs_backward_fused_loss leaves (a fixed order of its own).
template <int THREADS>
__device__ __forceinline__ void tracking_tail_body(const TailParams &t) {
    constexpr int WAVES = THREADS / 64;
    __shared__ float s[10][WAVES];   // [value][wave]: loss sums 0..3, pose gradient 4..9
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < t.loss.nblk; b += THREADS) {
        a[0] += t.loss.partial[4 * b]; a[1] += t.loss.partial[4 * b + 1];
        a[2] += t.loss.partial[4 * b + 2]; a[3] += t.loss.partial[4 * b + 3];
    }
    float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < t.tau_blocks; b += THREADS)
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] += t.tau_part[(size_t)b * 6 + k];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float v = wave_sum_to_lane63(a[k]);
        if (lane == 63) s[k][wave] = v;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const float v = wave_sum_to_lane63(g[k]);
        if (lane == 63) s[4 + k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        float v = s[threadIdx.x][0];
#pragma unroll
        for (int w = 1; w < WAVES; w++) v += s[threadIdx.x][w];   // ((s0 + s1) + s2) + s3 ...
        s[threadIdx.x][0] = v;
        if (threadIdx.x >= 4) t.dL_dtau[threadIdx.x - 4] = v;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (threadIdx.x == 0 && t.loss.loss) {   // (no loss block: a view whose loss value is finished elsewhere, lvdgs_map_view_tail(loss = NULL))
        t.loss.loss[0] = t.loss.w_rgb * (s[0][0] / (3.f * (float)t.loss.P)) + t.loss.w_d * (s[1][0] / (float)t.loss.P);
        if (t.loss.d_a) t.loss.d_a[0] = s[2][0];
        if (t.loss.d_b) t.loss.d_b[0] = s[3][0];
    }
    if (!t.has_pose) return;
    __threadfence_block();   // the wave reads dL_dtau (written by threads 4..9 before the barrier) and d_a / d_b (lane 0, just now) back
    __builtin_amdgcn_wave_barrier();
    pose_step_body(t.pose);
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS) tracking_tail_kernel(TailParams t) { tracking_tail_body<THREADS>(t); }

// A mapping view's last launch: workgroup 0 finishes the view's loss and pose gradient (the tail above, without a pose
// step), the others take the view's statistics (map_stats.hpp) -- one launch where lvdgs_tracking_tail + lvdgs_view_stats
// were two on every view's critical path.
__global__ void __launch_bounds__(1024) map_view_tail_kernel(TailParams t, ViewStats v) {
    if (blockIdx.x == 0) { tracking_tail_body<1024>(t); return; }
    const int i = ((int)blockIdx.x - 1) * 1024 + (int)threadIdx.x;
    if (i < v.N) view_stats_one(v, i);
}

// The tails of ALL views of a mapping window in one launch (lvdgs_map_view_tail_batch): workgroup k < n finishes view k's loss and
// pose gradient, the others take the statistics -- a thread per Gaussian walking the views IN ORDER, so that norm_sum receives the
// same additions in the same order as one tail per view (bit-identical).  Ten launches of ~5 us on the iteration's critical path less.
constexpr int MAP_TAIL_VIEWS = 12;
struct MapTailView { LossTail loss; const float *tau_part; int tau_blocks; float *dL_dtau; ViewStats st; };
struct MapTailBatch { int n; MapTailView v[MAP_TAIL_VIEWS]; };
static_assert(sizeof(MapTailBatch) <= 4000, "kernel arguments");
__global__ void __launch_bounds__(1024) map_view_tail_batch_kernel(MapTailBatch b) {
    if ((int)blockIdx.x < b.n) {
        const MapTailView &mv = b.v[blockIdx.x];
        TailParams t{};
        t.loss = mv.loss; t.tau_part = mv.tau_part; t.tau_blocks = mv.tau_blocks; t.dL_dtau = mv.dL_dtau; t.has_pose = 0;
        tracking_tail_body<1024>(t);
        return;
    }
    const int i = ((int)blockIdx.x - b.n) * 1024 + (int)threadIdx.x;
    for (int k = 0; k < b.n; k++)
        if (i < b.v[k].st.N) view_stats_one(b.v[k].st, i);
}

}  // namespace
}  // namespace lvdgs

using namespace lvdgs;

static int check_pose_args(const lvdgs_pose_step_args *a);

extern "C" int lvdgs_pose_step(const lvdgs_pose_step_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (int e = check_pose_args(a)) return e;
    PoseStepParams pp{*a};
    ProfScope ps("pose_step", s);
    hipLaunchKernelGGL(pose_step_kernel, dim3(1), dim3(64), 0, s, pp);
    LVDGS_LAUNCH_CHECK("pose_step", 0, s);
    return LVDGS_OK;
}

extern "C" int lvdgs_host_device_pointer(void *host, void **device) {
    if (!host || !device) { set_error("host_device_pointer: NULL argument"); return LVDGS_E_INVALID; }
    *device = nullptr;
    return check_hip(hipHostGetDevicePointer(device, host, 0), "hipHostGetDevicePointer (is the memory page-locked and mapped?)");
}

static int check_pose_args(const lvdgs_pose_step_args *a) {
    if (!a || !a->state || (a->R ? (!a->T || !a->cam_rot_delta || !a->cam_trans_delta) : (a->T != nullptr))) {
        set_error("pose step: state is NULL, or R is given without T / cam_rot_delta / cam_trans_delta (R and T both NULL = exposure only)");
        return LVDGS_E_INVALID;
    }
    if (!(a->beta1 >= 0.0 && a->beta1 < 1.0 && a->beta2 >= 0.0 && a->beta2 < 1.0)) { set_error("pose step: betas must lie in [0, 1)"); return LVDGS_E_INVALID; }
    return LVDGS_OK;
}

static int make_tail_params(const lvdgs_loss_args *loss, const lvdgs_args *bwd, const lvdgs_pose_step_args *pose, float *dL_dtau,
                            int32_t partials_per_tile, TailParams &t) {
    if (!bwd || !dL_dtau) { set_error("tracking tail: backward arguments / dL_dtau is NULL"); return LVDGS_E_INVALID; }
    if (pose)
        if (int e = check_pose_args(pose)) return e;
    t = TailParams{};
    if (loss) {
        if (int e = loss_tail_params(loss, partials_per_tile != 0, &t.loss)) return e;
    } else if (pose) { set_error("tracking tail: a pose step needs the loss block (its exposure gradients)"); return LVDGS_E_INVALID; }
    if (loss && partials_per_tile) {   // a band of tile rows (lvdgs_args.tile_row_*): the backward left partial sums for its tiles only
        int row0, row1;
        tile_row_band(*bwd, &row0, &row1);
        const int gx = cdiv(bwd->image_width, TILE);
        if (loss->width != bwd->image_width || loss->height != bwd->image_height) { set_error("tracking tail: image size differs between loss and backward arguments"); return LVDGS_E_INVALID; }
        t.loss.partial += 4 * (size_t)row0 * gx;
        t.loss.nblk = (row1 - row0) * gx;
    }
    const int N = bwd->num_gaussians;
    if (N < 0 || bwd->num_rendered < 0 || (N > 0 && !bwd->scratch)) { set_error("tracking tail: bad backward arguments"); return LVDGS_E_INVALID; }
    if (N > 0) {
        if (bwd->scratch_bytes < lvdgs_backward_scratch_bytes(N, N == 0 ? 0 : bwd->num_rendered)) { set_error("tracking tail: scratch too small"); return LVDGS_E_INVALID; }
        BwdScratch w;
        bwd_scratch_layout(N, bwd->num_rendered, &w, bwd->scratch);
        t.tau_part = w.tau_part;
        t.tau_blocks = cdiv(N, 256);
    }
    t.dL_dtau = dL_dtau;
    t.has_pose = pose != nullptr;
    if (pose) {
        t.pose = *pose;
        t.pose.grad_tau = dL_dtau; t.pose.grad_rot = t.pose.grad_trans = nullptr;
        t.pose.grad_exposure_a = pose->exposure_a ? loss->d_exposure_a : nullptr;
        t.pose.grad_exposure_b = pose->exposure_b ? loss->d_exposure_b : nullptr;
    }
    return LVDGS_OK;
}

extern "C" int lvdgs_tracking_tail(const lvdgs_loss_args *loss, const lvdgs_args *bwd, const lvdgs_pose_step_args *pose, float *dL_dtau,
                                   int32_t partials_per_tile, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    TailParams t;
    if (int e = make_tail_params(loss, bwd, pose, dL_dtau, partials_per_tile, t)) return e;
    ProfScope ps("tracking_tail", s);
    if (partials_per_tile) hipLaunchKernelGGL(tracking_tail_kernel<1024>, dim3(1), dim3(1024), 0, s, t);
    else hipLaunchKernelGGL(tracking_tail_kernel<256>, dim3(1), dim3(256), 0, s, t);
    LVDGS_LAUNCH_CHECK("tracking_tail", 0, s);
    return LVDGS_OK;
}

extern "C" int lvdgs_map_view_tail(const lvdgs_loss_args *loss, const lvdgs_args *bwd, float *dL_dtau, const lvdgs_view_stats_args *stats, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    TailParams t;
    if (int e = make_tail_params(loss, bwd, nullptr, dL_dtau, 1, t)) return e;
    const int N = bwd->num_gaussians;
    // (an empty map has no statistics to take: only workgroup 0 -- the loss and the pose gradient -- runs, as the header says)
    if (!stats || (N > 0 && (!bwd->radii || !stats->radii_max || (!stats->norm_sum && !stats->split_xy) || (stats->touched_row && !bwd->n_touched)))) {
        set_error("map view tail: a statistics pointer is NULL"); return LVDGS_E_INVALID;
    }
    const ViewStats v{N, bwd->radii, bwd->n_touched, bwd->dL_dmeans2D, stats->radii_max, stats->norm_sum, stats->vis_count, stats->touched_row, stats->split_xy};
    ProfScope ps("map_view_tail", s);
    hipLaunchKernelGGL(map_view_tail_kernel, dim3(1 + cdiv(N, 1024)), dim3(1024), 0, s, t, v);
    LVDGS_LAUNCH_CHECK("map_view_tail", 0, s);
    return LVDGS_OK;
}

extern "C" int lvdgs_map_view_tail_batch(const lvdgs_loss_args *const *losses, const lvdgs_args *const *bwds, float *const *dL_dtau,
                                         const lvdgs_view_stats_args *const *stats, int32_t count, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (count < 0 || (count > 0 && (!bwds || !dL_dtau || !stats))) { set_error("map view tail batch: bad view list"); return LVDGS_E_INVALID; }
    for (int32_t first = 0; first < count; first += MAP_TAIL_VIEWS) {
        const int n = count - first < MAP_TAIL_VIEWS ? count - first : MAP_TAIL_VIEWS;
        MapTailBatch b{};
        b.n = n;
        int max_n = 0;
        for (int k = 0; k < n; k++) {
            const int v = first + k;
            const lvdgs_args *bwd = bwds[v];
            const lvdgs_view_stats_args *st = stats[v];
            TailParams t;
            if (int e = make_tail_params(losses ? losses[v] : nullptr, bwd, nullptr, dL_dtau[v], 1, t)) return e;
            const int N = bwd->num_gaussians;
            if (!st || (N > 0 && (!bwd->radii || !st->radii_max || (!st->norm_sum && !st->split_xy) || (st->touched_row && !bwd->n_touched)))) {
                set_error("map view tail batch: a statistics pointer of view %d is NULL", v); return LVDGS_E_INVALID;
            }
            b.v[k] = MapTailView{t.loss, t.tau_part, t.tau_blocks, t.dL_dtau,
                                 ViewStats{N, bwd->radii, bwd->n_touched, bwd->dL_dmeans2D, st->radii_max, st->norm_sum, st->vis_count, st->touched_row, st->split_xy}};
            max_n = N > max_n ? N : max_n;
        }
        ProfScope ps("map_view_tail", s);
        hipLaunchKernelGGL(map_view_tail_batch_kernel, dim3(n + cdiv(max_n, 1024)), dim3(1024), 0, s, b);
        LVDGS_LAUNCH_CHECK("map_view_tail (batch)", 0, s);
    }
    return LVDGS_OK;
}

extern "C" int lvdgs_pose_step_batch(const lvdgs_pose_step_args *steps, int32_t count, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (count < 0 || (count > 0 && !steps)) { set_error("pose step batch: bad arguments"); return LVDGS_E_INVALID; }
    for (int32_t first = 0; first < count; first += POSE_BATCH_MAX) {
        const int n = count - first < POSE_BATCH_MAX ? count - first : POSE_BATCH_MAX;
        PoseBatchParams pp{};
        for (int i = 0; i < n; i++) {
            if (int e = check_pose_args(steps + first + i)) return e;
            pp.a[i] = steps[first + i];
        }
        ProfScope ps("pose_step", s);
        hipLaunchKernelGGL(pose_step_batch_kernel, dim3(n), dim3(64), 0, s, pp);
        LVDGS_LAUNCH_CHECK("pose_step", 0, s);
    }
    return LVDGS_OK;
}
