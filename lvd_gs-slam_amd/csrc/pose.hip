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

__global__ void pose_step_kernel(PoseStepParams pp) {
    const lvdgs_pose_step_args &a = pp.a;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float *st = a.state;  // [0..15]: (m, v) of rot xyz, trans xyz, a, b ; [16]: step count ; [17]: converged (sticky) ; [18]: iterations applied
    if (st[17] != 0.f) return;  // already converged: the host may have run ahead
    const float step = st[16] + 1.f;
    st[16] = step;
    st[18] = step;
    // beta^step by squaring (the step count is a small integer; the general pow() is several hundred dependent
    // double-precision instructions on this one thread -- a third of the launch).  It can differ from Python's
    // beta ** step in the last bits of the double, which the float32 step size and denominator never see.
    const double bc1 = 1.0 - int_pow(a.beta1, (unsigned)step);
    const double bc2_sqrt = sqrt(1.0 - int_pow(a.beta2, (unsigned)step));
    // ---- Adam (a parameter without a gradient is skipped, moments included, as torch.optim.Adam does) ----
    const float *g_rot = a.grad_tau ? a.grad_tau + 3 : a.grad_rot, *g_trans = a.grad_tau ? a.grad_tau : a.grad_trans;
    float rot[3] = {0.f, 0.f, 0.f}, trans[3] = {0.f, 0.f, 0.f};
    if (a.R) {
        for (int k = 0; k < 3; k++) {
            rot[k] = g_rot ? adam_update(a.cam_rot_delta[k], g_rot[k], st + 2 * k, st + 2 * k + 1, a.lr_rot, a.beta1, a.beta2, a.eps, bc1, bc2_sqrt)
                           : a.cam_rot_delta[k];
            trans[k] = g_trans ? adam_update(a.cam_trans_delta[k], g_trans[k], st + 6 + 2 * k, st + 6 + 2 * k + 1, a.lr_trans, a.beta1, a.beta2, a.eps, bc1, bc2_sqrt)
                               : a.cam_trans_delta[k];
        }
    }
    if (a.exposure_a && a.grad_exposure_a) *a.exposure_a = adam_update(*a.exposure_a, *a.grad_exposure_a, st + 12, st + 13, a.lr_exposure, a.beta1, a.beta2, a.eps, bc1, bc2_sqrt);
    if (a.exposure_b && a.grad_exposure_b) *a.exposure_b = adam_update(*a.exposure_b, *a.grad_exposure_b, st + 14, st + 15, a.lr_exposure, a.beta1, a.beta2, a.eps, bc1, bc2_sqrt);
    if (!a.R) return;  // exposure only
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
    float R0[9], T0[3], R1[9], T1[3];
    for (int i = 0; i < 9; i++) R0[i] = a.R[i];
    for (int i = 0; i < 3; i++) T0[i] = a.T[i];
    mat3_mul(dR, R0, R1);
    for (int i = 0; i < 3; i++) T1[i] = dR[3 * i] * T0[0] + dR[3 * i + 1] * T0[1] + dR[3 * i + 2] * T0[2] + dt[i];
    for (int i = 0; i < 9; i++) a.R[i] = R1[i];
    for (int i = 0; i < 3; i++) a.T[i] = T1[i];
    const float tau_norm = sqrtf(trans[0] * trans[0] + trans[1] * trans[1] + trans[2] * trans[2] + rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
    if (tau_norm < a.converged_threshold) st[17] = 1.f;
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
                for (int k = 0; k < 4; k++) s += view[4 * i + k] * a.projmatrix_raw[4 * k + j];
                a.projmatrix[4 * i + j] = s;
            }
    }
    if (a.campos)
        for (int i = 0; i < 3; i++) a.campos[i] = -(R1[i] * T1[0] + R1[3 + i] * T1[1] + R1[6 + i] * T1[2]);
}

}  // namespace
}  // namespace lvdgs

using namespace lvdgs;

extern "C" int lvdgs_pose_step(const lvdgs_pose_step_args *a, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!a || !a->state || (a->R ? (!a->T || !a->cam_rot_delta || !a->cam_trans_delta) : (a->T != nullptr))) {
        set_error("pose step: state is NULL, or R is given without T / cam_rot_delta / cam_trans_delta (R and T both NULL = exposure only)");
        return LVDGS_E_INVALID;
    }
    if (!(a->beta1 >= 0.0 && a->beta1 < 1.0 && a->beta2 >= 0.0 && a->beta2 < 1.0)) { set_error("pose step: betas must lie in [0, 1)"); return LVDGS_E_INVALID; }
    PoseStepParams pp{*a};
    ProfScope ps("pose_step", s);
    hipLaunchKernelGGL(pose_step_kernel, dim3(1), dim3(64), 0, s, pp);
    LVDGS_LAUNCH_CHECK("pose_step", 0, s);
    return LVDGS_OK;
}
