/*
 * lvdgs_oracle.c -- CPU restatement of the tile-based differentiable Gaussian rasterizer
 * (with depth / opacity / n_touched outputs and camera-pose gradient) that LVD-GS calls
 * through gaussian_splatting.gaussian_renderer.render().
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it.  The product path (lvd_gs-slam_amd/) never
 * links or imports anything under oracle/.
 *
 * PARITY UNPINNED.  The rasterizer's source (submodules/diff-gaussian-rasterization,
 * reference README.md:43) is not in /root/reference (.MISSING_LARGE_BLOBS:1 -> submodules.zip),
 * no version is pinned anywhere (no .gitmodules / lockfile) and the reference has no tests or
 * golden vectors for it (SURVEY.md sections 0, 4, 8(c)).  This file restates the *published*
 * algorithm -- 3D Gaussian Splatting (Kerbl et al., SIGGRAPH 2023, section 6 + appendix A/C) with the
 * MonoGS extensions (Matsuki et al., CVPR 2024, section 3.3: depth and opacity images, per-Gaussian
 * touch count, analytic Jacobian of the render w.r.t. a left SE(3) perturbation of the camera) --
 * anchored on the reference's call sites:
 *   render(viewpoint, gaussians, pipe, bg) -> dict keys      utils/slam_backend.py:98-116
 *   viewmatrix / projmatrix layouts (row-vector, transposed) utils/camera_utils.py:106-116
 *   projection_matrix = getProjectionMatrix2(znear=.01, ...) utils/slam_frontend.py:1743-1749
 *   pose delta tau = [rho, theta], T <- Exp(tau) T           utils/pose_utils.py:70-87
 *   consumers of depth / opacity / n_touched                 utils/slam_utils.py:42-134,
 *                                                            utils/slam_backend.py:147,311-315
 * It is pinned only internally: against a dense float64 autograd formulation
 * (tests/ref_torch.py) and finite differences.  Constants that upstream is believed to use but
 * that cannot be verified here are named below and marked UNPINNED.
 *
 * Build:  make -C oracle        (two libraries: real = float and real = double)
 * All floating-point expressions that decide integers (radius, tile rectangle, depth bits) are
 * written as plain IEEE operations in a fixed order; compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_F64
typedef double real;
#define R_EXP exp
#define R_SQRT sqrt
#define R_CEIL ceil
#define R_FABS fabs
#else
typedef float real;
#define R_EXP expf
#define R_SQRT sqrtf
#define R_CEIL ceilf
#define R_FABS fabsf
#endif

/* ---- constants (UNPINNED: recalled from the published implementations) ---- */
#define TILE_X 16
#define TILE_Y 16
#define NEAR_CULL ((real)0.2)       /* view-space z at or below this is culled            */
#define HOMOG_EPS ((real)0.0000001) /* added to w before the perspective divide           */
#define FOV_GUARD ((real)1.3)       /* clamp of x/z, y/z to 1.3 * tan(fov/2) in the EWA J  */
#define LOWPASS ((real)0.3)         /* added to the 2D covariance diagonal                */
#define LAMBDA_FLOOR ((real)0.1)    /* floor under the eigenvalue discriminant            */
#define ALPHA_MAX ((real)0.99)
#define ALPHA_MIN ((real)(1.0 / 255.0))
#define T_STOP ((real)0.0001)       /* stop before T would fall below this                */
#define T_TOUCH ((real)0.5)         /* a Gaussian "touches" a pixel if T after it > 0.5   */

static const real SH_C0 = (real)0.28209479177387814;
static const real SH_C1 = (real)0.4886025119029199;
static const real SH_C2[5] = {(real)1.0925484305920792, (real)-1.0925484305920792, (real)0.31539156525252005,
                              (real)-1.0925484305920792, (real)0.5462742152960396};
static const real SH_C3[7] = {(real)-0.5900435899266435, (real)2.890611442640554, (real)-0.4570457994644658,
                              (real)0.3731763325901154,  (real)-0.4570457994644658, (real)1.445305721320277,
                              (real)-0.5900435899266435};

typedef struct {
    /* sizes */
    int32_t N, W, H, sh_degree, M; /* M = SH coefficients per Gaussian (>= (deg+1)^2) */
    int32_t prefiltered;
    real tanfovx, tanfovy, scale_modifier;
    /* inputs (row-major, contiguous) */
    const real *means3D;        /* N*3 */
    const real *scales;         /* N*3 or NULL when cov3D_precomp */
    const real *rotations;      /* N*4 (r,x,y,z) or NULL */
    const real *opacities;      /* N */
    const real *shs;            /* N*M*3 or NULL */
    const real *colors_precomp; /* N*3 or NULL */
    const real *cov3D_precomp;  /* N*6 or NULL */
    const real *viewmatrix;     /* 16, row-vector layout: p_view = [p 1] * V   */
    const real *projmatrix;     /* 16, full projection = V * P                 */
    const real *projmatrix_raw; /* 16, P alone (pose gradient)                 */
    const real *campos;         /* 3 */
    const real *bg;             /* 3 */
    /* forward outputs (allocated by oracle_forward, freed by oracle_free) */
    int64_t num_rendered;
    real *out_color;   /* 3*H*W */
    real *out_depth;   /* H*W   */
    real *out_opacity; /* H*W   */
    int32_t *radii;    /* N */
    int32_t *n_touched; /* N */
    /* forward internals exposed for parity checks */
    real *means2D;        /* N*2 pixel coordinates */
    real *depths;         /* N */
    real *conic_opacity;  /* N*4 */
    real *rgb;            /* N*3 */
    real *cov3D;          /* N*6 */
    uint8_t *clamped;     /* N*3 */
    uint32_t *tiles_touched; /* N */
    int32_t *rect;           /* N*4: x0,y0,x1,y1 (tiles) */
    uint64_t *keys_sorted;   /* D: (tile << 32) | depth bits (float32 bits also in the f64 build) */
    uint32_t *ids_sorted;    /* D */
    uint32_t *ranges;        /* T*2 */
    real *final_T;           /* P */
    uint32_t *n_contrib;     /* P */
    uint8_t *fragile;        /* P: a threshold comparison on this pixel was within 1e-5 relative */
    uint32_t *n_contrib_lo;  /* P: n_contrib had every such comparison that decides the last contributor gone the other way: */
    uint32_t *n_contrib_hi;  /* P: ... the smallest / the largest value an implementation that rounds differently may report */
    /* backward inputs */
    const real *dL_dcolor;   /* 3*H*W */
    const real *dL_ddepth;   /* H*W or NULL */
    const real *dL_dopacity_img; /* H*W or NULL */
    /* backward outputs */
    real *dL_dmeans3D;  /* N*3 */
    real *dL_dmeans2D;  /* N*3: gradient w.r.t. NDC x,y (what viewspace_points.grad holds); z = 0 */
    real *dL_dscales;   /* N*3 */
    real *dL_drotations;/* N*4 */
    real *dL_dopacity;  /* N */
    real *dL_dcolors;   /* N*3 (w.r.t. colors_precomp, or the SH-evaluated rgb) */
    real *dL_dshs;      /* N*M*3 */
    real *dL_dcov3D;    /* N*6 */
    real *dL_dtau;      /* 6: [rho, theta] */
} oracle_ctx;

static void *zalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

/* OpenMP (liblvdgs_oracle_f32_omp.so only, built with -fopenmp -DORACLE_OMP): the same loops spread over the host's cores,
 * for bench.py's cpu_baseline leg -- "the CPU path on all cores of the box".  The scalar builds, which the tests check
 * the kernels against, do not see a single pragma or a single changed statement: every ORACLE_OMP difference below is a
 * different ORDER of the same additions (per-tile partial sums, atomics), never a different formula. */
#ifdef ORACLE_OMP
#include <omp.h>
#define OMP_PRAGMA(x) _Pragma(#x)
int oracle_threads(void) { return omp_get_max_threads(); }
#else
#define OMP_PRAGMA(x)
int oracle_threads(void) { return 1; }
#endif

/* ------------------------------------------------------------------------------------------ */
/* per-Gaussian projection                                                                     */

static void xform4x3(const real *p, const real *m, real *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform4x4(const real *p, const real *m, real *o) {
    xform4x3(p, m, o);
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

static void quat_rot(const real *q, real R[3][3]) {
    real r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = (real)1 - (real)2 * (y * y + z * z); R[0][1] = (real)2 * (x * y - r * z); R[0][2] = (real)2 * (x * z + r * y);
    R[1][0] = (real)2 * (x * y + r * z); R[1][1] = (real)1 - (real)2 * (x * x + z * z); R[1][2] = (real)2 * (y * z - r * x);
    R[2][0] = (real)2 * (x * z - r * y); R[2][1] = (real)2 * (y * z + r * x); R[2][2] = (real)1 - (real)2 * (x * x + y * y);
}

/* Sigma = R diag(s)^2 R^T, stored as xx,xy,xz,yy,yz,zz */
static void cov3d_from_scale_rot(const real *s, real mod, const real *q, real *c6) {
    real R[3][3], M[3][3];
    quat_rot(q, R);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M[i][j] = R[i][j] * (mod * s[j]);
    c6[0] = M[0][0] * M[0][0] + M[0][1] * M[0][1] + M[0][2] * M[0][2];
    c6[1] = M[0][0] * M[1][0] + M[0][1] * M[1][1] + M[0][2] * M[1][2];
    c6[2] = M[0][0] * M[2][0] + M[0][1] * M[2][1] + M[0][2] * M[2][2];
    c6[3] = M[1][0] * M[1][0] + M[1][1] * M[1][1] + M[1][2] * M[1][2];
    c6[4] = M[1][0] * M[2][0] + M[1][1] * M[2][1] + M[1][2] * M[2][2];
    c6[5] = M[2][0] * M[2][0] + M[2][1] * M[2][1] + M[2][2] * M[2][2];
}

/* EWA: T = J * Wrot (2x3), cov2 = T Sigma T^T.  t = view-space mean with x/z, y/z clamped. */
typedef struct { real T[2][3]; real t[3]; int clx, cly; real fx, fy; } ewa_t;

static void ewa_setup(const real *pv, const real *V, real fx, real fy, real tanx, real tany, ewa_t *e) {
    real limx = FOV_GUARD * tanx, limy = FOV_GUARD * tany;
    real txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    e->clx = (txtz < -limx) || (txtz > limx);
    e->cly = (tytz < -limy) || (tytz > limy);
    real cx = txtz < -limx ? -limx : (txtz > limx ? limx : txtz);
    real cy = tytz < -limy ? -limy : (tytz > limy ? limy : tytz);
    e->t[0] = cx * pv[2]; e->t[1] = cy * pv[2]; e->t[2] = pv[2];
    e->fx = fx; e->fy = fy;
    real j00 = fx / e->t[2], j02 = -(fx * e->t[0]) / (e->t[2] * e->t[2]);
    real j11 = fy / e->t[2], j12 = -(fy * e->t[1]) / (e->t[2] * e->t[2]);
    /* Wrot[r][c] = V[4*c + r] : rows of the world->camera rotation */
    for (int c = 0; c < 3; c++) {
        real w0 = V[4 * c + 0], w1 = V[4 * c + 1], w2 = V[4 * c + 2];
        e->T[0][c] = j00 * w0 + j02 * w2;
        e->T[1][c] = j11 * w1 + j12 * w2;
    }
}

static void sym_from6(const real *c6, real S[3][3]) {
    S[0][0] = c6[0]; S[0][1] = S[1][0] = c6[1]; S[0][2] = S[2][0] = c6[2];
    S[1][1] = c6[3]; S[1][2] = S[2][1] = c6[4]; S[2][2] = c6[5];
}

static void cov2d(const ewa_t *e, const real *c6, real *a, real *b, real *c) {
    real S[3][3], TS[2][3];
    sym_from6(c6, S);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) TS[i][j] = e->T[i][0] * S[0][j] + e->T[i][1] * S[1][j] + e->T[i][2] * S[2][j];
    *a = TS[0][0] * e->T[0][0] + TS[0][1] * e->T[0][1] + TS[0][2] * e->T[0][2] + LOWPASS;
    *b = TS[0][0] * e->T[1][0] + TS[0][1] * e->T[1][1] + TS[0][2] * e->T[1][2];
    *c = TS[1][0] * e->T[1][0] + TS[1][1] * e->T[1][1] + TS[1][2] * e->T[1][2] + LOWPASS;
}

static void sh_basis(int deg, const real *d, real *B) {
    real x = d[0], y = d[1], z = d[2];
    B[0] = SH_C0;
    if (deg > 0) { B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x; }
    if (deg > 1) {
        real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * ((real)2 * zz - xx - yy);
        B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
        if (deg > 2) {
            B[9] = SH_C3[0] * y * ((real)3 * xx - yy); B[10] = SH_C3[1] * xy * z;
            B[11] = SH_C3[2] * y * ((real)4 * zz - xx - yy);
            B[12] = SH_C3[3] * z * ((real)2 * zz - (real)3 * xx - (real)3 * yy);
            B[13] = SH_C3[4] * x * ((real)4 * zz - xx - yy); B[14] = SH_C3[5] * z * (xx - yy);
            B[15] = SH_C3[6] * x * (xx - (real)3 * yy);
        }
    }
}

/* d basis / d (x,y,z) of the unit direction, per coefficient */
static void sh_basis_grad(int deg, const real *d, real G[16][3]) {
    real x = d[0], y = d[1], z = d[2];
    memset(G, 0, sizeof(real) * 16 * 3);
    if (deg > 0) { G[1][1] = -SH_C1; G[2][2] = SH_C1; G[3][0] = -SH_C1; }
    if (deg > 1) {
        G[4][0] = SH_C2[0] * y; G[4][1] = SH_C2[0] * x;
        G[5][1] = SH_C2[1] * z; G[5][2] = SH_C2[1] * y;
        G[6][0] = SH_C2[2] * (real)-2 * x; G[6][1] = SH_C2[2] * (real)-2 * y; G[6][2] = SH_C2[2] * (real)4 * z;
        G[7][0] = SH_C2[3] * z; G[7][2] = SH_C2[3] * x;
        G[8][0] = SH_C2[4] * (real)2 * x; G[8][1] = SH_C2[4] * (real)-2 * y;
    }
    if (deg > 2) {
        real xx = x * x, yy = y * y, zz = z * z;
        G[9][0] = SH_C3[0] * (real)6 * x * y; G[9][1] = SH_C3[0] * ((real)3 * xx - (real)3 * yy);
        G[10][0] = SH_C3[1] * y * z; G[10][1] = SH_C3[1] * x * z; G[10][2] = SH_C3[1] * x * y;
        G[11][0] = SH_C3[2] * (real)-2 * x * y; G[11][1] = SH_C3[2] * ((real)4 * zz - xx - (real)3 * yy);
        G[11][2] = SH_C3[2] * (real)8 * y * z;
        G[12][0] = SH_C3[3] * (real)-6 * x * z; G[12][1] = SH_C3[3] * (real)-6 * y * z;
        G[12][2] = SH_C3[3] * ((real)6 * zz - (real)3 * xx - (real)3 * yy);
        G[13][0] = SH_C3[4] * ((real)4 * zz - (real)3 * xx - yy); G[13][1] = SH_C3[4] * (real)-2 * x * y;
        G[13][2] = SH_C3[4] * (real)8 * x * z;
        G[14][0] = SH_C3[5] * (real)2 * x * z; G[14][1] = SH_C3[5] * (real)-2 * y * z; G[14][2] = SH_C3[5] * (xx - yy);
        G[15][0] = SH_C3[6] * ((real)3 * xx - (real)3 * yy); G[15][1] = SH_C3[6] * (real)-6 * x * y;
    }
}

static uint32_t depth_bits(real z) {
    float f = (float)z; uint32_t u; memcpy(&u, &f, 4); return u;
}

/* stable merge sort of (key,id) pairs by key */
static void merge_sort_pairs(uint64_t *k, uint32_t *v, uint64_t *tk, uint32_t *tv, int64_t n) {
    for (int64_t w = 1; w < n; w *= 2) {
        OMP_PRAGMA(omp parallel for schedule(static))
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                if (k[j] < k[i]) { tk[o] = k[j]; tv[o++] = v[j++]; }
                else { tk[o] = k[i]; tv[o++] = v[i++]; }
            }
            while (i < mid) { tk[o] = k[i]; tv[o++] = v[i++]; }
            while (j < hi) { tk[o] = k[j]; tv[o++] = v[j++]; }
        }
        memcpy(k, tk, sizeof(uint64_t) * n); memcpy(v, tv, sizeof(uint32_t) * n);
    }
}

static int near_rel(real a, real b) { return R_FABS(a - b) <= (real)1e-5 * R_FABS(b); }

/* ------------------------------------------------------------------------------------------ */
int oracle_forward(oracle_ctx *c) {
    const int N = c->N, W = c->W, H = c->H, P = W * H;
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y, NT = gx * gy;
    const real fx = (real)W / ((real)2 * c->tanfovx), fy = (real)H / ((real)2 * c->tanfovy);

    c->out_color = zalloc(3 * (size_t)P, sizeof(real)); c->out_depth = zalloc(P, sizeof(real));
    c->out_opacity = zalloc(P, sizeof(real)); c->radii = zalloc(N, 4); c->n_touched = zalloc(N, 4);
    c->means2D = zalloc(2 * (size_t)N, sizeof(real)); c->depths = zalloc(N, sizeof(real));
    c->conic_opacity = zalloc(4 * (size_t)N, sizeof(real)); c->rgb = zalloc(3 * (size_t)N, sizeof(real));
    c->cov3D = zalloc(6 * (size_t)N, sizeof(real)); c->clamped = zalloc(3 * (size_t)N, 1);
    c->tiles_touched = zalloc(N, 4); c->rect = zalloc(4 * (size_t)N, 4);
    c->ranges = zalloc(2 * (size_t)NT, 4); c->final_T = zalloc(P, sizeof(real));
    c->n_contrib = zalloc(P, 4); c->fragile = zalloc(P, 1);
    c->n_contrib_lo = zalloc(P, 4); c->n_contrib_hi = zalloc(P, 4);

    /* ---- per-Gaussian projection ---- */
    int64_t D = 0;
    OMP_PRAGMA(omp parallel for reduction(+:D) schedule(static))
    for (int i = 0; i < N; i++) {
        const real *p = c->means3D + 3 * i;
        real pv[3], ph[4];
        xform4x3(p, c->viewmatrix, pv);
        if (pv[2] <= NEAR_CULL) continue;
        xform4x4(p, c->projmatrix, ph);
        real pw = (real)1 / (ph[3] + HOMOG_EPS);
        real ndc[2] = {ph[0] * pw, ph[1] * pw};
        real *c6 = c->cov3D + 6 * i;
        if (c->cov3D_precomp) memcpy(c6, c->cov3D_precomp + 6 * i, 6 * sizeof(real));
        else cov3d_from_scale_rot(c->scales + 3 * i, c->scale_modifier, c->rotations + 4 * i, c6);
        ewa_t e; real ca, cb, cc;
        ewa_setup(pv, c->viewmatrix, fx, fy, c->tanfovx, c->tanfovy, &e);
        cov2d(&e, c6, &ca, &cb, &cc);
        real det = ca * cc - cb * cb;
        if (det == (real)0) continue;
        real det_inv = (real)1 / det;
        real conic[3] = {cc * det_inv, -cb * det_inv, ca * det_inv};
        real mid = (real)0.5 * (ca + cc);
        real disc = mid * mid - det; if (disc < LAMBDA_FLOOR) disc = LAMBDA_FLOOR;
        real l1 = mid + R_SQRT(disc), l2 = mid - R_SQRT(disc);
        real lmax = l1 > l2 ? l1 : l2;
        int radius = (int)R_CEIL((real)3 * R_SQRT(lmax));
        real px = ((ndc[0] + (real)1) * (real)W - (real)1) * (real)0.5;
        real py = ((ndc[1] + (real)1) * (real)H - (real)1) * (real)0.5;
        int x0 = (int)((px - (real)radius) / (real)TILE_X), y0 = (int)((py - (real)radius) / (real)TILE_Y);
        int x1 = (int)((px + (real)radius + (real)(TILE_X - 1)) / (real)TILE_X);
        int y1 = (int)((py + (real)radius + (real)(TILE_Y - 1)) / (real)TILE_Y);
        x0 = x0 < 0 ? 0 : (x0 > gx ? gx : x0); x1 = x1 < 0 ? 0 : (x1 > gx ? gx : x1);
        y0 = y0 < 0 ? 0 : (y0 > gy ? gy : y0); y1 = y1 < 0 ? 0 : (y1 > gy ? gy : y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;

        real *rgb = c->rgb + 3 * i;
        if (c->colors_precomp) memcpy(rgb, c->colors_precomp + 3 * i, 3 * sizeof(real));
        else {
            real d[3] = {p[0] - c->campos[0], p[1] - c->campos[1], p[2] - c->campos[2]};
            real inv = (real)1 / R_SQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] *= inv; d[1] *= inv; d[2] *= inv;
            real B[16]; sh_basis(c->sh_degree, d, B);
            int nb = (c->sh_degree + 1) * (c->sh_degree + 1);
            for (int ch = 0; ch < 3; ch++) {
                real v = 0;
                for (int k = 0; k < nb; k++) v += B[k] * c->shs[((size_t)i * c->M + k) * 3 + ch];
                v += (real)0.5;
                c->clamped[3 * i + ch] = v < 0;
                rgb[ch] = v < 0 ? 0 : v;
            }
        }
        c->depths[i] = pv[2]; c->radii[i] = radius;
        c->means2D[2 * i] = px; c->means2D[2 * i + 1] = py;
        c->conic_opacity[4 * i] = conic[0]; c->conic_opacity[4 * i + 1] = conic[1];
        c->conic_opacity[4 * i + 2] = conic[2]; c->conic_opacity[4 * i + 3] = c->opacities[i];
        c->rect[4 * i] = x0; c->rect[4 * i + 1] = y0; c->rect[4 * i + 2] = x1; c->rect[4 * i + 3] = y1;
        c->tiles_touched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
        D += c->tiles_touched[i];
    }
    c->num_rendered = D;

    /* ---- duplicate with keys, sort, ranges ---- */
    uint64_t *keys = zalloc(D, 8), *tk = zalloc(D, 8);
    uint32_t *ids = zalloc(D, 4), *tv = zalloc(D, 4);
#ifdef ORACLE_OMP
    int64_t *first = zalloc((size_t)N + 1, 8);
    for (int i = 0; i < N; i++) first[i + 1] = first[i] + (c->radii[i] > 0 ? c->tiles_touched[i] : 0);
    OMP_PRAGMA(omp parallel for schedule(static))
    for (int i = 0; i < N; i++) {
        if (c->radii[i] <= 0) continue;
        const int32_t *r = c->rect + 4 * i;
        int64_t off = first[i];
        for (int y = r[1]; y < r[3]; y++)
            for (int x = r[0]; x < r[2]; x++) {
                keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | depth_bits(c->depths[i]);
                ids[off++] = (uint32_t)i;
            }
    }
    free(first);
#else
    int64_t off = 0;
    for (int i = 0; i < N; i++) {
        if (c->radii[i] <= 0) continue;
        const int32_t *r = c->rect + 4 * i;
        for (int y = r[1]; y < r[3]; y++)
            for (int x = r[0]; x < r[2]; x++) {
                keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | depth_bits(c->depths[i]);
                ids[off++] = (uint32_t)i;
            }
    }
#endif
    merge_sort_pairs(keys, ids, tk, tv, D);
    free(tk); free(tv);
    c->keys_sorted = keys; c->ids_sorted = ids;
    for (int64_t k = 0; k < D; k++) {
        uint32_t t = (uint32_t)(keys[k] >> 32);
        if (k == 0 || (uint32_t)(keys[k - 1] >> 32) != t) c->ranges[2 * t] = (uint32_t)k;
        if (k == D - 1 || (uint32_t)(keys[k + 1] >> 32) != t) c->ranges[2 * t + 1] = (uint32_t)(k + 1);
    }

    /* ---- per-pixel front-to-back compositing ---- */
    OMP_PRAGMA(omp parallel for collapse(2) schedule(dynamic, 4))
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            uint32_t beg = c->ranges[2 * (ty * gx + tx)], end = c->ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < TILE_Y; ly++)
                for (int lx = 0; lx < TILE_X; lx++) {
                    int x = tx * TILE_X + lx, y = ty * TILE_Y + ly;
                    if (x >= W || y >= H) continue;
                    int pix = y * W + x;
                    real T = 1, C[3] = {0, 0, 0}, Dp = 0;
                    uint32_t contributor = 0, last = 0; uint8_t frag = 0;
                    /* The last contributor under every way the near-threshold comparisons can fall.  An entry whose alpha is
                       within 1e-5 of 1/255 is composited by one implementation and skipped by another: skipped here, it may
                       be the other's last contributor (last_hi); composited here as the last one, the other may stop at the
                       one before (last_lo).  A transmittance within 1e-5 of the 1e-4 stop: stopped here, the other composites
                       this entry and stops at the next that qualifies (every alpha >= 1/255 takes T below the threshold for
                       good); not stopped here, the other stops before it. */
                    uint32_t last_lo = 0, last_hi = 0;
                    for (uint32_t k = beg; k < end; k++) {
                        contributor++;
                        uint32_t g = ids[k];
                        real dx = c->means2D[2 * g] - (real)x, dy = c->means2D[2 * g + 1] - (real)y;
                        const real *co = c->conic_opacity + 4 * g;
                        real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0) continue;
                        real raw = co[3] * R_EXP(power);
                        real alpha = raw < ALPHA_MAX ? raw : ALPHA_MAX;
                        int near_a = near_rel(alpha, ALPHA_MIN);
                        frag |= near_a;
                        if (alpha < ALPHA_MIN) { if (near_a) last_hi = contributor; continue; }
                        real test_T = T * ((real)1 - alpha);
                        int near_t = near_rel(test_T, T_STOP);
                        frag |= near_t;
                        if (test_T < T_STOP) { if (near_t) last_hi = contributor; break; }
                        real w = alpha * T;
                        for (int ch = 0; ch < 3; ch++) C[ch] += c->rgb[3 * g + ch] * w;
                        Dp += c->depths[g] * w;
                        frag |= near_rel(test_T, T_TOUCH);
                        if (test_T > T_TOUCH) {
                            OMP_PRAGMA(omp atomic)
                            c->n_touched[g]++;
                        }
                        T = test_T; last = contributor;
                        if (!near_a && !near_t) last_lo = contributor;
                    }
                    c->final_T[pix] = T; c->n_contrib[pix] = last; c->fragile[pix] = frag;
                    c->n_contrib_lo[pix] = last_lo; c->n_contrib_hi[pix] = last_hi > last ? last_hi : last;
                    for (int ch = 0; ch < 3; ch++) c->out_color[(size_t)ch * P + pix] = C[ch] + T * c->bg[ch];
                    c->out_depth[pix] = Dp; c->out_opacity[pix] = (real)1 - T;
                }
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Backward.  Per pixel back-to-front, as published: recover T by dividing out (1-alpha), keep
 * the running "colour behind" recurrences, accumulate per-Gaussian 2D gradients; then chain
 * each Gaussian through conic -> cov2D -> (cov3D, view-space mean, rotation) -> parameters and
 * the SE(3) perturbation of the camera.  Per-Gaussian sums are kept in double in both builds
 * (the GPU sums float32 in a different order; double keeps the oracle's own noise out). */
int oracle_backward(oracle_ctx *c) {
    const int N = c->N, W = c->W, H = c->H, P = W * H;
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const real fx = (real)W / ((real)2 * c->tanfovx), fy = (real)H / ((real)2 * c->tanfovy);
    const int M = c->M;

    c->dL_dmeans3D = zalloc(3 * (size_t)N, sizeof(real)); c->dL_dmeans2D = zalloc(3 * (size_t)N, sizeof(real));
    c->dL_dscales = zalloc(3 * (size_t)N, sizeof(real)); c->dL_drotations = zalloc(4 * (size_t)N, sizeof(real));
    c->dL_dopacity = zalloc(N, sizeof(real)); c->dL_dcolors = zalloc(3 * (size_t)N, sizeof(real));
    c->dL_dshs = zalloc((size_t)N * (M > 0 ? M : 1) * 3, sizeof(real)); c->dL_dcov3D = zalloc(6 * (size_t)N, sizeof(real));
    c->dL_dtau = zalloc(6, sizeof(real));

    /* 2D accumulators: [0,1] d/d pixel-mean, [2..4] d/d conic (a, b, c as true partials),
     * [5] d/d opacity, [6..8] d/d rgb, [9] d/d view depth */
    double *acc = zalloc(10 * (size_t)N, sizeof(double));

#ifdef ORACLE_OMP
    /* per tile: sums by list position in a buffer of the thread's own, added to the shared per-Gaussian sums once per
     * (tile, Gaussian) with atomics */
#define ACC(g, k, j) local[10 * (size_t)((k) - beg) + (j)]
    OMP_PRAGMA(omp parallel for collapse(2) schedule(dynamic, 4))
#else
#define ACC(g, k, j) acc[10 * (size_t)(g) + (j)]
#endif
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            uint32_t beg = c->ranges[2 * (ty * gx + tx)];
#ifdef ORACLE_OMP
            const uint32_t end_ = c->ranges[2 * (ty * gx + tx) + 1];
            double *local = calloc(10 * (size_t)(end_ > beg ? end_ - beg : 1), sizeof(double));
#endif
            for (int ly = 0; ly < TILE_Y; ly++)
                for (int lx = 0; lx < TILE_X; lx++) {
                    int x = tx * TILE_X + lx, y = ty * TILE_Y + ly;
                    if (x >= W || y >= H) continue;
                    int pix = y * W + x;
                    const real T_final = c->final_T[pix];
                    real T = T_final;
                    real gC[3] = {c->dL_dcolor[pix], c->dL_dcolor[(size_t)P + pix], c->dL_dcolor[2 * (size_t)P + pix]};
                    real gD = c->dL_ddepth ? c->dL_ddepth[pix] : 0;
                    real gO = c->dL_dopacity_img ? c->dL_dopacity_img[pix] : 0;
                    real behind[3] = {0, 0, 0}, behind_d = 0, last_alpha = 0, last_c[3] = {0, 0, 0}, last_d = 0;
                    real bg_dot = c->bg[0] * gC[0] + c->bg[1] * gC[1] + c->bg[2] * gC[2];
                    for (int64_t k = (int64_t)beg + c->n_contrib[pix] - 1; k >= (int64_t)beg; k--) {
                        uint32_t g = c->ids_sorted[k];
                        real dx = c->means2D[2 * g] - (real)x, dy = c->means2D[2 * g + 1] - (real)y;
                        const real *co = c->conic_opacity + 4 * g;
                        real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0) continue;
                        real G = R_EXP(power);
                        real raw = co[3] * G;
                        real alpha = raw < ALPHA_MAX ? raw : ALPHA_MAX;
                        if (alpha < ALPHA_MIN) continue;
                        T = T / ((real)1 - alpha);
                        real w = alpha * T;
                        real dL_dalpha = 0;
                        for (int ch = 0; ch < 3; ch++) {
                            behind[ch] = last_alpha * last_c[ch] + ((real)1 - last_alpha) * behind[ch];
                            last_c[ch] = c->rgb[3 * g + ch];
                            dL_dalpha += (last_c[ch] - behind[ch]) * gC[ch];
                            ACC(g, k, 6 + ch) += w * gC[ch];
                        }
                        behind_d = last_alpha * last_d + ((real)1 - last_alpha) * behind_d;
                        last_d = c->depths[g];
                        dL_dalpha += (last_d - behind_d) * gD;
                        ACC(g, k, 9) += w * gD;
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        /* background and the opacity image both see alpha only through T_final */
                        dL_dalpha += (-T_final / ((real)1 - alpha)) * (bg_dot - gO);
                        /* UNPINNED: the published backward differentiates alpha = o*G and ignores the
                         * min(0.99, .) clamp (the gradient is not masked where the clamp is active). */
                        real dL_dG = co[3] * dL_dalpha;
                        ACC(g, k, 5) += G * dL_dalpha;
                        /* power = -1/2 (a dx^2 + c dy^2) - b dx dy, d = mean - pixel */
                        real dG_ddx = G * (-co[0] * dx - co[1] * dy), dG_ddy = G * (-co[2] * dy - co[1] * dx);
                        ACC(g, k, 0) += dL_dG * dG_ddx;
                        ACC(g, k, 1) += dL_dG * dG_ddy;
                        ACC(g, k, 2) += dL_dG * G * ((real)-0.5 * dx * dx);
                        ACC(g, k, 3) += dL_dG * G * (-dx * dy);
                        ACC(g, k, 4) += dL_dG * G * ((real)-0.5 * dy * dy);
                    }
                }
#ifdef ORACLE_OMP
            for (uint32_t k = beg; k < end_; k++) {
                const uint32_t g = c->ids_sorted[k];
                for (int j = 0; j < 10; j++) {
                    const double v = local[10 * (size_t)(k - beg) + j];
                    if (v != 0) {
                        OMP_PRAGMA(omp atomic)
                        acc[10 * (size_t)g + j] += v;
                    }
                }
            }
            free(local);
#endif
        }
#undef ACC

    /* ---- per-Gaussian chain ---- */
    double tau[6] = {0, 0, 0, 0, 0, 0};
    const real *V = c->viewmatrix, *PM = c->projmatrix, *PR = c->projmatrix_raw;
    OMP_PRAGMA(omp parallel for reduction(+:tau[:6]) schedule(static))
    for (int i = 0; i < N; i++) {
        if (c->radii[i] <= 0) continue;
        const double *A = acc + 10 * (size_t)i;
        const real *p = c->means3D + 3 * i;
        real pv[3], ph[4];
        xform4x3(p, V, pv); xform4x4(p, PM, ph);
        real g_pix[2] = {(real)A[0], (real)A[1]};
        real g_con[3] = {(real)A[2], (real)A[3], (real)A[4]};
        real g_rgb[3] = {(real)A[6], (real)A[7], (real)A[8]};
        real g_z = (real)A[9];
        c->dL_dopacity[i] = (real)A[5];
        /* what viewspace_points.grad receives: gradient w.r.t. NDC x,y */
        c->dL_dmeans2D[3 * i] = g_pix[0] * (real)0.5 * (real)W;
        c->dL_dmeans2D[3 * i + 1] = g_pix[1] * (real)0.5 * (real)H;

        real g_pview[3] = {0, 0, g_z}; /* gradient w.r.t. the view-space mean (pose path) */
        real g_tau_sh[3] = {0, 0, 0};  /* d/d rho through the SH view direction */
        real g_world[3] = {0, 0, 0};   /* gradient w.r.t. the world mean */

        /* -- colour -- */
        if (c->colors_precomp) { for (int ch = 0; ch < 3; ch++) c->dL_dcolors[3 * i + ch] = g_rgb[ch]; }
        else {
            for (int ch = 0; ch < 3; ch++) { if (c->clamped[3 * i + ch]) g_rgb[ch] = 0; c->dL_dcolors[3 * i + ch] = g_rgb[ch]; }
            real d[3] = {p[0] - c->campos[0], p[1] - c->campos[1], p[2] - c->campos[2]};
            real len = R_SQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            real u[3] = {d[0] / len, d[1] / len, d[2] / len};
            real B[16], Gb[16][3]; sh_basis(c->sh_degree, u, B); sh_basis_grad(c->sh_degree, u, Gb);
            int nb = (c->sh_degree + 1) * (c->sh_degree + 1);
            real g_u[3] = {0, 0, 0};
            for (int k = 0; k < nb; k++)
                for (int ch = 0; ch < 3; ch++) {
                    real s = c->shs[((size_t)i * M + k) * 3 + ch];
                    c->dL_dshs[((size_t)i * M + k) * 3 + ch] = B[k] * g_rgb[ch];
                    for (int a = 0; a < 3; a++) g_u[a] += Gb[k][a] * s * g_rgb[ch];
                }
            /* u = d/|d| : du/dd = (I - u u^T)/|d|, d = mean - camera centre.  The view direction moves with the mean AND
             * with the camera: centre C = -R^T T, and under T_w2c <- Exp(tau) T_w2c (utils/pose_utils.py:70-87)
             * dC/drho = -R^T, dC/dtheta = 0 at tau = 0, so dL/drho += R g_d with g_d = dL/dd (checked against the dense
             * autograd formulation and finite differences in float64, tests/test_oracle_pinning.py).  Zero at SH degree 0. */
            real dot = u[0] * g_u[0] + u[1] * g_u[1] + u[2] * g_u[2];
            for (int a = 0; a < 3; a++) {
                real g_d = (g_u[a] - u[a] * dot) / len;
                g_world[a] += g_d;
                for (int j = 0; j < 3; j++) g_tau_sh[j] += V[4 * a + j] * g_d;
            }
        }

        /* -- depth image: z_view = row 2 of the view transform -- */
        /* (added to g_pview above; mapped to world and pose below) */

        /* -- conic -> cov2D -- */
        const real *c6 = c->cov3D + 6 * i;
        ewa_t e; real ca, cb, cc;
        ewa_setup(pv, V, fx, fy, c->tanfovx, c->tanfovy, &e);
        cov2d(&e, c6, &ca, &cb, &cc);
        real det = ca * cc - cb * cb, di = (real)1 / det;
        /* conic = [cc, -cb, ca]/det.  With Q = conic matrix and Gq its symmetric gradient
         * [[g0, g1/2],[g1/2, g2]]:  dL/dSigma2 = -Q Gq Q  (then off-diagonal counted twice). */
        real Q00 = cc * di, Q01 = -cb * di, Q11 = ca * di;
        real G00 = g_con[0], G01 = (real)0.5 * g_con[1], G11 = g_con[2];
        real QG00 = Q00 * G00 + Q01 * G01, QG01 = Q00 * G01 + Q01 * G11;
        real QG10 = Q01 * G00 + Q11 * G01, QG11 = Q01 * G01 + Q11 * G11;
        real S00 = -(QG00 * Q00 + QG01 * Q01), S01 = -(QG00 * Q01 + QG01 * Q11);
        real S11 = -(QG10 * Q01 + QG11 * Q11);
        /* symmetric gradient matrix of cov2D (entry gradient halves off the diagonal) = [[S00,S01],[S01,S11]] */

        /* -- cov2D = T Sigma T^T -- */
        real Sg[3][3]; sym_from6(c6, Sg);
        real g_S[3][3]; /* dL/dSigma3 (full symmetric matrix form) = T^T Gs T */
        real Gs[2][2] = {{S00, S01}, {S01, S11}};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                real v = 0;
                for (int r = 0; r < 2; r++) for (int s = 0; s < 2; s++) v += e.T[r][a] * Gs[r][s] * e.T[s][b];
                g_S[a][b] = v;
            }
        real g6[6] = {g_S[0][0], (real)2 * g_S[0][1], (real)2 * g_S[0][2], g_S[1][1], (real)2 * g_S[1][2], g_S[2][2]};
        for (int k = 0; k < 6; k++) c->dL_dcov3D[6 * i + k] = g6[k];
        /* dL/dT = 2 Gs T Sigma */
        real TS[2][3], g_T[2][3];
        for (int r = 0; r < 2; r++) for (int b = 0; b < 3; b++) TS[r][b] = e.T[r][0] * Sg[0][b] + e.T[r][1] * Sg[1][b] + e.T[r][2] * Sg[2][b];
        for (int r = 0; r < 2; r++) for (int b = 0; b < 3; b++) g_T[r][b] = (real)2 * (Gs[r][0] * TS[0][b] + Gs[r][1] * TS[1][b]);
        /* T = J Wrot, Wrot[r][c] = V[4c+r]:  dL/dJ = g_T Wrot^T ; dL/dWrot = J^T g_T */
        real g_J00 = 0, g_J02 = 0, g_J11 = 0, g_J12 = 0;
        for (int cidx = 0; cidx < 3; cidx++) {
            real w0 = V[4 * cidx + 0], w1 = V[4 * cidx + 1], w2 = V[4 * cidx + 2];
            g_J00 += g_T[0][cidx] * w0; g_J02 += g_T[0][cidx] * w2;
            g_J11 += g_T[1][cidx] * w1; g_J12 += g_T[1][cidx] * w2;
        }
        real tz = e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        real j00 = fx / tz, j02 = -(fx * e.t[0]) / tz2, j11 = fy / tz, j12 = -(fy * e.t[1]) / tz2;
        real g_W[3][3]; /* rows r, cols c of Wrot */
        for (int cidx = 0; cidx < 3; cidx++) {
            g_W[0][cidx] = j00 * g_T[0][cidx];
            g_W[1][cidx] = j11 * g_T[1][cidx];
            g_W[2][cidx] = j02 * g_T[0][cidx] + j12 * g_T[1][cidx];
        }
        real g_t[3];
        g_t[0] = e.clx ? 0 : -(fx / tz2) * g_J02;
        g_t[1] = e.cly ? 0 : -(fy / tz2) * g_J12;
        g_t[2] = -(fx / tz2) * g_J00 - (fy / tz2) * g_J11 + ((real)2 * fx * e.t[0] / tz3) * g_J02 + ((real)2 * fy * e.t[1] / tz3) * g_J12;
        for (int a = 0; a < 3; a++) g_pview[a] += g_t[a];

        /* -- pixel mean through the full projection -- */
        real pw = (real)1 / (ph[3] + HOMOG_EPS);
        real g_ndc[2] = {g_pix[0] * (real)0.5 * (real)W, g_pix[1] * (real)0.5 * (real)H};
        real g_hom[4] = {g_ndc[0] * pw, g_ndc[1] * pw, 0, -(g_ndc[0] * ph[0] + g_ndc[1] * ph[1]) * pw * pw};
        for (int a = 0; a < 3; a++) /* p_hom[k] = sum_a p[a] PM[4a+k] + PM[12+k] */
            g_world[a] += PM[4 * a + 0] * g_hom[0] + PM[4 * a + 1] * g_hom[1] + PM[4 * a + 3] * g_hom[3];
        /* the same through the raw projection gives the view-space gradient for the pose */
        real g_pview_proj[3];
        for (int a = 0; a < 3; a++) g_pview_proj[a] = PR[4 * a + 0] * g_hom[0] + PR[4 * a + 1] * g_hom[1] + PR[4 * a + 3] * g_hom[3];

        /* view-space gradient (covariance + depth) back to the world mean: p_view[r] = sum_a p[a] V[4a+r] */
        for (int a = 0; a < 3; a++) g_world[a] += V[4 * a + 0] * g_pview[0] + V[4 * a + 1] * g_pview[1] + V[4 * a + 2] * g_pview[2];
        for (int a = 0; a < 3; a++) c->dL_dmeans3D[3 * i + a] = g_world[a];

        /* -- pose: T' = Exp(tau) T.  p_view' = p_view + rho + theta x p_view ;
         *          Wrot' = (I + [theta]x) Wrot  => each column w_c moves by theta x w_c      */
        real gv[3] = {g_pview[0] + g_pview_proj[0], g_pview[1] + g_pview_proj[1], g_pview[2] + g_pview_proj[2]};
        tau[0] += gv[0] + g_tau_sh[0]; tau[1] += gv[1] + g_tau_sh[1]; tau[2] += gv[2] + g_tau_sh[2];
        tau[3] += pv[1] * gv[2] - pv[2] * gv[1];
        tau[4] += pv[2] * gv[0] - pv[0] * gv[2];
        tau[5] += pv[0] * gv[1] - pv[1] * gv[0];
        for (int cidx = 0; cidx < 3; cidx++) {
            real w[3] = {V[4 * cidx + 0], V[4 * cidx + 1], V[4 * cidx + 2]};
            real g[3] = {g_W[0][cidx], g_W[1][cidx], g_W[2][cidx]};
            tau[3] += w[1] * g[2] - w[2] * g[1];
            tau[4] += w[2] * g[0] - w[0] * g[2];
            tau[5] += w[0] * g[1] - w[1] * g[0];
        }

        /* -- Sigma3 = (R S)(R S)^T -> scale, quaternion -- */
        if (!c->cov3D_precomp) {
            const real *s = c->scales + 3 * i, *q = c->rotations + 4 * i;
            real R[3][3]; quat_rot(q, R);
            real sm[3] = {c->scale_modifier * s[0], c->scale_modifier * s[1], c->scale_modifier * s[2]};
            /* M = R diag(sm); dL/dM = 2 g_S M */
            real Mm[3][3], g_M[3][3];
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Mm[a][b] = R[a][b] * sm[b];
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++)
                g_M[a][b] = (real)2 * (g_S[a][0] * Mm[0][b] + g_S[a][1] * Mm[1][b] + g_S[a][2] * Mm[2][b]);
            real g_R[3][3];
            for (int b = 0; b < 3; b++) {
                real v = 0;
                for (int a = 0; a < 3; a++) { v += g_M[a][b] * R[a][b]; g_R[a][b] = g_M[a][b] * sm[b]; }
                c->dL_dscales[3 * i + b] = v * c->scale_modifier;
            }
            real r = q[0], x = q[1], y = q[2], z = q[3];
            c->dL_drotations[4 * i + 0] = (real)2 * (-z * g_R[0][1] + y * g_R[0][2] + z * g_R[1][0] - x * g_R[1][2] - y * g_R[2][0] + x * g_R[2][1]);
            c->dL_drotations[4 * i + 1] = (real)2 * (y * g_R[0][1] + z * g_R[0][2] + y * g_R[1][0] - (real)2 * x * g_R[1][1] - r * g_R[1][2] + z * g_R[2][0] + r * g_R[2][1] - (real)2 * x * g_R[2][2]);
            c->dL_drotations[4 * i + 2] = (real)2 * (-(real)2 * y * g_R[0][0] + x * g_R[0][1] + r * g_R[0][2] + x * g_R[1][0] + z * g_R[1][2] - r * g_R[2][0] + z * g_R[2][1] - (real)2 * y * g_R[2][2]);
            c->dL_drotations[4 * i + 3] = (real)2 * (-(real)2 * z * g_R[0][0] - r * g_R[0][1] + x * g_R[0][2] + r * g_R[1][0] - (real)2 * z * g_R[1][1] + y * g_R[1][2] + x * g_R[2][0] + y * g_R[2][1]);
        }
    }
    for (int k = 0; k < 6; k++) c->dL_dtau[k] = (real)tau[k];
    free(acc);
    return 0;
}

/* frustum test used by GaussianRasterizer.markVisible */
int oracle_mark_visible(int N, const real *means3D, const real *viewmatrix, uint8_t *present) {
    for (int i = 0; i < N; i++) {
        real pv[3]; xform4x3(means3D + 3 * i, viewmatrix, pv);
        present[i] = pv[2] > NEAR_CULL;
    }
    return 0;
}

void oracle_free(oracle_ctx *c) {
    void **ptrs[] = {(void **)&c->out_color, (void **)&c->out_depth, (void **)&c->out_opacity, (void **)&c->radii,
                     (void **)&c->n_touched, (void **)&c->means2D, (void **)&c->depths, (void **)&c->conic_opacity,
                     (void **)&c->rgb, (void **)&c->cov3D, (void **)&c->clamped, (void **)&c->tiles_touched,
                     (void **)&c->rect, (void **)&c->keys_sorted, (void **)&c->ids_sorted, (void **)&c->ranges,
                     (void **)&c->final_T, (void **)&c->n_contrib, (void **)&c->fragile, (void **)&c->n_contrib_lo, (void **)&c->n_contrib_hi, (void **)&c->dL_dmeans3D,
                     (void **)&c->dL_dmeans2D, (void **)&c->dL_dscales, (void **)&c->dL_drotations,
                     (void **)&c->dL_dopacity, (void **)&c->dL_dcolors, (void **)&c->dL_dshs, (void **)&c->dL_dcov3D,
                     (void **)&c->dL_dtau};
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) { free(*ptrs[i]); *ptrs[i] = NULL; }
}

int oracle_real_bytes(void) { return (int)sizeof(real); }
