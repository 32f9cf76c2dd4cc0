/*
 * ekf_oracle.c -- TEST INFRASTRUCTURE (see ekf_oracle.h for status and scope).
 *
 * Dense fp64 restatement of FBUS-EKF's ImuUpdate / MeasureUpdate in both
 * dialects.  Deliberately NOT optimised: every matrix product is a plain
 * triple loop over the full n x n matrices, as the reference's Eigen
 * MatrixXd / Matlab expressions do, so that it doubles as a representative
 * CPU baseline.  Each function cites the reference lines it follows.
 */
#include "ekf_oracle.h"
#include "vision_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small dense helpers                                                 */
/* ------------------------------------------------------------------ */
static void mat_mul(const double* A, const double* B, double* C, int m, int k, int n)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int l = 0; l < k; ++l) acc += A[i * k + l] * B[l * n + j];
            C[i * n + j] = acc;
        }
}

/* C = A * B' with A m x k, B n x k */
static void mat_mul_bt(const double* A, const double* B, double* C, int m, int k, int n)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int l = 0; l < k; ++l) acc += A[i * k + l] * B[j * k + l];
            C[i * n + j] = acc;
        }
}

static void mat3_vec(const double* R, const double* x, double* y)
{
    for (int i = 0; i < 3; ++i) y[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2];
}

static void mat3t_vec(const double* R, const double* x, double* y)
{
    for (int i = 0; i < 3; ++i) y[i] = R[i] * x[0] + R[3 + i] * x[1] + R[6 + i] * x[2];
}

static double norm3(const double* x) { return sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); }
static double norm4(const double* x) { return sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]); }

static void symmetrise(double* P, int n)
{   /* ImuUpdate.m:81, MeasureUpdate.m:102, filter.cpp:614-615,738-739 */
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            double m = (P[i * n + j] + P[j * n + i]) / 2.0;
            P[i * n + j] = m;
            P[j * n + i] = m;
        }
}

/* in-place inverse by Gauss-Jordan with partial pivoting (stands in for
 * Matlab's inv, MeasureUpdate.m:84). Returns 0 on success.                */
static int mat_inv(double* A, int n)
{
    /* scratch sized by the ACTUAL n: the reference path (7 rows) needs 7 x 14 doubles on the stack, not the 1 MiB
     * FBO_MMAX would reserve in every worker thread; above FBO_STACK_ROWS rows (stacked corner / pixel models) the
     * scratch comes from the heap. */
    if (n > FBO_MMAX || n <= 0) return -1;
    const int on_stack = (n <= FBO_STACK_ROWS);
    double Wstack[on_stack ? n * 2 * n : 1];
    double* W = on_stack ? Wstack : (double*)malloc(sizeof(double) * (size_t)n * 2 * n);
    if (!W) return -3;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            W[i * 2 * n + j] = A[i * n + j];
            W[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (fabs(W[r * 2 * n + c]) > fabs(W[piv * 2 * n + c])) piv = r;
        if (W[piv * 2 * n + c] == 0.0) { if (!on_stack) free(W); return -2; }
        if (piv != c)
            for (int j = 0; j < 2 * n; ++j) {
                double t = W[c * 2 * n + j];
                W[c * 2 * n + j] = W[piv * 2 * n + j];
                W[piv * 2 * n + j] = t;
            }
        double d = 1.0 / W[c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) W[c * 2 * n + j] *= d;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = W[r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; ++j) W[r * 2 * n + j] -= f * W[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[i * n + j] = W[i * 2 * n + n + j];
    if (!on_stack) free(W);
    return 0;
}

/* solve S X = B (S m x m SPD, B m x n) by LDL' without pivoting; stands in
 * for Eigen's S.ldlt().solve(H*P), filter.cpp:711.  X overwrites B.        */
static void ldlt_solve(const double* S, double* B, int m, int n)
{
    const int on_stack = (m <= FBO_STACK_ROWS);
    double Lstack[on_stack ? m * m : 1], D[FBO_MMAX];
    double* L = on_stack ? Lstack : (double*)malloc(sizeof(double) * (size_t)m * m);
    if (!L) return;
    memset(L, 0, sizeof(double) * (size_t)m * m);
    for (int j = 0; j < m; ++j) {
        double d = S[j * m + j];
        for (int k = 0; k < j; ++k) d -= L[j * m + k] * L[j * m + k] * D[k];
        D[j] = d;
        L[j * m + j] = 1.0;
        for (int i = j + 1; i < m; ++i) {
            double v = S[i * m + j];
            for (int k = 0; k < j; ++k) v -= L[i * m + k] * L[j * m + k] * D[k];
            L[i * m + j] = v / d;
        }
    }
    for (int c = 0; c < n; ++c) {
        for (int i = 0; i < m; ++i) {          /* L y = b */
            double v = B[i * n + c];
            for (int k = 0; k < i; ++k) v -= L[i * m + k] * B[k * n + c];
            B[i * n + c] = v;
        }
        for (int i = 0; i < m; ++i) B[i * n + c] /= D[i];
        for (int i = m - 1; i >= 0; --i) {     /* L' x = z */
            double v = B[i * n + c];
            for (int k = i + 1; k < m; ++k) v -= L[k * m + i] * B[k * n + c];
            B[i * n + c] = v;
        }
    }
    if (!on_stack) free(L);
}

/* ------------------------------------------------------------------ */
/* L0 helpers                                                          */
/* ------------------------------------------------------------------ */
void fbo_quat_mul(const double p[4], const double q[4], double out[4])
{   /* quaternion_add.m:22-28 (Hamilton product, wxyz) */
    double o0 = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
    double o1 = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
    double o2 = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
    double o3 = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
    out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3;
}

void fbo_axisangle_to_quat(const double axis[3], double angle, double q[4])
{   /* axisangle_to_quaternion.m:22-29 ; matrix_math.hpp:90-99 (NaN at zero axis, kept) */
    double n = norm3(axis);
    q[0] = cos(angle / 2);
    q[1] = axis[0] / n * sin(angle / 2);
    q[2] = axis[1] / n * sin(angle / 2);
    q[3] = axis[2] / n * sin(angle / 2);
}

void fbo_quat_to_rotmat(const double q[4], double R[9])
{   /* quaternion_to_rotmat.m:22-33 */
    R[0] = q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3];
    R[1] = 2 * (q[1] * q[2] - q[0] * q[3]);
    R[2] = 2 * (q[1] * q[3] + q[0] * q[2]);
    R[3] = 2 * (q[1] * q[2] + q[0] * q[3]);
    R[4] = q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3];
    R[5] = 2 * (q[2] * q[3] - q[0] * q[1]);
    R[6] = 2 * (q[1] * q[3] - q[0] * q[2]);
    R[7] = 2 * (q[2] * q[3] + q[0] * q[1]);
    R[8] = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
}

void fbo_quat_to_rotmat_eigen(const double q[4], double R[9])
{   /* Eigen Quaterniond::toRotationMatrix (call sites filter.cpp:542,562,564) */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

void fbo_rotmat_to_quat(const double R[9], double q[4])
{   /* trace-based conversion as Eigen's Quaterniond(Matrix3d) (filter.cpp:630);
     * fixes the sign that Matlab's eig-based rotmat_to_quaternion.m:22-44 leaves open */
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t;
        q[2] = (R[2] - R[6]) * t;
        q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}

void fbo_quat_left_matrix(const double q[4], double L[16])
{   /* quaternion_left_product_matrix.m:22-40 ; matrix_math.hpp:38-62 */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double M[16] = { w, -x, -y, -z,
                     x,  w, -z,  y,
                     y,  z,  w, -x,
                     z, -y,  x,  w };
    memcpy(L, M, sizeof(M));
}

void fbo_quat_right_matrix(const double q[4], double Rm[16])
{   /* quaternion_right_product_matrix.m:22-40 ; matrix_math.hpp:64-88 */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double M[16] = { w, -x, -y, -z,
                     x,  w,  z, -y,
                     y, -z,  w,  x,
                     z,  y, -x,  w };
    memcpy(Rm, M, sizeof(M));
}

void fbo_skew(const double v[3], double M[9])
{   /* vector_to_crossmat.m:22-30 ; matrix_math.hpp:26-36 */
    M[0] = 0;     M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2];  M[4] = 0;     M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0];  M[8] = 0;
}

void fbo_expm_so3_neg(const double w[3], double dt, double E[9])
{   /* expm(-[w]x*dt), ImuUpdate.m:68.  Closed form (Rodrigues) of the matrix
     * exponential of a skew matrix; a Taylor branch covers phi -> 0.          */
    double u[3] = { -w[0] * dt, -w[1] * dt, -w[2] * dt };
    double phi = norm3(u);
    double K[9], K2[9];
    fbo_skew(u, K);
    mat_mul(K, K, K2, 3, 3, 3);
    double a, b;
    if (phi < 1e-6) {
        a = 1.0 - phi * phi / 6.0;
        b = 0.5 - phi * phi / 24.0;
    } else {
        a = sin(phi) / phi;
        b = (1.0 - cos(phi)) / (phi * phi);
    }
    for (int i = 0; i < 9; ++i) E[i] = a * K[i] + b * K2[i];
    E[0] += 1.0; E[4] += 1.0; E[8] += 1.0;
}

/* ------------------------------------------------------------------ */
/* constants                                                           */
/* ------------------------------------------------------------------ */
void fbo_set_camera(fbo_params* prm, const double TSC_raw[16])
{   /* FBUS_EKF.m:68 / filter.hpp:67-70 (flip), MeasureUpdate.m:45-48 / filter.cpp:629-632 */
    double T[16];
    memcpy(T, TSC_raw, sizeof(T));
    for (int j = 0; j < 4; ++j) { T[j] = -T[j]; T[4 + j] = -T[4 + j]; }
    double t[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) prm->R_IL[3 * i + j] = T[4 * i + j];
        t[i] = T[4 * i + 3];
    }
    double tmp[3];
    mat3t_vec(prm->R_IL, t, tmp);
    for (int i = 0; i < 3; ++i) prm->P_IL[i] = -tmp[i];
    fbo_rotmat_to_quat(prm->R_IL, prm->Q_IL);
}

int fbo_add_marker(fbo_params* prm, int id, const double pos[3], const double rot[9])
{
    if (prm->n_markers >= FBO_MAX_MARKERS) return -1;
    int k = prm->n_markers++;
    prm->marker_id[k] = id;
    memcpy(prm->marker_pos[k], pos, 3 * sizeof(double));
    fbo_rotmat_to_quat(rot, prm->marker_quat[k]);   /* MeasureUpdate.m:64, common.hpp MarkerPose */
    return k;
}

void fbo_default_params(fbo_params* prm, int dialect, int nstate)
{
    memset(prm, 0, sizeof(*prm));
    prm->dialect = dialect;
    prm->nstate = nstate;
    /* FBUS_EKF.m:36-39,103-106 ; paramconfig.yml:46-49 via filter.hpp:108-115 */
    prm->q_diag[0] = 1e-3; prm->q_diag[1] = 1e-4; prm->q_diag[2] = 1e-3; prm->q_diag[3] = 1e-4;
    if (dialect == FBO_DIALECT_MATLAB) { prm->r_pos = 0.01;  prm->r_quat = 0.01;  }  /* FBUS_EKF.m:32-33 */
    else                               { prm->r_pos = 0.001; prm->r_quat = 0.001; }  /* paramconfig.yml:53-54 */
    prm->switch_thres = 0.5;                                                         /* paramconfig.yml:57 */
    prm->cov_form = FBO_COV_SIMPLE;
    /* matlab/config/camerainfo.yml:11-15 == C++/config/camerainfo1.yml (left TSC, raw) */
    static const double TSC[16] = { -0.999862, 0.015685, -0.00548,  0.059967,
                                    -0.015639, -0.999843, -0.00827, 0.000127837,
                                    -0.005609, -0.008183, 0.999951, -0.002,
                                     0, 0, 0, 1 };
    fbo_set_camera(prm, TSC);
    /* GetMarkerMap.m:1-63 == C++/config/markersetup.yml */
    static const double I3[9]  = { 1, 0, 0,  0, 1, 0,  0, 0, 1 };
    static const double RA[9]  = { 1, 0, 0,  0, 0, -1, 0, 1, 0 };
    static const double RB[9]  = { 1, 0, 0,  0, -1, 0, 0, 0, -1 };
    static const struct { int id; double pos[3]; const double* rot; } map[12] = {
        { 0,  { 0, 0, 0 },          I3 }, { 1,  { 0, 0.61, 0.285 },  RA },
        { 2,  { 0, 0.61, 1.185 },   RA }, { 3,  { 0, 0.61, 2.085 },  RA },
        { 4,  { 0, 0.61, 2.985 },   RA }, { 5,  { 0, 0.265, 4.12 },  RB },
        { 6,  { 0, -0.635, 4.12 },  RB }, { 7,  { 0, -1.535, 4.12 }, RB },
        { 8,  { 0, -2.435, 4.12 },  RB }, { 16, { 0, -2.7, 0 },      I3 },
        { 17, { 0, -1.8, 0 },       I3 }, { 18, { 0, -0.9, 0 },      I3 } };
    for (int k = 0; k < 12; ++k) fbo_add_marker(prm, map[k].id, map[k].pos, map[k].rot);
}

void fbo_default_P0(const fbo_params* prm, double* P)
{
    int n = prm->nstate;
    /* FBUS_EKF.m:88-99 vs filter.hpp:29-34 */
    static const double m[6] = { 1e-4, 0.1,  1e-4, 1e-3, 1e-3, 100.0 };
    static const double c[6] = { 1e-4, 1e-2, 1e-4, 1e-2, 1e-2, 100.0 };
    const double* d = (prm->dialect == FBO_DIALECT_MATLAB) ? m : c;
    memset(P, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) P[i * n + i] = d[i / 3];
}

static int find_marker(const fbo_params* prm, int id)
{
    for (int k = 0; k < prm->n_markers; ++k)
        if (prm->marker_id[k] == id) return k;
    return -1;
}

/* ------------------------------------------------------------------ */
/* predict                                                             */
/* ------------------------------------------------------------------ */
/* Fx of ImuUpdate.m:63-69 / filter.cpp:597-604 at the pre-step state (a, w bias-corrected).  fbo_predict uses
 * exactly this matrix; it is exported (fbo_transition) so that a test can check it against finite differences of
 * the nominal kinematics, i.e. the transcription is pinned by something other than itself. */
static void build_Fx(const fbo_state* s, const fbo_params* prm, const double a[3], const double w[3], double dt,
                     double* Fx)
{
    const int n = prm->nstate;
    const int cpp = (prm->dialect == FBO_DIALECT_CPP);
    memset(Fx, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) Fx[i * n + i] = 1.0;
    double ax[9], Ra[9];
    fbo_skew(a, ax);
    mat_mul(s->R, ax, Ra, 3, 3, 3);         /* carried rotateMat / rotmatI2G */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            if (i == j) Fx[(0 + i) * n + 3 + j] = dt;                 /* (p,v)   */
            Fx[(3 + i) * n + 6 + j] = -Ra[3 * i + j] * dt;            /* (v,th)  */
            Fx[(3 + i) * n + 9 + j] = -s->R[3 * i + j] * dt;          /* (v,ba)  */
            if (n == 18 && i == j) Fx[(3 + i) * n + 15 + j] = dt;     /* (v,g)   */
            if (i == j) Fx[(6 + i) * n + 12 + j] = -dt;               /* (th,bg) */
        }
    double Th[9];
    if (cpp) {                              /* filter.cpp:603 : I - [w]x dt */
        double wx[9];
        fbo_skew(w, wx);
        for (int i = 0; i < 9; ++i) Th[i] = -wx[i] * dt;
        Th[0] += 1; Th[4] += 1; Th[8] += 1;
    } else {                                /* ImuUpdate.m:68 : expm(-[w]x dt) */
        fbo_expm_so3_neg(w, dt, Th);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Fx[(6 + i) * n + 6 + j] = Th[3 * i + j];
}

void fbo_transition(const fbo_state* s, const fbo_params* prm, const double accel[3], const double gyro[3],
                    double dt, double* Fx)
{
    double a[3], w[3];
    for (int i = 0; i < 3; ++i) { a[i] = accel[i] - s->ba[i]; w[i] = gyro[i] - s->bg[i]; }
    build_Fx(s, prm, a, w, dt, Fx);
}

void fbo_predict(fbo_state* s, const fbo_params* prm,
                 const double accel[3], const double gyro[3], double dt)
{
    const int n = prm->nstate;
    const int cpp = (prm->dialect == FBO_DIALECT_CPP);
    double a[3], w[3];
    for (int i = 0; i < 3; ++i) {           /* ImuUpdate.m:37-38 ; filter.cpp:539,568,594-595 */
        a[i] = accel[i] - s->ba[i];
        w[i] = gyro[i] - s->bg[i];
    }

    /* ---- covariance (reads the pre-step state; filter.cpp:588-616, ImuUpdate.m:63-73) ---- */
    double Fx[FBO_NMAX * FBO_NMAX];
    build_Fx(s, prm, a, w, dt, Fx);

    double FP[FBO_NMAX * FBO_NMAX], Pn[FBO_NMAX * FBO_NMAX] = { 0 };
    mat_mul(Fx, s->P, FP, n, n, n);
    mat_mul_bt(FP, Fx, Pn, n, n, n);
    for (int i = 3; i < 15; ++i) Pn[i * n + i] += prm->q_diag[(i - 3) / 3];   /* Fi*Q*Fi' */
    symmetrise(Pn, n);

    /* ---- nominal state (ImuUpdate.m:41-60,76-79 ; filter.cpp:533-582) ---- */
    double qT[4], qH[4], R0[9], RH[9], RT[9];
    if (!cpp) {
        double wdt[3] = { w[0] * dt, w[1] * dt, w[2] * dt };
        double dtheta = norm3(wdt);
        double dq[4];
        fbo_axisangle_to_quat(w, dtheta, dq);
        fbo_quat_mul(s->q, dq, qT);
        fbo_axisangle_to_quat(w, dtheta / 2, dq);
        fbo_quat_mul(s->q, dq, qH);
        memcpy(R0, s->R, sizeof(R0));       /* carried, possibly stale (ImuUpdate.m:46) */
        fbo_quat_to_rotmat(qH, RH);
        fbo_quat_to_rotmat(qT, RT);
    } else {
        double wn = norm3(w);
        fbo_quat_to_rotmat_eigen(s->q, R0); /* fresh (filter.cpp:542) */
        double dq[4];
        if (wn > 10e-5) {
            double axis[3] = { w[0] / wn, w[1] / wn, w[2] / wn };
            double ang = wn * dt / 2;
            dq[0] = cos(ang / 2);
            for (int i = 0; i < 3; ++i) dq[1 + i] = sin(ang / 2) * axis[i];
            fbo_quat_mul(s->q, dq, qH);
            ang = wn * dt;
            dq[0] = cos(ang / 2);
            for (int i = 0; i < 3; ++i) dq[1 + i] = sin(ang / 2) * axis[i];
            fbo_quat_mul(s->q, dq, qT);
        } else {                            /* filter.cpp:553-560 */
            dq[0] = 1;
            for (int i = 0; i < 3; ++i) dq[1 + i] = 0.5 * dt * w[i] / 2;
            fbo_quat_mul(s->q, dq, qH);
            for (int i = 0; i < 3; ++i) dq[1 + i] = 0.5 * dt * w[i];
            fbo_quat_mul(s->q, dq, qT);
        }
        double nh = norm4(qH), nt = norm4(qT);
        for (int i = 0; i < 4; ++i) { qH[i] /= nh; qT[i] /= nt; }
        fbo_quat_to_rotmat_eigen(qH, RH);
        fbo_quat_to_rotmat_eigen(qT, RT);
    }
    double kv1[3], kv2[3], kv4[3];
    mat3_vec(R0, a, kv1);
    mat3_vec(RH, a, kv2);
    mat3_vec(RT, a, kv4);
    double vnew[3], pnew[3];
    for (int i = 0; i < 3; ++i) {
        kv1[i] += s->g[i]; kv2[i] += s->g[i]; kv4[i] += s->g[i];
        double kv3 = kv2[i];
        vnew[i] = s->v[i] + dt / 6 * (kv1[i] + 2 * kv2[i] + 2 * kv3 + kv4[i]);
        double kp1 = s->v[i];
        double kp2 = s->v[i] + kv1[i] * dt / 2;
        double kp3 = s->v[i] + kv2[i] * dt / 2;
        double kp4 = s->v[i] + kv3 * dt / 2;          /* dt/2, sic (ImuUpdate.m:59, filter.cpp:580) */
        pnew[i] = s->p[i] + dt / 6 * (kp1 + 2 * kp2 + 2 * kp3 + kp4);
    }
    if (!cpp) {                             /* ImuUpdate.m:76 */
        double nt = norm4(qT);
        for (int i = 0; i < 4; ++i) s->q[i] = qT[i] / nt;
    } else {
        memcpy(s->q, qT, sizeof(qT));
    }
    memcpy(s->R, RT, sizeof(RT));           /* ImuUpdate.m:77 ; filter.cpp:564 */
    memcpy(s->v, vnew, sizeof(vnew));
    memcpy(s->p, pnew, sizeof(pnew));
    memcpy(s->P, Pn, sizeof(double) * n * n);
}

/* ------------------------------------------------------------------ */
/* correct                                                             */
/* ------------------------------------------------------------------ */
/* rows for one marker: h(x), H (7 x n), residual r (7).                */
static void marker_rows_h(const fbo_state* s, const fbo_params* prm, int slot,
                          const double* yp, const double* yq,
                          double* H /*7 x n*/, double* r /*7*/, double* h /*7, may be NULL*/)
{
    const int n = prm->nstate;
    const int cpp = (prm->dialect == FBO_DIALECT_CPP);
    const double* Pm = prm->marker_pos[slot];
    const double* Qm = prm->marker_quat[slot];
    const double* R = s->R;

    /* hp = R_IL R' (Pm - p - R P_IL)   MeasureUpdate.m:67 ; filter.cpp:684-685 */
    double RP[3], d[3], t[3], hp[3];
    mat3_vec(R, prm->P_IL, RP);
    for (int i = 0; i < 3; ++i) d[i] = Pm[i] - s->p[i] - RP[i];
    mat3t_vec(R, d, t);
    mat3_vec(prm->R_IL, t, hp);
    /* hq = Q_IL (x) q* (x) Qm          MeasureUpdate.m:68 ; filter.cpp:686 */
    double qc[4] = { s->q[0], -s->q[1], -s->q[2], -s->q[3] }, tmp[4], hq[4];
    fbo_quat_mul(prm->Q_IL, qc, tmp);
    fbo_quat_mul(tmp, Qm, hq);

    memset(H, 0, sizeof(double) * 7 * n);
    /* H(1:3,1:3) = -R_IL R'            MeasureUpdate.m:72 ; filter.cpp:691 */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += prm->R_IL[3 * i + k] * R[3 * j + k];
            H[i * n + j] = -acc;
        }
    /* H(1:3,7:9) = R_IL [R'(Pm - p)]x  MeasureUpdate.m:73 ; filter.cpp:692 */
    double dm[3], u[3], ux[9], B[9];
    for (int i = 0; i < 3; ++i) dm[i] = Pm[i] - s->p[i];
    mat3t_vec(R, dm, u);
    fbo_skew(u, ux);
    mat_mul(prm->R_IL, ux, B, 3, 3, 3);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) H[i * n + 6 + j] = B[3 * i + j];
    /* H(4:7,7:9) = Rq(Qm) Lq(Q_IL) L2 Lq(q) L1   MeasureUpdate.m:74-75 ; filter.cpp:693-694 */
    double Rq[16], Lil[16], Lq[16], L2[16], L1[12], A[16], Bq[16], C[16], D[12];
    fbo_quat_right_matrix(Qm, Rq);
    fbo_quat_left_matrix(prm->Q_IL, Lil);
    fbo_quat_left_matrix(s->q, Lq);
    memset(L2, 0, sizeof(L2));
    L2[0] = 1; L2[5] = -1; L2[10] = -1; L2[15] = -1;
    memset(L1, 0, sizeof(L1));
    L1[1 * 3 + 0] = 0.5; L1[2 * 3 + 1] = 0.5; L1[3 * 3 + 2] = 0.5;
    mat_mul(Rq, Lil, A, 4, 4, 4);
    mat_mul(A, L2, Bq, 4, 4, 4);
    mat_mul(Bq, Lq, C, 4, 4, 4);
    mat_mul(C, L1, D, 4, 4, 3);
    /* sign unification   MeasureUpdate.m:77-81 ; filter.cpp:698-706 */
    double k1 = 0, k2 = 0;
    for (int i = 0; i < 4; ++i) {
        k1 += (yq[i] - hq[i]) * (yq[i] - hq[i]);
        k2 += (yq[i] + hq[i]) * (yq[i] + hq[i]);
    }
    int flip = cpp ? (k1 > k2) : (sqrt(k1) > sqrt(k2));
    double sg = flip ? -1.0 : 1.0;
    for (int i = 0; i < 4; ++i) hq[i] *= sg;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) H[(3 + i) * n + 6 + j] = sg * D[3 * i + j];
    for (int i = 0; i < 3; ++i) r[i] = yp[i] - hp[i];
    for (int i = 0; i < 4; ++i)             /* MeasureUpdate.m:87-88 zeroes these rows */
        r[3 + i] = cpp ? (yq[i] - hq[i]) : 0.0;
    if (h) {
        for (int i = 0; i < 3; ++i) h[i] = hp[i];
        for (int i = 0; i < 4; ++i) h[3 + i] = hq[i];      /* after the sign unification */
    }
}

static void marker_rows(const fbo_state* s, const fbo_params* prm, int slot,
                        const double* yp, const double* yq, double* H, double* r)
{
    marker_rows_h(s, prm, slot, yp, yq, H, r, NULL);
}

/* exported for the finite-difference test of the Jacobian: predicted measurement h (7, quaternion part after the
 * sign unification against yq), H (7 x n), residual r (7) of the marker with ArUco id `id`, exactly what
 * fbo_correct uses.  Returns 0 if the id is not in the map. */
int fbo_measurement(const fbo_state* s, const fbo_params* prm, int id, const double* yp, const double* yq,
                    double* h, double* H, double* r)
{
    const int k = find_marker(prm, id);
    if (k < 0) return 0;
    marker_rows_h(s, prm, k, yp, yq, H, r, h);
    return 1;
}

/* Dense update shared by the pose-row and corner-row measurement models, exactly the reference's
 * algebra: S = H P H' + R, K = P H' inv(S) (Matlab) / K' = ldlt(S).solve(H P) (C++), dx = K r,
 * inject, P = (I - K H) P [Joseph option], symmetrise.  Rd = per-row measurement noise. */
static void dense_update(fbo_state* s, const fbo_params* prm, int m, const double* H, const double* r,
                         const double* Rd)
{
    const int n = prm->nstate;
    const int cpp = (prm->dialect == FBO_DIALECT_CPP);
    /* scratch by the actual row count m (see mat_inv): stack up to FBO_STACK_ROWS rows, heap above */
    const int on_stack = (m <= FBO_STACK_ROWS);
    const size_t need = (size_t)m * n * 4 + (size_t)m * m;                 /* HP, K, PHt, X, S */
    double scratch_stack[on_stack ? need : 1];
    double* scratch = on_stack ? scratch_stack : (double*)malloc(sizeof(double) * need);
    if (!scratch) return;
    double* HP = scratch; double* K = HP + (size_t)m * n; double* PHt = K + (size_t)m * n; double* X = PHt + (size_t)m * n;
    double* S = X + (size_t)m * n;
    /* S = H P H' + Rm    MeasureUpdate.m:84 ; filter.cpp:709-710 */
    mat_mul(H, s->P, HP, m, n, n);
    mat_mul_bt(HP, H, S, m, n, m);
    for (int j = 0; j < m; ++j) S[j * m + j] += Rd[j];
    if (!cpp) {                             /* K = P H' inv(S) */
        mat_mul_bt(s->P, H, PHt, n, n, m);
        mat_inv(S, m);
        mat_mul(PHt, S, K, n, m, m);
    } else {                                /* K' = S.ldlt().solve(H P) */
        memcpy(X, HP, sizeof(double) * m * n);
        ldlt_solve(S, X, m, n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j) K[i * m + j] = X[j * n + i];
    }
    double dx[FBO_NMAX];
    mat_mul(K, r, dx, n, m, 1);             /* MeasureUpdate.m:89 ; filter.cpp:723 */

    /* inject   MeasureUpdate.m:92-98 ; filter.cpp:726-733 */
    for (int i = 0; i < 3; ++i) { s->p[i] += dx[i]; s->v[i] += dx[3 + i]; }
    double dq[4], qn[4];
    fbo_axisangle_to_quat(dx + 6, norm3(dx + 6), dq);
    fbo_quat_mul(s->q, dq, qn);
    double nq = norm4(qn);
    for (int i = 0; i < 4; ++i) s->q[i] = qn[i] / nq;
    for (int i = 0; i < 3; ++i) { s->ba[i] += dx[9 + i]; s->bg[i] += dx[12 + i]; }
    if (n == 18) for (int i = 0; i < 3; ++i) s->g[i] += dx[15 + i];

    /* covariance   MeasureUpdate.m:101-102 ; filter.cpp:735-739 */
    double IKH[FBO_NMAX * FBO_NMAX], Pn[FBO_NMAX * FBO_NMAX];
    mat_mul(K, H, IKH, n, m, n);
    for (int i = 0; i < n * n; ++i) IKH[i] = -IKH[i];
    for (int i = 0; i < n; ++i) IKH[i * n + i] += 1.0;
    mat_mul(IKH, s->P, Pn, n, n, n);
    if (prm->cov_form == FBO_COV_JOSEPH) {
        double T[FBO_NMAX * FBO_NMAX];
        mat_mul_bt(Pn, IKH, T, n, n, n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double acc = 0;
                for (int k = 0; k < m; ++k) acc += K[i * m + k] * Rd[k] * K[j * m + k];
                Pn[i * n + j] = T[i * n + j] + acc;
            }
    }
    symmetrise(Pn, n);
    memcpy(s->P, Pn, sizeof(double) * n * n);
    if (!on_stack) free(scratch);
    /* rotateMat / rotmatI2G deliberately NOT refreshed (both dialects) */
}

int fbo_correct(fbo_state* s, const fbo_params* prm, int M,
                const int* ids, const double* pos, const double* quat, int mode)
{
    const int n = prm->nstate;
    const int cpp = (prm->dialect == FBO_DIALECT_CPP);
    int sel[FBO_MAX_VISIBLE], slot[FBO_MAX_VISIBLE], nsel = 0;
    if (M > FBO_MAX_VISIBLE) M = FBO_MAX_VISIBLE;

    if (mode == FBO_MODE_NEAREST) {
        /* MeasureUpdate.m:51-60 ; filter.cpp:639-664 */
        int min_i = -1, prev_i = -1;
        double min_d = 10.0, prev_d = 0.0;
        for (int i = 0; i < M; ++i) {
            if (ids[i] < 0) continue;       /* absent slot of the batched interface */
            double dist = norm3(pos + 3 * i);
            if (dist < min_d) { min_d = dist; min_i = i; }
            if (cpp && ids[i] == s->prev_id) { prev_d = dist; prev_i = i; }
        }
        if (min_i < 0) return 0;
        if (cpp && fabs(prev_d - min_d) < prm->switch_thres && prev_d != 0.0) min_i = prev_i;
        int k = find_marker(prm, ids[min_i]);
        if (k < 0) return 0;                /* filter.cpp:671-673 */
        if (cpp) s->prev_id = ids[min_i];   /* filter.cpp:675 */
        sel[0] = min_i; slot[0] = k; nsel = 1;
    } else {
        for (int i = 0; i < M; ++i) {
            if (ids[i] < 0) continue;
            int k = find_marker(prm, ids[i]);
            if (k < 0) continue;
            sel[nsel] = i; slot[nsel] = k; ++nsel;
        }
        if (nsel == 0) return 0;
    }

    const int m = 7 * nsel;
    double H[FBO_MMAX * FBO_NMAX], r[FBO_MMAX], Rd[FBO_MMAX];
    for (int j = 0; j < nsel; ++j)
        marker_rows(s, prm, slot[j], pos + 3 * sel[j], quat + 4 * sel[j], H + 7 * j * n, r + 7 * j);
    for (int j = 0; j < m; ++j) Rd[j] = ((j % 7) < 3) ? prm->r_pos : prm->r_quat;
    dense_update(s, prm, m, H, r, Rd);
    return 1;
}

/* ------------------------------------------------------------------ */
/* corner-row measurement model (north-star extension, NO reference    */
/* counterpart: parity unpinned by construction)                       */
/* ------------------------------------------------------------------ */
/* Measurement = the four triangulated corner positions of a marker in the left camera frame (3 rows per
 * corner, 12 per marker).  Corner k sits at c_k = {(0,0,0),(0,s,0),(s,s,0),(s,0,0)} in the marker frame -- the
 * frame VISION::ComputeMarkerPose builds (vision.cpp:736-759: origin = corner 0, axes along corner0->corner3 and
 * corner0->corner1; verified on the recordings).  h_k = R_IL R' (P_m + R_m c_k - p - R P_IL), the Jacobian has
 * the structure of the reference's position rows (MeasureUpdate.m:72-73) with the corner in place of the marker
 * origin; noise r_pos per row.  corners: M x 12. */
int fbo_correct_corners(fbo_state* s, const fbo_params* prm, int M, const int* ids, const double* corners,
                        double size, int mode)
{
    const int n = prm->nstate;
    const int cpp = (prm->dialect == FBO_DIALECT_CPP);
    int sel[FBO_MAX_VISIBLE], slot[FBO_MAX_VISIBLE], nsel = 0;
    if (M > FBO_MAX_VISIBLE) M = FBO_MAX_VISIBLE;
    if (mode == FBO_MODE_NEAREST) {                 /* same selection rule as fbo_correct, on corner 0 */
        int min_i = -1, prev_i = -1;
        double min_d = 10.0, prev_d = 0.0;
        for (int i = 0; i < M; ++i) {
            if (ids[i] < 0) continue;
            double dist = norm3(corners + 12 * i);
            if (dist < min_d) { min_d = dist; min_i = i; }
            if (cpp && ids[i] == s->prev_id) { prev_d = dist; prev_i = i; }
        }
        if (min_i < 0) return 0;
        if (cpp && fabs(prev_d - min_d) < prm->switch_thres && prev_d != 0.0) min_i = prev_i;
        int k = find_marker(prm, ids[min_i]);
        if (k < 0) return 0;
        if (cpp) s->prev_id = ids[min_i];
        sel[0] = min_i; slot[0] = k; nsel = 1;
    } else {
        for (int i = 0; i < M; ++i) {
            if (ids[i] < 0) continue;
            int k = find_marker(prm, ids[i]);
            if (k < 0) continue;
            sel[nsel] = i; slot[nsel] = k; ++nsel;
        }
        if (nsel == 0) return 0;
    }
    const int m = 12 * nsel;
    double H[FBO_MMAX * FBO_NMAX], r[FBO_MMAX], Rd[FBO_MMAX];
    memset(H, 0, sizeof(double) * m * n);
    const double ck[4][3] = { { 0, 0, 0 }, { 0, size, 0 }, { size, size, 0 }, { size, 0, 0 } };
    for (int j = 0; j < nsel; ++j) {
        double Rm[9];
        fbo_quat_to_rotmat(prm->marker_quat[slot[j]], Rm);
        for (int k = 0; k < 4; ++k) {
            double cw[3], RP[3], d[3], dm[3], t[3], hp[3], u[3], ux[9], Bm[9];
            mat3_vec(Rm, ck[k], cw);
            for (int i = 0; i < 3; ++i) cw[i] += prm->marker_pos[slot[j]][i];      /* corner in the world */
            mat3_vec(s->R, prm->P_IL, RP);
            for (int i = 0; i < 3; ++i) { dm[i] = cw[i] - s->p[i]; d[i] = dm[i] - RP[i]; }
            mat3t_vec(s->R, d, t);
            mat3_vec(prm->R_IL, t, hp);
            mat3t_vec(s->R, dm, u);
            fbo_skew(u, ux);
            mat_mul(prm->R_IL, ux, Bm, 3, 3, 3);
            double* Hk = H + (size_t)(12 * j + 3 * k) * n;
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) {
                    double acc = 0;
                    for (int c = 0; c < 3; ++c) acc += prm->R_IL[3 * a + c] * s->R[3 * b + c];
                    Hk[a * n + b] = -acc;
                    Hk[a * n + 6 + b] = Bm[3 * a + b];
                }
                r[12 * j + 3 * k + a] = corners[12 * sel[j] + 3 * k + a] - hp[a];
                Rd[12 * j + 3 * k + a] = prm->r_pos;
            }
        }
    }
    dense_update(s, prm, m, H, r, Rd);
    return 1;
}

static void load_state(fbo_state* s, int n, const double* nom, const double* rot, const double* P, int prev);
static void store_state(const fbo_state* s, int n, double* nom, double* rot, double* P, int* prev);

/* ------------------------------------------------------------------ */
/* pixel-row measurement model: the north star's "flat-port refractive  */
/* stereo reprojection of ArUco corners, per-corner 2 x N Jacobians".   */
/* NO reference counterpart (the reference has no forward projection):  */
/* parity unpinned by construction; the projection itself is pinned     */
/* through the reference's back-projection (vision_oracle.c).           */
/* ------------------------------------------------------------------ */
/* Measurement = the normalised image points of the four corners of every visible marker, left camera (2 rows per
 * corner) or both cameras (4 rows per corner).  h = pi(X_k(x)), X_k = R_IL R'(P_m + R_m c_k - p - R P_IL) the corner in the
 * left camera frame (as fbo_correct_corners), pi = fbv_project_stereo.  H = (d pi / d X) [ -R_IL R' | R_IL [R'(c_w - p)]x ]
 * with d pi / d X by CENTRAL DIFFERENCES of the projection (deliberately not the analytic form the device uses).
 * All visible markers, one linearisation point, noise r_pix per row. */
static int correct_pixels_impl(fbo_state* s, const fbo_params* prm, const void* vision_params, int M, const int* ids,
                               const double* left /*M x 8*/, const double* right /*M x 8 or NULL*/, double size, double r_pix,
                               int analytic)
{
    const fbv_params* vp = (const fbv_params*)vision_params;
    const int n = prm->nstate, rows_c = right ? 4 : 2;
    int sel[FBO_MAX_VISIBLE], slot[FBO_MAX_VISIBLE], nsel = 0;
    if (M > FBO_MAX_VISIBLE) M = FBO_MAX_VISIBLE;
    for (int i = 0; i < M; ++i) {
        if (ids[i] < 0) continue;
        int k = find_marker(prm, ids[i]);
        if (k < 0) continue;
        sel[nsel] = i; slot[nsel] = k; ++nsel;
    }
    if (nsel == 0) return 0;
    static __thread double H[FBO_MMAX * FBO_NMAX], r[FBO_MMAX], Rd[FBO_MMAX];
    const double ck[4][3] = { { 0, 0, 0 }, { 0, size, 0 }, { size, size, 0 }, { size, 0, 0 } };
    int m = 0;
    for (int j = 0; j < nsel; ++j) {
        double Rm[9];
        fbo_quat_to_rotmat(prm->marker_quat[slot[j]], Rm);
        for (int k = 0; k < 4; ++k) {
            double cw[3], RP[3], d[3], dm[3], t[3], X[3], u[3], ux[9], Bm[9], J3[3 * FBO_NMAX];
            mat3_vec(Rm, ck[k], cw);
            for (int i = 0; i < 3; ++i) cw[i] += prm->marker_pos[slot[j]][i];
            mat3_vec(s->R, prm->P_IL, RP);
            for (int i = 0; i < 3; ++i) { dm[i] = cw[i] - s->p[i]; d[i] = dm[i] - RP[i]; }
            mat3t_vec(s->R, d, t);
            mat3_vec(prm->R_IL, t, X);                                   /* corner in the left camera frame */
            mat3t_vec(s->R, dm, u);
            fbo_skew(u, ux);
            mat_mul(prm->R_IL, ux, Bm, 3, 3, 3);
            memset(J3, 0, sizeof(double) * 3 * n);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double acc = 0;
                    for (int c = 0; c < 3; ++c) acc += prm->R_IL[3 * a + c] * s->R[3 * b + c];
                    J3[a * n + b] = -acc;
                    J3[a * n + 6 + b] = Bm[3 * a + b];
                }
            /* each camera on its own: a corner outside one camera's view still gives the other camera's rows */
            double uvL[2], uvR[2], Jp[4][3];
            int okL, okR;
            if (analytic) {                 /* d pi / d X in closed form (vision_oracle.c::fbv_project_camera_jac) */
                okL = fbv_project_camera_jac(vp, X, 0, uvL, (double*)Jp);
                okR = right ? fbv_project_camera_jac(vp, X, 1, uvR, (double*)Jp + 6) : 0;
            } else {
                okL = fbv_project_camera(vp, X, 0, uvL);
                okR = right ? fbv_project_camera(vp, X, 1, uvR) : 0;
            }
            const double eps = 1e-6;
            for (int c = 0; c < 3 && !analytic; ++c) {
                double Xp[3] = { X[0], X[1], X[2] }, Xm[3] = { X[0], X[1], X[2] }, a[2], b[2];
                Xp[c] += eps; Xm[c] -= eps;
                if (okL) {
                    fbv_project_camera(vp, Xp, 0, a); fbv_project_camera(vp, Xm, 0, b);
                    Jp[0][c] = (a[0] - b[0]) / (2 * eps); Jp[1][c] = (a[1] - b[1]) / (2 * eps);
                }
                if (okR) {
                    fbv_project_camera(vp, Xp, 1, a); fbv_project_camera(vp, Xm, 1, b);
                    Jp[2][c] = (a[0] - b[0]) / (2 * eps); Jp[3][c] = (a[1] - b[1]) / (2 * eps);
                }
            }
            const double* yl = left + 8 * sel[j] + 2 * k;
            const double* yr = right ? right + 8 * sel[j] + 2 * k : NULL;
            const double res[4] = { yl[0] - uvL[0], yl[1] - uvL[1], yr ? yr[0] - uvR[0] : 0, yr ? yr[1] - uvR[1] : 0 };
            for (int q = 0; q < 4; ++q) {
                if ((q < 2 && !okL) || (q >= 2 && !okR)) continue;
                double* Hq = H + (size_t)m * n;
                for (int c = 0; c < n; ++c) Hq[c] = Jp[q][0] * J3[c] + Jp[q][1] * J3[n + c] + Jp[q][2] * J3[2 * n + c];
                r[m] = res[q];
                Rd[m] = r_pix;
                ++m;
            }
        }
    }
    (void)rows_c;
    if (m == 0) return 1;                  /* markers of the map were seen, none of their corners is in view: a no-op update */
    dense_update(s, prm, m, H, r, Rd);
    return 1;
}

int fbo_correct_pixels(fbo_state* s, const fbo_params* prm, const void* vision_params, int M, const int* ids,
                       const double* left, const double* right, double size, double r_pix)
{
    return correct_pixels_impl(s, prm, vision_params, M, ids, left, right, size, r_pix, 0);
}

/* The same update with d pi / d X in CLOSED FORM (round 6): the second, independent pixel oracle.  The central-difference one
 * above carries the O(eps^2) + O(ulp / eps) error of its differences (1e-10 relative on the rows, up to 1e-7 on the posterior of
 * a sharply observed state) -- what capped the fp64 device gates at 1e-6; this one is exact to rounding.  The two are
 * cross-checked against each other on the CPU (tests/test_oracle_pixels_cpu.py). */
int fbo_correct_pixels_analytic(fbo_state* s, const fbo_params* prm, const void* vision_params, int M, const int* ids,
                                const double* left, const double* right, double size, double r_pix)
{
    return correct_pixels_impl(s, prm, vision_params, M, ids, left, right, size, r_pix, 1);
}

static void correct_pixels_batch_impl(int B, double* nominal, double* rot, double* P, int* prev, const fbo_params* prm,
                                      const void* vision_params, int M, const int* ids, const double* left, const double* right,
                                      double size, double r_pix, int* applied, int analytic)
{
    const int n = prm->nstate;
    for (int b = 0; b < B; ++b) {
        fbo_state s;
        load_state(&s, n, nominal + (size_t)b * 19, rot + (size_t)b * 9, P + (size_t)b * n * n, prev[b]);
        applied[b] = correct_pixels_impl(&s, prm, vision_params, M, ids + (size_t)b * M, left + (size_t)b * M * 8,
                                         right ? right + (size_t)b * M * 8 : NULL, size, r_pix, analytic);
        store_state(&s, n, nominal + (size_t)b * 19, rot + (size_t)b * 9, P + (size_t)b * n * n, prev + b);
    }
}

void fbo_correct_pixels_batch(int B, double* nominal, double* rot, double* P, int* prev, const fbo_params* prm,
                              const void* vision_params, int M, const int* ids, const double* left, const double* right,
                              double size, double r_pix, int* applied)
{
    correct_pixels_batch_impl(B, nominal, rot, P, prev, prm, vision_params, M, ids, left, right, size, r_pix, applied, 0);
}

void fbo_correct_pixels_analytic_batch(int B, double* nominal, double* rot, double* P, int* prev, const fbo_params* prm,
                                       const void* vision_params, int M, const int* ids, const double* left, const double* right,
                                       double size, double r_pix, int* applied)
{
    correct_pixels_batch_impl(B, nominal, rot, P, prev, prm, vision_params, M, ids, left, right, size, r_pix, applied, 1);
}

void fbo_correct_corners_batch(int B, double* nominal, double* rot, double* P, int* prev, const fbo_params* prm,
                               int M, const int* ids, const double* corners, double size, int mode, int* applied)
{
    const int n = prm->nstate;
    fbo_state s;
    for (int b = 0; b < B; ++b) {
        int pv = prev ? prev[b] : 0;
        memset(&s, 0, sizeof(s));
        memcpy(s.p, nominal + 19 * (size_t)b, 3 * sizeof(double));       memcpy(s.v, nominal + 19 * (size_t)b + 3, 3 * sizeof(double));
        memcpy(s.q, nominal + 19 * (size_t)b + 6, 4 * sizeof(double));   memcpy(s.ba, nominal + 19 * (size_t)b + 10, 3 * sizeof(double));
        memcpy(s.bg, nominal + 19 * (size_t)b + 13, 3 * sizeof(double)); memcpy(s.g, nominal + 19 * (size_t)b + 16, 3 * sizeof(double));
        memcpy(s.R, rot + 9 * (size_t)b, 9 * sizeof(double));
        memcpy(s.P, P + (size_t)n * n * b, sizeof(double) * n * n);
        s.prev_id = pv;
        int ok = fbo_correct_corners(&s, prm, M, ids + (size_t)M * b, corners + 12 * (size_t)M * b, size, mode);
        if (applied) applied[b] = ok;
        memcpy(nominal + 19 * (size_t)b, s.p, 3 * sizeof(double));       memcpy(nominal + 19 * (size_t)b + 3, s.v, 3 * sizeof(double));
        memcpy(nominal + 19 * (size_t)b + 6, s.q, 4 * sizeof(double));   memcpy(nominal + 19 * (size_t)b + 10, s.ba, 3 * sizeof(double));
        memcpy(nominal + 19 * (size_t)b + 13, s.bg, 3 * sizeof(double)); memcpy(nominal + 19 * (size_t)b + 16, s.g, 3 * sizeof(double));
        memcpy(P + (size_t)n * n * b, s.P, sizeof(double) * n * n);
        if (prev) prev[b] = s.prev_id;
    }
}

/* ------------------------------------------------------------------ */
/* batched drivers                                                     */
/* ------------------------------------------------------------------ */
static void load_state(fbo_state* s, int n, const double* nom, const double* rot, const double* P, int prev)
{
    memcpy(s->p, nom, 3 * sizeof(double));       memcpy(s->v, nom + 3, 3 * sizeof(double));
    memcpy(s->q, nom + 6, 4 * sizeof(double));   memcpy(s->ba, nom + 10, 3 * sizeof(double));
    memcpy(s->bg, nom + 13, 3 * sizeof(double)); memcpy(s->g, nom + 16, 3 * sizeof(double));
    memcpy(s->R, rot, 9 * sizeof(double));
    memcpy(s->P, P, sizeof(double) * n * n);
    s->prev_id = prev;
}

static void store_state(const fbo_state* s, int n, double* nom, double* rot, double* P, int* prev)
{
    memcpy(nom, s->p, 3 * sizeof(double));       memcpy(nom + 3, s->v, 3 * sizeof(double));
    memcpy(nom + 6, s->q, 4 * sizeof(double));   memcpy(nom + 10, s->ba, 3 * sizeof(double));
    memcpy(nom + 13, s->bg, 3 * sizeof(double)); memcpy(nom + 16, s->g, 3 * sizeof(double));
    memcpy(rot, s->R, 9 * sizeof(double));
    memcpy(P, s->P, sizeof(double) * n * n);
    *prev = s->prev_id;
}

typedef struct {
    int lo, hi, is_correct, is_frame, K, B;
    int nframes, reps; const int* Ks;
    double *nominal, *rot, *P;
    int* prev;
    const fbo_params* prm;
    const double *accel, *gyro, *dt;
    int dt_stride;
    int M, mode;
    const int* ids;
    const double *pos, *quat;
    int* applied;
} batch_job;

static void* batch_worker(void* arg)
{
    batch_job* j = (batch_job*)arg;
    const int n = j->prm->nstate;
    fbo_state s;
    for (int b = j->lo; b < j->hi; ++b) {
        int prev = j->prev ? j->prev[b] : 0;
        load_state(&s, n, j->nominal + 19 * (size_t)b, j->rot + 9 * (size_t)b, j->P + (size_t)n * n * b, prev);
        if (j->nframes > 0) {       /* whole schedule: reps x (frames of K_f predicts + one correct) */
            for (int rep = 0; rep < j->reps; ++rep) {
                int k0 = 0;
                for (int f = 0; f < j->nframes; ++f) {
                    for (int k = 0; k < j->Ks[f]; ++k, ++k0)
                        fbo_predict(&s, j->prm, j->accel + 3 * ((size_t)k0 * j->B + b), j->gyro + 3 * ((size_t)k0 * j->B + b),
                                    j->dt[k0]);
                    fbo_correct(&s, j->prm, j->M, j->ids + ((size_t)f * j->B + b) * j->M,
                                j->pos + 3 * ((size_t)f * j->B + b) * j->M, j->quat + 4 * ((size_t)f * j->B + b) * j->M,
                                j->mode);
                }
            }
        } else if (j->is_frame) {
            for (int k = 0; k < j->K; ++k)
                fbo_predict(&s, j->prm, j->accel + 3 * ((size_t)k * j->B + b), j->gyro + 3 * ((size_t)k * j->B + b),
                            j->dt[k]);
            if (j->M > 0)
                fbo_correct(&s, j->prm, j->M, j->ids + (size_t)j->M * b, j->pos + 3 * (size_t)j->M * b,
                            j->quat + 4 * (size_t)j->M * b, j->mode);
        } else if (!j->is_correct) {
            fbo_predict(&s, j->prm, j->accel + 3 * (size_t)b, j->gyro + 3 * (size_t)b,
                        j->dt[(size_t)b * j->dt_stride]);
        } else {
            int ok = fbo_correct(&s, j->prm, j->M, j->ids + (size_t)j->M * b, j->pos + 3 * (size_t)j->M * b,
                                 j->quat + 4 * (size_t)j->M * b, j->mode);
            if (j->applied) j->applied[b] = ok;
        }
        store_state(&s, n, j->nominal + 19 * (size_t)b, j->rot + 9 * (size_t)b, j->P + (size_t)n * n * b, &prev);
        if (j->prev) j->prev[b] = prev;
    }
    return 0;
}

static void run_batch(batch_job* proto, int B, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if (nthreads > B) nthreads = B > 0 ? B : 1;
    batch_job jobs[256];
    pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = *proto;
        jobs[t].lo = (int)((long long)B * t / nthreads);
        jobs[t].hi = (int)((long long)B * (t + 1) / nthreads);
    }
    if (nthreads == 1) { batch_worker(&jobs[0]); return; }
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], 0, batch_worker, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
}

void fbo_predict_batch(int B, double* nominal, double* rot, double* P, int* prev,
                       const fbo_params* prm, const double* accel, const double* gyro,
                       const double* dt, int dt_stride, int nthreads)
{
    batch_job j;
    memset(&j, 0, sizeof(j));
    j.nominal = nominal; j.rot = rot; j.P = P; j.prev = prev; j.prm = prm;
    j.accel = accel; j.gyro = gyro; j.dt = dt; j.dt_stride = dt_stride;
    run_batch(&j, B, nthreads);
}

void fbo_correct_batch(int B, double* nominal, double* rot, double* P, int* prev,
                       const fbo_params* prm, int M, const int* ids, const double* pos,
                       const double* quat, int mode, int* applied, int nthreads)
{
    batch_job j;
    memset(&j, 0, sizeof(j));
    j.is_correct = 1;
    j.nominal = nominal; j.rot = rot; j.P = P; j.prev = prev; j.prm = prm;
    j.M = M; j.ids = ids; j.pos = pos; j.quat = quat; j.mode = mode; j.applied = applied;
    run_batch(&j, B, nthreads);
}

void fbo_frame_batch(int B, double* nominal, double* rot, double* P, int* prev,
                     const fbo_params* prm, int K, const double* accel, const double* gyro,
                     const double* dt, int M, const int* ids, const double* pos,
                     const double* quat, int mode, int nthreads)
{
    batch_job j;
    memset(&j, 0, sizeof(j));
    j.is_frame = 1; j.K = K; j.B = B;
    j.nominal = nominal; j.rot = rot; j.P = P; j.prev = prev; j.prm = prm;
    j.accel = accel; j.gyro = gyro; j.dt = dt;
    j.M = M; j.ids = ids; j.pos = pos; j.quat = quat; j.mode = mode;
    run_batch(&j, B, nthreads);
}

/* ------------------------------------------------------------------ */
/* init / reset                          */
/* ------------------------------------------------------------------ */
void fbo_init_gravity_bias(int T, const double* accel /*T x 3*/, const double* gyro /*T x 3*/,
                           double g[3], double bg[3])
{   /* InitGravityAndGyrobias.m:36-40 ; filter.cpp:263-276 */
    double ma[3] = { 0, 0, 0 }, mg[3] = { 0, 0, 0 };
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 3; ++i) { ma[i] += accel[3 * t + i]; mg[i] += gyro[3 * t + i]; }
    for (int i = 0; i < 3; ++i) { ma[i] /= T; bg[i] = mg[i] / T; }
    g[0] = 0; g[1] = 0; g[2] = -norm3(ma);
}

/* nearest marker, start threshold 10; -1 if none (shared by init / reset / vision-only) */
static int nearest_marker(int M, const int* ids, const double* pos, double* dist)
{
    int mi = -1;
    double md = 10.0;
    for (int i = 0; i < M; ++i) {
        if (ids[i] < 0) continue;
        double d = norm3(pos + 3 * i);
        if (d < md) { md = d; mi = i; }
    }
    *dist = md;
    return mi;
}

/* IMU pose from one marker measurement:
 * Q_IG = Q_MG (x) Q_ML* (x) Q_IL ; P_IG = -R_IG R_IL' P_ML + P_MG - R_IG P_IL
 * (InitPositionAndQuaternion.m:63-72, ResetState.m:63-72, ComputeVisionOnlyResults.m:64-73,
 *  filter.cpp:374-384,449-456).  normalise = ComputeVisionOnlyResults.m:67 only. */
static void pose_from_marker(const fbo_params* prm, int slot, const double* yp, const double* yq,
                             int normalise, double p[3], double q[4], double R[9])
{
    double qc[4] = { yq[0], -yq[1], -yq[2], -yq[3] }, t[4];
    fbo_quat_mul(prm->marker_quat[slot], qc, t);
    fbo_quat_mul(t, prm->Q_IL, q);
    if (normalise) { double n = norm4(q); for (int i = 0; i < 4; ++i) q[i] /= n; }
    if (prm->dialect == FBO_DIALECT_CPP) fbo_quat_to_rotmat_eigen(q, R);
    else fbo_quat_to_rotmat(q, R);
    double a[3], b[3], c[3];
    mat3t_vec(prm->R_IL, yp, a);            /* R_IL' P_ML */
    mat3_vec(R, a, b);
    mat3_vec(R, prm->P_IL, c);
    for (int i = 0; i < 3; ++i) p[i] = -b[i] + prm->marker_pos[slot][i] - c[i];
}

/* what: 0 = init (InitPositionAndQuaternion.m:38-80 / FILTER::InitializePose filter.cpp:291-399),
 *       1 = reset (ResetState.m:37-80 / FILTER::ResetSystemState filter.cpp:405-477),
 *       2 = vision-only pose into out7 = [p3 q4], state untouched (ComputeVisionOnlyResults.m:39-79).
 * max_dist <= 0 disables the C++ range check (filter.cpp:343-347,432-436).  Returns 1 if applied. */
int fbo_pose_init(fbo_state* s, const fbo_params* prm, int M, const int* ids, const double* pos,
                  const double* quat, int what, double max_dist, double* out7)
{
    double dist;
    int mi = nearest_marker(M, ids, pos, &dist);
    if (mi < 0) return 0;
    if (prm->dialect == FBO_DIALECT_CPP && max_dist > 0 && dist > max_dist) return 0;
    int slot = find_marker(prm, ids[mi]);
    if (slot < 0) return 0;
    double p[3], q[4], R[9];
    pose_from_marker(prm, slot, pos + 3 * mi, quat + 4 * mi, what == 2, p, q, R);
    if (what == 2) {
        memcpy(out7, p, sizeof(p));
        memcpy(out7 + 3, q, sizeof(q));
        return 1;
    }
    memcpy(s->p, p, sizeof(p));
    memcpy(s->q, q, sizeof(q));
    if (what == 0) {
        memcpy(s->R, R, sizeof(R));
        s->g[0] = 9.8; s->g[1] = 0; s->g[2] = 0;           /* InitPositionAndQuaternion.m:79 ; filter.cpp:387 */
    } else {
        for (int i = 0; i < 3; ++i) { s->v[i] = 0; s->ba[i] = 0; }
        if (prm->dialect == FBO_DIALECT_CPP) for (int i = 0; i < 3; ++i) s->bg[i] = 0;   /* filter.cpp:470 */
        else memcpy(s->R, R, sizeof(R));                    /* ResetState.m:77 ; C++ leaves rotmatI2G stale */
    }
    return 1;
}

void fbo_pose_init_batch(int B, double* nominal, double* rot, const fbo_params* prm, int M, const int* ids,
                         const double* pos, const double* quat, int what, double max_dist,
                         const unsigned char* mask, double* out7, int* applied)
{
    fbo_state s;
    for (int b = 0; b < B; ++b) {
        if (applied) applied[b] = 0;
        if (mask && !mask[b]) continue;
        memset(&s, 0, sizeof(s));
        load_state(&s, prm->nstate, nominal + 19 * (size_t)b, rot + 9 * (size_t)b, s.P, 0);
        int ok = fbo_pose_init(&s, prm, M, ids + (size_t)M * b, pos + 3 * (size_t)M * b, quat + 4 * (size_t)M * b,
                               what, max_dist, out7 ? out7 + 7 * (size_t)b : 0);
        if (applied) applied[b] = ok;
        if (what != 2) {
            double Pd[FBO_NMAX * FBO_NMAX];
            int prev;
            store_state(&s, prm->nstate, nominal + 19 * (size_t)b, rot + 9 * (size_t)b, Pd, &prev);
        }
    }
}

/* IMU pre-filter of the live pipeline: y[t] = 0.9 y[t-1] + 0.1 x[t], first sample passed through
 * (FILTER::SetImuData, filter.cpp:36-47).  x: T x 6 (accel, gyro), carry: previous filtered sample or
 * NULL at stream start; in place. */
void fbo_imu_ema(int T, double* x, double* carry, int have_carry)
{
    const double c = 0.1;
    double prev[6];
    if (have_carry) memcpy(prev, carry, sizeof(prev));
    for (int t = 0; t < T; ++t) {
        if (t > 0 || have_carry)
            for (int i = 0; i < 6; ++i) x[6 * t + i] = prev[i] * (1 - c) + x[6 * t + i] * c;
        memcpy(prev, x + 6 * t, sizeof(prev));
    }
    if (T > 0 && carry) memcpy(carry, prev, sizeof(prev));
}

/* CPU-baseline driver: `reps` repetitions of a schedule of nframes camera frames (frame f = Ks[f] predicts +
 * one correct) per filter, one thread team for the whole call.  accel/gyro: sum(Ks) x B x 3, dt: sum(Ks),
 * ids/pos/quat: nframes x B x M x {1,3,4}. */
void fbo_schedule_batch(int B, double* nominal, double* rot, double* P, int* prev, const fbo_params* prm,
                        int nframes, const int* Ks, int reps, const double* accel, const double* gyro,
                        const double* dt, int M, const int* ids, const double* pos, const double* quat,
                        int mode, int nthreads)
{
    batch_job j;
    memset(&j, 0, sizeof(j));
    j.nframes = nframes; j.Ks = Ks; j.reps = reps; j.B = B;
    j.nominal = nominal; j.rot = rot; j.P = P; j.prev = prev; j.prm = prm;
    j.accel = accel; j.gyro = gyro; j.dt = dt;
    j.M = M; j.ids = ids; j.pos = pos; j.quat = quat; j.mode = mode;
    run_batch(&j, B, nthreads);
}
