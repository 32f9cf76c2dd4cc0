/*
 * vision_oracle.c -- TEST INFRASTRUCTURE (see vision_oracle.h).
 * fp64 restatement of C++/src/vision.cpp:395-759 (triangulation + marker pose).
 */
#include "vision_oracle.h"

#include <math.h>
#include <string.h>

#define FBV_REF_PI 3.1415926   /* C++/include/common.hpp:14 redefines M_PI to this value */

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double nrm3(const double* a) { return sqrt(dot3(a, a)); }
static void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static void m3v(const double* R, const double* x, double* y)
{
    for (int i = 0; i < 3; ++i) y[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2];
}
static double det3cols(const double* c0, const double* c1, const double* c2)
{
    return c0[0] * (c1[1] * c2[2] - c1[2] * c2[1])
         - c1[0] * (c0[1] * c2[2] - c0[2] * c2[1])
         + c2[0] * (c0[1] * c1[2] - c0[2] * c1[1]);
}

/* cyclic Jacobi for a symmetric n x n matrix (n <= 4); V columns = eigenvectors */
static void jacobi_sym(double* A, double* V, int n)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

static void rotmat_to_quat(const double R[9], double q[4])
{   /* Eigen Quaterniond(Matrix3d), vision.cpp:758 */
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

void fbv_default_params(fbv_params* p)
{
    /* C++/config/camerainfo1.yml (== matlab/config/camerainfo.yml) TSC blocks, raw */
    static const double RL[9] = { -0.999862, 0.015685, -0.00548,
                                  -0.015639, -0.999843, -0.00827,
                                  -0.005609, -0.008183, 0.999951 };
    static const double PL[3] = { 0.059967, 0.000127837, -0.002 };
    static const double RR[9] = { -0.999826, 0.00929485, -0.0161445,
                                  -0.00937869, -0.999942, 0.00514829,
                                  -0.0160959, 0.00529897, 0.999857 };
    static const double PR[3] = { -0.0601272, 0.000124714, -0.002 };
    memcpy(p->R_IL, RL, sizeof(RL)); memcpy(p->P_LI, PL, sizeof(PL));
    memcpy(p->R_IR, RR, sizeof(RR)); memcpy(p->P_RI, PR, sizeof(PR));
    /* C++/config/paramconfig.yml:31-42 */
    p->n_air = 1.00; p->n_glass = 1.49; p->n_water = 1.32;
    p->d_air = 0.002; p->d_glass = 0.02;
    p->normal[0] = 0; p->normal[1] = 0; p->normal[2] = 1;
}

/* one Snell refraction of unit ray r at an interface with normal nv; `first`
 * selects which comparison the reference makes (vision.cpp:513-522 vs 532-541) */
static void refract(const double* r, const double* nv, double alpha, int sqrt_minus, double* out, double* v_out)
{
    double v = dot3(r, nv);
    double root = sqrt(1 - alpha * alpha * (1 - v * v));
    double beta = sqrt_minus ? (root - alpha * v) : (alpha * v - root);
    for (int i = 0; i < 3; ++i) out[i] = alpha * r[i] + beta * nv[i];
    *v_out = v;
}

void fbv_refraction_triangulate(const fbv_params* p, const double left[8], const double right[8],
                                double corners[12])
{
    /* vision.cpp:476-481 */
    double R_RL[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += p->R_IL[3 * i + k] * p->R_IR[3 * j + k];
            R_RL[3 * i + j] = acc;
        }
    double tmp[3], P_LR[3];
    m3v(R_RL, p->P_RI, tmp);
    for (int i = 0; i < 3; ++i) P_LR[i] = p->P_LI[i] - tmp[i];

    for (int c = 0; c < 4; ++c) {
        double pl[3] = { left[2 * c], left[2 * c + 1], 1.0 };
        double pr[3] = { right[2 * c], right[2 * c + 1], 1.0 };
        double nl = nrm3(pl), nr = nrm3(pr);
        double r0L[3], r0R[3];
        for (int i = 0; i < 3; ++i) { r0L[i] = pl[i] / nl; r0R[i] = pr[i] / nr; }

        double alpha0 = p->n_air / p->n_glass;          /* :508-524 */
        double r1L[3], r1R[3], v0L, v0R;
        refract(r0L, p->normal, alpha0, p->n_air < p->n_glass, r1L, &v0L);
        refract(r0R, p->normal, alpha0, p->n_air < p->n_glass, r1R, &v0R);
        double alpha1 = p->n_glass / p->n_water;        /* :527-543 */
        double r2L[3], r2R[3], v1L, v1R;
        refract(r1L, p->normal, alpha1, p->n_glass > p->n_water, r2L, &v1L);
        refract(r1R, p->normal, alpha1, p->n_glass > p->n_water, r2R, &v1R);

        double P1L[3], P1R[3];                          /* :546-552 */
        for (int i = 0; i < 3; ++i) {
            P1L[i] = (p->d_air / v0L) * r0L[i] + (p->d_glass / v1L) * r1L[i];
            P1R[i] = (p->d_air / v0R) * r0R[i] + (p->d_glass / v1R) * r1R[i];
        }
        double r2RL[3], P1RL[3];                        /* :555-556 */
        m3v(R_RL, r2R, r2RL);
        m3v(R_RL, P1R, P1RL);
        for (int i = 0; i < 3; ++i) P1RL[i] += P_LR[i];

        double cr[3], dP[3];                            /* :559-595 (Cramer) */
        cross3(r2L, r2RL, cr);
        for (int i = 0; i < 3; ++i) dP[i] = P1RL[i] - P1L[i];
        double d1 = det3cols(cr, dP, r2RL);
        double d2 = det3cols(cr, r2L, dP);
        double d3 = det3cols(cr, r2L, r2RL);
        double t1 = d1 / d3, t2 = -d2 / d3;
        for (int i = 0; i < 3; ++i) {
            double P = 0.5 * (P1L[i] + t1 * r2L[i] + P1RL[i] + t2 * r2RL[i]);
            corners[3 * c + i] = (i < 2) ? -P : P;      /* :597-599 */
        }
    }
}

void fbv_normal_triangulate(const fbv_params* p, const double left[8], const double right[8],
                            double corners[12])
{
    /* vision.cpp:399-409 */
    double Rlr[9], t[3], tmp[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += p->R_IR[3 * i + k] * p->R_IL[3 * j + k];
            Rlr[3 * i + j] = acc;
        }
    m3v(Rlr, p->P_RI, tmp);
    for (int i = 0; i < 3; ++i) t[i] = p->P_LI[i] - tmp[i];
    for (int c = 0; c < 4; ++c) {
        double l[3] = { left[2 * c], left[2 * c + 1], 1.0 };
        double r[3] = { right[2 * c], right[2 * c + 1], 1.0 };
        double A[24];                                   /* 6 x 4, :424-426 */
        double Sl[9] = { 0, -l[2], l[1],  l[2], 0, -l[0],  -l[1], l[0], 0 };
        double Sr[9] = { 0, -r[2], r[1],  r[2], 0, -r[0],  -r[1], r[0], 0 };
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) A[i * 4 + j] = Sl[3 * i + j];
            A[i * 4 + 3] = 0;
            for (int j = 0; j < 3; ++j) {
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += Sr[3 * i + k] * Rlr[3 * k + j];
                A[(3 + i) * 4 + j] = acc;
            }
            A[(3 + i) * 4 + 3] = Sr[3 * i] * t[0] + Sr[3 * i + 1] * t[1] + Sr[3 * i + 2] * t[2];
        }
        /* right singular vector of the smallest singular value (:429-434)
         * = eigenvector of A'A with the smallest eigenvalue */
        double G[16], V[16];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0;
                for (int k = 0; k < 6; ++k) acc += A[k * 4 + i] * A[k * 4 + j];
                G[i * 4 + j] = acc;
            }
        jacobi_sym(G, V, 4);
        int kmin = 0;
        for (int k = 1; k < 4; ++k) if (G[k * 4 + k] < G[kmin * 4 + kmin]) kmin = k;
        double X[4] = { V[kmin], V[4 + kmin], V[8 + kmin], V[12 + kmin] };
        double Pn[3] = { -X[0] / X[3], -X[1] / X[3], X[2] / X[3] };   /* :441-444 */
        double sg = (Pn[2] < 0) ? -1.0 : 1.0;                         /* :445, Signum */
        for (int i = 0; i < 3; ++i) corners[3 * c + i] = sg * Pn[i];
    }
}

void fbv_marker_pose(const double corners[12], double pos[3], double quat[4], double rot[9])
{
    const double* C0 = corners;
    const double* C1 = corners + 3;
    const double* C2 = corners + 6;
    const double* C3 = corners + 9;
    /* vision.cpp:634-675 : scatter of the six corner differences */
    const double* a[6] = { C1, C2, C3, C2, C3, C3 };
    const double* b[6] = { C0, C0, C0, C1, C1, C2 };
    double M[9] = { 0 };
    for (int k = 0; k < 6; ++k) {
        double v[3] = { a[k][0] - b[k][0], a[k][1] - b[k][1], a[k][2] - b[k][2] };
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M[3 * i + j] += v[i] * v[j];
    }
    /* :677-709 : eigenvector of the smallest eigenvalue = plane normal, sign rule */
    double V[9];
    jacobi_sym(M, V, 3);
    int kmin = 0;
    for (int k = 1; k < 3; ++k) if (M[4 * k] < M[4 * kmin]) kmin = k;
    double Z[3] = { V[kmin], V[3 + kmin], V[6 + kmin] };
    double zn = nrm3(Z);
    for (int i = 0; i < 3; ++i) Z[i] /= zn;
    if (Z[2] > 0.1) {
        for (int i = 0; i < 3; ++i) Z[i] = -Z[i];
    } else if (Z[2] < -0.1) {
        /* keep */
    } else {
        double s = -((C0[0] < 0) ? -1.0 : 1.0) * ((Z[0] < 0) ? -1.0 : 1.0);
        for (int i = 0; i < 3; ++i) Z[i] *= s;
    }
    /* :711-721 : plane offset and corner projections */
    double sum[3] = { C0[0] + C1[0] + C2[0] + C3[0], C0[1] + C1[1] + C2[1] + C3[1], C0[2] + C1[2] + C2[2] + C3[2] };
    double D = 0.25 * dot3(Z, sum);
    double P1[3], P2[3], P4[3];
    double t1 = dot3(Z, C0) - D, t2 = dot3(Z, C1) - D, t4 = dot3(Z, C3) - D;
    for (int i = 0; i < 3; ++i) {
        P1[i] = C0[i] - t1 * Z[i];
        P2[i] = C1[i] - t2 * Z[i];
        P4[i] = C3[i] - t4 * Z[i];
    }
    /* :736-751 : in-plane axes */
    double V12[3], V14[3], m[3];
    for (int i = 0; i < 3; ++i) { V12[i] = P2[i] - P1[i]; V14[i] = P4[i] - P1[i]; }
    double n12 = nrm3(V12), n14 = nrm3(V14);
    for (int i = 0; i < 3; ++i) m[i] = V12[i] / n12 + V14[i] / n14;
    double ang = -FBV_REF_PI / 4, c = cos(ang), s = sin(ang);
    double Rm[9];                       /* Eigen AngleAxisd(ang, Z).matrix() */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rm[3 * i + j] = (1 - c) * Z[i] * Z[j] + ((i == j) ? c : 0.0);
    Rm[1] -= s * Z[2]; Rm[2] += s * Z[1];
    Rm[3] += s * Z[2]; Rm[5] -= s * Z[0];
    Rm[6] -= s * Z[1]; Rm[7] += s * Z[0];
    double X[3], Y[3], mn = nrm3(m);
    m3v(Rm, m, X);
    for (int i = 0; i < 3; ++i) X[i] /= mn;
    cross3(Z, X, Y);
    /* :753-762 */
    for (int i = 0; i < 3; ++i) { rot[3 * i] = X[i]; rot[3 * i + 1] = Y[i]; rot[3 * i + 2] = Z[i]; }
    rotmat_to_quat(rot, quat);
    memcpy(pos, P1, 3 * sizeof(double));
}

/* ------------------------------------------------------------------------------------------------------------
 * Forward flat-port projection (NOT in the reference, which only back-projects): the normalised image point
 * whose ray -- refracted air -> glass -> water exactly as RefractionTriangulation builds it (vision.cpp:505-552) --
 * passes through a given point.  The ray stays in the plane spanned by the port normal and the point, so this
 * is one monotone scalar equation in t = tan(theta_air):
 *     rho = d_air t + d_glass tan(theta_glass) + (z - d_air - d_glass) tan(theta_water),
 *     sin(theta_glass) = (n_air / n_glass) sin(theta_air),  sin(theta_water) = (n_air / n_water) sin(theta_air),
 * with z the depth of the point along the normal and rho its distance from the axis; solved by Newton.
 * Pinned through the reference's own back-projection: project -> fbv_refraction_triangulate returns the point
 * (tests/test_oracle_cpu.py), and the recorded corners.txt pixels are recovered from their triangulated points.
 * Xp: point in the camera's refraction frame (the frame of the rays of vision.cpp:496-552, i.e. before the axis
 * flip of :597-599).  Returns 0 if the point is not in front of the port. */
static double lateral_offset(const fbv_params* p, double t, double z, double* dLdt)
{
    const double a0 = p->n_air / p->n_glass, a1 = p->n_air / p->n_water;
    const double s = t / sqrt(1 + t * t), dsdt = 1 / ((1 + t * t) * sqrt(1 + t * t));
    const double cg = sqrt(1 - a0 * a0 * s * s), cw = sqrt(1 - a1 * a1 * s * s);
    const double zw = z - p->d_air - p->d_glass;
    if (dLdt) *dLdt = p->d_air + (p->d_glass * a0 / (cg * cg * cg) + zw * a1 / (cw * cw * cw)) * dsdt;
    return p->d_air * t + p->d_glass * a0 * s / cg + zw * a1 * s / cw;
}

int fbv_refraction_project(const fbv_params* p, const double Xp[3], double uv[2])
{
    const double* n = p->normal;
    const double z = dot3(Xp, n);
    if (!(z > p->d_air + p->d_glass)) return 0;
    double lat[3] = { Xp[0] - z * n[0], Xp[1] - z * n[1], Xp[2] - z * n[2] };
    const double rho = nrm3(lat);
    /* field of view of the flat port: in water the ray cannot lean further than asin(n_air / n_water) (49.3 deg) however
     * steep it leaves the camera; points beyond 0.9 of that limit are treated as not visible (no rows) */
    {
        const double a1 = p->n_air / p->n_water;
        if (!(rho < 0.9 * (z - p->d_air - p->d_glass) * a1 / sqrt(1 - a1 * a1))) return 0;
    }
    double t = rho / z;                               /* pin-hole start; the water bends the ray towards the axis */
    for (int it = 0; it < 60; ++it) {
        double dL, L = lateral_offset(p, t, z, &dL);
        double step = (L - rho) / dL;
        t -= step;
        if (t < 0) t = 0;
        if (fabs(step) <= 1e-16 * (1 + t)) break;
    }
    const double k = (rho > 0) ? t / rho : 0.0;
    const double D[3] = { n[0] + k * lat[0], n[1] + k * lat[1], n[2] + k * lat[2] };
    uv[0] = D[0] / D[2];
    uv[1] = D[1] / D[2];
    return 1;
}

/* a point given in the LEFT camera frame as the triangulation returns it (after the flip of vision.cpp:597-599)
 * -> normalised image point of ONE camera (which = 0 left, 1 right) */
int fbv_project_camera(const fbv_params* p, const double Xcam[3], int which, double uv[2])
{
    const double XL[3] = { -Xcam[0], -Xcam[1], Xcam[2] };            /* undo :597-599 */
    if (which == 0) return fbv_refraction_project(p, XL, uv);
    double R_RL[9], tmp[3], P_LR[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += p->R_IL[3 * i + k] * p->R_IR[3 * j + k];
            R_RL[3 * i + j] = acc;
        }
    m3v(R_RL, p->P_RI, tmp);
    for (int i = 0; i < 3; ++i) P_LR[i] = p->P_LI[i] - tmp[i];
    /* X_L = R_RL X_R + P_LR (vision.cpp:555-556).  The calibration files carry 6-digit rotation matrices, so R_RL is
     * orthonormal to 1e-5 only: the exact inverse is used, not the transpose, to stay consistent with the back-projection */
    double d[3] = { XL[0] - P_LR[0], XL[1] - P_LR[1], XL[2] - P_LR[2] }, XR[3];
    const double c0[3] = { R_RL[0], R_RL[3], R_RL[6] }, c1[3] = { R_RL[1], R_RL[4], R_RL[7] }, c2[3] = { R_RL[2], R_RL[5], R_RL[8] };
    const double det = det3cols(c0, c1, c2);
    XR[0] = det3cols(d, c1, c2) / det; XR[1] = det3cols(c0, d, c2) / det; XR[2] = det3cols(c0, c1, d) / det;
    return fbv_refraction_project(p, XR, uv);
}

/* ------------------------------------------------------------------------------------------------------------
 * ANALYTIC Jacobian of the forward projection (round 6: the second pixel oracle, beside the central-difference one of
 * fbo_correct_pixels).  Written from the forward model above by the implicit-function theorem -- not from the device code:
 *     F(t; z, rho) = L(t, z) - rho = 0,   L = d_air t + d_glass a0 s / cg + (z - d_air - d_glass) a1 s / cw   (lateral_offset)
 *     dt = (d rho - L_z dz) / L_t,        L_z = a1 s / cw,   L_t = lateral_offset's dLdt
 *     z = n . X,  lat = X - z n,  rho = |lat|:   dz = n' dX,   d rho = lat' dX / rho   (lat is orthogonal to n)
 *     k = t / rho,  D = n + k lat:               dk = dt / rho - t d rho / rho^2,   dD = lat dk + k (I - n n') dX
 *     uv = (D0, D1) / D2:                        d uv = [ dD0 - uv0 dD2, dD1 - uv1 dD2 ] / D2
 * On the axis (rho -> 0) k tends to 1 / (d_air + d_glass a0 + z_w a1) and the lat dk term vanishes.
 * The ray geometry is that of vision.cpp:505-552 run forward.  J = d uv / d Xp, 2 x 3 row-major. */
int fbv_refraction_project_jac(const fbv_params* p, const double Xp[3], double uv[2], double J[6])
{
    if (!fbv_refraction_project(p, Xp, uv)) return 0;
    const double* n = p->normal;
    const double a0 = p->n_air / p->n_glass, a1 = p->n_air / p->n_water;
    const double z = dot3(Xp, n), zw = z - p->d_air - p->d_glass;
    const double lat[3] = { Xp[0] - z * n[0], Xp[1] - z * n[1], Xp[2] - z * n[2] };
    const double rho = nrm3(lat);
    double dD[9];                                   /* dD / dX, 3 x 3 */
    double k;
    if (rho > 1e-12 * (1 + fabs(z))) {
        double t;
        {   /* the scalar solve of fbv_refraction_project once more (cheap; keeps this function independent of how uv was formed) */
            t = rho / z;
            for (int it = 0; it < 60; ++it) {
                double dL, L = lateral_offset(p, t, z, &dL);
                double step = (L - rho) / dL;
                t -= step;
                if (t < 0) t = 0;
                if (fabs(step) <= 1e-16 * (1 + t)) break;
            }
        }
        double Lt;
        (void)lateral_offset(p, t, z, &Lt);
        const double s = t / sqrt(1 + t * t), cw = sqrt(1 - a1 * a1 * s * s);
        const double Lz = a1 * s / cw;
        k = t / rho;
        for (int j = 0; j < 3; ++j) {
            const double drho = lat[j] / rho, dz = n[j];
            const double dt = (drho - Lz * dz) / Lt;
            const double dk = dt / rho - t * drho / (rho * rho);
            for (int i = 0; i < 3; ++i) dD[3 * i + j] = lat[i] * dk + k * ((i == j ? 1.0 : 0.0) - n[i] * n[j]);
        }
    } else {
        k = 1.0 / (p->d_air + p->d_glass * a0 + zw * a1);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) dD[3 * i + j] = k * ((i == j ? 1.0 : 0.0) - n[i] * n[j]);
    }
    const double D2 = n[2] + k * lat[2];
    for (int j = 0; j < 3; ++j) {
        J[j] = (dD[j] - uv[0] * dD[6 + j]) / D2;
        J[3 + j] = (dD[3 + j] - uv[1] * dD[6 + j]) / D2;
    }
    return 1;
}

/* as fbv_project_camera, with J = d uv / d Xcam (2 x 3): the chain through the axis flip (vision.cpp:597-599) and, for the right
 * camera, the exact inverse of X_L = R_RL X_R + P_LR (vision.cpp:555-556) */
int fbv_project_camera_jac(const fbv_params* p, const double Xcam[3], int which, double uv[2], double J[6])
{
    const double XL[3] = { -Xcam[0], -Xcam[1], Xcam[2] };
    const double flip[3] = { -1.0, -1.0, 1.0 };
    double Jp[6];
    if (which == 0) {
        if (!fbv_refraction_project_jac(p, XL, uv, Jp)) return 0;
        for (int r = 0; r < 2; ++r)
            for (int j = 0; j < 3; ++j) J[3 * r + j] = Jp[3 * r + j] * flip[j];
        return 1;
    }
    double R_RL[9], tmp[3], P_LR[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += p->R_IL[3 * i + k] * p->R_IR[3 * j + k];
            R_RL[3 * i + j] = acc;
        }
    m3v(R_RL, p->P_RI, tmp);
    for (int i = 0; i < 3; ++i) P_LR[i] = p->P_LI[i] - tmp[i];
    const double d[3] = { XL[0] - P_LR[0], XL[1] - P_LR[1], XL[2] - P_LR[2] };
    const double c0[3] = { R_RL[0], R_RL[3], R_RL[6] }, c1[3] = { R_RL[1], R_RL[4], R_RL[7] }, c2[3] = { R_RL[2], R_RL[5], R_RL[8] };
    const double det = det3cols(c0, c1, c2);
    const double XR[3] = { det3cols(d, c1, c2) / det, det3cols(c0, d, c2) / det, det3cols(c0, c1, d) / det };
    if (!fbv_refraction_project_jac(p, XR, uv, Jp)) return 0;
    /* inverse of R_RL column by column (Cramer on the unit vectors): Ainv[:, j] = solve(R_RL, e_j) */
    double Ainv[9];
    for (int j = 0; j < 3; ++j) {
        double e[3] = { 0, 0, 0 };
        e[j] = 1.0;
        Ainv[j] = det3cols(e, c1, c2) / det; Ainv[3 + j] = det3cols(c0, e, c2) / det; Ainv[6 + j] = det3cols(c0, c1, e) / det;
    }
    for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += Jp[3 * r + k] * Ainv[3 * k + j];
            J[3 * r + j] = acc * flip[j];
        }
    return 1;
}

/* both cameras; returns 1 only if the point is in view of every camera asked for */
int fbv_project_stereo(const fbv_params* p, const double Xcam[3], double uvL[2], double uvR[2])
{
    int ok = fbv_project_camera(p, Xcam, 0, uvL);
    if (uvR) ok = fbv_project_camera(p, Xcam, 1, uvR) && ok;
    return ok;
}
