// vision_device.hpp -- device arithmetic of the step in front of correct():
// flat-port refractive stereo triangulation of the four marker corners and the
// marker pose fit.  One marker per lane.
//
// Reference (paths relative to the upstream repository, C++/src/vision.cpp):
//   RefractionTriangulation :472-618   (air -> glass -> water Snell ray trace, ray mid-point by Cramer)
//   NormalTriangulation     :395-466   (6x4 DLT, null vector)
//   ComputeMarkerPose       :624-759   (plane normal = smallest eigenvector of the corner scatter,
//                                       in-plane axes by a -pi/4 rotation, Eigen quaternion)
#pragma once
#include <hip/hip_runtime.h>

#include "ekf_device.hpp"

namespace fbus {

enum { VIS_REFRACTIVE = 0, VIS_PINHOLE = 1, VIS_CORNERS3D = 2 };

template <typename T>
struct VisConst {
    T R_RL[9], P_LR[3];         // right -> left: R_IL R_IR', P_LI - R_RL P_RI          (vision.cpp:476-481)
    T R_LRn[9], t_LRn[3];       // NormalTriangulation's T_L_R: R_IR R_IL', P_LI - R_LRn P_RI (:402-409)
    T R_RL_inv[9];              // exact inverse of R_RL (the calibration's 6-digit matrices are orthonormal to 1e-5 only)
    T alpha0, alpha1;           // n_air/n_glass, n_glass/n_water
    int sqrt_minus0, sqrt_minus1;   // which branch of :513-522 / :532-541 applies
    T d_air, d_glass;
    T nrm[3];
    // (round 6) constants of the square-port triangulation in the tangent (ekf_meas.hpp::tri_corners_refractive), a = alpha0 alpha1:
    //   tri[2 i], tri[2 i + 1] = a R_RL(i, 0), a R_RL(i, 1);  tri[6 + i] = R_RL(i, 2) (d_air + d_glass) + P_LR(i);
    //   tri[9] = d_air / a,  tri[10] = d_glass alpha0 / a,  tri[11] = a
    T tri[12];
};

template <typename T> __device__ __forceinline__ T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename T> __device__ __forceinline__ void cross3(const T* a, const T* b, T* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T> __device__ __forceinline__ void m3v(const T* R, const T* x, T* y)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) y[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2];
}
template <typename T> __device__ __forceinline__ T det3cols(const T* c0, const T* c1, const T* c2)
{
    return c0[0] * (c1[1] * c2[2] - c1[2] * c2[1]) - c1[0] * (c0[1] * c2[2] - c0[2] * c2[1])
         + c2[0] * (c0[1] * c1[2] - c0[2] * c1[1]);
}

// fp32: square root, reciprocal square root and reciprocal from the 1-ulp hardware estimates (v_rsq_f32 / v_rcp_f32) + one Newton
// step, 3-4 instructions instead of the 10-20 of the IEEE sequences; the triangulation of one corner has 6 square roots and 8
// divisions among its ~300 other instructions (r3: correct_corners at 65 536 filters x 16 slots 68.9 -> see DESIGN.md 4.2c).
// Arguments are sums of squares of unit-ish vectors, cosines of refracted rays and a Cramer determinant: positive, far from 0.
__device__ __forceinline__ float vs_rsq(float x) { const float r = __builtin_amdgcn_rsqf(x); return r * (1.5f - 0.5f * x * r * r); }
__device__ __forceinline__ double vs_rsq(double x) { return 1.0 / sqrt(x); }                // fp64: as before
__device__ __forceinline__ float vs_sqrt(float x)
{
    const float r = __builtin_amdgcn_rsqf(x), s_ = x * r;
    return x >= 1.17549435e-38f ? s_ + 0.5f * r * (x - s_ * s_) : 0.0f;      // (v_rsq_f32 of a denormal is +inf: treated as 0)
}
__device__ __forceinline__ double vs_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float vs_div(float a, float x) { const float r = __builtin_amdgcn_rcpf(x); return a * (r * (2.0f - x * r)); }
__device__ __forceinline__ double vs_div(double a, double x) { return a / x; }          // fp64: the reference's own operation

// One Snell refraction at the flat port (unit ray r, interface normal nv).
template <typename T>
__device__ __forceinline__ void refract(const T* r, const T* nv, T alpha, int sqrt_minus, T* out, T& v)
{
    v = dot3(r, nv);
    const T root = vs_sqrt(T(1) - alpha * alpha * (T(1) - v * v));
    const T beta = sqrt_minus ? (root - alpha * v) : (alpha * v - root);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = alpha * r[i] + beta * nv[i];
}

// vision.cpp:496-599 for one corner: left/right normalised image points -> point in the left camera frame.
template <typename T>
__device__ __forceinline__ void refraction_corner(const VisConst<T>& vc, T xl, T yl, T xr, T yr, T* out)
{
    const T pl[3] = { xl, yl, T(1) }, pr[3] = { xr, yr, T(1) };
    const T il = vs_rsq(dot3(pl, pl)), ir = vs_rsq(dot3(pr, pr));
    const T r0L[3] = { pl[0] * il, pl[1] * il, pl[2] * il }, r0R[3] = { pr[0] * ir, pr[1] * ir, pr[2] * ir };
    T r1L[3], r1R[3], r2L[3], r2R[3], v0L, v0R, v1L, v1R;
    refract(r0L, vc.nrm, vc.alpha0, vc.sqrt_minus0, r1L, v0L);
    refract(r0R, vc.nrm, vc.alpha0, vc.sqrt_minus0, r1R, v0R);
    refract(r1L, vc.nrm, vc.alpha1, vc.sqrt_minus1, r2L, v1L);
    refract(r1R, vc.nrm, vc.alpha1, vc.sqrt_minus1, r2R, v1R);
    T P1L[3], P1R[3];
    const T aL = vs_div(vc.d_air, v0L), gL = vs_div(vc.d_glass, v1L), aR = vs_div(vc.d_air, v0R), gR = vs_div(vc.d_glass, v1R);
#pragma unroll
    for (int i = 0; i < 3; ++i) {                        // :546-552 exit points on the outer glass face
        P1L[i] = aL * r0L[i] + gL * r1L[i];
        P1R[i] = aR * r0R[i] + gR * r1R[i];
    }
    T r2RL[3], P1RL[3];                                  // :555-556 right ray in the left frame
    m3v(vc.R_RL, r2R, r2RL);
    m3v(vc.R_RL, P1R, P1RL);
    T cr[3], dP[3];
    cross3(r2L, r2RL, cr);
#pragma unroll
    for (int i = 0; i < 3; ++i) { P1RL[i] += vc.P_LR[i]; dP[i] = P1RL[i] - P1L[i]; }
    const T d3 = det3cols(cr, r2L, r2RL);                // :559-595 Cramer (the determinant is |r2L x r2R|^2 > 0)
    const T t1 = vs_div(det3cols(cr, dP, r2RL), d3), t2 = -vs_div(det3cols(cr, r2L, dP), d3);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T P = T(0.5) * (P1L[i] + t1 * r2L[i] + P1RL[i] + t2 * r2RL[i]);
        out[i] = (i < 2) ? -P : P;                       // :597-599
    }
}

// (The forward flat-port projection with its Jacobian -- the north-star extension the reference has no counterpart of -- lives in
// ekf_meas.hpp since round 4: closed-form thin-port start + one Halley step in double.)

// cyclic Jacobi on a symmetric NxN matrix held in registers (N = 3 or 4): A -> diag, V = eigenvectors (columns).
template <typename T, int N>
__device__ __forceinline__ void jacobi_sym(T (&A)[N * N], T (&V)[N * N])
{
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i * N + j] = (i == j) ? T(1) : T(0);
    for (int sweep = 0; sweep < 12; ++sweep) {
        T off = T(0), dg = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            dg += A[i * N + i] * A[i * N + i];
#pragma unroll
            for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j];
        }
        if (!(off > dg * T(sizeof(T) == 4 ? 1e-14 : 1e-30))) break;
#pragma unroll
        for (int p = 0; p < N; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                const T apq = A[p * N + q];
                const T theta = (A[q * N + q] - A[p * N + p]) / (T(2) * apq);
                T t = T(1) / (fb_abs(theta) + fb_sqrt(theta * theta + T(1)));
                t = (theta < T(0)) ? -t : t;
                t = (apq == T(0)) ? T(0) : t;            // nothing to rotate
                const T c = T(1) / fb_sqrt(t * t + T(1)), s = t * c;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const T akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - s * akq;
                    A[k * N + q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const T apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - s * aqk;
                    A[q * N + k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const T vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - s * vkq;
                    V[k * N + q] = s * vkp + c * vkq;
                }
            }
    }
}

// vision.cpp:411-446 for one corner.
template <typename T>
__device__ __forceinline__ void pinhole_corner(const VisConst<T>& vc, T xl, T yl, T xr, T yr, T* out)
{
    const T l[3] = { xl, yl, T(1) }, r[3] = { xr, yr, T(1) };
    T A[24];
    const T Sl[9] = { T(0), -l[2], l[1], l[2], T(0), -l[0], -l[1], l[0], T(0) };
    const T Sr[9] = { T(0), -r[2], r[1], r[2], T(0), -r[0], -r[1], r[0], T(0) };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A[i * 4 + j] = Sl[3 * i + j];
            A[(3 + i) * 4 + j] = Sr[3 * i] * vc.R_LRn[j] + Sr[3 * i + 1] * vc.R_LRn[3 + j] + Sr[3 * i + 2] * vc.R_LRn[6 + j];
        }
        A[i * 4 + 3] = T(0);
        A[(3 + i) * 4 + 3] = Sr[3 * i] * vc.t_LRn[0] + Sr[3 * i + 1] * vc.t_LRn[1] + Sr[3 * i + 2] * vc.t_LRn[2];
    }
    T G[16], V[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T acc = T(0);
#pragma unroll
            for (int k = 0; k < 6; ++k) acc += A[k * 4 + i] * A[k * 4 + j];
            G[i * 4 + j] = acc;
        }
    jacobi_sym<T, 4>(G, V);
    T best = G[0], X[4] = { V[0], V[4], V[8], V[12] };
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const bool lt = G[k * 4 + k] < best;
        best = lt ? G[k * 4 + k] : best;
#pragma unroll
        for (int i = 0; i < 4; ++i) X[i] = lt ? V[i * 4 + k] : X[i];
    }
    const T Pn[3] = { -X[0] / X[3], -X[1] / X[3], X[2] / X[3] };
    const T sg = (Pn[2] < T(0)) ? T(-1) : T(1);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = sg * Pn[i];
}

template <typename T>
__device__ __forceinline__ void rotmat_to_quat_dev(const T* R, T* q)
{   // Eigen Quaterniond(Matrix3d), vision.cpp:758
    T t = R[0] + R[4] + R[8];
    if (t > T(0)) {
        t = fb_sqrt(t + T(1));
        q[0] = T(0.5) * t;
        t = T(0.5) / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {           // i = 0
        t = fb_sqrt(R[0] - R[4] - R[8] + T(1));
        q[1] = T(0.5) * t; t = T(0.5) / t;
        q[0] = (R[7] - R[5]) * t; q[2] = (R[3] + R[1]) * t; q[3] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {            // i = 1
        t = fb_sqrt(R[4] - R[8] - R[0] + T(1));
        q[2] = T(0.5) * t; t = T(0.5) / t;
        q[0] = (R[2] - R[6]) * t; q[3] = (R[7] + R[5]) * t; q[1] = (R[1] + R[3]) * t;
    } else {                                             // i = 2
        t = fb_sqrt(R[8] - R[0] - R[4] + T(1));
        q[3] = T(0.5) * t; t = T(0.5) / t;
        q[0] = (R[3] - R[1]) * t; q[1] = (R[2] + R[6]) * t; q[2] = (R[5] + R[7]) * t;
    }
}

// vision.cpp:634-759 : four corner positions (C[12]) -> marker position, quaternion (wxyz).
template <typename T>
__device__ __forceinline__ void marker_pose(const T* C, T* pos, T* quat)
{
    T M[9] = { T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0) };
    constexpr int ia[6] = { 1, 2, 3, 2, 3, 3 }, ib[6] = { 0, 0, 0, 1, 1, 2 };
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const T v[3] = { C[3 * ia[k]] - C[3 * ib[k]], C[3 * ia[k] + 1] - C[3 * ib[k] + 1], C[3 * ia[k] + 2] - C[3 * ib[k] + 2] };
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M[3 * i + j] += v[i] * v[j];
    }
    T V[9];
    jacobi_sym<T, 3>(M, V);
    T best = M[0], Z[3] = { V[0], V[3], V[6] };
#pragma unroll
    for (int k = 1; k < 3; ++k) {
        const bool lt = M[4 * k] < best;
        best = lt ? M[4 * k] : best;
#pragma unroll
        for (int i = 0; i < 3; ++i) Z[i] = lt ? V[3 * i + k] : Z[i];
    }
    const T zi = T(1) / fb_sqrt(dot3(Z, Z));
    T sgn;                                               // :700-709 sign rule
    if (Z[2] * zi > T(0.1)) sgn = T(-1);
    else if (Z[2] * zi < T(-0.1)) sgn = T(1);
    else sgn = -((C[0] < T(0)) ? T(-1) : T(1)) * ((Z[0] < T(0)) ? T(-1) : T(1));
#pragma unroll
    for (int i = 0; i < 3; ++i) Z[i] *= zi * sgn;
    const T sum[3] = { C[0] + C[3] + C[6] + C[9], C[1] + C[4] + C[7] + C[10], C[2] + C[5] + C[8] + C[11] };
    const T D = T(0.25) * dot3(Z, sum);
    const T t1 = dot3(Z, C) - D, t2 = dot3(Z, C + 3) - D, t4 = dot3(Z, C + 9) - D;
    T P1[3], V12[3], V14[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P1[i] = C[i] - t1 * Z[i];
        V12[i] = (C[3 + i] - t2 * Z[i]) - P1[i];
        V14[i] = (C[9 + i] - t4 * Z[i]) - P1[i];
    }
    const T i12 = T(1) / fb_sqrt(dot3(V12, V12)), i14 = T(1) / fb_sqrt(dot3(V14, V14));
    const T m[3] = { V12[0] * i12 + V14[0] * i14, V12[1] * i12 + V14[1] * i14, V12[2] * i12 + V14[2] * i14 };
    // Eigen AngleAxisd(-M_PI/4, Z) with the reference's M_PI = 3.1415926 (common.hpp:14)
    const T c = T(0.707106790659974), s = T(-0.707106771713121);
    T Rm[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Rm[3 * i + j] = (T(1) - c) * Z[i] * Z[j] + ((i == j) ? c : T(0));
    Rm[1] -= s * Z[2]; Rm[2] += s * Z[1];
    Rm[3] += s * Z[2]; Rm[5] -= s * Z[0];
    Rm[6] -= s * Z[1]; Rm[7] += s * Z[0];
    T X[3], Y[3];
    m3v(Rm, m, X);
    const T im = T(1) / fb_sqrt(dot3(m, m));
#pragma unroll
    for (int i = 0; i < 3; ++i) X[i] *= im;
    cross3(Z, X, Y);
    const T R[9] = { X[0], Y[0], Z[0], X[1], Y[1], Z[1], X[2], Y[2], Z[2] };
    rotmat_to_quat_dev(R, quat);
#pragma unroll
    for (int i = 0; i < 3; ++i) pos[i] = P1[i];
}

}  // namespace fbus
