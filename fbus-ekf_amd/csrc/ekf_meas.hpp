// ekf_meas.hpp -- correct() from corner PIXELS (round 4): the north star's own MeasureUpdate -- flat-port refractive stereo
// reprojection of the ArUco corners, per-corner 2 x N Jacobians -- as a first-class kernel.
//
// Reference geometry and algebra (paths relative to the upstream repository):
//   the port model            C++/src/vision.cpp:496-599   (air -> glass -> water, Snell twice; here run FORWARD)
//   the rows                  matlab/MeasureUpdate.m:67,72-73 ; C++/src/filter.cpp:684-685,691-692 with a marker corner in place of
//                             the marker origin:  X_k = R_IL R'(P_m + R_m c_k - p - R P_IL),  H = (d pi/d X)[ -R_IL R' | R_IL [R'(c_w - p)]x ]
//   the update                matlab/MeasureUpdate.m:84-102 ; filter.cpp:709-739   K = P H'(H P H' + R)^-1, dx = K r, P = (I - K H) P
//
// What round 3 had (ekf_kernels.hpp::correct_pixels_kernel; deleted in round 4, commit fdeea42) and why it was replaced:
//   * fp32 throughout.  128-256 rows at sigma_pix = 1e-3 shrink the pose variances by 4-5 decades in ONE update; P - k (P h')(P h')'
//     then cancels to 1e-5 of its terms, the 6 x 6 information matrix accumulated in fp32 is perturbed by eps * cond(Lam), and a
//     predicted image point known to 6e-8 moves a weakly observed block (gravity, sigma 10) by 5e-5 of itself:
//     block-wise covariance error 2e-4 .. 3e-3, literal state error 2e-5 .. 4e-5 (tools/emul_pixels_precision.py decomposes it).
//   * ~550 VALU instructions per projection on one dependent chain: Newton from the paraxial start with a wave vote per step,
//     IEEE divisions and a square root (52 / 85 cycles each, tools/exp_issue_rates.hip) in the epilogue, 87 instructions per row to
//     fold it.  0.53 VALU issue.
// Round 4:
//   * fp64 costs what fp32 costs on this part except for rsq / rcp (v_fma_f64 4.7 cycles against 4.6, v_rsq_f64 16.5 against 8.5:
//     profiles/r04_issue_rates.txt), so everything that decides the posterior runs in double: the corner geometry, ONE final
//     evaluation of the port equation with its Newton correction (the iteration itself stays in fp32 and only has to come within
//     1e-3: the correction squares that twice), the residual, the Jacobian, the fold, the 6 x 6 algebra.  fp32 is left where it
//     is harmless: the Newton iterations and the N x N covariance arithmetic.
//   * the rows regrouped per corner (exact algebra, as PoseFold did for the pose rows): with a = (J_q Mc)' in the IMU frame,
//     h_p = -R a and h_theta = a x ru (ru = R'(c_w - p)), so  Lam_pp = R S_aa R', Lam_pt = -R S_ac, Lam_tt = S_cc and per corner only
//     N' = sum a a', n' = sum a r are accumulated row by row (9 FMA per row); S_ac += N'[ru]x, S_cc += [ru]x' N' [ru]x per corner.
//   * J_q = alpha_q e' + beta_q n' + k g_q': no 3 x 3 dD/dX, no divisions.
//   * the Newton iteration starts from the MEASURED image point (the innovation is a few 1e-3: one or two steps) and runs the
//     four corners x one or two cameras of a marker in lock step: 4-8 independent chains per vote instead of one.
//   * the update in the non-cancelling form: with Lam = Lc Lc', Mt = I + Lc' P_JJ Lc = Cm Cm', Z = Lc Cm^-T,
//         Sinv = (Lam^-1 + P_JJ)^-1 = Z Z',   G = (I + P_JJ Lam)^-1 = I - P_JJ Sinv  (the difference taken in double),
//     the p / theta rows of the posterior are the PRODUCT  G P(J, :)  -- no subtraction of nearly equal numbers -- and only the
//     block of the states the rows do not touch takes  P_rr - W W',  W = P_rJ Z, where the cancellation is the physical one.
//     dx = P(:, J) G' b.
#pragma once
#include "ekf_kernels.hpp"
#include "ekf_launch.hpp"          // MeasConst
#include "ekf_team.hpp"            // CovMap: storage index -> (row, column)


namespace {

// workgroup barrier that does not drain the global loads in flight (ekf_team.hpp::team_barrier)
__device__ __forceinline__ void meas_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- double-precision reciprocal square root / reciprocal from the hardware estimates (v_rsq_f64 / v_rcp_f64: ~2^-23 relative)
// + NS Newton steps: one gives 2^-45 (what an fp32 posterior needs many times over), two the full double
template <int NS = 2>
__device__ __forceinline__ double md_rsq(double x)
{
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double h = 0.5 * x * y;
        y = __builtin_fma(y, __builtin_fma(-h, y, 0.5), y);
    }
    return y;
}
template <int NS = 2>
__device__ __forceinline__ double md_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
#pragma unroll
    for (int i = 0; i < NS; ++i) y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
    return y;
}

// Batched forms: the same operation on NQ independent values, written OPERATION BY OPERATION across the batch.  One wave per SIMD
// has nothing but its own independent instructions to fill the latency of a dependent chain (v_fma_f64 issues every 4.7 cycles
// but a dependent one every 5.6, v_rsq_f64 16.5, profiles/r04_issue_rates.txt), and the machine scheduler keeps the source order
// of long unrolled bodies: written projection by projection the four corners of a marker ran one after the other
// (round 4, first version: 6.0 cycles per instruction).
template <int NS, int NQ>
__device__ __forceinline__ void md_rsq_n(const double (&x)[NQ], double (&y)[NQ])
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) y[q] = __builtin_amdgcn_rsq(x[q]);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double h[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) h[q] = 0.5 * x[q] * y[q];
#pragma unroll
        for (int q = 0; q < NQ; ++q) h[q] = __builtin_fma(-h[q], y[q], 0.5);
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] = __builtin_fma(y[q], h[q], y[q]);
    }
}
template <int NS, int NQ>
__device__ __forceinline__ void md_rcp_n(const double (&x)[NQ], double (&y)[NQ])
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) y[q] = __builtin_amdgcn_rcp(x[q]);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double h[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) h[q] = __builtin_fma(-x[q], y[q], 1.0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] = __builtin_fma(y[q], h[q], y[q]);
    }
}

// the port equation  rho = L(t) = d_air t + d_glass tan(theta_glass) + zw tan(theta_water),  t = tan(theta_air)   (vision_device.hpp)
// IN THE TANGENT (round 6): sin(theta_m) = a_m sin(theta_air) gives tan(theta_m) = a_m t / sqrt(1 + (1 - a_m^2) t^2), so with
// x_g = 1 + (1 - a0^2) t^2, x_w = 1 + (1 - a1^2) t^2, i_g = x_g^-1/2, i_w = x_w^-1/2, G = d_glass a0, W = zw a1:
//     L   = t (d_air + G i_g + W i_w)
//     L'  = d_air + G i_g^3 + W i_w^3                          (d/dt [t x^-1/2] = x^-3/2)
//     L'' = -3 t ((1 - a0^2) G i_g^5 + (1 - a1^2) W i_w^5)     (< 0: L is concave)
//     L_z = a1 t i_w,   d L_z / dt = a1 i_w^3
// -- two reciprocal square roots per evaluation and no sine: rounds 4-5 went through s = sin(theta_air) = t (1 + t^2)^-1/2 and
// 1 / cos(theta_m) = (1 - a_m^2 s^2)^-1/2, three of them and 18 more instructions per projection (EXPERIMENTS -1.10; the same function,
// tests/test_port_solver_cpu.py holds both forms against each other).
// dt: HALLEY's step  -2 f L' / (2 L'^2 - f L''),  f = L - rho  (cubic convergence).  The denominator is kept >= L'^2 (far below the
// root f L'' > 0 could eat it: the step then is at most twice Newton's).
// Lzt, Ltt: d L_z / dt, d L_t / dt (to carry L_z, L_t along the step to first order).
template <typename S, int NQ> struct PortEvalN { S Lt[NQ], Lz[NQ], dt[NQ], Ltt[NQ], Lzt[NQ]; };
template <int NS, typename S, int NQ>
__device__ __forceinline__ void port_eval_n(S a0, S a1, S d_air, S G, const S (&W)[NQ], const S (&rho)[NQ], const S (&t)[NQ],
                                            PortEvalN<S, NQ>& o)
{
    const S qg = S(1) - a0 * a0, qw = S(1) - a1 * a1;
    S t2[NQ], x[NQ], ig[NQ], iw[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) t2[q] = t[q] * t[q];
#pragma unroll
    for (int q = 0; q < NQ; ++q) x[q] = S(1) + qg * t2[q];
    md_rsq_n<NS, NQ>(x, ig);
#pragma unroll
    for (int q = 0; q < NQ; ++q) x[q] = S(1) + qw * t2[q];
    md_rsq_n<NS, NQ>(x, iw);
    S ig2[NQ], iw2[NQ], g1[NQ], w1[NQ], g3[NQ], w3[NQ], f[NQ], Lt2[NQ], den[NQ], h[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { ig2[q] = ig[q] * ig[q]; iw2[q] = iw[q] * iw[q]; g1[q] = G * ig[q]; w1[q] = W[q] * iw[q]; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) { g3[q] = g1[q] * ig2[q]; w3[q] = w1[q] * iw2[q]; f[q] = (g1[q] + w1[q]) + d_air; o.Lz[q] = a1 * t[q] * iw[q]; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) { o.Lt[q] = (g3[q] + w3[q]) + d_air; h[q] = qg * g3[q] * ig2[q] + qw * w3[q] * iw2[q]; f[q] = t[q] * f[q] - rho[q]; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) { o.Ltt[q] = S(-3) * t[q] * h[q]; o.Lzt[q] = a1 * iw[q] * iw2[q]; Lt2[q] = o.Lt[q] * o.Lt[q]; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) den[q] = fmax(S(2) * Lt2[q] - f[q] * o.Ltt[q], Lt2[q]);
    md_rcp_n<NS, NQ>(den, h);
#pragma unroll
    for (int q = 0; q < NQ; ++q) o.dt[q] = -S(2) * f[q] * o.Lt[q] * h[q];
}

// ---- the START of the port equation's solution, in packed fp32 (round 6) ---------------------------------------------------------
// The thin-port solution in closed form, twice (see pixel_fold_marker): a starting value for the Halley step in double that follows, good
// to 1e-4 by construction -- so fp32 arithmetic (1e-7) costs nothing, and the hardware's fp32 estimates issue in half the time of the
// double ones (v_rsq_f32 8.5 ticks against v_rsq_f64 16.5, profiles/r04_issue_rates.txt) while two projections share one v_pk_* slot:
// ~104 double instructions (24 of them v_rsq / v_rcp_f64) per four projections become ~80 fp32 ones (EXPERIMENTS -1.10).
//   rho, zs: lateral offset and water depth of the NP projections (out of view: 0 and 1);  t: tan(theta_air), the start
using f32x2 = float __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_bc(float a) { return f32x2{ a, a }; }
__device__ __forceinline__ f32x2 pk_rsq(f32x2 a) { return f32x2{ __builtin_amdgcn_rsqf(a.x), __builtin_amdgcn_rsqf(a.y) }; }
__device__ __forceinline__ f32x2 pk_rcp(f32x2 a) { return f32x2{ __builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y) }; }
__device__ __forceinline__ f32x2 pk_max(f32x2 a, float lo) { return f32x2{ __builtin_fmaxf(a.x, lo), __builtin_fmaxf(a.y, lo) }; }
template <int NP>
__device__ __forceinline__ void port_start_f32(const MeasConst& mc, const double (&rho)[NP], const double (&zs)[NP], double (&t)[NP])
{
    static_assert(NP % 2 == 0, "projections come in pairs");
    constexpr int H = NP / 2;
    const float a1sq = mc.st[1], q1 = mc.st[2], qg = mc.st[3], dair = mc.st[4], Gd0 = mc.st[5], c0 = mc.st[6];
    f32x2 rh[H], z[H], u[H], w[H], tt[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { rh[h] = f32x2{ (float)rho[2 * h], (float)rho[2 * h + 1] }; z[h] = f32x2{ (float)zs[2 * h], (float)zs[2 * h + 1] }; }
    // t0 = u / sqrt(a1^2 - (1 - a1^2) u^2),  u = rho / z_e,  z_e = z_w + (d_air + d_glass a0) / a1
#pragma unroll
    for (int h = 0; h < H; ++h) w[h] = pk_rcp(z[h] + pk_bc(c0));
#pragma unroll
    for (int h = 0; h < H; ++h) { u[h] = rh[h] * w[h]; w[h] = pk_max(pk_fma2(pk_bc(-q1) * u[h], u[h], pk_bc(a1sq)), 1e-6f); }
#pragma unroll
    for (int h = 0; h < H; ++h) w[h] = pk_rsq(w[h]);
#pragma unroll
    for (int h = 0; h < H; ++h) { tt[h] = u[h] * w[h]; w[h] = pk_fma2(pk_bc(qg) * tt[h], tt[h], pk_bc(1.f)); }
#pragma unroll
    for (int h = 0; h < H; ++h) { w[h] = pk_rsq(w[h]); z[h] = pk_rcp(z[h]); }     // tan(theta_glass) / (a0 t0) (see port_eval_n), 1 / z_w
    // the port's offsets at t0 taken off rho, the thin-port equation once more for the water alone
#pragma unroll
    for (int h = 0; h < H; ++h) {
        u[h] = pk_fma2(-tt[h], pk_fma2(pk_bc(Gd0), w[h], pk_bc(dair)), rh[h]);
        u[h] = pk_max(u[h] * z[h], 0.f);
        w[h] = pk_max(pk_fma2(pk_bc(-q1) * u[h], u[h], pk_bc(a1sq)), 1e-6f);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) w[h] = pk_rsq(w[h]);
#pragma unroll
    for (int h = 0; h < H; ++h) { tt[h] = u[h] * w[h]; t[2 * h] = (double)tt[h].x; t[2 * h + 1] = (double)tt[h].y; }
}

// ---- the sums of the fold (double) --------------------------------------------------------------------------------------
struct PixAcc {
    double Saa[6], Sac[9], Scc[6], sa[3], sc[3];          // symmetric ones in the order 00 01 02 11 12 22
    static constexpr int NVAL = 27;
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < 6; ++i) { Saa[i] = 0.0; Scc[i] = 0.0; }
#pragma unroll
        for (int i = 0; i < 9; ++i) Sac[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) { sa[i] = 0.0; sc[i] = 0.0; }
    }
    __device__ __forceinline__ double& at(int k)
    {
        return k < 6 ? Saa[k] : (k < 15 ? Sac[k - 6] : (k < 21 ? Scc[k - 15] : (k < 24 ? sa[k - 21] : sc[k - 24])));
    }
    // one corner: N' = sum a a' (sym, 00 01 02 11 12 22), n' = sum a r over its rows, r = ru = R'(c_w - p)
    __device__ __forceinline__ void add_corner(const double* Np, const double* np, const double* r)
    {
        // full symmetric N'
        const double N00 = Np[0], N01 = Np[1], N02 = Np[2], N11 = Np[3], N12 = Np[4], N22 = Np[5];
        const double Nf[9] = { N00, N01, N02, N01, N11, N12, N02, N12, N22 };
        // Q = N' [r]x :  column 0 = N'(:,1) r2 - N'(:,2) r1, column 1 = N'(:,2) r0 - N'(:,0) r2, column 2 = N'(:,0) r1 - N'(:,1) r0
        double Q[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Q[3 * i + 0] = Nf[3 * i + 1] * r[2] - Nf[3 * i + 2] * r[1];
            Q[3 * i + 1] = Nf[3 * i + 2] * r[0] - Nf[3 * i + 0] * r[2];
            Q[3 * i + 2] = Nf[3 * i + 0] * r[1] - Nf[3 * i + 1] * r[0];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) Saa[i] += Np[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) Sac[i] += Q[i];
        // [r]x' Q :  row 0 = r2 Q(1,:) - r1 Q(2,:), row 1 = r0 Q(2,:) - r2 Q(0,:), row 2 = r1 Q(0,:) - r0 Q(1,:)
        Scc[0] += r[2] * Q[3] - r[1] * Q[6];
        Scc[1] += r[2] * Q[4] - r[1] * Q[7];
        Scc[2] += r[2] * Q[5] - r[1] * Q[8];
        Scc[3] += r[0] * Q[7] - r[2] * Q[1];
        Scc[4] += r[0] * Q[8] - r[2] * Q[2];
        Scc[5] += r[1] * Q[2] - r[0] * Q[5];
#pragma unroll
        for (int i = 0; i < 3; ++i) sa[i] += np[i];
        sc[0] += np[1] * r[2] - np[2] * r[1];           // n' x r
        sc[1] += np[2] * r[0] - np[0] * r[2];
        sc[2] += np[0] * r[1] - np[1] * r[0];
    }
    // Corners whose rows all share ONE matrix N' (the corner-position update: N' = R_IL' R_IL for every corner): S_aa, S_ac and S_cc are
    // then linear in the count, in sum r and in sum r r' -- 18 operations per corner here instead of add_corner's 70, and the sums
    // themselves once per filter (expand_const).  Until then the slots hold: Saa[0] = count, Sac[0..2] = sum r, Scc = sum r r'
    // (00 01 02 11 12 22); sa, sc as always.  (Sums of sums: the roles' partial sums add up the same way.)
    __device__ __forceinline__ void add_corner_const(const double* np, const double* r)
    {
        Saa[0] += 1.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) Sac[i] += r[i];
        Scc[0] += r[0] * r[0]; Scc[1] += r[0] * r[1]; Scc[2] += r[0] * r[2];
        Scc[3] += r[1] * r[1]; Scc[4] += r[1] * r[2]; Scc[5] += r[2] * r[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) sa[i] += np[i];
        sc[0] += np[1] * r[2] - np[2] * r[1];
        sc[1] += np[2] * r[0] - np[0] * r[2];
        sc[2] += np[0] * r[1] - np[1] * r[0];
    }
    // count, sum r, sum r r'  ->  S_aa = count N',  S_ac = N' [sum r]x,  S_cc = sum [r]x' N' [r]x  (linear in sum r r')
    __device__ __forceinline__ void expand_const(const double* Np)
    {
        const double cnt = Saa[0], r0 = Sac[0], r1 = Sac[1], r2 = Sac[2];
        const double w00 = Scc[0], w01 = Scc[1], w02 = Scc[2], w11 = Scc[3], w12 = Scc[4], w22 = Scc[5];
        const double n00 = Np[0], n01 = Np[1], n02 = Np[2], n11 = Np[3], n12 = Np[4], n22 = Np[5];
#pragma unroll
        for (int i = 0; i < 6; ++i) Saa[i] = cnt * Np[i];
        const double Nf[9] = { n00, n01, n02, n01, n11, n12, n02, n12, n22 };
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Sac[3 * i + 0] = Nf[3 * i + 1] * r2 - Nf[3 * i + 2] * r1;
            Sac[3 * i + 1] = Nf[3 * i + 2] * r0 - Nf[3 * i + 0] * r2;
            Sac[3 * i + 2] = Nf[3 * i + 0] * r1 - Nf[3 * i + 1] * r0;
        }
        Scc[0] = n11 * w22 - 2.0 * n12 * w12 + n22 * w11;
        Scc[1] = -n01 * w22 + n12 * w02 + n02 * w12 - n22 * w01;
        Scc[2] = n01 * w12 - n11 * w02 - n02 * w11 + n12 * w01;
        Scc[3] = n00 * w22 - 2.0 * n02 * w02 + n22 * w00;
        Scc[4] = -n00 * w12 + n01 * w02 + n02 * w01 - n12 * w00;
        Scc[5] = n00 * w11 - 2.0 * n01 * w01 + n11 * w00;
    }
    // one row a (its theta part is c = a x r, r = ru = R'(c_w - p)) with residual res, accumulated directly: 33 operations per row --
    // cheaper than add_corner's detour over N', n' while a corner has two rows (left camera alone: 66 against 18 + 70), dearer from
    // four rows on (stereo: 132 against 36 + 70)
    __device__ __forceinline__ void add_row(const double* a, double res, const double* r)
    {
        const double c[3] = { a[1] * r[2] - a[2] * r[1], a[2] * r[0] - a[0] * r[2], a[0] * r[1] - a[1] * r[0] };
        Saa[0] += a[0] * a[0]; Saa[1] += a[0] * a[1]; Saa[2] += a[0] * a[2];
        Saa[3] += a[1] * a[1]; Saa[4] += a[1] * a[2]; Saa[5] += a[2] * a[2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Sac[3 * i + j] += a[i] * c[j];
        Scc[0] += c[0] * c[0]; Scc[1] += c[0] * c[1]; Scc[2] += c[0] * c[2];
        Scc[3] += c[1] * c[1]; Scc[4] += c[1] * c[2]; Scc[5] += c[2] * c[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) { sa[i] += a[i] * res; sc[i] += c[i] * res; }
    }
    // (round 6) the two rows of a projection in the camera frame (pixel_fold_marker<..., CF>): the radial one (three components) and the
    // tangential one (third component zero: 24 operations instead of 33).  Sums of two products are spelled out as one product and one fma --
    // which product the compiler fuses otherwise depends on the kernel around it, and the window of frames must equal the sequence of frames
    // bit for bit (see direct_update_part).
    static __device__ __forceinline__ double dm2(double a, double b, double c, double d) { return __builtin_fma(a, b, -(c * d)); }   // a b - c d
    static __device__ __forceinline__ double dp3(double a0, double b0, double a1, double b1, double a2, double b2)
    {
        return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0));
    }
    __device__ __forceinline__ void add_row_cf(const double* a, double res, const double* r)
    {
        const double c[3] = { dm2(a[1], r[2], a[2], r[1]), dm2(a[2], r[0], a[0], r[2]), dm2(a[0], r[1], a[1], r[0]) };
        Saa[0] += a[0] * a[0]; Saa[1] += a[0] * a[1]; Saa[2] += a[0] * a[2];
        Saa[3] += a[1] * a[1]; Saa[4] += a[1] * a[2]; Saa[5] += a[2] * a[2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Sac[3 * i + j] += a[i] * c[j];
        Scc[0] += c[0] * c[0]; Scc[1] += c[0] * c[1]; Scc[2] += c[0] * c[2];
        Scc[3] += c[1] * c[1]; Scc[4] += c[1] * c[2]; Scc[5] += c[2] * c[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) { sa[i] += a[i] * res; sc[i] += c[i] * res; }
    }
    __device__ __forceinline__ void add_row_xy(double a0, double a1, double res, const double* r)
    {
        const double c[3] = { a1 * r[2], -(a0 * r[2]), dm2(a0, r[1], a1, r[0]) };
        Saa[0] += a0 * a0; Saa[1] += a0 * a1; Saa[3] += a1 * a1;
#pragma unroll
        for (int j = 0; j < 3; ++j) { Sac[j] += a0 * c[j]; Sac[3 + j] += a1 * c[j]; }
        Scc[0] += c[0] * c[0]; Scc[1] += c[0] * c[1]; Scc[2] += c[0] * c[2];
        Scc[3] += c[1] * c[1]; Scc[4] += c[1] * c[2]; Scc[5] += c[2] * c[2];
        sa[0] += a0 * res; sa[1] += a1 * res;
#pragma unroll
        for (int i = 0; i < 3; ++i) sc[i] += c[i] * res;
    }
    // (round 6) sums folded in a CAMERA frame (rows j, their theta parts c' = j x Y with Y = M^-T r) -> what finish() expects.  The IMU-frame
    // rows are a = M' j, c = a x r = adj(M) c'  ((A u) x (A v) = cof(A) (u x v)), so  S_aa = M' S_jj M,  S_ac = M' S_jc' adj(M)',
    // S_cc = adj(M) S_c'c' adj(M)',  s_a = M' s_j,  s_c = adj(M) s_c'.  The M' ... M parts are not formed: finish() rotates S_aa, S_ac, s_a by
    // R anyway, and R (M' . M) R' = (R M') . (R M')' -- the caller hands finish() RM = R M' (camera_rotation) instead of R.  Here only the
    // adj(M) parts: S_ac <- S_jc' adj(M)', S_cc <- adj(M) S_c'c' adj(M)', s_c <- adj(M) s_c'  (~110 operations once per filter).
    __device__ __forceinline__ void to_imu_frame(const double* A)
    {
        {
            double T1[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) T1[3 * i + k] = dp3(Sac[3 * i], A[3 * k], Sac[3 * i + 1], A[3 * k + 1], Sac[3 * i + 2], A[3 * k + 2]);
#pragma unroll
            for (int i = 0; i < 9; ++i) Sac[i] = T1[i];
        }
        {
            const double Sf[9] = { Scc[0], Scc[1], Scc[2], Scc[1], Scc[3], Scc[4], Scc[2], Scc[4], Scc[5] };
            double T1[9];                                                        // S_c'c' adj(M)'
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) T1[3 * i + k] = dp3(Sf[3 * i], A[3 * k], Sf[3 * i + 1], A[3 * k + 1], Sf[3 * i + 2], A[3 * k + 2]);
            int o = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = i; k < 3; ++k) Scc[o++] = dp3(A[3 * i], T1[k], A[3 * i + 1], T1[3 + k], A[3 * i + 2], T1[6 + k]);
        }
        const double sp[3] = { sc[0], sc[1], sc[2] };
#pragma unroll
        for (int i = 0; i < 3; ++i) sc[i] = dp3(A[3 * i], sp[0], A[3 * i + 1], sp[1], A[3 * i + 2], sp[2]);
    }
    // RM = R M' (see to_imu_frame)
    static __device__ __forceinline__ void camera_rotation(const double* R, const double* M, double* RM)
    {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) RM[3 * i + k] = dp3(R[3 * i], M[3 * k], R[3 * i + 1], M[3 * k + 1], R[3 * i + 2], M[3 * k + 2]);
    }
    // -> the 6 x 6 information matrix (upper triangle, lidx order) and vector of the stacked rows
    //    Lam_pp = w R S_aa R',  Lam_pt = -w R S_ac,  Lam_tt = w S_cc,  b_p = -w R s_a,  b_t = w s_c
    __device__ __forceinline__ void finish(const double* R, double w, double* Lam, double* b) const
    {
        const double Sf[9] = { Saa[0], Saa[1], Saa[2], Saa[1], Saa[3], Saa[4], Saa[2], Saa[4], Saa[5] };
        double T1[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) T1[3 * i + k] = R[3 * i] * Sf[k] + R[3 * i + 1] * Sf[3 + k] + R[3 * i + 2] * Sf[6 + k];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = i; j < 3; ++j)
                Lam[lidx(i, j)] = w * (T1[3 * i] * R[3 * j] + T1[3 * i + 1] * R[3 * j + 1] + T1[3 * i + 2] * R[3 * j + 2]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                Lam[lidx(i, 3 + j)] = -w * (R[3 * i] * Sac[j] + R[3 * i + 1] * Sac[3 + j] + R[3 * i + 2] * Sac[6 + j]);
            b[i] = -w * (R[3 * i] * sa[0] + R[3 * i + 1] * sa[1] + R[3 * i + 2] * sa[2]);
            b[3 + i] = w * sc[i];
        }
        Lam[lidx(3, 3)] = w * Scc[0]; Lam[lidx(3, 4)] = w * Scc[1]; Lam[lidx(3, 5)] = w * Scc[2];
        Lam[lidx(4, 4)] = w * Scc[3]; Lam[lidx(4, 5)] = w * Scc[4]; Lam[lidx(5, 5)] = w * Scc[5];
    }
};

// The reference's h(x) is R_IL R'(c_w - p - R P_IL) (MeasureUpdate.m:67 ; filter.cpp:684) with the CARRIED rotation matrix, which
// is a rotation only to its fp32 rounding: R'(R P_IL) is not P_IL but 6e-8 |P_IL| off, i.e. 1e-8 in an image point -- visible in
// the fp64 kernels (2e-7 against the oracle before this was taken over literally).  Per filter: pil = R'(R P_IL); then
// R'(c_w - p - R P_IL) = R'(c_w - p) - pil.
__device__ __forceinline__ void filter_pil(const double* R, const double* P_IL, double* pil)
{
    double RP[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) RP[i] = R[3 * i] * P_IL[0] + R[3 * i + 1] * P_IL[1] + R[3 * i + 2] * P_IL[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) pil[i] = R[i] * RP[0] + R[3 + i] * RP[1] + R[6 + i] * RP[2];
}

// ---- one marker: 4 corners x NCAM cameras in lock step --------------------------------------------------------------------
// p, R: the filter's position and carried rotation (double copies of the record's values); pil = R'(R P_IL) (see filter_pil);
// mkc: the map slot (corner 0, x axis, y axis); yl / yr: the 8 + 8 measured image coordinates.  Every stage is written across the NP = 4 NCAM projections (see md_rsq_n).
// One evaluation of the port equation per projection in the common case (see below).
// NZ: the port is square to the camera, normal = (0, 0, 1) exactly -- the reference's configuration (paramconfig.yml:39-42,
// refractinfo.yml:10-13: normal_vector 0 0 1).  Then z = X_z, the lateral offset is (X_x, X_y, 0), D_z = 1 (no division), and in the
// row  a = alpha e'M + beta n'M + k (M_r - uv_r M_z)  the two k uv_r M_z terms cancel: a = c1 e_r (e'M) + k M_r - c2 e_r M_z --
// ~60 of ~360 instructions per corner less.  Any other normal takes the general form (tests/test_pixels_gpu.py: a tilted port).
// NK, K0 (round 6): the corners K0 .. K0 + NK - 1 of the marker only (NK = 4, K0 = 0: all of them).  The stereo fold of the one-wave
// kernels takes a marker as two passes of two corners x two cameras (pixel_fold_marker_stereo_halves): four projections in lock step, the
// left-camera fold's working set, instead of eight -- the eight-projection stage spilled ~300 v_accvgpr moves per marker (EXPERIMENTS -1.7:
// 2388 -> 2161 instructions per marker in the compiled loop, 16 slots stereo 100-102 -> 93-96 us).
// wgt (round 6): 1 for a marker of the map, 0 for a slot whose id is not in it -- the callers fold EVERY slot (with slot 0's frame for the
// unknown ones, rows of weight 0) instead of branching around the fold per lane: the divergent branch cost ~70 instructions per marker
// (exec-mask bookkeeping and zero-initialised merge values for the 27 sums) and saved work only when all 64 filters of a wave skip.
// CF (round 6; left camera, square port): the rows are folded IN THE CAMERA FRAME, rotated into the radial and the tangential direction of the
// image point (the pixel noise is isotropic: sum a a' and sum a res do not change).  There they are sparse --
//     j_rad = (e_0 / L_t, e_1 / L_t, -c2),   j_tan = (-k e_1, k e_0, 0)          (c1 + k = 1 / L_t; on the axis e = (1, 0))
// -- no product with M per row (a = M' j: 32 operations per projection) and a short tangential row; their theta parts are c' = j x Y with
// Y = M^-T R'(c_w - p) (three more 3 x 3 products per marker), and PixAcc::to_imu_frame turns the sums into the IMU-frame ones once per filter:
// 906 -> 825 arithmetic instructions per marker in the compiled loop (EXPERIMENTS -1.12).  The same rows for BOTH cameras (the right camera's
// taken to the left camera's frame, 15 operations per projection) was built and is slower: the stereo fold goes corner by corner over N' =
// sum j j' (PixAcc::add_corner), where the sparse rows save little, and its two halves would each form the Y_k.
template <int NCAM, typename T, bool NZ, int NK = 4, int K0 = 0, bool CF = false>
__device__ __forceinline__ void pixel_fold_marker(PixAcc& acc, const double* p, const double* R, const double* pil,
                                                  const MeasConst& mc, const double* mkc, const T* yl, const T* yr, double size, double wgt = 1.0)
{
    static_assert(NK >= 1 && K0 >= 0 && K0 + NK <= 4, "corners of one marker");
    static_assert(!CF || (NZ && NCAM == 1), "the camera-frame fold: left camera, square port");
    constexpr int NS = sizeof(T) == 8 ? 2 : 1;           // Newton steps behind v_rsq_f64 / v_rcp_f64
    constexpr int NP = NK * NCAM;
    double ru[4][3], rAx[3], rAy[3];
    {
        double u0[3], ru0[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) u0[i] = mkc[i] - p[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ru0[i] = R[i] * u0[0] + R[3 + i] * u0[1] + R[6 + i] * u0[2];                                   // R'(C0 - p)
            rAx[i] = size * (R[i] * mkc[3] + R[3 + i] * mkc[4] + R[6 + i] * mkc[5]);
            rAy[i] = size * (R[i] * mkc[6] + R[3 + i] * mkc[7] + R[6 + i] * mkc[8]);
        }
        // corners c_k = (0,0,0), (0,s,0), (s,s,0), (s,0,0) of the marker frame (vision.cpp:736-759)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ru[0][i] = ru0[i];
            ru[1][i] = ru0[i] + rAy[i];
            ru[2][i] = ru0[i] + rAx[i] + rAy[i];
            ru[3][i] = ru0[i] + rAx[i];
        }
    }
    const double* n = mc.n;
    // per projection q = (corner k, camera c): lateral offset, depth, visibility, and the start of the port equation's solution
    double lat[NP][3], rho[NP], irho[NP], Wd[NP], vis[NP], t[NP], Yc[CF ? NK : 1][3];
    bool offax[NP];                                       // the point is off the camera's axis (on it: rho = 0, lat = 0, 1 / rho stands for 1)
    {
        double X[NP][3], z[NP], r2[NP], zwq[NP], r2s[NP], xs[NP], ir0[NP], zsq[NP];
        bool ok[NP];
        // X = M_c (ru_k - pil) + t_c: corner 0 and the two edge vectors once per camera, the other corners by addition
#pragma unroll
        for (int c = 0; c < NCAM; ++c) {
            const double* M = c ? mc.McR : mc.McL;
            const double tI[3] = { ru[0][0] - pil[0], ru[0][1] - pil[1], ru[0][2] - pil[2] };
            double X0[3], MAx[3], MAy[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                X0[i] = M[3 * i] * tI[0] + M[3 * i + 1] * tI[1] + M[3 * i + 2] * tI[2] + (c ? mc.tR[i] : 0.0);
                MAx[i] = M[3 * i] * rAx[0] + M[3 * i + 1] * rAx[1] + M[3 * i + 2] * rAx[2];
                MAy[i] = M[3 * i] * rAy[0] + M[3 * i + 1] * rAy[1] + M[3 * i + 2] * rAy[2];
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int kk = K0 + k;                   // corner 0: X0, 1: X0 + MAy, 2: (X0 + MAy) + MAx, 3: X0 + MAx
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    X[k * NCAM + c][i] = kk == 0 ? X0[i] : (kk == 1 ? X0[i] + MAy[i] : (kk == 2 ? X0[i] + MAy[i] + MAx[i] : X0[i] + MAx[i]));
            }
        }
        if constexpr (CF) {
            // Y_k = McL^-T ru_k, as X: corner 0 and the two edges, the others by addition
            double Y0[3], YAx[3], YAy[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Y0[i] = PixAcc::dp3(mc.MiTL[3 * i], ru[0][0], mc.MiTL[3 * i + 1], ru[0][1], mc.MiTL[3 * i + 2], ru[0][2]);
                YAx[i] = PixAcc::dp3(mc.MiTL[3 * i], rAx[0], mc.MiTL[3 * i + 1], rAx[1], mc.MiTL[3 * i + 2], rAx[2]);
                YAy[i] = PixAcc::dp3(mc.MiTL[3 * i], rAy[0], mc.MiTL[3 * i + 1], rAy[1], mc.MiTL[3 * i + 2], rAy[2]);
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int kk = K0 + k;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    Yc[k][i] = kk == 0 ? Y0[i] : (kk == 1 ? Y0[i] + YAy[i] : (kk == 2 ? Y0[i] + YAy[i] + YAx[i] : Y0[i] + YAx[i]));
            }
        }
        if constexpr (NZ) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                z[q] = X[q][2];
                lat[q][0] = X[q][0]; lat[q][1] = X[q][1]; lat[q][2] = 0.0;
                r2[q] = X[q][0] * X[q][0] + X[q][1] * X[q][1];
                zwq[q] = z[q] - (mc.d_air + mc.d_glass);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NP; ++q) z[q] = X[q][0] * n[0] + X[q][1] * n[1] + X[q][2] * n[2];
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int i = 0; i < 3; ++i) lat[q][i] = X[q][i] - z[q] * n[i];
#pragma unroll
            for (int q = 0; q < NP; ++q) { r2[q] = lat[q][0] * lat[q][0] + lat[q][1] * lat[q][1] + lat[q][2] * lat[q][2]; zwq[q] = z[q] - mc.d_air - mc.d_glass; }
        }
        // in front of the port and inside its field of view (in water no ray leans further than asin(n_air / n_water); 0.9 of that
        // limit, as the oracle): otherwise the corner contributes no rows to this camera.  A point out of view is replaced by a
        // harmless one (rho = 0, one metre of water; 1 / rho stands for 1 wherever rho = 0): everything below stays finite, its rows get weight 0
        const double klim = mc.klim;                                             // (0.9 a1)^2 / (1 - a1^2), from the host
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            ok[q] = (zwq[q] > 0.0) & (r2[q] < klim * zwq[q] * zwq[q]);      // (&, not &&: the short circuit compiled to two exec-mask regions per projection)
            vis[q] = ok[q] ? wgt : 0.0;
            r2s[q] = ok[q] ? r2[q] : 0.0;
            zsq[q] = ok[q] ? zwq[q] : 1.0;
            Wd[q] = zsq[q] * mc.a1;
            xs[q] = r2s[q] > 0.0 ? r2s[q] : 1.0;
        }
        md_rsq_n<NS, NP>(xs, ir0);
#pragma unroll
        for (int q = 0; q < NP; ++q) { offax[q] = r2s[q] > 0.0; irho[q] = ir0[q]; rho[q] = r2s[q] * irho[q]; }
        // Start: the THIN-port solution in closed form, twice.  With the port's own offsets folded into an effective water depth
        // z_e = z_w + (d_air + d_glass a0) / a1 (exact in the paraxial limit) the equation is rho = z_e tan(theta_water) with
        // sin(theta_water) = a1 sin(theta_air):  t0 = u / sqrt(a1^2 - (1 - a1^2) u^2),  u = rho / z_e  -- within 1.4e-3 of the root for
        // 0.25 .. 2 m of water and tangents up to 1.4, 2.7e-2 up to 2.5, 1e-1 at the rim of the admitted field of view (tangent 3.16).
        // Then the port's offsets AT t0 are taken off rho and the thin-port equation is solved once more for the water alone:
        //     u1 = (rho - d_air t0 - d_glass tan(theta_glass(t0))) / z_w,   t1 = u1 / sqrt(a1^2 - (1 - a1^2) u1^2)
        // -- within 1.2e-4 inside tangent 1.4, 6e-4 up to 2, 3.4e-3 up to 2.5, 1.8e-2 at the rim (tests/test_port_solver_cpu.py
        // restates this solver in numpy and asserts the figures).  Measured in one run of round 4 (65 536 filters x 16 slots, left / stereo): second pass always 79.4 / 120.0 us,
        // only for the waves that hold a tangent > 1.3 (voted, applied per lane) 78.3 / 124.0 -- the branch costs more than the
        // pass --, never 76.0 / 113.2 (and 1.6e-4 off at the rim).
        // (round 6) in packed fp32: port_start_f32 -- a start good to 1e-4 does not need double
        port_start_f32<NP>(mc, rho, zsq, t);
    }
    // ONE Halley step in double from there (cubic): 1.4e-13 left inside tangent 1.4 (every lens; the recordings and test scenes stay
    // below 0.8), 7e-12 up to 2, 9e-10 up to 2.5, 4.4e-8 = 1.4e-8 relative at the very rim -- below the fp32 rounding of the image
    // point it is compared with; fp64 records take a second step, which reaches double precision (5e-15) everywhere.  No iteration,
    // no wave vote, and nothing depends on the measured image point.
    // (Round-4 history: Newton in fp32 from the paraxial start, 3-4 voted steps; from the measured ray 2-3; Halley from the
    // measured ray 1-2, then one in double; this form: one evaluation of the port equation.)
    double iLt[NP], c2[NP];
    {
        constexpr int NFIN = sizeof(T) == 8 ? 2 : 1;
        PortEvalN<double, NP> f;
        const double Gd = mc.d_glass * mc.a0;
#pragma unroll
        for (int rep = 0; rep < NFIN; ++rep) {
            port_eval_n<NS, double, NP>(mc.a0, mc.a1, mc.d_air, Gd, Wd, rho, t, f);
#pragma unroll
            for (int q = 0; q < NP; ++q) t[q] = fmax(t[q] + f.dt[q], 0.0);
        }
        // L_t and L_z for the Jacobian, carried along the last step to first order (they were evaluated in front of it)
#pragma unroll
        for (int q = 0; q < NP; ++q) { f.Lt[q] += f.Ltt[q] * f.dt[q]; f.Lz[q] += f.Lzt[q] * f.dt[q]; }
        md_rcp_n<NS, NP>(f.Lt, iLt);
#pragma unroll
        for (int q = 0; q < NP; ++q) c2[q] = f.Lz[q] * iLt[q];
    }
    double kk[NP], uv[NP][2], a[NP][2][3], res[NP][2];
#pragma unroll
    for (int q = 0; q < NP; ++q) kk[q] = offax[q] ? t[q] * irho[q] : iLt[q];                 // t / rho; on the axis its limit 1 / L_t
    if constexpr (CF) {
        double e[NP][2];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            uv[q][0] = kk[q] * lat[q][0];
            uv[q][1] = kk[q] * lat[q][1];
            e[q][0] = offax[q] ? lat[q][0] * irho[q] : 1.0;                 // on the axis any direction will do
            e[q][1] = lat[q][1] * irho[q];
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int k = K0 + q;
            const double r0 = (double)yl[2 * k] - uv[q][0], r1 = (double)yl[2 * k + 1] - uv[q][1];
            const double jr = iLt[q] * vis[q], kv = kk[q] * vis[q];
            const double jrad[3] = { jr * e[q][0], jr * e[q][1], -(c2[q] * vis[q]) };
            acc.add_row_cf(jrad, __builtin_fma(e[q][1], r1, e[q][0] * r0), Yc[q]);                                  // res_rad = e . res
            acc.add_row_xy(-(kv * e[q][1]), kv * e[q][0], PixAcc::dm2(e[q][0], r1, e[q][1], r0), Yc[q]);            // res_tan = e_perp . res
        }
        return;
    }
    if constexpr (NZ) {
        double e[NP][2], eM[NP][3];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            uv[q][0] = kk[q] * lat[q][0];
            uv[q][1] = kk[q] * lat[q][1];
            e[q][0] = lat[q][0] * irho[q];
            e[q][1] = lat[q][1] * irho[q];
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double* M = (q % NCAM) ? mc.McR : mc.McL;
#pragma unroll
            for (int j = 0; j < 3; ++j) eM[q][j] = e[q][0] * M[j] + e[q][1] * M[3 + j];
        }
        // a_r = vis (c1 e_r (e'M) + k M_r - c2 e_r M_z)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int k = K0 + q / NCAM, c = q % NCAM;
            const double* M = c ? mc.McR : mc.McL;
            const T* y = c ? yr : yl;
            const double c1v = (iLt[q] - kk[q]) * vis[q], c2v = c2[q] * vis[q], kv = kk[q] * vis[q];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const double w1 = c1v * e[q][r], w3 = -c2v * e[q][r];
#pragma unroll
                for (int j = 0; j < 3; ++j) a[q][r][j] = w1 * eM[q][j] + kv * M[3 * r + j] + w3 * M[6 + j];
                res[q][r] = (double)y[2 * k + r] - uv[q][r];
            }
        }
    } else {
        double Dz[NP], iDz[NP], e[NP][3], eM[NP][3];
#pragma unroll
        for (int q = 0; q < NP; ++q) Dz[q] = n[2] + kk[q] * lat[q][2];
        md_rcp_n<NS, NP>(Dz, iDz);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            uv[q][0] = (n[0] + kk[q] * lat[q][0]) * iDz[q];
            uv[q][1] = (n[1] + kk[q] * lat[q][1]) * iDz[q];
#pragma unroll
            for (int i = 0; i < 3; ++i) e[q][i] = lat[q][i] * irho[q];
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double* M = (q % NCAM) ? mc.McR : mc.McL;
#pragma unroll
            for (int j = 0; j < 3; ++j) eM[q][j] = e[q][0] * M[j] + e[q][1] * M[3 + j] + e[q][2] * M[6 + j];
        }
        // rows: J_r = alpha e' + beta n' + k g',  g = (unit_r - uv_r unit_z) / D_z;  a = (J_r Mc)' masked by visibility
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int k = K0 + q / NCAM, c = q % NCAM;
            const double* M = c ? mc.McR : mc.McL;
            const double* nM = c ? mc.nMR : mc.nML;
            const T* y = c ? yr : yl;
            const double c1 = iLt[q] - kk[q];
            const double kz = kk[q] * iDz[q] * vis[q];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const double ge = (e[q][r] - uv[q][r] * e[q][2]) * iDz[q], gn = (n[r] - uv[q][r] * n[2]) * iDz[q];
                const double am = ge * c1 * vis[q], bm = -(c2[q] * ge + kk[q] * gn) * vis[q];
#pragma unroll
                for (int j = 0; j < 3; ++j) a[q][r][j] = am * eM[q][j] + bm * nM[j] + kz * (M[3 * r + j] - uv[q][r] * M[6 + j]);
                res[q][r] = (double)y[2 * k + r] - uv[q][r];
            }
        }
    }
    if constexpr (NCAM == 1) {
        // two rows per corner: straight into the sums (PixAcc::add_row)
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int r = 0; r < 2; ++r) acc.add_row(a[k][r], res[k][r], ru[K0 + k]);
    } else {
        double Np[NK][6], np[NK][3];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int i = 0; i < 6; ++i) Np[k][i] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) np[k][i] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < NCAM; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const double* ar = a[k * NCAM + c][r];
                    Np[k][0] += ar[0] * ar[0]; Np[k][1] += ar[0] * ar[1]; Np[k][2] += ar[0] * ar[2];
                    Np[k][3] += ar[1] * ar[1]; Np[k][4] += ar[1] * ar[2]; Np[k][5] += ar[2] * ar[2];
#pragma unroll
                    for (int j = 0; j < 3; ++j) np[k][j] += ar[j] * res[k * NCAM + c][r];
                }
#pragma unroll
        for (int k = 0; k < NK; ++k) acc.add_corner(Np[k], np[k], ru[K0 + k]);
    }
}

// one marker, both cameras, as two passes of two corners each (see NK, K0 above): the same projections, rows and sums, corner by corner
// in the same order -- bit-equal to pixel_fold_marker<2, T, NZ>
template <typename T, bool NZ>
__device__ __forceinline__ void pixel_fold_marker_stereo_halves(PixAcc& acc, const double* p, const double* R, const double* pil,
                                                                const MeasConst& mc, const double* mkc, const T* yl, const T* yr, double size,
                                                                double wgt = 1.0)
{
    pixel_fold_marker<2, T, NZ, 2, 0>(acc, p, R, pil, mc, mkc, yl, yr, size, wgt);
    order_fence();
    pixel_fold_marker<2, T, NZ, 2, 2>(acc, p, R, pil, mc, mkc, yl, yr, size, wgt);
}


// ---- correct() from stereo CORNERS: triangulation through the port in double, 3 position-type rows per corner ----------------
// vision.cpp:496-599 for the four corners of one marker (the arithmetic of vision_device.hpp::refraction_corner, in double and
// written across the 8 rays / 4 corners -- see md_rsq_n): left / right normalised image points -> points in the left camera frame.
// NZ: the port square to the camera (normal exactly (0, 0, 1), see pixel_fold_marker): v = r_z, and a refraction changes r_z only.
template <typename T, bool NZ>
__device__ __forceinline__ void tri_corners_refractive(const VisConst<double>& vc, const T* yl, const T* yr, double (&C)[4][3])
{
    constexpr int NS = sizeof(T) == 8 ? 2 : 1;
    constexpr int NQ = 8;                                 // ray q = 2 k + c: corner k, camera c
    double rL[4][3], rR[4][3], PL[4][3], PR[4][3];        // per corner: the two rays in the water and their exit points on the outer glass face, left frame
    // mid-point of the two rays by Cramer (vision.cpp:559-595).  Inlined behind each of the two ray constructions: the square-port one hands it
    // left rays with z component exactly 1 (a literal the compiler sees: four operations per corner less in the cross products)
    auto midpoints = [&]() __attribute__((always_inline)) {
        double d3[4], id3[4], t1[4], t2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double dP[3], cr[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) dP[i] = PR[k][i] - PL[k][i];
            cross3(rL[k], rR[k], cr);
            d3[k] = cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2];       // det(cr, rL, rR) = cr . (rL x rR) = |cr|^2
            t1[k] = det3cols(cr, dP, rR[k]);
            t2[k] = -det3cols(cr, rL[k], dP);
        }
        md_rcp_n<NS, 4>(d3, id3);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t1[k] *= id3[k]; t2[k] *= id3[k];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double Pm = 0.5 * (PL[k][i] + t1[k] * rL[k][i] + PR[k][i] + t2[k] * rR[k][i]);
                C[k][i] = (i < 2) ? -Pm : Pm;                 // vision.cpp:597-599
            }
        }
    };
    if (NZ && vc.sqrt_minus0 && vc.sqrt_minus1) {
        // The port square to the camera and both refractions towards the normal's side (the reference's configuration; wave-uniform):
        // everything follows from the image point (x, y) -- t^2 = x^2 + y^2 is the squared tangent in air -- and TWO reciprocal square roots
        // (round 6, in the tangent as port_eval_n: tan(theta_m) = a_m t / sqrt(1 + (1 - a_m^2) t^2), a_glass = alpha0, a_water = a = alpha0 alpha1):
        //   x_g = 1 + (1 - alpha0^2) t^2,  x_w = 1 + (1 - a^2) t^2,
        //   ray ~ (a x, a y, sqrt(x_w))          -- the unit vector of vision.cpp:524-543 times 1 / cos(theta_air): the mid-point of two rays
        //                                           does not depend on their lengths (the left one is taken at z = 1: times 1 / sqrt(x_w))
        //   exit = ((d_air + d_glass alpha0 / sqrt(x_g)) x, (same) y, d_air + d_glass)
        // and the right camera's pair in the left frame (vision.cpp:555-556) from ONE product u = a R_RL(:, 0:1) (x, y):
        //   ray_R = u + R_RL(:, 2) sqrt(x_w),   exit_R = ((d_air + d_glass alpha0 / sqrt(x_g)) / a) u + (R_RL(:, 2) (d_air + d_glass) + P_LR)
        // Rounds 4-5 went through cos(theta_air) = 1 / |(x, y, 1)| and the two cosines root0, root1 of vision.cpp:505-543 (three reciprocal
        // square roots per ray) and rotated ray and exit point separately: 821 -> 653 instructions per marker in isolation, tools/stage_meas_isa.hip (EXPERIMENTS -1.11).
        double xx[NQ], yy[NQ], t2[NQ], xg[NQ], xw[NQ], ig[NQ], iw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const T* p = (q & 1) ? yr : yl;
            xx[q] = (double)p[2 * (q >> 1)]; yy[q] = (double)p[2 * (q >> 1) + 1];
            t2[q] = xx[q] * xx[q] + yy[q] * yy[q];
        }
        const double a01 = vc.tri[11], qg = 1.0 - vc.alpha0 * vc.alpha0, qw = 1.0 - a01 * a01;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { xg[q] = 1.0 + qg * t2[q]; xw[q] = 1.0 + qw * t2[q]; }
        md_rsq_n<NS, NQ>(xg, ig);
        md_rsq_n<NS, NQ>(xw, iw);
        const double zP = vc.d_air + vc.d_glass, ga = vc.d_glass * vc.alpha0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int qL = 2 * k, qR = 2 * k + 1;
            const double cpL = vc.d_air + ga * ig[qL], cpR = vc.tri[9] + vc.tri[10] * ig[qR], swR = xw[qR] * iw[qR];
            const double aL = a01 * iw[qL];                                  // the left ray scaled to z = 1: 1 / sqrt(x_w) = i_w
            rL[k][0] = aL * xx[qL]; rL[k][1] = aL * yy[qL]; rL[k][2] = 1.0;
            PL[k][0] = cpL * xx[qL]; PL[k][1] = cpL * yy[qL]; PL[k][2] = zP;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double u = vc.tri[2 * i] * xx[qR] + vc.tri[2 * i + 1] * yy[qR];
                rR[k][i] = u + vc.R_RL[3 * i + 2] * swR;
                PR[k][i] = cpR * u + vc.tri[6 + i];
            }
        }
        midpoints();
    } else {
        const double* n = vc.nrm;
        double r2[NQ][3], P1[NQ][3];                      // the ray in the water and its exit point, each in its own camera's frame
        double r0[NQ][3], r1[NQ][3], v0[NQ], v1[NQ], x[NQ], y[NQ], iv0[NQ], iv1[NQ];
    #pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const T* p = (q & 1) ? yr : yl;
            r0[q][0] = (double)p[2 * (q >> 1)]; r0[q][1] = (double)p[2 * (q >> 1) + 1]; r0[q][2] = 1.0;
            x[q] = r0[q][0] * r0[q][0] + r0[q][1] * r0[q][1] + 1.0;
        }
        md_rsq_n<NS, NQ>(x, y);
        // NZ: v0 = r0_z = 1 / |(x, y, 1)|, so 1 / v0 = |.|^2 / |.| costs one product; and where the refracted ray keeps the normal's side
        // (sqrt_minus: the lower index first, as in air -> glass) v1 = alpha0 v0 + beta IS the root below, whose reciprocal is the
        // reciprocal square root already taken -- no division for the two path lengths d_air / v0, d_glass / v1
        const bool free_iv = NZ && vc.sqrt_minus0;           // wave-uniform
        if constexpr (NZ) {
    #pragma unroll
            for (int q = 0; q < NQ; ++q) iv0[q] = x[q] * y[q];
        }
    #pragma unroll
        for (int q = 0; q < NQ; ++q) {
    #pragma unroll
            for (int i = 0; i < 3; ++i) r0[q][i] *= y[q];
            v0[q] = NZ ? r0[q][2] : r0[q][0] * n[0] + r0[q][1] * n[1] + r0[q][2] * n[2];
            x[q] = 1.0 - vc.alpha0 * vc.alpha0 * (1.0 - v0[q] * v0[q]);
        }
        md_rsq_n<NS, NQ>(x, y);                              // air -> glass   (vision.cpp:505-522)
    #pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double root = x[q] * y[q];
            const double beta = vc.sqrt_minus0 ? (root - vc.alpha0 * v0[q]) : (vc.alpha0 * v0[q] - root);
    #pragma unroll
            for (int i = 0; i < 3; ++i) r1[q][i] = (NZ && i < 2) ? vc.alpha0 * r0[q][i] : vc.alpha0 * r0[q][i] + beta * (NZ ? 1.0 : n[i]);
            v1[q] = NZ ? r1[q][2] : r1[q][0] * n[0] + r1[q][1] * n[1] + r1[q][2] * n[2];
            iv1[q] = y[q];                                   // (= 1 / v1 if free_iv)
            x[q] = 1.0 - vc.alpha1 * vc.alpha1 * (1.0 - v1[q] * v1[q]);
        }
        md_rsq_n<NS, NQ>(x, y);                              // glass -> water (vision.cpp:524-543)
        if constexpr (!NZ) md_rcp_n<NS, NQ>(v0, iv0);
        if (!free_iv) md_rcp_n<NS, NQ>(v1, iv1);
    #pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double root = x[q] * y[q];
            const double beta = vc.sqrt_minus1 ? (root - vc.alpha1 * v1[q]) : (vc.alpha1 * v1[q] - root);
            const double aq = vc.d_air * iv0[q], gq = vc.d_glass * iv1[q];
    #pragma unroll
            for (int i = 0; i < 3; ++i) {
                r2[q][i] = (NZ && i < 2) ? vc.alpha1 * r1[q][i] : vc.alpha1 * r1[q][i] + beta * (NZ ? 1.0 : n[i]);
                P1[q][i] = aq * r0[q][i] + gq * r1[q][i];      // exit point on the outer glass face (vision.cpp:546-552)
            }
        }
        // the right ray in the left frame (vision.cpp:555-556)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int qL = 2 * k, qR = 2 * k + 1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                rL[k][i] = r2[qL][i]; PL[k][i] = P1[qL][i];
                rR[k][i] = vc.R_RL[3 * i] * r2[qR][0] + vc.R_RL[3 * i + 1] * r2[qR][1] + vc.R_RL[3 * i + 2] * r2[qR][2];
                PR[k][i] = vc.R_RL[3 * i] * P1[qR][0] + vc.R_RL[3 * i + 1] * P1[qR][1] + vc.R_RL[3 * i + 2] * P1[qR][2] + vc.P_LR[i];
            }
        }
        midpoints();
    }
}

// the four corner positions C (left camera frame, as triangulated) of one marker as 12 position-type rows:
//   h_k = R_IL (ru_k - P_IL),  rows a_i = (R_IL)_i' for every corner:  N' = R_IL' R_IL (constant, mc.NI),  n' = R_IL' (C_k - h_k)
// (MeasureUpdate.m:67,72-73 with the corner in place of the marker origin; oracle: fbo_correct_corners)
__device__ __forceinline__ void corner_fold_marker(PixAcc& acc, const double* p, const double* R, const double* pil,
                                                   const MeasConst& mc, const double* mkc, const double (&C)[4][3], double size)
{
    double ru[4][3];
    {
        double u0[3], ru0[3], rAx[3], rAy[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) u0[i] = mkc[i] - p[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ru0[i] = R[i] * u0[0] + R[3 + i] * u0[1] + R[6 + i] * u0[2];
            rAx[i] = size * (R[i] * mkc[3] + R[3 + i] * mkc[4] + R[6 + i] * mkc[5]);
            rAy[i] = size * (R[i] * mkc[6] + R[3 + i] * mkc[7] + R[6 + i] * mkc[8]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ru[0][i] = ru0[i];
            ru[1][i] = ru0[i] + rAy[i];
            ru[2][i] = ru0[i] + rAx[i] + rAy[i];
            ru[3][i] = ru0[i] + rAx[i];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double tI[3] = { ru[k][0] - pil[0], ru[k][1] - pil[1], ru[k][2] - pil[2] };
        double fr[3], np[3];                              // F (C - h): McL = F R_IL, so R_IL' res = McL' (F res)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double XL = mc.McL[3 * i] * tI[0] + mc.McL[3 * i + 1] * tI[1] + mc.McL[3 * i + 2] * tI[2];
            fr[i] = (i < 2 ? -C[k][i] : C[k][i]) - XL;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) np[j] = mc.McL[j] * fr[0] + mc.McL[3 + j] * fr[1] + mc.McL[6 + j] * fr[2];
        acc.add_corner_const(np, ru[k]);                  // (the caller expands the sums with mc.NI before the 6 x 6 stage)
    }
}

// ---- the 6 x 6 stage in double ----------------------------------------------------------------------------------------------
// in: Lam (21, lidx order), b (6), PJJ (36, full symmetric);  out: G = (I + P_JJ Lam)^-1 (36), Sinv = (Lam^-1 + P_JJ)^-1 (21,
// lidx order), m = G' b (6)
__host__ __device__ constexpr int ltx(int i, int j) { return i * (i + 1) / 2 + j; }      // lower triangle, j <= i
template <typename T>
__device__ __forceinline__ void info_solve(const double* Lam, const double* b, const double* PJJ, T* G, T* Sinv, T* m)
{
    // Lam = Lc Lc' (no pivoting: Lam is positive semi-definite; a pivot that is not clearly positive relative to its original
    // diagonal carries no information and its column is dropped -- joint_factor's rule)
    double Lc[21];
    {
        double A[21];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) A[ltx(i, j)] = Lam[lidx(j, i)];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double piv = A[ltx(a, a)];
            const bool ok = piv > 4e-15 * Lam[lidx(a, a)];
            const double s0 = md_rsq(ok ? piv : 1.0);
            const double s = ok ? s0 : 0.0;
#pragma unroll
            for (int i = a; i < 6; ++i) Lc[ltx(i, a)] = A[ltx(i, a)] * s;
#pragma unroll
            for (int i = a + 1; i < 6; ++i)
#pragma unroll
                for (int j = a + 1; j <= i; ++j) A[ltx(i, j)] -= Lc[ltx(i, a)] * Lc[ltx(j, a)];
        }
    }
    // Y = P_JJ Lc ;  Mt = I + Lc' Y  (symmetric, eigenvalues >= 1) = Cm Cm'
    double Y[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < 6; ++k) s += PJJ[6 * i + k] * Lc[ltx(k, j)];
            Y[6 * i + j] = s;
        }
    double Cm[21], iC[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = i; k < 6; ++k) s += Lc[ltx(k, i)] * Y[6 * k + j];
            Cm[ltx(i, j)] = s;
        }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double s = md_rsq(Cm[ltx(a, a)]);
        iC[a] = s;
#pragma unroll
        for (int i = a; i < 6; ++i) Cm[ltx(i, a)] *= s;
#pragma unroll
        for (int i = a + 1; i < 6; ++i)
#pragma unroll
            for (int j = a + 1; j <= i; ++j) Cm[ltx(i, j)] -= Cm[ltx(i, a)] * Cm[ltx(j, a)];
    }
    // Z Cm' = Lc, row by row: z_j = (lc_ij - sum_{k<j} z_k Cm(j,k)) / Cm(j,j)
    double Zd[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double s = (j <= i) ? Lc[ltx(i, j)] : 0.0;
#pragma unroll
            for (int k = 0; k < j; ++k) s -= Zd[6 * i + k] * Cm[ltx(j, k)];
            Zd[6 * i + j] = s * iC[j];
        }
    // Sinv = Z Z' ;  G = I - P_JJ Sinv ;  m = G' b
    double Si[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += Zd[6 * i + k] * Zd[6 * j + k];
            Si[6 * i + j] = s; Si[6 * j + i] = s;
        }
    double Gd[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s -= PJJ[6 * i + k] * Si[6 * k + j];
            Gd[6 * i + j] = s;
        }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) s += Gd[6 * i + j] * b[i];
        m[j] = (T)s;
    }
#pragma unroll
    for (int i = 0; i < 36; ++i) G[i] = (T)Gd[i];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) Sinv[lidx(i, j)] = (T)Si[6 * i + j];
}

// ---- the update of the N x N covariance and the error state, type T, whole covariance resident --------------------------
// J = the p and theta columns (state indices jcol(0..5)); r = the other N - 6; x_c = P(J, c).
//   dx       = P(:, J) m
//   P(a, c) -= x_a' Sinv x_c                  a <= c in r                     (the physical cancellation only)
//   P(J, c)  = G x_c                          all c: a product, nothing is subtracted
__host__ __device__ constexpr int rcol(int k) { return k < 3 ? 3 + k : 6 + k; }           // k-th state index outside J
// the 63 coefficients of the update (G 36, Sinv 21, m 6), in registers beside the resident covariance.  (fp64 records: 342 registers
// of covariance leave too little, the kernels spill 650-780 bytes per lane; keeping the coefficients in LDS and loading the
// covariance after the 6 x 6 stage were both tried in round 4 and changed nothing, see EXPERIMENTS.md)
template <typename T>
struct RegCoef {
    T g[36], s[21], m_[6];
    __device__ __forceinline__ T G(int i, int j) const { return g[6 * i + j]; }
    __device__ __forceinline__ T S(int i, int j) const { return s[i <= j ? lidx(i, j) : lidx(j, i)]; }
    __device__ __forceinline__ T m(int k) const { return m_[k]; }
    __device__ __forceinline__ void set(const T* G_, const T* S_, const T* M_)
    {
#pragma unroll
        for (int i = 0; i < 36; ++i) g[i] = G_[i];
#pragma unroll
        for (int i = 0; i < 21; ++i) s[i] = S_[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) m_[i] = M_[i];
    }
};
template <typename T, int N, int LO, int HI, bool WANT_DX, typename COEF, int PHASES = 3, bool PKON = true>
__device__ __forceinline__ void direct_update_part(T* P, T* dx, const COEF& cf);
// (round 6) the whole covariance = the part form over the full storage range: ONE body (below) for every kernel that applies the update.
// PKON = false: the same operations without the packed column pairs (bit-equal: every half of a packed operation is the scalar one) --
// the frame-window kernels, which sit at 512 registers, spill 28-212 bytes with the pairs and nothing without
template <typename T, int N, typename COEF, bool PKON = true>
__device__ __forceinline__ void direct_update(T* P, T* dx, const COEF& cf)
{
    direct_update_part<T, N, 0, Lay<N>::NP, true, COEF, 3, PKON>(P, dx, cf);
}

// ---- the update in two PARTS of the packed covariance (round 5) ----------------------------------------------------------------
// EARLY part = storage [0, late_start): rows 0..8 and the collected diagonals -- every x_c = P(J, c) lives there; LATE part = the
// chunks behind it, which hold only elements P(a, c) with a, c >= 9 (type "outside J": P(a, c) -= x_a' S^-1 x_c).  Used by the
// divided-update kernel (ekf_meas_split.hpp: the two parts on two waves) and by the fp64 tail below (one wave, one part after the other:
// 171 doubles + 63 coefficients do not fit 512 registers at once).
// first storage index of the LATE part: a multiple of the chunk size behind which every element has row >= 9 (and so column >= 9:
// no element of the J rows / columns, nothing the 6 x 6 stage reads)
template <typename T, int N>
constexpr int late_start()
{
    constexpr int EPC = Rec<T, N>::EPC, NP = Lay<N>::NP;
    int e0 = NP;
    for (int e = NP - 1; e >= 0 && cov_row<N>(e) >= 9; --e) e0 = e;
    return (e0 + EPC - 1) / EPC * EPC;
}
constexpr bool in_J(int s) { return s < 3 || (s >= 6 && s < 9); }
// what the SOLVER reads of the early part, by covariance chunk cc (0 = the first chunk behind the nominal state):
//   SEL_JJ   a chunk that holds an element of P(J, J): the 6 x 6 stage's input, requested in front of the exchange
//   SEL_XL   a chunk that holds an element P(J, c >= 9) = part of an x_c of the late columns (and no P(J, J) element: those it has
//            already); requested behind the 6 x 6 stage, whose doubles leave no room for them
enum { SEL_JJ = 1, SEL_XL = 2 };
template <typename T, int N>
constexpr int chunk_sel(int cc)
{
    constexpr int EPC = Rec<T, N>::EPC, NP = Lay<N>::NP;
    bool jj = false, xl = false;
    for (int k = 0; k < EPC; ++k) {
        const int e = cc * EPC + k;
        if (e >= NP) continue;
        const int i = cov_row<N>(e), j = cov_col<N>(e);
        jj = jj || (in_J(i) && in_J(j));
        xl = xl || (in_J(i) && j >= 9);
    }
    return jj ? SEL_JJ : (xl ? SEL_XL : 0);
}
// chunks [C0, C1) of the covariance whose selector is SEL -> P (storage order; the others stay untouched)
template <typename T, int N, int C0, int C1, int SEL, int AUX = AUX_DEFAULT>
__device__ __forceinline__ void load_cov_chunks(__amdgpu_buffer_rsrc_t rs, unsigned lane, T* P)
{
    using RC = Rec<T, N>;
    static_for<C0, C1>([&](auto cc_) {
        constexpr int cc = decltype(cc_)::value;
        if constexpr (chunk_sel<T, N>(cc) == SEL)
            load_chunks<T, N, RC::CH_NOM + cc, RC::CH_NOM + cc + 1, AUX>(rs, lane, P + cc * RC::EPC);
    });
}

#ifndef FBUS_X_PACK_UPDATE
#define FBUS_X_PACK_UPDATE 1
#endif
// coef * x + acc on a register pair as ONE fused operation per half, spelled out: which products of a sum the compiler contracts into
// FMAs differs from kernel to kernel for the packed form (the fused frame's update and the per-call update stopped agreeing bit for bit
// when this was left to it), and the explicit fma is what the scalar form's contraction gives
__device__ __forceinline__ f32x2 pk_fma(float coef, f32x2 x, f32x2 acc) { return __builtin_elementwise_fma(f32x2{ coef, coef }, x, acc); }
// ... and the scalar chains of the update the same way: s = a0 b0, then s = fma(a_k, b_k, s) -- in `a0 b0 + a1 b1` either product may be the
// one the compiler fuses, and it chose differently in different kernels once the update had one body for all of them
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
// columns rcol(c), rcol(c + 1) outside J are neighbours that start on an even state index: an aligned storage pair in every row above them
template <int N>
constexpr bool upd_pair_head(int c) { return c + 1 < N - 6 && rcol(c) % 2 == 0 && rcol(c + 1) == rcol(c) + 1; }
// which elements of the three groups of direct_update lie in the storage range [LO, HI)
template <int N, int LO, int HI> constexpr bool in_part(int i, int j) { return pidx<N>(i, j) >= LO && pidx<N>(i, j) < HI; }
template <int N, int LO, int HI> constexpr bool part_any_rr(int c) { bool a_ = false; for (int a = 0; a <= c; ++a) a_ = a_ || in_part<N, LO, HI>(rcol(a), rcol(c)); return a_; }
template <int N, int LO, int HI> constexpr bool part_any_jj() { bool a_ = false; for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) a_ = a_ || in_part<N, LO, HI>(jcol(i), jcol(j)); return a_; }
template <int N, int LO, int HI> constexpr bool part_all_jj() { bool a_ = true; for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) a_ = a_ && in_part<N, LO, HI>(jcol(i), jcol(j)); return a_; }
template <int N, int LO, int HI> constexpr bool part_any_jr(int c) { bool a_ = false; for (int i = 0; i < 6; ++i) a_ = a_ || in_part<N, LO, HI>(jcol(i), rcol(c)); return a_; }
template <int N, int LO, int HI> constexpr bool part_all_jr(int c) { bool a_ = true; for (int i = 0; i < 6; ++i) a_ = a_ && in_part<N, LO, HI>(jcol(i), rcol(c)); return a_; }

// direct_update (ekf_meas.hpp) restricted to the covariance elements with storage index in [LO, HI): the same operations on every
// element it touches, in the same order.  P is the full-size array; every element of the J rows / columns that the touched elements
// need must be loaded (x_c = P(J, c)); WANT_DX: dx = P(:, J) m as well (needs all of the J rows).  PHASES: 1 = the block outside J
// (reads S^-1), 2 = the J rows (reads G), 3 = both -- the fp64 tail runs them one after the other with only that phase's coefficients live.
template <typename T, int N, int LO, int HI, bool WANT_DX, typename COEF, int PHASES, bool PKON>
__device__ __forceinline__ void direct_update_part(T* P, T* dx, const COEF& cf)
{
#define PS(i, j) P[pidx<N>((i), (j))]
#define INR(i, j) (in_part<N, LO, HI>((i), (j)))
    constexpr int NR_ = N - 6;
    if constexpr (WANT_DX) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T s = PS(i, jcol(0)) * cf.m(0);
#pragma unroll
            for (int k = 1; k < 6; ++k) s = fma_t(PS(i, jcol(k)), cf.m(k), s);
            dx[i] = s;
        }
    }
    // (round 6) fp32, pair-aligned storage: two neighbouring columns (c, c + 1) of the block outside J / of the J rows at a time on packed
    // instructions (v_pk_mul / v_pk_fma_f32: two columns per issue slot) -- every element sees the operations of the scalar form in the
    // same order (each half of a packed operation IS the scalar operation), so the results are the scalar form's bit for bit
    constexpr bool PK = PKON && PackedMath<T, N>::on && FBUS_X_PACK_UPDATE;
    // the block outside J, column by column: t = Sinv x_c, then P(a, c) -= x_a . t for the columns a <= c (the x are still the old ones)
    static_for<0, NR_>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        constexpr bool head = PK && upd_pair_head<N>(c), tail = PK && c >= 1 && upd_pair_head<N>(c - 1);
        if constexpr (tail) {
            // (done with column c - 1)
        } else if constexpr (head && (PHASES & 1) && (part_any_rr<N, LO, HI>(c) || part_any_rr<N, LO, HI>(c + 1))) {
            if constexpr (sizeof(T) == 4) {
                f32x2 xv[6], tv[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) xv[j] = f32x2{ PS(jcol(j), rcol(c)), PS(jcol(j), rcol(c + 1)) };
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    f32x2 sv = cf.S(0, i) * xv[0];
#pragma unroll
                    for (int j = 1; j < 6; ++j) sv = pk_fma(cf.S(j, i), xv[j], sv);
                    tv[i] = sv;
                }
                static_for<0, c + 2>([&](auto a_) {
                    constexpr int a = decltype(a_)::value;
                    constexpr bool in0 = a <= c && INR(rcol(a), rcol(c)), in1 = INR(rcol(a), rcol(c + 1));
                    if constexpr (a <= c && (in0 || in1)) {
                        f32x2 sv = PS(jcol(0), rcol(a)) * tv[0];
#pragma unroll
                        for (int k = 1; k < 6; ++k) sv = pk_fma(PS(jcol(k), rcol(a)), tv[k], sv);
                        if constexpr (in0 && in1 && is_pair<N>(rcol(a), rcol(c))) {
                            st_pair<N>(P, rcol(a), rcol(c), ld_pair<N>(P, rcol(a), rcol(c)) - sv);
                        } else {
                            if constexpr (in0) PS(rcol(a), rcol(c)) -= sv.x;
                            if constexpr (in1) PS(rcol(a), rcol(c + 1)) -= sv.y;
                        }
                    } else if constexpr (a == c + 1 && in1) {
                        T sc = PS(jcol(0), rcol(a)) * tv[0].y;
#pragma unroll
                        for (int k = 1; k < 6; ++k) sc = __builtin_fmaf(PS(jcol(k), rcol(a)), tv[k].y, sc);
                        PS(rcol(a), rcol(c + 1)) -= sc;
                    }
                });
            }
        } else if constexpr (part_any_rr<N, LO, HI>(c) && (PHASES & 1)) {
            T t[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                T s = cf.S(0, i) * PS(jcol(0), rcol(c));
#pragma unroll
                for (int j = 1; j < 6; ++j) s = fma_t(cf.S(j, i), PS(jcol(j), rcol(c)), s);
                t[i] = s;
            }
            static_for<0, c + 1>([&](auto a_) {
                constexpr int a = decltype(a_)::value;
                if constexpr (INR(rcol(a), rcol(c))) {
                    T s = PS(jcol(0), rcol(a)) * t[0];
#pragma unroll
                    for (int k = 1; k < 6; ++k) s = fma_t(PS(jcol(k), rcol(a)), t[k], s);
                    PS(rcol(a), rcol(c)) -= s;
                }
            });
        }
    });
    // the J x J block from the old values (upper triangle of G P_JJ), then the J x r columns in place
    if constexpr (part_any_jj<N, LO, HI>() && (PHASES & 2)) {
        // (the J x J block lies in ONE part: the early one)
        T nj[21];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) {
                T s = cf.G(i, 0) * PS(jcol(0), jcol(j));
#pragma unroll
                for (int k = 1; k < 6; ++k) s = fma_t(cf.G(i, k), PS(jcol(k), jcol(j)), s);
                nj[lidx(i, j)] = s;
            }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) PS(jcol(i), jcol(j)) = nj[lidx(i, j)];
        static_assert(part_all_jj<N, LO, HI>(), "the J x J block must lie in one part");
    }
    static_for<0, NR_>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        constexpr bool head = PK && upd_pair_head<N>(c), tail = PK && c >= 1 && upd_pair_head<N>(c - 1);
        if constexpr (tail) {
            // (done with column c - 1)
        } else if constexpr (head && (PHASES & 2) && (part_any_jr<N, LO, HI>(c) || part_any_jr<N, LO, HI>(c + 1))) {
            static_assert(part_all_jr<N, LO, HI>(c) && part_all_jr<N, LO, HI>(c + 1), "a pair of columns of the J rows must lie in one part");
            if constexpr (sizeof(T) == 4) {
                f32x2 xv[6], yv[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) xv[j] = f32x2{ PS(jcol(j), rcol(c)), PS(jcol(j), rcol(c + 1)) };
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    f32x2 sv = cf.G(i, 0) * xv[0];
#pragma unroll
                    for (int j = 1; j < 6; ++j) sv = pk_fma(cf.G(i, j), xv[j], sv);
                    yv[i] = sv;
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) { PS(jcol(i), rcol(c)) = yv[i].x; PS(jcol(i), rcol(c + 1)) = yv[i].y; }
            }
        } else if constexpr (part_any_jr<N, LO, HI>(c) && (PHASES & 2)) {
            static_assert(part_all_jr<N, LO, HI>(c), "a column of the J rows must lie in one part");
            T x[6], y[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) x[j] = PS(jcol(j), rcol(c));
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                T s = cf.G(i, 0) * x[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) s = fma_t(cf.G(i, j), x[j], s);
                y[i] = s;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) PS(jcol(i), rcol(c)) = y[i];
        }
    });
#undef INR
#undef PS
}

// coefficient views of the phased fp64 tail: S^-1 and m in registers while the block outside J is updated, G back from LDS for the J rows
template <typename T>
struct CoefSm {
    T s[21], m_[6];
    __device__ __forceinline__ T G(int, int) const { return T(0); }
    __device__ __forceinline__ T S(int i, int j) const { return s[i <= j ? lidx(i, j) : lidx(j, i)]; }
    __device__ __forceinline__ T m(int k) const { return m_[k]; }
};
template <typename T>
struct CoefG {
    T g[36];
    __device__ __forceinline__ T G(int i, int j) const { return g[6 * i + j]; }
    __device__ __forceinline__ T S(int, int) const { return T(0); }
    __device__ __forceinline__ T m(int) const { return T(0); }
};

// the sums -> information matrix -> 6 x 6 stage (double) -> one-shot update of the resident covariance; dx = the error state
// (the resident kernels; meas_update_tail below holds the same statements in place)
template <typename T, int N, bool PKON = true>
__device__ __forceinline__ void meas_solve_update(T* P, const PixAcc& acc, const double* Rd, double w, T* dx)
{
    double Lam[21], bv[6];
    acc.finish(Rd, w, Lam, bv);
    T G[36], Sinv[21], m[6];
    {
        double PJJ[36];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) PJJ[6 * i + j] = (double)P[pidx<N>(jcol(i), jcol(j))];
        info_solve<T>(Lam, bv, PJJ, G, Sinv, m);
    }
    RegCoef<T> cf;
    cf.set(G, Sinv, m);
    direct_update<T, N, RegCoef<T>, PKON>(P, dx, cf);
}

// the tail both per-call measurement kernels share: the sums -> information matrix -> 6 x 6 stage -> update -> injection -> stores
// (Round 6 tried two things around it, both measured slower and kept as patches only: the covariance through LDS while the fold runs --
// gfx950's direct-to-LDS 16-byte loads, requested at the top of the kernel or behind the last marker's fetch, tools/patches/
// r06_meas_lds_prefetch.diff -- and the left camera's markers two at a time, tools/patches/r06_meas_pair.diff; EXPERIMENTS -1.5.)
template <typename T, int N>
__device__ __forceinline__ void meas_update_tail(const __amdgpu_buffer_rsrc_t rs, unsigned lane, const PixAcc& acc, const double* Rd, double w,
                                                 int new_prev, T* gpark /* fp64 records: this lane's column of 36 x 64 values in LDS */)
{
    using L = Lay<N>;
    using RC = Rec<T, N>;
    if constexpr (sizeof(T) == 8) {
        // fp64 records (round 5): 171 covariance doubles are 342 registers, the 6 x 6 stage wants ~220 and the update 63 coefficients
        // (126 registers) beside the covariance: held all at once the kernels spilled 650-790 bytes per lane.  So:
        //   * only the chunks that hold P(J, J) in front of the 6 x 6 stage;
        //   * the update in two parts (direct_update_part) and two phases: the LATE part first (it reads the old x_c = P(J, c >= 9) of the
        //     early part, S^-1 in registers, and is stored at once), then the early part -- dx and its elements outside J with S^-1, then
        //     the J rows with G, which has waited in LDS (gpark: 36 doubles per lane) -- and the injection.
        constexpr int EPC = RC::EPC, CN = RC::CH_NOM, E0 = late_start<T, N>(), C_E = E0 / EPC;
        T P[RC::NCOVP];
        load_cov_chunks<T, N, 0, C_E, SEL_JJ>(rs, lane, P);
        CoefSm<T> cs;
        {
            double Lam[21], bv[6];
            acc.finish(Rd, w, Lam, bv);
            T G[36];
            {
                double PJJ[36];
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) PJJ[6 * i + j] = (double)P[pidx<N>(jcol(i), jcol(j))];
                info_solve<T>(Lam, bv, PJJ, G, cs.s, cs.m_);
            }
#pragma unroll
            for (int i = 0; i < 36; ++i) gpark[i * 64] = G[i];
        }
        order_fence();
        load_cov_chunks<T, N, 0, C_E, SEL_XL, AUX_NT>(rs, lane, P);
        load_chunks<T, N, CN + C_E, RC::NCH, AUX_NT>(rs, lane, P + E0);
        order_fence();
        T dx[N];
        direct_update_part<T, N, E0, L::NP, false, CoefSm<T>, 1>(P, dx, cs);
        if (new_prev >= 0) P[L::OFF_PREV - L::OFF_COV] = (T)new_prev;
        store_chunks<T, N, CN + C_E, RC::NCH, FBUS_X_CORRECT_ST>(rs, lane, P + E0);
        order_fence();
        // (the rest of the early part only now: requested in front of the late update its load targets spill)
        load_cov_chunks<T, N, 0, C_E, 0, AUX_NT>(rs, lane, P);
        order_fence();
        direct_update_part<T, N, 0, E0, true, CoefSm<T>, 1>(P, dx, cs);
        order_fence();
        {
            CoefG<T> cg;
#pragma unroll
            for (int i = 0; i < 36; ++i) cg.g[i] = gpark[i * 64];
            direct_update_part<T, N, 0, E0, false, CoefG<T>, 2>(P, dx, cg);
        }
        order_fence();
        store_chunks<T, N, CN, CN + C_E, FBUS_X_CORRECT_ST>(rs, lane, P);
        order_fence();
        T nom[L::NNOM];
        load_chunks<T, N, 0, CN>(rs, lane, nom);
        inject<T, N>(nom, dx);
        store_chunks<T, N, 0, RC::CH_PQ, FBUS_X_CORRECT_ST>(rs, lane, nom);
        store_chunks<T, N, RC::CH_PQR, CN, FBUS_X_CORRECT_ST>(rs, lane, nom + L::NPQR);
        return;
    }
    // the covariance is requested here: it arrives under the 6 x 6 stage
    T P[RC::NCOVP];
    load_chunks<T, N, RC::CH_NOM, RC::NCH, AUX_NT>(rs, lane, P);
    double Lam[21], bv[6];
    acc.finish(Rd, w, Lam, bv);
    T dx[N];
    {   // (meas_solve_update's body, kept in place: calling it here changed the register allocation of the tuned per-call kernels)
        T G[36], Sinv[21], m[6];
        {
            double PJJ[36];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) PJJ[6 * i + j] = (double)P[pidx<N>(jcol(i), jcol(j))];
            info_solve<T>(Lam, bv, PJJ, G, Sinv, m);
        }
        RegCoef<T> cf;
        cf.set(G, Sinv, m);
        direct_update<T, N>(P, dx, cf);
    }
    T nom[L::NNOM];
    load_chunks<T, N, 0, RC::CH_NOM>(rs, lane, nom);
    inject<T, N>(nom, dx);
    if (new_prev >= 0) P[L::OFF_PREV - L::OFF_COV] = (T)new_prev;
    // write-through (sc1) as correct_kernel: the lines reach the Infinity Cache at once instead of being written back from the L2s
    // under the tail of the launch
    store_chunks<T, N, 0, RC::CH_PQ, FBUS_X_CORRECT_ST>(rs, lane, nom);
    store_chunks<T, N, RC::CH_PQR, RC::CH_NOM, FBUS_X_CORRECT_ST>(rs, lane, nom + L::NPQR);
    store_chunks<T, N, RC::CH_NOM, RC::NCH, FBUS_X_CORRECT_ST>(rs, lane, P);
}

// Marker map of the pixel fold in LDS: id -> slot (the table of the other kernels) and the double-precision corner frame of
// every slot.
struct alignas(16) MeasLDS {
    short id2slot[FBUS_MAX_MARKER_ID + 1];
    double mkc[FBUS_MAX_MARKERS * MKC_STRIDE];
};

// =================================================================================
// correct() from corner pixels: all visible markers, 2 (left camera) or 4 (stereo) reprojection rows per corner, one
// linearisation point.  One filter per lane; NR waves ("roles") per 64-filter tile divide the markers among themselves
// (role r folds markers r, r + NR, ...; their sums meet in LDS, in role order) and role 0 applies the update.
// =================================================================================
template <typename T, int N, int NR, bool NZ, int CAM = 0>
__global__ void __launch_bounds__(64 * NR)
correct_pixels2_kernel(T* __restrict__ recs, int B, int M, const int* __restrict__ ids, const T* __restrict__ left,
                       const T* __restrict__ right, double size, double r_pix, const unsigned char* __restrict__ skip,
                       unsigned char* __restrict__ applied, const short* __restrict__ id2slot, MeasConst mc)
{
    using L = Lay<N>;
    using RC = Rec<T, N>;
    constexpr int NT = 64 * NR;
    const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned tile = blockIdx.x;
    const int b = (int)(tile * 64u + lane);
    const bool live = b < B && !(skip && skip[b < B ? b : 0]);
    const int bc = b < B ? b : (int)(tile * 64u);
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, tile);
    __shared__ MeasLDS tbl;
    // the roles' partial sums; fp64 records: behind them (role 0 has added them up by then) G waits here while the block outside J is updated
    constexpr int PART_N = NR > 1 ? (NR - 1) * (PixAcc::NVAL + 1) * 64 : 1, GPARK_N = sizeof(T) == 8 ? 36 * 64 : 1;
    __shared__ double part_mem[PART_N > GPARK_N ? PART_N : GPARK_N];
    T* gpark_mem = reinterpret_cast<T*>(part_mem);
    // CAM: 0 = left camera or stereo by the right pointer (one kernel for both), 1 = left camera only, 2 = stereo only: compiled
    // apart, the left-camera kernel carries neither the second camera's image points nor the stereo fold's register pressure
    struct Meas { int id; T l[8], r[CAM == 1 ? 1 : 8]; };
    const bool stereo = CAM == 0 ? right != nullptr : CAM == 2;
    // the id and the 8 (+ 8) image coordinates of marker slot i: 16-byte loads (a slot's 8 coordinates are 32 / 64 contiguous bytes)
    auto fetch = [&](int i, Meas& mm) __attribute__((always_inline)) {
        const size_t o = (size_t)bc * M + i;
        constexpr int EP = 16 / (int)sizeof(T);
        mm.id = ids[o];
        const u32x4* pl = reinterpret_cast<const u32x4*>(left + o * 8);
        const u32x4* pr = reinterpret_cast<const u32x4*>((stereo ? right : left) + o * 8);
#pragma unroll
        for (int c = 0; c < 8 / EP; ++c) {
            const u32x4 vl = pl[c];
            const T* el = reinterpret_cast<const T*>(&vl);
#pragma unroll
            for (int k = 0; k < EP; ++k) mm.l[c * EP + k] = el[k];
            if constexpr (CAM != 1) {
                const u32x4 vr = pr[c];
                const T* er = reinterpret_cast<const T*>(&vr);
#pragma unroll
                for (int k = 0; k < EP; ++k) mm.r[c * EP + k] = er[k];
            }
        }
    };
    Meas cur, nxt;
    T pqr[L::NPQR];
    if constexpr (NR == 1) simd_stagger<FBUS_X_STAGGER_MEAS>();
    {
        // the marker map -> LDS (all threads), this role's first marker, the pose part of the nominal state
        constexpr int NI = (int)sizeof(short) * (FBUS_MAX_MARKER_ID + 1) / 16, NM = (int)sizeof(double) * FBUS_MAX_MARKERS * MKC_STRIDE / 16;
        constexpr int PI = (NI + NT - 1) / NT, PM = (NM + NT - 1) / NT;
        const u32x4* si = reinterpret_cast<const u32x4*>(id2slot);
        const u32x4* sm = reinterpret_cast<const u32x4*>(mc.mkc);
        u32x4* di = reinterpret_cast<u32x4*>(tbl.id2slot);
        u32x4* dm = reinterpret_cast<u32x4*>(tbl.mkc);
        u32x4 vi[PI], vm[PM];
#pragma unroll
        for (int q = 0; q < PI; ++q) { const int i = threadIdx.x + q * NT; vi[q] = si[i < NI ? i : 0]; }
#pragma unroll
        for (int q = 0; q < PM; ++q) { const int i = threadIdx.x + q * NT; vm[q] = sm[i < NM ? i : 0]; }
        order_fence();
        if (M > 0) fetch((int)role < M ? (int)role : M - 1, cur);
        order_fence();
        load_chunks<T, N, 0, RC::CH_PQR>(rs, lane, pqr);
        order_fence();
#pragma unroll
        for (int q = 0; q < PI; ++q) { const int i = threadIdx.x + q * NT; if (i < NI) di[i] = vi[q]; }
#pragma unroll
        for (int q = 0; q < PM; ++q) { const int i = threadIdx.x + q * NT; if (i < NM) dm[i] = vm[q]; }
        order_fence();
    }
    if constexpr (NR > 1) meas_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double pd[3], Rd[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) pd[i] = (double)pqr[L::OFF_P3 + i];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rd[i] = (double)pqr[L::OFF_R + i];
    double pil[3];
    filter_pil(Rd, mc.P_IL, pil);
    PixAcc acc;
    acc.clear();
    double nfold = 0.0;                                          // markers of the map this role has folded
    const int last = live ? M : 0;
#pragma unroll 1
    for (int i = (int)role; i < last; i += NR) {
        fetch(i + NR < M ? i + NR : M - 1, nxt);                // always a fresh load (no conditional merge of the two records)
        const bool ok = cur.id >= 0 && cur.id <= FBUS_MAX_MARKER_ID;
        const int slot = ok ? (int)tbl.id2slot[ok ? cur.id : 0] : -1;
        {
            // every slot is folded; one whose id is not in the map with slot 0's frame and weight 0 (see pixel_fold_marker)
            const double wgt = slot >= 0 ? 1.0 : 0.0;
            const int sl = slot >= 0 ? slot : 0;
            double mk[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) mk[q] = tbl.mkc[sl * MKC_STRIDE + q];
            // the image points of a slot that carries no marker of the map may be anything (padding is the caller's: NaN included), and
            // 0 x NaN would poison the sums: such a slot is folded with zeros
            constexpr int NRR = sizeof(cur.r) / sizeof(T);
            T yl_[8], yr_[NRR];
#pragma unroll
            for (int k = 0; k < 8; ++k) yl_[k] = slot >= 0 ? cur.l[k] : T(0);
#pragma unroll
            for (int k = 0; k < NRR; ++k) yr_[k] = slot >= 0 ? cur.r[k] : T(0);
            if constexpr (CAM == 1) pixel_fold_marker<1, T, NZ, 4, 0, true>(acc, pd, Rd, pil, mc, mk, yl_, yl_, size, wgt);
            else if constexpr (CAM == 2) pixel_fold_marker_stereo_halves<T, NZ>(acc, pd, Rd, pil, mc, mk, yl_, yr_, size, wgt);
            else if (stereo) pixel_fold_marker_stereo_halves<T, NZ>(acc, pd, Rd, pil, mc, mk, yl_, yr_, size, wgt);
            else pixel_fold_marker<1, T, NZ>(acc, pd, Rd, pil, mc, mk, yl_, yl_, size, wgt);
            nfold += wgt;
        }
        cur = nxt;
    }
    if constexpr (NR > 1) {
        // partial sums through LDS: value i of role r at part_mem[((r - 1) * (NVAL + 1) + i) * 64 + lane]
        if (role != 0) {
            double* part = part_mem + ((role - 1) * (PixAcc::NVAL + 1)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < PixAcc::NVAL; ++i) part[i * 64] = acc.at(i);
            part[PixAcc::NVAL * 64] = nfold;
            meas_barrier();
            return;
        }
        meas_barrier();
#pragma unroll
        for (int r = 1; r < NR; ++r) {
            const double* part = part_mem + ((r - 1) * (PixAcc::NVAL + 1)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < PixAcc::NVAL; ++i) acc.at(i) += part[i * 64];
            nfold += part[PixAcc::NVAL * 64];
        }
    }
    if (!live || nfold == 0.0) { if (b < B) applied[b] = 0; return; }
    if constexpr (CAM == 1) {                                           // (the left-only kernel folds in the camera frame)
        acc.to_imu_frame(mc.adjL);
        double RM[9];
        PixAcc::camera_rotation(Rd, mc.McL, RM);
        meas_update_tail<T, N>(rs, lane, acc, RM, 1.0 / r_pix, -1, gpark_mem + lane);
        applied[b] = 1;
        return;
    }
    meas_update_tail<T, N>(rs, lane, acc, Rd, 1.0 / r_pix, -1, gpark_mem + lane);
    applied[b] = 1;
}


// =================================================================================
// correct() from stereo corners (12 position-type rows per marker; north-star extension B2, oracle: fbo_correct_corners):
// nearest marker (by its first corner; C++ dialect: hysteresis against the previous one, filter.cpp:639-664) or all of them.
// Same structure as correct_pixels2_kernel; the triangulation of the refractive geometry runs in double.
// =================================================================================
template <typename T, int N, int NR, bool NZ>
__global__ void __launch_bounds__(64 * NR)
correct_corners2_kernel(T* __restrict__ recs, int B, int M, const int* __restrict__ ids, const T* __restrict__ left,
                        const T* __restrict__ right, int geometry, int mode, int dialect, double size, double r_pos,
                        double switch_thres, const unsigned char* __restrict__ skip, unsigned char* __restrict__ applied,
                        const short* __restrict__ id2slot, MeasConst mc, VisConst<double> vc, VisConst<T> vct)
{
    using L = Lay<N>;
    using RC = Rec<T, N>;
    constexpr int NT = 64 * NR;
    const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned tile = blockIdx.x;
    const int b = (int)(tile * 64u + lane);
    const bool live = b < B && !(skip && skip[b < B ? b : 0]);
    const int bc = b < B ? b : (int)(tile * 64u);
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, tile);
    __shared__ MeasLDS tbl;
    // the roles' partial sums; fp64 records: behind them (role 0 has added them up by then) G waits here while the block outside J is updated
    constexpr int PART_N = NR > 1 ? (NR - 1) * (PixAcc::NVAL + 1) * 64 : 1, GPARK_N = sizeof(T) == 8 ? 36 * 64 : 1;
    __shared__ double part_mem[PART_N > GPARK_N ? PART_N : GPARK_N];
    T* gpark_mem = reinterpret_cast<T*>(part_mem);
    struct Meas { int id; T l[12], r[8]; };
    const bool c3d = geometry == VIS_CORNERS3D;
    const int lw = c3d ? 12 : 8;
    auto fetch = [&](int i, Meas& mm) __attribute__((always_inline)) {
        const size_t o = (size_t)bc * M + i;
        constexpr int EP = 16 / (int)sizeof(T);
        mm.id = ids[o];
        const u32x4* pl = reinterpret_cast<const u32x4*>(left + o * lw);
        const u32x4* pr = reinterpret_cast<const u32x4*>((c3d ? left : right) + o * 8);
#pragma unroll
        for (int c = 0; c < 12 / EP; ++c) {
            const u32x4 vl = pl[(c < 8 / EP || c3d) ? c : 0];
            const T* el = reinterpret_cast<const T*>(&vl);
#pragma unroll
            for (int k = 0; k < EP; ++k) mm.l[c * EP + k] = el[k];
        }
#pragma unroll
        for (int c = 0; c < 8 / EP; ++c) {
            const u32x4 vr = pr[c];
            const T* er = reinterpret_cast<const T*>(&vr);
#pragma unroll
            for (int k = 0; k < EP; ++k) mm.r[c * EP + k] = er[k];
        }
    };
    // the four corners of a measured marker in the left camera frame
    auto corners = [&](const Meas& mm, double (&C)[4][3]) __attribute__((always_inline)) {
        if (geometry == VIS_REFRACTIVE) { tri_corners_refractive<T, NZ>(vc, mm.l, mm.r, C); return; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (geometry == VIS_CORNERS3D) {
#pragma unroll
                for (int i = 0; i < 3; ++i) C[k][i] = (double)mm.l[3 * k + i];
            } else {
                T o3[3];                                 // pin-hole DLT (vision.cpp:395-466): the 4 x 4 eigen-solver stays in T
                pinhole_corner(vct, mm.l[2 * k], mm.l[2 * k + 1], mm.r[2 * k], mm.r[2 * k + 1], o3);
#pragma unroll
                for (int i = 0; i < 3; ++i) C[k][i] = (double)o3[i];
            }
        }
    };
    Meas cur, nxt;
    T pqr[L::NPQR];
    T prev_raw = T(0);
    if constexpr (NR == 1) simd_stagger<FBUS_X_STAGGER_MEAS>();
    {
        constexpr int NI = (int)sizeof(short) * (FBUS_MAX_MARKER_ID + 1) / 16, NM = (int)sizeof(double) * FBUS_MAX_MARKERS * MKC_STRIDE / 16;
        constexpr int PI = (NI + NT - 1) / NT, PM = (NM + NT - 1) / NT;
        const u32x4* si = reinterpret_cast<const u32x4*>(id2slot);
        const u32x4* sm = reinterpret_cast<const u32x4*>(mc.mkc);
        u32x4* di = reinterpret_cast<u32x4*>(tbl.id2slot);
        u32x4* dm = reinterpret_cast<u32x4*>(tbl.mkc);
        u32x4 vi[PI], vm[PM];
#pragma unroll
        for (int q = 0; q < PI; ++q) { const int i = threadIdx.x + q * NT; vi[q] = si[i < NI ? i : 0]; }
#pragma unroll
        for (int q = 0; q < PM; ++q) { const int i = threadIdx.x + q * NT; vm[q] = sm[i < NM ? i : 0]; }
        order_fence();
        if (M > 0) fetch((int)role < M ? (int)role : M - 1, cur);
        if (mode == MODE_NEAREST && dialect == DIALECT_CPP) prev_raw = recs[elem_index<T, N>(bc, L::OFF_PREV)];
        order_fence();
        load_chunks<T, N, 0, RC::CH_PQR>(rs, lane, pqr);
        order_fence();
#pragma unroll
        for (int q = 0; q < PI; ++q) { const int i = threadIdx.x + q * NT; if (i < NI) di[i] = vi[q]; }
#pragma unroll
        for (int q = 0; q < PM; ++q) { const int i = threadIdx.x + q * NT; if (i < NM) dm[i] = vm[q]; }
        order_fence();
    }
    if constexpr (NR > 1) meas_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double pd[3], Rd[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) pd[i] = (double)pqr[L::OFF_P3 + i];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rd[i] = (double)pqr[L::OFF_R + i];
    double pil[3];
    filter_pil(Rd, mc.P_IL, pil);
    PixAcc acc;
    acc.clear();
    double nfold = 0.0;
    int new_prev = -1;
    auto fold_marker = [&](const Meas& mm) __attribute__((always_inline)) {
        const bool ok = mm.id >= 0 && mm.id <= FBUS_MAX_MARKER_ID;
        const int slot = ok ? (int)tbl.id2slot[ok ? mm.id : 0] : -1;
        if (slot < 0) return false;
        double mk[9], C[4][3];
#pragma unroll
        for (int q = 0; q < 9; ++q) mk[q] = tbl.mkc[slot * MKC_STRIDE + q];
        corners(mm, C);
        corner_fold_marker(acc, pd, Rd, pil, mc, mk, C, size);
        return true;
    };
    if (mode == MODE_NEAREST) {
        // one role only (the launcher): nearest visible marker by its first corner
        if (live) {
            const int prev_id = (int)prev_raw;
            int min_i = -1, prev_i = -1;
            double min_d = 10.0, prev_d = 0.0;
#pragma unroll 1
            for (int i = 0; i < M; ++i) {
                fetch(i + 1 < M ? i + 1 : M - 1, nxt);
                if (cur.id >= 0) {
                    double C[4][3];
                    corners(cur, C);
                    const double dist = sqrt(C[0][0] * C[0][0] + C[0][1] * C[0][1] + C[0][2] * C[0][2]);
                    if (dist < min_d) { min_d = dist; min_i = i; }
                    if (dialect == DIALECT_CPP && cur.id == prev_id) { prev_d = dist; prev_i = i; }
                }
                cur = nxt;
            }
            if (min_i >= 0) {
                if (dialect == DIALECT_CPP && prev_i >= 0 && fabs(prev_d - min_d) < switch_thres && prev_d != 0.0) min_i = prev_i;
                fetch(min_i, cur);
                if (fold_marker(cur)) { nfold = 1.0; if (dialect == DIALECT_CPP) new_prev = cur.id; }
            }
        }
    } else {
        const int last = live ? M : 0;
#pragma unroll 1
        for (int i = (int)role; i < last; i += NR) {
            fetch(i + NR < M ? i + NR : M - 1, nxt);
            if (fold_marker(cur)) nfold += 1.0;
            cur = nxt;
        }
    }
    if constexpr (NR > 1) {
        if (role != 0) {
            double* part = part_mem + ((role - 1) * (PixAcc::NVAL + 1)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < PixAcc::NVAL; ++i) part[i * 64] = acc.at(i);
            part[PixAcc::NVAL * 64] = nfold;
            meas_barrier();
            return;
        }
        meas_barrier();
#pragma unroll
        for (int r = 1; r < NR; ++r) {
            const double* part = part_mem + ((r - 1) * (PixAcc::NVAL + 1)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < PixAcc::NVAL; ++i) acc.at(i) += part[i * 64];
            nfold += part[PixAcc::NVAL * 64];
        }
    }
    if (!live || nfold == 0.0) { if (b < B) applied[b] = 0; return; }
    acc.expand_const(mc.NI);
    meas_update_tail<T, N>(rs, lane, acc, Rd, 1.0 / r_pos, new_prev, gpark_mem + lane);
    applied[b] = 1;
}

// =================================================================================
// One camera frame in ONE launch with the north star's own MeasureUpdate (round 5): K ImuUpdates with the record resident in
// registers (predict_steps: the arithmetic of K fbus_ekf_predict calls), then correct() from corner pixels (KIND = MEAS_PIXELS:
// correct_pixels2_kernel's fold) or from stereo corners (MEAS_CORNERS: correct_corners2_kernel's), then the one-shot update --
// the reference's BatchImuProcessing + ObservationUpdate (filter.cpp:232-235) with the measurement model of BASELINE.json's
// north_star in place of the pose rows.  Same device functions in the same order per filter as K predict launches + one
// correct_pixels / correct_corners launch with one wave per tile: equal to that sequence to fp32 rounding (the single-step gate; the
// predict loop is built with FBUS_X_PACK_FMEAS, the per-call predict with FBUS_X_PACK, and FMA contraction differs between the
// kernels); the update alone (K = 0) and window-vs-frames ARE bit-equal (tests/test_frame_meas_gpu.py).
//
// Registers: the predict loop holds the whole record (28 + 172 values) beside its coefficients; the double-precision fold needs
// ~390 registers WITHOUT the covariance.  So between the last ImuUpdate and the update the covariance waits outside the register
// file: the chunks ImuUpdate can change (N = 18: 33 of 43; N = 15: all 31) in LDS -- 33 KiB per wave, lane-consecutive 16-byte
// slots, conflict-free; with the 4.5 KiB marker table 37.5 KiB per one-wave workgroup, four of which fit a CU's 160 KiB: one wave
// per SIMD, the occupancy the 400-register fold allows anyway -- and the predict-invariant tail (N = 18: ba / bg / g block, 10
// chunks) is not kept at all: it is in the record as it was, and comes back from L2 under the 6 x 6 stage.  The nominal state
// stays in registers (the fold reads p, R from it; 12 values more than the per-call kernel's pqr).
// What the fusion saves against predict_n + correct_pixels: one launch, the record's way out and in again between them (the
// per-call update waits ~8 us for p, q, R and again for the covariance), and 2 x 800 B of traffic per filter and frame.
// =================================================================================
template <typename T> struct QDiag { T qd[4]; };      // (MEAS_PIXELS / MEAS_CORNERS: ekf_launch.hpp)

template <typename T, int N, int DIALECT, int KIND, bool NZ, bool WINDOW = false, int CAM = 0>
__global__ void __launch_bounds__(64)
frame_meas_kernel(T* __restrict__ recs, int B, int F, FrameCounts kc, const T* __restrict__ accel, const T* __restrict__ gyro,
                  const T* __restrict__ dt, int dt_stride, int M, const int* __restrict__ ids, const T* __restrict__ left,
                  const T* __restrict__ right, int geometry, int mode, double size, double r_meas, double switch_thres,
                  const unsigned char* __restrict__ skip, unsigned char* __restrict__ applied, const short* __restrict__ id2slot,
                  MeasConst mc, VisConst<double> vc, VisConst<T> vct, QDiag<T> qd)
{
    using L = Lay<N>;
    using RC = Rec<T, N>;
    constexpr int EPC = RC::EPC, CN = RC::CH_NOM;
    constexpr int PCH = RC::CH_VAR_END - CN;             // covariance chunks parked in LDS across the fold
    constexpr int NT = 64;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned tile = blockIdx.x;
    const int b = (int)(tile * 64u + lane);
    const int bc = b < B ? b : (int)(tile * 64u);
    size_t fo = 0;                                       // f * B: where frame f's measurements start (WINDOW)
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, tile);
    __shared__ MeasLDS tbl;
    __shared__ u32x4 park_mem[PCH * 64];
    u32x4* park = park_mem + lane;
    // CAM (pixel rows): 0 = left camera or stereo by the right pointer, 1 = left camera only, 2 = stereo only (correct_pixels2_kernel's)
    const bool stereo = (KIND == MEAS_PIXELS && CAM != 0) ? CAM == 2 : right != nullptr;
    const bool c3d = KIND == MEAS_CORNERS && geometry == VIS_CORNERS3D;
    const int lw = c3d ? 12 : 8;
    struct Meas { int id; T l[KIND == MEAS_CORNERS ? 12 : 8], r[(KIND == MEAS_PIXELS && CAM == 1) ? 1 : 8]; };
    // the id and the image coordinates of marker slot i (16-byte loads; the layouts of correct_pixels2 / correct_corners2_kernel)
    auto fetch = [&](int i, Meas& mm) __attribute__((always_inline)) {
        const size_t o = (fo + (size_t)bc) * M + i;
        constexpr int EP = 16 / (int)sizeof(T);
        mm.id = ids[o];
        const u32x4* pl = reinterpret_cast<const u32x4*>(left + o * lw);
        const u32x4* pr = reinterpret_cast<const u32x4*>(((KIND == MEAS_CORNERS ? c3d : !stereo) ? left : right) + o * 8);
        if constexpr (KIND == MEAS_PIXELS) {
#pragma unroll
            for (int c = 0; c < 8 / EP; ++c) {
                const u32x4 vl = pl[c];
                const T* el = reinterpret_cast<const T*>(&vl);
#pragma unroll
                for (int k = 0; k < EP; ++k) mm.l[c * EP + k] = el[k];
                if constexpr (CAM != 1) {
                    const u32x4 vr = pr[c];
                    const T* er = reinterpret_cast<const T*>(&vr);
#pragma unroll
                    for (int k = 0; k < EP; ++k) mm.r[c * EP + k] = er[k];
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 12 / EP; ++c) {
                const u32x4 vl = pl[(c < 8 / EP || c3d) ? c : 0];
                const T* el = reinterpret_cast<const T*>(&vl);
#pragma unroll
                for (int k = 0; k < EP; ++k) mm.l[c * EP + k] = el[k];
            }
#pragma unroll
            for (int c = 0; c < 8 / EP; ++c) {
                const u32x4 vr = pr[c];
                const T* er = reinterpret_cast<const T*>(&vr);
#pragma unroll
                for (int k = 0; k < EP; ++k) mm.r[c * EP + k] = er[k];
            }
        }
    };
    // the four corners of a measured marker in the left camera frame (MEAS_CORNERS)
    auto corners = [&](const Meas& mm, double (&C)[4][3]) __attribute__((always_inline)) {
        if (geometry == VIS_REFRACTIVE) { tri_corners_refractive<T, NZ>(vc, mm.l, mm.r, C); return; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (geometry == VIS_CORNERS3D) {
#pragma unroll
                for (int i = 0; i < 3; ++i) C[k][i] = (double)mm.l[(3 * k + i) % (KIND == MEAS_CORNERS ? 12 : 8)];
            } else {
                T o3[3];
                pinhole_corner(vct, mm.l[2 * k], mm.l[2 * k + 1], mm.r[2 * k], mm.r[2 * k + 1], o3);
#pragma unroll
                for (int i = 0; i < 3; ++i) C[k][i] = (double)o3[i];
            }
        }
    };
    T nom[L::NNOM], P[RC::NCOVP];
    Meas cur, nxt;
    T prev_raw = T(0);
    simd_stagger<FBUS_X_STAGGER_FRAME>();
    {
        {
            // the marker map -> LDS, the record behind it (lanes past B load their existing tile too: no branch in front of the loads)
            constexpr int NI = (int)sizeof(short) * (FBUS_MAX_MARKER_ID + 1) / 16, NM = (int)sizeof(double) * FBUS_MAX_MARKERS * MKC_STRIDE / 16;
            constexpr int PI = (NI + NT - 1) / NT, PM = (NM + NT - 1) / NT;
            const u32x4* si = reinterpret_cast<const u32x4*>(id2slot);
            const u32x4* sm = reinterpret_cast<const u32x4*>(mc.mkc);
            u32x4* di = reinterpret_cast<u32x4*>(tbl.id2slot);
            u32x4* dm = reinterpret_cast<u32x4*>(tbl.mkc);
            u32x4 vi[PI], vm[PM];
#pragma unroll
            for (int q = 0; q < PI; ++q) { const int i = threadIdx.x + q * NT; vi[q] = si[i < NI ? i : 0]; }
#pragma unroll
            for (int q = 0; q < PM; ++q) { const int i = threadIdx.x + q * NT; vm[q] = sm[i < NM ? i : 0]; }
            order_fence();
            load_chunks<T, N, 0, CN, AUX_NT>(rs, lane, nom);
            load_chunks<T, N, CN, RC::NCH, AUX_NT>(rs, lane, P);
            order_fence();
#pragma unroll
            for (int q = 0; q < PI; ++q) { const int i = threadIdx.x + q * NT; if (i < NI) di[i] = vi[q]; }
#pragma unroll
            for (int q = 0; q < PM; ++q) { const int i = threadIdx.x + q * NT; if (i < NM) dm[i] = vm[q]; }
            order_fence();
        }
    }
    if (b >= B) return;
    // WINDOW: F times { K_f ImuUpdates, the update } with the record resident from the first load to the last store (offline replay of a
    // recorded stretch of corners.txt, FBUS_EKF.m:151-210); otherwise one pass (F = 1).  Same device functions in the same order per filter
    // as F launches of the frame form: bit-identical results.
    int k0 = 0;
    bool did = false;
#pragma unroll 1
    for (int f = 0; f < (WINDOW ? F : 1); ++f) {
    const int K = kc.k[f];
    fo = (size_t)f * B;
    const bool live = M > 0 && !(skip && skip[fo + b]);
    {
        // K ImuUpdates; under the covariance stages of the last one the first marker's image points are requested
        auto first_marker = [&]() __attribute__((always_inline)) { if (M > 0) fetch(0, cur); };
        // (the C++-dialect stereo frame -- 256 + 256 registers -- spilled 36 bytes with the packed nominal step; its window takes the same mask:
        // the window of frames equals the sequence of frames bit for bit only while both run the same ImuUpdate instructions)
        constexpr int PKM = (KIND == MEAS_PIXELS && CAM == 2 && DIALECT == 1) ? FBUS_X_PACK_FMEAS_CPP_STEREO : FBUS_X_PACK_FMEAS;
        predict_steps<T, N, DIALECT, decltype(first_marker), NoStepPark, PKM>(nom, P, K, accel + (size_t)k0 * B * 3, gyro + (size_t)k0 * B * 3, dt + (size_t)k0 * (dt_stride ? B : 1),
                                     dt_stride, B, b, qd.qd, first_marker);
        k0 += K;
        order_fence();
        if (KIND == MEAS_CORNERS && mode == MODE_NEAREST && DIALECT == DIALECT_CPP) prev_raw = P[L::OFF_PREV - L::OFF_COV];
        // the covariance as predicted -> LDS (what ImuUpdate can change; the rest is in the record as it was)
#pragma unroll
        for (int c = 0; c < PCH; ++c) {
            u32x4 v;
            T* e = reinterpret_cast<T*>(&v);
#pragma unroll
            for (int k = 0; k < EPC; ++k) e[k] = P[c * EPC + k];
            park[c * 64] = v;
        }
        order_fence();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double pd[3], Rd[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) pd[i] = (double)nom[L::OFF_P3 + i];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rd[i] = (double)nom[L::OFF_R + i];
    double pil[3];
    filter_pil(Rd, mc.P_IL, pil);
    PixAcc acc;
    acc.clear();
    double nfold = 0.0;
    int new_prev = -1;
    if constexpr (KIND == MEAS_PIXELS) {
        const int last = live ? M : 0;
#pragma unroll 1
        for (int i = 0; i < last; ++i) {
            fetch(i + 1 < M ? i + 1 : M - 1, nxt);
            const bool ok = cur.id >= 0 && cur.id <= FBUS_MAX_MARKER_ID;
            const int slot = ok ? (int)tbl.id2slot[ok ? cur.id : 0] : -1;
            {
                const double wgt = slot >= 0 ? 1.0 : 0.0;            // (every slot is folded: see pixel_fold_marker)
                const int sl = slot >= 0 ? slot : 0;
                double mk[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) mk[q] = tbl.mkc[sl * MKC_STRIDE + q];
                constexpr int NRR = sizeof(cur.r) / sizeof(T);          // (a slot without a marker of the map: zeros for its image points)
                T yl_[8], yr_[NRR];
#pragma unroll
                for (int k = 0; k < 8; ++k) yl_[k] = slot >= 0 ? cur.l[k] : T(0);
#pragma unroll
                for (int k = 0; k < NRR; ++k) yr_[k] = slot >= 0 ? cur.r[k] : T(0);
                if constexpr (CAM == 1) pixel_fold_marker<1, T, NZ, 4, 0, true>(acc, pd, Rd, pil, mc, mk, yl_, yl_, size, wgt);
                else if constexpr (CAM == 2) pixel_fold_marker_stereo_halves<T, NZ>(acc, pd, Rd, pil, mc, mk, yl_, yr_, size, wgt);
                else if (stereo) pixel_fold_marker_stereo_halves<T, NZ>(acc, pd, Rd, pil, mc, mk, yl_, yr_, size, wgt);
                else pixel_fold_marker<1, T, NZ>(acc, pd, Rd, pil, mc, mk, yl_, yl_, size, wgt);
                nfold += wgt;
            }
            cur = nxt;
        }
    } else {
        auto fold_marker = [&](const Meas& mm) __attribute__((always_inline)) {
            const bool ok = mm.id >= 0 && mm.id <= FBUS_MAX_MARKER_ID;
            const int slot = ok ? (int)tbl.id2slot[ok ? mm.id : 0] : -1;
            if (slot < 0) return false;
            double mk[9], C[4][3];
#pragma unroll
            for (int q = 0; q < 9; ++q) mk[q] = tbl.mkc[slot * MKC_STRIDE + q];
            corners(mm, C);
            corner_fold_marker(acc, pd, Rd, pil, mc, mk, C, size);
            return true;
        };
        if (mode == MODE_NEAREST) {
            if (live) {
                const int prev_id = (int)prev_raw;
                int min_i = -1, prev_i = -1;
                double min_d = 10.0, prev_d = 0.0;
#pragma unroll 1
                for (int i = 0; i < M; ++i) {
                    fetch(i + 1 < M ? i + 1 : M - 1, nxt);
                    if (cur.id >= 0) {
                        double C[4][3];
                        corners(cur, C);
                        const double dist = sqrt(C[0][0] * C[0][0] + C[0][1] * C[0][1] + C[0][2] * C[0][2]);
                        if (dist < min_d) { min_d = dist; min_i = i; }
                        if (DIALECT == DIALECT_CPP && cur.id == prev_id) { prev_d = dist; prev_i = i; }
                    }
                    cur = nxt;
                }
                if (min_i >= 0) {
                    if (DIALECT == DIALECT_CPP && prev_i >= 0 && fabs(prev_d - min_d) < switch_thres && prev_d != 0.0) min_i = prev_i;
                    fetch(min_i, cur);
                    if (fold_marker(cur)) { nfold = 1.0; if (DIALECT == DIALECT_CPP) new_prev = cur.id; }
                }
            }
        } else {
            const int last = live ? M : 0;
#pragma unroll 1
            for (int i = 0; i < last; ++i) {
                fetch(i + 1 < M ? i + 1 : M - 1, nxt);
                if (fold_marker(cur)) nfold += 1.0;
                cur = nxt;
            }
        }
    }
    order_fence();
    did = live && nfold != 0.0;
    if (!did) {
        // no update for this filter (skipped, no usable marker, M = 0): the record as predicted -- what ImuUpdate writes
#pragma unroll
        for (int c = 0; c < PCH; ++c) {
            const u32x4 v = park[c * 64];
            const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int k = 0; k < EPC; ++k) P[c * EPC + k] = e[k];
        }
        if constexpr (!WINDOW) {
            store_chunks<T, N, 0, RC::CH_KIN, FBUS_X_FMEAS_ST>(rs, lane, nom);
            store_chunks<T, N, CN, RC::CH_VAR_END, FBUS_X_FMEAS_ST>(rs, lane, P);
            if (M > 0) applied[b] = 0;
            return;
        } else {
            // the next frame's ImuUpdates read the predict-invariant tail as well: back from the record (current: see below)
            if constexpr (RC::CH_VAR_END < RC::NCH) load_chunks<T, N, RC::CH_VAR_END, RC::NCH>(rs, lane, P + PCH * EPC);
            continue;
        }
    }
    // the predict-invariant tail from the record (L2-hot: this wave read it at the top), the rest back from LDS
    if constexpr (RC::CH_VAR_END < RC::NCH) load_chunks<T, N, RC::CH_VAR_END, RC::NCH>(rs, lane, P + PCH * EPC);
#pragma unroll
    for (int c = 0; c < PCH; ++c) {
        const u32x4 v = park[c * 64];
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int k = 0; k < EPC; ++k) P[c * EPC + k] = e[k];
    }
    if constexpr (KIND == MEAS_CORNERS) acc.expand_const(mc.NI);
    T dx[N];
    if constexpr (KIND == MEAS_PIXELS && CAM == 1) {                            // (the left-only kernel folds in the camera frame)
        acc.to_imu_frame(mc.adjL);
        double RM[9];
        PixAcc::camera_rotation(Rd, mc.McL, RM);
        meas_solve_update<T, N, !WINDOW && NZ>(P, acc, RM, 1.0 / r_meas, dx);
    } else
        meas_solve_update<T, N, !WINDOW && NZ>(P, acc, Rd, 1.0 / r_meas, dx);  // (packed pairs: the one-frame square-port kernels)
    inject<T, N>(nom, dx);
    if (new_prev >= 0) P[L::OFF_PREV - L::OFF_COV] = (T)new_prev;
    if constexpr (!WINDOW) {
        store_chunks<T, N, 0, CN, FBUS_X_FMEAS_ST>(rs, lane, nom);
        store_chunks<T, N, CN, RC::NCH, FBUS_X_FMEAS_ST>(rs, lane, P);
        applied[b] = 1;
        return;
    } else {
        // the update changed the predict-invariant tail too, and the next frame's tail fetches it from the record again: out it goes
        // (ten chunks; a later load of this wave from the same addresses returns what was stored: vector memory operations of one wave
        // are served in order -- frame2_kernel relies on the same)
        if constexpr (RC::CH_VAR_END < RC::NCH) {
            if (f + 1 < F) store_chunks<T, N, RC::CH_VAR_END, RC::NCH, FBUS_X_FMEAS_ST>(rs, lane, P + PCH * EPC);
        }
        order_fence();
    }
    }   // frames
    if constexpr (WINDOW) {
        store_chunks<T, N, 0, CN, FBUS_X_FMEAS_ST>(rs, lane, nom);
        store_chunks<T, N, CN, RC::NCH, FBUS_X_FMEAS_ST>(rs, lane, P);
        if (M > 0 && F > 0) applied[b] = did ? 1 : 0;
    }
}


}  // namespace
