// ekf_meas_split.hpp -- correct() from corner pixels with the UPDATE divided between the waves of a tile (round 5).
//
// Reference algebra (paths relative to the upstream repository): the rows of matlab/MeasureUpdate.m:67,72-73 with a marker corner in
// place of the marker origin, through the flat-port model of C++/src/vision.cpp:496-599 run forward (ekf_meas.hpp); the update
// matlab/MeasureUpdate.m:84-102 ; filter.cpp:709-739 in the one-shot form of ekf_meas.hpp::direct_update.
//
// Why.  correct_pixels2_kernel (ekf_meas.hpp) keeps a filter's whole update in one wave: ~400 registers, one wave per SIMD.  At one wave
// per SIMD an fp64 instruction issues every 5.3 cycles (4.45 with two waves, profiles/r04_issue_rates.txt) and every wave of the launch
// walks through the same phases at the same time -- all fold (memory idle), then all wait for their covariance, then all store.  Its NR
// "roles" divide the FOLD of a filter's markers over NR waves, but the tail -- sums -> 6 x 6 stage -> update of 171 covariance elements
// -> injection -> 50 chunk stores -- stays with role 0 while the others have left (round 4 measured ~9 us of a 17 us config-3 launch
// there), and a 256-register form whose tail ran row-split through LDS on one wave lost (profiles/r04_meas_two_wave.txt).
//
// Here every wave stays below 256 registers (two waves per SIMD at 65 536 filters with two waves per tile) and the tail is divided:
//   fold      role r folds markers r, r + NR, ...  corner by corner (pixel_fold_corners_nz: the arithmetic of pixel_fold_marker's
//             square-port path, one corner's NCAM projections per pass of a loop that is not unrolled: ~230 registers instead of ~390)
//   exchange  roles >= 1 leave their 27 sums in LDS; barrier
//   SOLVER    (role 0) adds the sums in role order, forms Lam, b, solves the 6 x 6 stage in double from P_JJ (it has requested the
//             chunks that hold P(J, J) and P(J, c >= 9) before the fold ended), leaves G, S^-1, m (63 values) in LDS; barrier;
//             then updates and stores the LATE part of the covariance -- the chunks that hold only elements P(a, c), a, c >= 9
//             (N = 18: storage [132, 172), the predict-invariant tail; all of type "outside J": P(a, c) -= x_a' S^-1 x_c)
//   UPDATER   (role 1) has requested the EARLY part (rows 0..8 and the collected diagonals: every x_c = P(J, c) lives there) and the
//             nominal state while the solver worked; behind the second barrier it reads the 63 coefficients, forms dx = P(:, J) m,
//             updates the early part (the J rows as the product G P(J, :), the rest by subtraction), injects dx, stores
//   roles >= 2 (four waves per tile: small launches) fold their share and leave behind the second barrier.
// Per element the operations are those of direct_update in the same order (direct_update_part below is that function restricted to a
// storage range): the posterior equals the one-wave kernel's bit for bit given the same sums; the sums are added in role order as in
// correct_pixels2_kernel<., ., NR, .>.
// Square port (normal = (0, 0, 1), the reference's configuration) and fp32 records only: everything else keeps correct_pixels2_kernel.
#pragma once
#include "ekf_meas.hpp"
#include "ekf_team.hpp"            // CovMap: storage index -> (row, column)

namespace {

// ---- who owns what -----------------------------------------------------------------------------------------------------
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

// which elements of the three groups of direct_update lie in the storage range [LO, HI)
template <int N, int LO, int HI> constexpr bool in_part(int i, int j) { return pidx<N>(i, j) >= LO && pidx<N>(i, j) < HI; }
template <int N, int LO, int HI> constexpr bool part_any_rr(int c) { bool a_ = false; for (int a = 0; a <= c; ++a) a_ = a_ || in_part<N, LO, HI>(rcol(a), rcol(c)); return a_; }
template <int N, int LO, int HI> constexpr bool part_any_jj() { bool a_ = false; for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) a_ = a_ || in_part<N, LO, HI>(jcol(i), jcol(j)); return a_; }
template <int N, int LO, int HI> constexpr bool part_all_jj() { bool a_ = true; for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) a_ = a_ && in_part<N, LO, HI>(jcol(i), jcol(j)); return a_; }
template <int N, int LO, int HI> constexpr bool part_any_jr(int c) { bool a_ = false; for (int i = 0; i < 6; ++i) a_ = a_ || in_part<N, LO, HI>(jcol(i), rcol(c)); return a_; }
template <int N, int LO, int HI> constexpr bool part_all_jr(int c) { bool a_ = true; for (int i = 0; i < 6; ++i) a_ = a_ && in_part<N, LO, HI>(jcol(i), rcol(c)); return a_; }

// direct_update (ekf_meas.hpp) restricted to the covariance elements with storage index in [LO, HI): the same operations on every
// element it touches, in the same order.  P is the full-size array; every element of the J rows / columns that the touched elements
// need must be loaded (x_c = P(J, c)); WANT_DX: dx = P(:, J) m as well (needs all of the J rows).
template <typename T, int N, int LO, int HI, bool WANT_DX, typename COEF>
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
            for (int k = 1; k < 6; ++k) s += PS(i, jcol(k)) * cf.m(k);
            dx[i] = s;
        }
    }
    // the block outside J, column by column: t = Sinv x_c, then P(a, c) -= x_a . t for the columns a <= c (the x are still the old ones)
    static_for<0, NR_>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        if constexpr (part_any_rr<N, LO, HI>(c)) {
            T t[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                T s = cf.S(0, i) * PS(jcol(0), rcol(c));
#pragma unroll
                for (int j = 1; j < 6; ++j) s += cf.S(j, i) * PS(jcol(j), rcol(c));
                t[i] = s;
            }
            static_for<0, c + 1>([&](auto a_) {
                constexpr int a = decltype(a_)::value;
                if constexpr (INR(rcol(a), rcol(c))) {
                    T s = PS(jcol(0), rcol(a)) * t[0];
#pragma unroll
                    for (int k = 1; k < 6; ++k) s += PS(jcol(k), rcol(a)) * t[k];
                    PS(rcol(a), rcol(c)) -= s;
                }
            });
        }
    });
    // the J x J block from the old values (upper triangle of G P_JJ), then the J x r columns in place
    if constexpr (part_any_jj<N, LO, HI>()) {
        // (the J x J block lies in ONE part: the early one)
        T nj[21];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) {
                T s = cf.G(i, 0) * PS(jcol(0), jcol(j));
#pragma unroll
                for (int k = 1; k < 6; ++k) s += cf.G(i, k) * PS(jcol(k), jcol(j));
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
        if constexpr (part_any_jr<N, LO, HI>(c)) {
            static_assert(part_all_jr<N, LO, HI>(c), "a column of the J rows must lie in one part");
            T x[6], y[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) x[j] = PS(jcol(j), rcol(c));
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                T s = cf.G(i, 0) * x[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) s += cf.G(i, j) * x[j];
                y[i] = s;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) PS(jcol(i), rcol(c)) = y[i];
        }
    });
#undef INR
#undef PS
}

// ---- one marker, corner by corner: the square-port path of pixel_fold_marker in ~230 registers --------------------------------
// Same arithmetic per projection as pixel_fold_marker<NCAM, T, true> (ekf_meas.hpp; see there for the geometry, the closed-form start of
// the port equation and the Halley step); one corner's NCAM projections per pass.  Corner k in the IMU frame by addition from corner 0
// and the two edge vectors as there, the flags wave-uniform; its camera-frame position from the corner itself (see below).
template <int NCAM, typename T>
__device__ __forceinline__ void pixel_fold_corners_nz(PixAcc& acc, const double* p, const double* R, const double* pil, const MeasConst& mc,
                                                      const double* mkc, const T* yl, const T* yr, double size)
{
    constexpr int NS = sizeof(T) == 8 ? 2 : 1;
    constexpr int NP = NCAM;
    double ru0[3], rAx[3], rAy[3];
    {
        double u0[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) u0[i] = mkc[i] - p[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ru0[i] = R[i] * u0[0] + R[3 + i] * u0[1] + R[6 + i] * u0[2];
            rAx[i] = size * (R[i] * mkc[3] + R[3 + i] * mkc[4] + R[6 + i] * mkc[5]);
            rAy[i] = size * (R[i] * mkc[6] + R[3 + i] * mkc[7] + R[6 + i] * mkc[8]);
        }
    }
    const double c0 = (mc.d_air + mc.d_glass * mc.a0) / mc.a1;
    const double klim = 0.81 * mc.a1 * mc.a1 / (1.0 - mc.a1 * mc.a1);
    const double q1 = 1.0 - mc.a1 * mc.a1, a12 = mc.a1 * mc.a1, Gd0 = mc.d_glass * mc.a0;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        // corners c_k = (0,0,0), (0,s,0), (s,s,0), (s,0,0) of the marker frame (vision.cpp:736-759)
        const bool by = (k == 1 || k == 2), bx = (k >= 2);
        double ru[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // ru0 / ru0 + rAy / ru0 + rAx + rAy / ru0 + rAx, with pixel_fold_marker's association
            const double a_ = by ? (bx ? rAx[i] + rAy[i] : rAy[i]) : (bx ? rAx[i] : 0.0);
            ru[i] = (by || bx) ? ru0[i] + a_ : ru0[i];
        }
        double lat[NP][2], rho[NP], irho[NP], Wd[NP], vis[NP], t[NP];
        {
            double X[NP][3], r2[NP], zwq[NP], r2s[NP], xs[NP], ir0[NP], ze[NP];
            bool ok[NP];
            // the corner in the refraction frame of each camera: X = M_c (ru_k - pil) + t_c, from the corner itself (pixel_fold_marker adds the
            // edge vectors in the camera frames instead -- 36 doubles that would have to stay live across the corners; same value to rounding)
            const double tI[3] = { ru[0] - pil[0], ru[1] - pil[1], ru[2] - pil[2] };
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const double* M = q ? mc.McR : mc.McL;
#pragma unroll
                for (int i = 0; i < 3; ++i) X[q][i] = M[3 * i] * tI[0] + M[3 * i + 1] * tI[1] + M[3 * i + 2] * tI[2] + (q ? mc.tR[i] : 0.0);
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                lat[q][0] = X[q][0]; lat[q][1] = X[q][1];
                r2[q] = X[q][0] * X[q][0] + X[q][1] * X[q][1];
                zwq[q] = X[q][2] - (mc.d_air + mc.d_glass);
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                ok[q] = (zwq[q] > 0.0) && (r2[q] < klim * zwq[q] * zwq[q]);
                vis[q] = ok[q] ? 1.0 : 0.0;
                r2s[q] = ok[q] ? r2[q] : 0.0;
                const double zs = ok[q] ? zwq[q] : 1.0;
                Wd[q] = zs * mc.a1;
                ze[q] = zs + c0;
                xs[q] = r2s[q] > 0.0 ? r2s[q] : 1.0;
            }
            md_rsq_n<NS, NP>(xs, ir0);
#pragma unroll
            for (int q = 0; q < NP; ++q) { irho[q] = (r2s[q] > 0.0) ? ir0[q] : 0.0; rho[q] = r2s[q] * irho[q]; }
            double u[NP], w_[NP], r_[NP], s_[NP], izw[NP];
            md_rcp_n<0, NP>(ze, u);
#pragma unroll
            for (int q = 0; q < NP; ++q) { u[q] *= rho[q]; w_[q] = fmax(a12 - q1 * u[q] * u[q], 1e-6); }
            md_rsq_n<0, NP>(w_, w_);
#pragma unroll
            for (int q = 0; q < NP; ++q) { t[q] = u[q] * w_[q]; r_[q] = 1.0 + t[q] * t[q]; izw[q] = Wd[q]; }
            md_rsq_n<0, NP>(r_, r_);
            md_rcp_n<0, NP>(izw, izw);
#pragma unroll
            for (int q = 0; q < NP; ++q) { s_[q] = t[q] * r_[q]; w_[q] = 1.0 - mc.a0 * mc.a0 * s_[q] * s_[q]; }
            md_rsq_n<0, NP>(w_, w_);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                u[q] = fmax((rho[q] - mc.d_air * t[q] - Gd0 * s_[q] * w_[q]) * izw[q] * mc.a1, 0.0);
                w_[q] = fmax(a12 - q1 * u[q] * u[q], 1e-6);
            }
            md_rsq_n<0, NP>(w_, w_);
#pragma unroll
            for (int q = 0; q < NP; ++q) t[q] = u[q] * w_[q];
        }
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
#pragma unroll
            for (int q = 0; q < NP; ++q) { f.Lt[q] += f.Ltt[q] * f.dt[q]; f.Lz[q] += f.Lzt[q] * f.dt[q]; }
            md_rcp_n<NS, NP>(f.Lt, iLt);
#pragma unroll
            for (int q = 0; q < NP; ++q) c2[q] = f.Lz[q] * iLt[q];
        }
        double kk[NP], uv[NP][2], a[NP][2][3], res[NP][2], e[NP][2], eM[NP][3];
#pragma unroll
        for (int q = 0; q < NP; ++q) kk[q] = (irho[q] > 0.0) ? t[q] * irho[q] : iLt[q];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            uv[q][0] = kk[q] * lat[q][0];
            uv[q][1] = kk[q] * lat[q][1];
            e[q][0] = lat[q][0] * irho[q];
            e[q][1] = lat[q][1] * irho[q];
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double* M = q ? mc.McR : mc.McL;
#pragma unroll
            for (int j = 0; j < 3; ++j) eM[q][j] = e[q][0] * M[j] + e[q][1] * M[3 + j];
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double* M = q ? mc.McR : mc.McL;
            const T* y = q ? yr : yl;
            const double c1v = (iLt[q] - kk[q]) * vis[q], c2v = c2[q] * vis[q], kv = kk[q] * vis[q];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const double w1 = c1v * e[q][r], w3 = -c2v * e[q][r];
#pragma unroll
                for (int j = 0; j < 3; ++j) a[q][r][j] = w1 * eM[q][j] + kv * M[3 * r + j] + w3 * M[6 + j];
                // y[2 k + r] with a wave-uniform k: a chain of selects (no indexed register access)
                const T y0 = y[r], y1 = y[2 + r], y2 = y[4 + r], y3 = y[6 + r];
                const T yk = k == 0 ? y0 : (k == 1 ? y1 : (k == 2 ? y2 : y3));
                res[q][r] = (double)yk - uv[q][r];
            }
        }
        if constexpr (NCAM == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r) acc.add_row(a[0][r], res[0][r], ru);
        } else {
            double Np[6], np[3];
#pragma unroll
            for (int i = 0; i < 6; ++i) Np[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) np[i] = 0.0;
#pragma unroll
            for (int c = 0; c < NCAM; ++c)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const double* ar = a[c][r];
                    Np[0] += ar[0] * ar[0]; Np[1] += ar[0] * ar[1]; Np[2] += ar[0] * ar[2];
                    Np[3] += ar[1] * ar[1]; Np[4] += ar[1] * ar[2]; Np[5] += ar[2] * ar[2];
#pragma unroll
                    for (int j = 0; j < 3; ++j) np[j] += ar[j] * res[c][r];
                }
            acc.add_corner(Np, np, ru);
        }
    }
}

// LDS image of the 63 coefficients + the verdict (64 values of T per lane)
constexpr int SPLIT_NCOEF = 64;

// =================================================================================
// correct() from corner pixels, update divided between the waves of a tile (see the head of this file).
// NR = 2 (KG1: the 256-register fold, two waves per SIMD when the launch has two waves per tile on every SIMD) or 4.
// =================================================================================
template <typename T, int N, int NR, bool KG1>
__global__ void __launch_bounds__(64 * NR, KG1 ? 2 : 1)
correct_pixels_split_kernel(T* __restrict__ recs, int B, int M, const int* __restrict__ ids, const T* __restrict__ left,
                            const T* __restrict__ right, double size, double r_pix, const unsigned char* __restrict__ skip,
                            unsigned char* __restrict__ applied, const short* __restrict__ id2slot, MeasConst mc)
{
    static_assert(sizeof(T) == 4 && NR >= 2, "fp32 records, at least a solver and an updater");
    using L = Lay<N>;
    using RC = Rec<T, N>;
    constexpr int NT = 64 * NR, EPC = RC::EPC, CN = RC::CH_NOM;
    constexpr int E0 = late_start<T, N>();                        // early part = storage [0, E0), late part = [E0, NCOVP)
    constexpr int C_E = E0 / EPC, C_ALL = RC::NCH - CN;           // covariance chunks [0, C_E) early, [C_E, C_ALL) late
    constexpr bool HAS_LATE = C_E < C_ALL;
    const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned tile = blockIdx.x;
    const int b = (int)(tile * 64u + lane);
    const bool live = b < B && !(skip && skip[b < B ? b : 0]);
    const int bc = b < B ? b : (int)(tile * 64u);
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, tile);
    __shared__ MeasLDS tbl;
    // roles >= 1 leave their sums here; then the solver's coefficients take the same memory (the solver has read the sums by then)
    constexpr int PART_BYTES = (NR - 1) * (PixAcc::NVAL + 1) * 64 * 8, COEF_BYTES = SPLIT_NCOEF * 64 * (int)sizeof(T);
    __shared__ double xch_mem[(PART_BYTES > COEF_BYTES ? PART_BYTES : COEF_BYTES) / 8];
    double* part_mem = xch_mem;
    T* coef_mem = reinterpret_cast<T*>(xch_mem);
    struct Meas { int id; T l[8], r[8]; };
    const bool stereo = right != nullptr;
    auto fetch = [&](int i, Meas& mm) __attribute__((always_inline)) {
        const size_t o = (size_t)bc * M + i;
        constexpr int EP = 16 / (int)sizeof(T);
        mm.id = ids[o];
        const u32x4* pl = reinterpret_cast<const u32x4*>(left + o * 8);
        const u32x4* pr = reinterpret_cast<const u32x4*>((stereo ? right : left) + o * 8);
#pragma unroll
        for (int c = 0; c < 8 / EP; ++c) {
            const u32x4 vl = pl[c], vr = pr[c];
            const T* el = reinterpret_cast<const T*>(&vl);
            const T* er = reinterpret_cast<const T*>(&vr);
#pragma unroll
            for (int k = 0; k < EP; ++k) { mm.l[c * EP + k] = el[k]; mm.r[c * EP + k] = er[k]; }
        }
    };
    Meas cur, nxt;
    T pqr[L::NPQR];
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
        order_fence();
        load_chunks<T, N, 0, RC::CH_PQR>(rs, lane, pqr);
        order_fence();
#pragma unroll
        for (int q = 0; q < PI; ++q) { const int i = threadIdx.x + q * NT; if (i < NI) di[i] = vi[q]; }
#pragma unroll
        for (int q = 0; q < PM; ++q) { const int i = threadIdx.x + q * NT; if (i < NM) dm[i] = vm[q]; }
        order_fence();
    }
    meas_barrier();
    double Rd[9];
    PixAcc acc;
    double nfold = 0.0;
    {
        double pd[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pd[i] = (double)pqr[L::OFF_P3 + i];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rd[i] = (double)pqr[L::OFF_R + i];
        double pil[3];
        filter_pil(Rd, mc.P_IL, pil);
        acc.clear();
        const int last = live ? M : 0;
#pragma unroll 1
        for (int i = (int)role; i < last; i += NR) {
            fetch(i + NR < M ? i + NR : M - 1, nxt);
            const bool ok = cur.id >= 0 && cur.id <= FBUS_MAX_MARKER_ID;
            const int slot = ok ? (int)tbl.id2slot[ok ? cur.id : 0] : -1;
            if (slot >= 0) {
                double mk[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) mk[q] = tbl.mkc[slot * MKC_STRIDE + q];
                if constexpr (KG1) {
                    if (stereo) pixel_fold_corners_nz<2, T>(acc, pd, Rd, pil, mc, mk, cur.l, cur.r, size);
                    else pixel_fold_corners_nz<1, T>(acc, pd, Rd, pil, mc, mk, cur.l, cur.l, size);
                } else {
                    if (stereo) pixel_fold_marker<2, T, true>(acc, pd, Rd, pil, mc, mk, cur.l, cur.r, size);
                    else pixel_fold_marker<1, T, true>(acc, pd, Rd, pil, mc, mk, cur.l, cur.l, size);
                }
                nfold += 1.0;
            }
            cur = nxt;
        }
    }
    order_fence();
    if (role != 0) {
        double* part = part_mem + ((role - 1) * (PixAcc::NVAL + 1)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < PixAcc::NVAL; ++i) part[i * 64] = acc.at(i);
        part[PixAcc::NVAL * 64] = nfold;
    }
    if (role == 0) {
        // ---------------- SOLVER: the chunks of P(J, J), requested in front of the exchange
        T P[RC::NCOVP];
        load_cov_chunks<T, N, 0, C_E, SEL_JJ>(rs, lane, P);
        order_fence();
        meas_barrier();                                                // (1) the other roles' sums are in LDS
#pragma unroll
        for (int r = 1; r < NR; ++r) {
            const double* part = part_mem + ((r - 1) * (PixAcc::NVAL + 1)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < PixAcc::NVAL; ++i) acc.at(i) += part[i * 64];
            nfold += part[PixAcc::NVAL * 64];
        }
        const bool apply = live && nfold != 0.0;
        RegCoef<T> cf;
        {
            double Lam[21], bv[6];
            acc.finish(Rd, 1.0 / r_pix, Lam, bv);
            T G[36], Sinv[21], m[6];
            {
                double PJJ[36];
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) PJJ[6 * i + j] = (double)P[pidx<N>(jcol(i), jcol(j))];
                info_solve<T>(Lam, bv, PJJ, G, Sinv, m);
            }
            cf.set(G, Sinv, m);
        }
        order_fence();
        // its own part of the update: the late chunks and the x_c of the late columns (requested here: the 6 x 6 stage's doubles
        // left no registers for them; they arrive while the coefficients go to LDS and the workgroup meets)
        if constexpr (HAS_LATE) {
            load_cov_chunks<T, N, 0, C_E, SEL_XL>(rs, lane, P);
            load_chunks<T, N, CN + C_E, RC::NCH, AUX_NT>(rs, lane, P + E0);
        }
        order_fence();
        {
            // the sums have been read (every lane its own column): their memory takes the coefficients
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            T* co = coef_mem + lane;
#pragma unroll
            for (int i = 0; i < 36; ++i) co[i * 64] = cf.g[i];
#pragma unroll
            for (int i = 0; i < 21; ++i) co[(36 + i) * 64] = cf.s[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) co[(57 + i) * 64] = cf.m_[i];
            co[63 * 64] = apply ? T(1) : T(0);
        }
        meas_barrier();                                                // (2) the coefficients are in LDS
        if (b < B) applied[b] = apply ? 1 : 0;
        if constexpr (HAS_LATE) {
            if (apply) {
                T dx_[1];
                direct_update_part<T, N, E0, L::NP, false>(P, dx_, cf);
                store_chunks<T, N, CN + C_E, RC::NCH, FBUS_X_CORRECT_ST>(rs, lane, P + E0);
            }
        }
        return;
    }
    if (role == 1) {
        // ---------------- UPDATER: the early part and the nominal state, requested while the solver works
        T P[RC::NCOVP], nom[L::NNOM];
        load_chunks<T, N, CN, CN + C_E, AUX_NT>(rs, lane, P);
        load_chunks<T, N, 0, CN>(rs, lane, nom);
        order_fence();
        meas_barrier();                                                // (1)
        meas_barrier();                                                // (2)
        RegCoef<T> cf;
        const T* co = coef_mem + lane;
#pragma unroll
        for (int i = 0; i < 36; ++i) cf.g[i] = co[i * 64];
#pragma unroll
        for (int i = 0; i < 21; ++i) cf.s[i] = co[(36 + i) * 64];
#pragma unroll
        for (int i = 0; i < 6; ++i) cf.m_[i] = co[(57 + i) * 64];
        const bool apply = co[63 * 64] != T(0);
        if (!apply) return;
        T dx[N];
        direct_update_part<T, N, 0, E0, true>(P, dx, cf);
        inject<T, N>(nom, dx);
        store_chunks<T, N, 0, RC::CH_PQ, FBUS_X_CORRECT_ST>(rs, lane, nom);
        store_chunks<T, N, RC::CH_PQR, CN, FBUS_X_CORRECT_ST>(rs, lane, nom + L::NPQR);
        store_chunks<T, N, CN, CN + C_E, FBUS_X_CORRECT_ST>(rs, lane, P);
        return;
    }
    // roles >= 2: folded their share; stay for the two barriers of the workgroup
    meas_barrier();
    meas_barrier();
}

}  // namespace
