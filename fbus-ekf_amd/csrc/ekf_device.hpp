// ekf_device.hpp -- device-side arithmetic of the batched error-state EKF (gfx950).
//
// Mapping: ONE FILTER PER LANE.  A lane keeps its filter's whole record --
// nominal state (19), carried rotation (9), previous-marker id (1) and the
// packed upper triangle of the N x N covariance (171 for N = 18) -- in VGPRs;
// every loop below has compile-time bounds and is fully unrolled so that all
// indexing is static (no scratch).  The structure of the reference's matrices
// is exploited instead of forming dense N x N products:
//
//   predict   F = E_th * E_v * E_p  (three elementary block-row operations), so
//             F P F' is applied as three in-place symmetric congruences
//             (~800 FMA instead of ~23 k for the dense product);
//   correct   the 7 rows of one marker (R is diagonal) are applied as 7
//             sequential scalar updates at one linearisation point, which is
//             algebraically the reference's K = P H' (H P H' + R)^-1 block
//             update; H has non-zeros only in the p and theta columns.
//
// What the arithmetic reproduces (paths relative to the upstream repository):
//   predict : matlab/ImuUpdate.m:36-82 ; C++/src/filter.cpp:533-616
//   correct : matlab/MeasureUpdate.m:37-103 ; C++/src/filter.cpp:622-741
//   helpers : matlab/quaternion_*.m, vector_to_crossmat.m ; C++/include/matrix_math.hpp:26-99
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

// fp32 ImuUpdate on packed instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue in the slot of one scalar instruction
// and do two: profiles/r04_issue_rates.txt) -- bit 0: the two rotation increments of the nominal step (full and half angle) as
// the halves of one pair; bit 1: rows p; bit 2: rows theta; bit 3: the v,v block of rows v.  0 = the scalar forms (A/B builds).
// The stages take the mask as their last template argument: pairs want aligned register pairs, and the kernels written for 256
// registers (two waves per SIMD) or already at 512 take the subset that leaves them without scratch.
#ifndef FBUS_X_PACK
#define FBUS_X_PACK 15
#endif
#ifndef FBUS_X_PACK_2W
#define FBUS_X_PACK_2W 9        // frame2_kernel<float> (256 registers): bit 1 costs it 16-24 bytes of scratch
#endif
#ifndef FBUS_X_PACK_TEAM
#define FBUS_X_PACK_TEAM 9      // the team kernels (ekf_team.hpp): all four bits cost frames_team_kernel 48 bytes of scratch
#endif
#ifndef FBUS_X_PACK_FMEAS_CPP_STEREO
#define FBUS_X_PACK_FMEAS_CPP_STEREO 0
#endif
#ifndef FBUS_X_PACK_FMEAS
#define FBUS_X_PACK_FMEAS 1     // frame_meas_kernel (512 registers around the fp64 fold): the nominal step only -- bits 1-3 add 12-76 bytes of scratch; bit 0 keeps its nominal arithmetic the one of predict_n (the fused frame and predict_n + update then agree to the single-step gate: tests/test_frame_meas_gpu.py)
#endif

namespace fbus {

enum { DIALECT_MATLAB = 0, DIALECT_CPP = 1 };
enum { MODE_NEAREST = 0, MODE_STACKED = 1, MODE_MEAS_VEC = 0x100 /* kernel-internal flag, see MarkerGroup::fetch */ };
enum { COV_SIMPLE = 0, COV_JOSEPH = 1 };

constexpr int MK_STRIDE = 8;    // per marker-map slot: pos3 quat4 pad1

// ---- record layout (elements of T inside one filter's record) ----------------
// Order inside a record (NOT the API order): [p3 q4 R9 | v3 ba3 bg3 g3 | P packed | prev].
// The first 16 elements are all that building the measurement rows needs, the next 12 are
// touched only by the kinematics / the final injection, so a kernel can bring the groups
// in when it needs them instead of holding them in registers across the covariance work;
// ba, bg, g are never written by predict and their chunks are not stored back.
template <int N>
struct Lay {
    static constexpr int NP = N * (N + 1) / 2;
    static constexpr int OFF_P3 = 0, OFF_Q = 3, OFF_R = 7, OFF_V = 16, OFF_BA = 19, OFF_BG = 22, OFF_G = 25;
    static constexpr int NNOM = 28;                 // nominal + carried rotation
    static constexpr int NPQR = 16;                 // p, q, R
    static constexpr int NKIN = 19;                 // p, q, R, v: everything predict writes
    static constexpr int OFF_COV = 28, OFF_PREV = OFF_COV + NP;
    static constexpr int NREC = OFF_PREV + 1;
};

// number of leading covariance elements predict may change (the rest is invariant under ImuUpdate; see the
// storage order below)
template <int N>
__host__ __device__ constexpr int cov_variant_count() { return N == 18 ? 132 : N * (N + 1) / 2; }

template <typename T, int N>
struct Rec {
    static constexpr int EPC = 16 / (int)sizeof(T);                       // elements per 16-byte chunk
    static constexpr int NRECP = (Lay<N>::NREC + EPC - 1) / EPC * EPC;    // padded record length
    static constexpr int NCH = NRECP / EPC;                               // chunks per record
    static constexpr int CH_NOM = Lay<N>::NNOM / EPC;                     // chunks [0, CH_NOM): nominal + R
    static constexpr int CH_PQR = Lay<N>::NPQR / EPC;                     // chunks [0, CH_PQR): p, q, R
    static constexpr int CH_KIN = (Lay<N>::NKIN + EPC - 1) / EPC;         // chunks predict must store back
    static constexpr int CH_PQ = (7 + EPC - 1) / EPC;                     // chunks holding p and q
    static constexpr int NCOVP = NRECP - Lay<N>::NNOM;                    // P + prev + padding
    // chunks [CH_NOM, CH_VAR_END) hold every covariance element predict can change (all of them unless N = 18)
    static constexpr int CH_VAR_END = (cov_variant_count<N>() % EPC == 0 && N == 18)
                                          ? CH_NOM + cov_variant_count<N>() / EPC : NCH;
    static_assert(Lay<N>::NNOM % EPC == 0 && Lay<N>::NPQR % EPC == 0, "groups must end on chunk boundaries");
};

template <typename T> struct Vec16;
template <> struct Vec16<float>  { using type = float4;  };
template <> struct Vec16<double> { using type = double2; };

// Packed storage order of the upper triangle.
//  odd N  : plain row-major upper triangle.
//  even N : "pair-aligned rows" -- every row is stored from an EVEN column on, so that (P(i,c), P(i,c+1)),
//           c even, are neighbours at an even offset and one v_pk_fma_f32 updates both: even rows start at
//           their diagonal, odd rows at column i+1, the diagonals of the odd rows are collected behind the rows.
//  N = 18 : the same idea, plus: the elements ImuUpdate never changes come LAST.  F's rows for ba, bg, g are
//           identity rows (ImuUpdate.m:63-69), so P(i,j) with 9 <= i < j and the gravity diagonals are
//           invariant under predict (only the ba/bg diagonals get + Q): 39 of the 171 elements.  Order:
//             [rows 0..8 pair-aligned (122)] [diagonals of rows 1,3,5,7 and 9..14 (10)]   <- 132 = 33 chunks, written by predict
//             [rows 9..16 off-diagonal pairs, (16,16)] [(10,11) (12,13) (14,15) (15,15) (17,17)] <- 39, never written by predict
//           predict therefore stores 33 of the 43 covariance chunks (-160 B per step, -10 % traffic).
//  Always exactly N(N+1)/2 elements, no duplicates.
template <int N>
__host__ __device__ constexpr int pair_cols(int i) { return (i & 1) ? (N - i - 1) : (N - i); }
template <int N>
__host__ __device__ constexpr int row_base(int i)
{
    int s = 0;
    for (int r = 0; r < i; ++r) s += pair_cols<N>(r);
    return s;
}
__host__ __device__ constexpr int pidx18(int i, int j)          // i <= j, N = 18
{
    if (i <= 8) return ((i & 1) && j == i) ? 122 + (i - 1) / 2 : row_base<18>(i) + (j - i - (i & 1));
    if (j == i) return (i <= 14) ? 126 + (i - 9) : (i == 15 ? 169 : (i == 16 ? 164 : 170));
    switch (i) {
        case 9:  return 132 + (j - 10);
        case 10: return (j == 11) ? 166 : 140 + (j - 12);
        case 11: return 146 + (j - 12);
        case 12: return (j == 13) ? 167 : 152 + (j - 14);
        case 13: return 156 + (j - 14);
        case 14: return (j == 15) ? 168 : 160 + (j - 16);
        case 15: return 162 + (j - 16);
        default: return 165;                                    // (16,17)
    }
}
template <int N>
__host__ __device__ constexpr int pidx_ord(int i, int j)        // i <= j
{
    if (N == 18) return pidx18(i, j);
    if (N % 2) return i * N - (i * (i - 1)) / 2 + (j - i);
    if ((i & 1) && j == i) return (N * (N + 1) / 2 - N / 2) + (i - 1) / 2;
    return row_base<N>(i) + (j - i - (i & 1));
}
template <int N>
__host__ __device__ constexpr int pidx(int i, int j) { return (i <= j) ? pidx_ord<N>(i, j) : pidx_ord<N>(j, i); }
// compile-time proof that the N = 18 order is a bijection onto 0..170 with the variant elements first
__host__ __device__ constexpr bool pidx18_ok()
{
    bool seen[171] = {};
    for (int i = 0; i < 18; ++i)
        for (int j = i; j < 18; ++j) {
            const int k = pidx18(i, j);
            if (k < 0 || k >= 171 || seen[k]) return false;
            seen[k] = true;
            const bool invariant = (i >= 9) && !(i == j && i <= 14);
            if (invariant != (k >= 132)) return false;
        }
    return true;
}
static_assert(pidx18_ok(), "N = 18 covariance order must be a bijection with the predict-invariant elements last");
// (P(r,c), P(r,c+1)) with r <= c is an aligned storage pair
template <int N>
__host__ __device__ constexpr bool is_pair(int r, int c)
{
    return (N % 2 == 0) && c + 1 < N && r <= c && (pidx<N>(r, c) % 2 == 0) && (pidx<N>(r, c + 1) == pidx<N>(r, c) + 1);
}
template <typename T, int N> struct PackedMath { static constexpr bool on = false; };
template <int N> struct PackedMath<float, N> { static constexpr bool on = (N % 2 == 0); };

// ---- compile-time loop (the body sees its index as a constant expression) ---------
template <int I, int E, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}
// hook called by the LAST rank-1 pass of a correct step when row I of the covariance is final (the kernels use it to
// store the finished chunks while the pass is still running); the default does nothing
struct NoRowHook {
    template <int I> __device__ __forceinline__ void row_done() const {}
};

// ---- scalar helpers -------------------------------------------------------------
__device__ __forceinline__ void fb_sincos(float x, float& s, float& c) { sincosf(x, &s, &c); }
// sin/cos of the rotation increments: of x and of x/2 (x = |w| dt / 2).  They are a few 1e-3 rad in this filter, where
// the library sincosf spends ~100 instructions (and several branches) per call on an argument reduction it does not
// need: for |x| <= 0.5 the Taylor polynomials through x^7 / x^8 truncate at 1.1e-8 / 2.7e-10 relative, so the error is
// the ~0.5 ulp of the Horner evaluation (5.9e-8 / 4.6e-8 measured over [-0.5, 0.5]); larger arguments take the
// library path, both angles behind one branch.
__device__ __forceinline__ void fb_sincos_poly(float x, float& s, float& c)
{
    const float x2 = x * x;
    s = x + x * (x2 * (-1.6666667163e-1f + x2 * (8.3333337680e-3f + x2 * -1.9841270114e-4f)));
    c = 1.0f + x2 * (-0.5f + x2 * (4.1666667908e-2f + x2 * (-1.3888889225e-3f + x2 * 2.4801587642e-5f)));
}
__device__ __forceinline__ void fb_sincos_x_halfx(float x, float& s, float& c, float& sh, float& ch)
{
    if (__builtin_fabsf(x) <= 0.5f) {
        fb_sincos_poly(x, s, c);
        fb_sincos_poly(0.5f * x, sh, ch);
    } else {
        sincosf(x, &s, &c);
        sincosf(0.5f * x, &sh, &ch);
    }
}
__device__ __forceinline__ void fb_sincos_x_halfx(double x, double& s, double& c, double& sh, double& ch)
{
    sincos(x, &s, &c);
    sincos(0.5 * x, &sh, &ch);
}
__device__ __forceinline__ void fb_sincos(double x, double& s, double& c) { sincos(x, &s, &c); }
__device__ __forceinline__ float fb_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double fb_sqrt(double x) { return sqrt(x); }
// 1 / sqrt(x), x > 0.  fp32: the 1-ulp hardware estimate + one Newton step (4 instructions; sqrtf followed by an IEEE division is ~25),
// fp64: the reference's own two operations.
// v_rsq_f32 does not handle denormal inputs (the estimate of 0 < x < 1.18e-38 is +inf and the Newton step turns it into NaN): callers
// that can see such an x -- the squared norm of a rate or of a rotation increment below 1.1e-19 -- test x >= FB_RSQRT_MIN, not x > 0
constexpr float FB_RSQRT_MIN = 1.17549435e-38f;
__device__ __forceinline__ float fb_rsqrt(float x) { const float r = __builtin_amdgcn_rsqf(x); return r * (1.5f - 0.5f * x * r * r); }
__device__ __forceinline__ double fb_rsqrt(double x) { return 1.0 / sqrt(x); }
// 1 / x.  fp32: hardware estimate + one Newton step (3 instructions instead of the ~10 of the IEEE sequence); fp64: a division.
__device__ __forceinline__ float fb_rcp1(float x) { const float r = __builtin_amdgcn_rcpf(x); return r * (2.0f - x * r); }
__device__ __forceinline__ double fb_rcp1(double x) { return 1.0 / x; }
__device__ __forceinline__ float fb_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double fb_abs(double x) { return fabs(x); }

template <typename T>
__device__ __forceinline__ void quat_mul(const T* p, const T* q, T* o)
{   // quaternion_add.m:22-28
    o[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
    o[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
    o[2] = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
    o[3] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
}

template <typename T>
__device__ __forceinline__ void quat_to_rotmat_m(const T* q, T* R)
{   // quaternion_to_rotmat.m:22-33
    const T w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z);           R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);           R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);           R[7] = 2 * (y * z + w * x);           R[8] = w * w - x * x - y * y + z * z;
}

template <typename T>
__device__ __forceinline__ void quat_to_rotmat_e(const T* q, T* R)
{   // Eigen Quaternion::toRotationMatrix (filter.cpp:542,562,564)
    const T w = q[0], x = q[1], y = q[2], z = q[3];
    const T tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const T twx = tx * w, twy = ty * w, twz = tz * w;
    const T txx = tx * x, txy = ty * x, txz = tz * x;
    const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

template <typename T>
__device__ __forceinline__ void quat_normalize(T* q)
{
    const T inv = fb_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}

// ---- the same helpers on register pairs: .x = the full rotation increment (T), .y = the half one (H) ------------------
// Every half of a packed operation is the scalar operation (same order, same rounding); what differs from the scalar
// helpers is which products the compiler contracts into FMAs.
using f32x2 = float __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fb_sincos_x_halfx_pk(float x, f32x2& s, f32x2& c)
{
    if (__builtin_fabsf(x) <= 0.5f) {
        const f32x2 xp = { x, 0.5f * x };
        const f32x2 x2 = xp * xp;
        s = xp + xp * (x2 * (-1.6666667163e-1f + x2 * (8.3333337680e-3f + x2 * -1.9841270114e-4f)));
        c = 1.0f + x2 * (-0.5f + x2 * (4.1666667908e-2f + x2 * (-1.3888889225e-3f + x2 * 2.4801587642e-5f)));
    } else {
        float s0, c0, s1, c1;
        sincosf(x, &s0, &c0);
        sincosf(0.5f * x, &s1, &c1);
        s = f32x2{ s0, s1 }; c = f32x2{ c0, c1 };
    }
}
__device__ __forceinline__ void quat_mul_pk(const float* p, const f32x2* q, f32x2* o)
{
    o[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
    o[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
    o[2] = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
    o[3] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
}
__device__ __forceinline__ void quat_to_rotmat_m_pk(const f32x2* q, f32x2* R)
{
    const f32x2 w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z);           R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);           R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);           R[7] = 2 * (y * z + w * x);           R[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ void quat_to_rotmat_e_pk(const f32x2* q, f32x2* R)
{
    const f32x2 w = q[0], x = q[1], y = q[2], z = q[3];
    const f32x2 tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const f32x2 twx = tx * w, twy = ty * w, twz = tz * w;
    const f32x2 txx = tx * x, txy = ty * x, txz = tz * x;
    const f32x2 tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void quat_normalize_pk(f32x2* q)
{
    const f32x2 n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const f32x2 inv = { fb_rsqrt(n2.x), fb_rsqrt(n2.y) };
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}

// ---- constants handed to the kernels by value (wave-uniform -> SGPRs) -------------
template <typename T>
struct DevConst {
    T qd[4];                // process noise on v, theta, ba, bg diagonals
    T r_pos, r_quat;
    T R_IL[9], P_IL[3], Q_IL[4];
    T switch_thres;
    int cov_form;
    T CL[16];               // Lq(Q_IL) * L2 (MeasureUpdate.m:39-44,74), row-major 4x4
    const T* mk;            // [n_slots][MK_STRIDE]: marker position (3) and quaternion (4, wxyz)
    const short* id2slot;   // [FBUS_MAX_MARKER_ID + 1], -1 = not in the map
};

// Accumulates coef * (P(r,c), P(r,c+1)) for an even column c of the symmetric packed covariance: one v_pk_fma_f32 where
// the storage holds the two elements as an aligned pair (r <= c, pair-aligned rows), two scalar FMAs otherwise (r > c
// reads the transposed elements).  All indices are compile-time after unrolling, so the flags fold away.
template <int N>
struct PairAcc {
    f32x2 v = { 0.f, 0.f };
    float lo = 0.f, hi = 0.f;
    bool vset = false, sset = false;
    __device__ __forceinline__ void add(const float* P, float coef, int r, int c)
    {
        if (is_pair<N>(r, c)) {
            const int o = pidx<N>(r, c);
            const f32x2 t = f32x2{ P[o], P[o + 1] };
            v = vset ? v + coef * t : coef * t;
            vset = true;
        } else {
            const float t0 = P[pidx<N>(r, c)], t1 = P[pidx<N>(r, c + 1)];
            lo = sset ? lo + coef * t0 : coef * t0;
            hi = sset ? hi + coef * t1 : coef * t1;
            sset = true;
        }
    }
    __device__ __forceinline__ f32x2 get() const
    {
        return vset ? (sset ? v + f32x2{ lo, hi } : v) : f32x2{ lo, hi };
    }
};

// (P(r,c), P(r,c+1)) as one value.  Meant for aligned storage pairs (is_pair): there it is a register pair as it stands.
template <int N>
__device__ __forceinline__ f32x2 ld_pair(const float* P, int r, int c) { return f32x2{ P[pidx<N>(r, c)], P[pidx<N>(r, c + 1)] }; }
template <int N>
__device__ __forceinline__ void st_pair(float* P, int r, int c, f32x2 v) { P[pidx<N>(r, c)] = v.x; P[pidx<N>(r, c + 1)] = v.y; }
// rows (a, b) of a 3 x 3 coefficient block, column by column: what multiplies ONE covariance element on its way into TWO
// neighbouring outputs
__device__ __forceinline__ void coef_rows(const float* M, int a, int b, f32x2 (&o)[3])
{
#pragma unroll
    for (int m = 0; m < 3; ++m) o[m] = f32x2{ M[3 * a + m], M[3 * b + m] };
}

// ================================================================================
// predict
// ================================================================================
// ImuUpdate is written as four stages so that a kernel can chase its own loads and store results as soon as they
// are final (the per-call kernel does; predict_n and the fused frame kernel just run them back to back):
//   predict_nominal   kinematics + the coefficient blocks of F            needs the nominal state and the IMU sample
//   cov_stage_p       rows p of  F P F'                                    needs covariance rows p, v
//   cov_stage_v       rows v                                               needs rows theta, ba, g (and P(bg,g))
//   cov_stage_th      rows theta, + Fi Q Fi' on the theta/ba/bg diagonals  needs rows bg
// F = E_theta * E_v * E_p (three elementary block-row operations: p += dt v ; v += A th + Bm ba + dt g ;
// th = Th th - dt bg, each reading not-yet-updated rows), so F P F' is three in-place symmetric congruences;
// the column part of E_v and E_theta on rows p (rows v) only reads rows p (rows v), which is why rows p are
// final after cov_stage_p and rows v after cov_stage_v.
template <typename T>
struct PredictCoef {
    T A[9];     // F(v,theta)  = -R [a]x dt      ImuUpdate.m:65 ; filter.cpp:600
    T Bm[9];    // F(v,ba)     = -R dt           ImuUpdate.m:66 ; filter.cpp:601
    T Th[9];    // F(theta,theta)                ImuUpdate.m:68 ; filter.cpp:603
    T dt;
};

// lowest storage index used by any covariance element of rows >= r: everything below it is final once rows < r are
template <int N>
__host__ __device__ constexpr int cov_final_before_row(int r)
{
    int m = N * (N + 1) / 2;
    for (int i = r; i < N; ++i)
        for (int j = i; j < N; ++j)
            if (pidx<N>(i, j) < m) m = pidx<N>(i, j);
    return m;
}

#define PS(i, j) P[pidx<N>((i), (j))]
// rows p:  E_p, then the v- and theta-columns of rows p
template <typename T, int N, int PK = FBUS_X_PACK>
__device__ __forceinline__ void cov_stage_p(T* P, const PredictCoef<T>& k)
{
    constexpr bool G = (N == 18);
    const T dt = k.dt;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j)
            PS(i, j) += dt * (PS(j, 3 + i) + PS(i, 3 + j)) + dt * dt * PS(3 + i, 3 + j);
    if constexpr (PackedMath<T, N>::on && (PK & 2)) {
        // the same operations, two neighbouring columns of a row at a time: the row operation with dt on both halves, the column
        // operations with rows (1, 2) of A, Bm / rows (0, 1) of Theta on the halves and the covariance element on both
        f32x2 A12[3], B12[3], T01[3];
        coef_rows(k.A, 1, 2, A12); coef_rows(k.Bm, 1, 2, B12); coef_rows(k.Th, 0, 1, T01);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            PS(i, 3) += dt * PS(3 + i, 3);
            if (is_pair<N>(i, 4) && is_pair<N>(3 + i, 4)) st_pair<N>(P, i, 4, ld_pair<N>(P, i, 4) + dt * ld_pair<N>(P, 3 + i, 4));
            else { PS(i, 4) += dt * PS(3 + i, 4); PS(i, 5) += dt * PS(3 + i, 5); }
#pragma unroll
            for (int c = 6; c < N; c += 2) st_pair<N>(P, i, c, ld_pair<N>(P, i, c) + dt * ld_pair<N>(P, 3 + i, c));
        }
        // E_v, column v of rows p: P(p,v) += P(p,theta) A' + P(p,ba) Bm' + dt P(p,g)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            PS(c, 3) += k.A[0] * PS(c, 6) + k.A[1] * PS(c, 7) + k.A[2] * PS(c, 8)
                      + k.Bm[0] * PS(c, 9) + k.Bm[1] * PS(c, 10) + k.Bm[2] * PS(c, 11) + (G ? dt * PS(c, 15) : T(0));
            f32x2 u = A12[0] * PS(c, 6) + A12[1] * PS(c, 7) + A12[2] * PS(c, 8) + B12[0] * PS(c, 9) + B12[1] * PS(c, 10) + B12[2] * PS(c, 11);
            if (G) u += dt * ld_pair<N>(P, c, 16);
            st_pair<N>(P, c, 4, ld_pair<N>(P, c, 4) + u);
        }
        // E_theta, column theta of rows p: P(p,theta) = P(p,theta) Th' - dt P(p,bg)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const T o0 = PS(c, 6), o1 = PS(c, 7), o2 = PS(c, 8);
            st_pair<N>(P, c, 6, T01[0] * o0 + T01[1] * o1 + T01[2] * o2 - dt * ld_pair<N>(P, c, 12));
            PS(c, 8) = k.Th[6] * o0 + k.Th[7] * o1 + k.Th[8] * o2 - dt * PS(c, 14);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 3; c < N; ++c) PS(i, c) += dt * PS(3 + i, c);
    // E_v, column v of rows p: P(p,v) += P(p,theta) A' + P(p,ba) Bm' + dt P(p,g)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            PS(c, 3 + i) += k.A[3 * i] * PS(c, 6) + k.A[3 * i + 1] * PS(c, 7) + k.A[3 * i + 2] * PS(c, 8)
                          + k.Bm[3 * i] * PS(c, 9) + k.Bm[3 * i + 1] * PS(c, 10) + k.Bm[3 * i + 2] * PS(c, 11)
                          + (G ? dt * PS(c, 15 + i) : T(0));
    // E_theta, column theta of rows p: P(p,theta) = P(p,theta) Th' - dt P(p,bg)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const T o0 = PS(c, 6), o1 = PS(c, 7), o2 = PS(c, 8);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            PS(c, 6 + i) = k.Th[3 * i] * o0 + k.Th[3 * i + 1] * o1 + k.Th[3 * i + 2] * o2 - dt * PS(c, 12 + i);
    }
}

// rows v:  E_v (row part and the symmetric v,v block), then the theta-column of rows v, + Q on the v diagonal
template <typename T, int N, int PK = FBUS_X_PACK>
__device__ __forceinline__ void cov_stage_v(T* P, const PredictCoef<T>& k, const T* qd)
{
    constexpr bool G = (N == 18);
    constexpr int NC = N - 6;                       // columns theta .. end
    const T dt = k.dt;
    if constexpr (sizeof(T) == 4) {
        // U(i, c) = A_i . P(theta, c) + Bm_i . P(ba, c) + dt P(g_i, c), c = 6 .. N-1; row by row (r3): U(i, :), then row i of
        // D = (P(v,theta) + U_theta/2) A' + (P(v,ba) + U_ba/2) Bm' + dt (P(v,g) + U_g/2) from the PRE-update row i, then
        // P(v_i, theta..) += U(i, :) -- 12 products live at a time instead of 36 (the same operations in the same order per element)
        T D[9];
        f32x2 A12[3], B12[3];
        coef_rows(k.A, 1, 2, A12); coef_rows(k.Bm, 1, 2, B12);
        static_for<0, 3>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            T U[NC];
            if constexpr (PackedMath<T, N>::on) {
                // the sums on aligned column pairs (6+c, 7+c): 26 of the 42 pair-terms per row are one v_pk_fma_f32
#pragma unroll
                for (int c = 0; c < NC; c += 2) {
                    PairAcc<N> acc;
                    acc.add(P, k.A[3 * i], 6, 6 + c); acc.add(P, k.A[3 * i + 1], 7, 6 + c); acc.add(P, k.A[3 * i + 2], 8, 6 + c);
                    acc.add(P, k.Bm[3 * i], 9, 6 + c); acc.add(P, k.Bm[3 * i + 1], 10, 6 + c); acc.add(P, k.Bm[3 * i + 2], 11, 6 + c);
                    if (G) acc.add(P, dt, 15 + i, 6 + c);
                    const f32x2 r = acc.get();
                    U[c] = r.x; U[c + 1] = r.y;
                }
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    U[c] = k.A[3 * i] * PS(6, 6 + c) + k.A[3 * i + 1] * PS(7, 6 + c) + k.A[3 * i + 2] * PS(8, 6 + c);
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    U[c] += k.Bm[3 * i] * PS(9, 6 + c) + k.Bm[3 * i + 1] * PS(10, 6 + c) + k.Bm[3 * i + 2] * PS(11, 6 + c);
                if (G) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) U[c] += dt * PS(15 + i, 6 + c);
                }
            }
            if constexpr (PackedMath<T, N>::on && (PK & 8)) {
                // t = P(v_i, .) + U(i, .)/2 on the column pairs of the row; D(i, 1), D(i, 2) as the halves of one value (rows (1, 2) of
                // A and Bm on the halves, t on both), D(i, 0) alone -- the same products in the same order per element
                T t[NC];
#pragma unroll
                for (int c = 0; c < NC; c += 2) {
                    if (c == 6) continue;                                     // columns bg take no part
                    if (is_pair<N>(3 + i, 6 + c)) {
                        const f32x2 r = ld_pair<N>(P, 3 + i, 6 + c) + T(0.5) * f32x2{ U[c], U[c + 1] };
                        t[c] = r.x; t[c + 1] = r.y;
                    } else {
                        t[c] = PS(3 + i, 6 + c) + T(0.5) * U[c]; t[c + 1] = PS(3 + i, 7 + c) + T(0.5) * U[c + 1];
                    }
                }
                T d0 = T(0);
                f32x2 d12 = { T(0), T(0) };
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    d0 += t[m] * k.A[m];
                    d0 += t[3 + m] * k.Bm[m];
                    d12 += t[m] * A12[m];
                    d12 += t[3 + m] * B12[m];
                }
                if (G) { d0 += dt * t[9]; d12 += dt * f32x2{ t[10], t[11] }; }
                D[3 * i] = d0; D[3 * i + 1] = d12.x; D[3 * i + 2] = d12.y;
            } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                T acc = T(0);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    acc += (PS(3 + i, 6 + m) + T(0.5) * U[m]) * k.A[3 * j + m];
                    acc += (PS(3 + i, 9 + m) + T(0.5) * U[3 + m]) * k.Bm[3 * j + m];
                }
                if (G) acc += dt * (PS(3 + i, 15 + j) + T(0.5) * U[9 + j]);
                D[3 * i + j] = acc;
            }
            }
            if constexpr (PackedMath<T, N>::on) {
#pragma unroll
                for (int c = 0; c < NC; c += 2) {
                    if (is_pair<N>(3 + i, 6 + c)) {
                        const int o = pidx<N>(3 + i, 6 + c);
                        const f32x2 r = f32x2{ P[o], P[o + 1] } + f32x2{ U[c], U[c + 1] };
                        P[o] = r.x; P[o + 1] = r.y;
                    } else {
                        PS(3 + i, 6 + c) += U[c]; PS(3 + i, 7 + c) += U[c + 1];
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) PS(3 + i, 6 + c) += U[c];
            }
        });
        // v,v block: P(v,v) += D + D'
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) PS(3 + i, 3 + j) += D[3 * i + j] + D[3 * j + i];
    } else {
        // fp64 (at the 512-register limit): the block form -- all of U, then D, then the row updates; the row-by-row order above
        // costs this instantiation 20-52 bytes of scratch
        // U(i, c) = A_i . P(theta, c) + Bm_i . P(ba, c) + dt P(g_i, c), c = 6 .. N-1, accumulated source by source in
        // the order the rows arrive (theta rows, ba rows, then the g/bg bits)
        T U[3 * NC];
        if constexpr (PackedMath<T, N>::on) {
            // the same sums on aligned column pairs (6+c, 7+c): 26 of the 42 pair-terms per row are one v_pk_fma_f32
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < NC; c += 2) {
                    PairAcc<N> acc;
                    acc.add(P, k.A[3 * i], 6, 6 + c); acc.add(P, k.A[3 * i + 1], 7, 6 + c); acc.add(P, k.A[3 * i + 2], 8, 6 + c);
                    acc.add(P, k.Bm[3 * i], 9, 6 + c); acc.add(P, k.Bm[3 * i + 1], 10, 6 + c); acc.add(P, k.Bm[3 * i + 2], 11, 6 + c);
                    if (G) acc.add(P, dt, 15 + i, 6 + c);
                    const f32x2 r = acc.get();
                    U[NC * i + c] = r.x; U[NC * i + c + 1] = r.y;
                }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    U[NC * i + c] = k.A[3 * i] * PS(6, 6 + c) + k.A[3 * i + 1] * PS(7, 6 + c) + k.A[3 * i + 2] * PS(8, 6 + c);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    U[NC * i + c] += k.Bm[3 * i] * PS(9, 6 + c) + k.Bm[3 * i + 1] * PS(10, 6 + c) + k.Bm[3 * i + 2] * PS(11, 6 + c);
            if (G) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int c = 0; c < NC; ++c) U[NC * i + c] += dt * PS(15 + i, 6 + c);
            }
        }
        // v,v block: P(v,v) += D + D',  D = (P(v,theta) + U_theta/2) A' + (P(v,ba) + U_ba/2) Bm' + dt (P(v,g) + U_g/2)
        T D[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                T acc = T(0);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    acc += (PS(3 + i, 6 + m) + T(0.5) * U[NC * i + m]) * k.A[3 * j + m];
                    acc += (PS(3 + i, 9 + m) + T(0.5) * U[NC * i + 3 + m]) * k.Bm[3 * j + m];
                }
                if (G) acc += dt * (PS(3 + i, 15 + j) + T(0.5) * U[NC * i + 9 + j]);
                D[3 * i + j] = acc;
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) PS(3 + i, 3 + j) += D[3 * i + j] + D[3 * j + i];
        if constexpr (PackedMath<T, N>::on) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < NC; c += 2) {
                    if (is_pair<N>(3 + i, 6 + c)) {
                        const int o = pidx<N>(3 + i, 6 + c);
                        const f32x2 r = f32x2{ P[o], P[o + 1] } + f32x2{ U[NC * i + c], U[NC * i + c + 1] };
                        P[o] = r.x; P[o + 1] = r.y;
                    } else {
                        PS(3 + i, 6 + c) += U[NC * i + c]; PS(3 + i, 7 + c) += U[NC * i + c + 1];
                    }
                }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < NC; ++c) PS(3 + i, 6 + c) += U[NC * i + c];
        }
    }
    // E_theta, column theta of rows v: P(v,theta) = P(v,theta) Th' - dt P(v,bg)
    if constexpr (PackedMath<T, N>::on && (PK & 8)) {
        f32x2 T01[3];
        coef_rows(k.Th, 0, 1, T01);
#pragma unroll
        for (int c = 3; c < 6; ++c) {
            const T o0 = PS(c, 6), o1 = PS(c, 7), o2 = PS(c, 8);
            st_pair<N>(P, c, 6, T01[0] * o0 + T01[1] * o1 + T01[2] * o2 - dt * ld_pair<N>(P, c, 12));
            PS(c, 8) = k.Th[6] * o0 + k.Th[7] * o1 + k.Th[8] * o2 - dt * PS(c, 14);
        }
    } else {
#pragma unroll
    for (int c = 3; c < 6; ++c) {
        const T o0 = PS(c, 6), o1 = PS(c, 7), o2 = PS(c, 8);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            PS(c, 6 + i) = k.Th[3 * i] * o0 + k.Th[3 * i + 1] * o1 + k.Th[3 * i + 2] * o2 - dt * PS(c, 12 + i);
    }
    }
#pragma unroll
    for (int i = 3; i < 6; ++i) PS(i, i) += qd[0];
}

// columns (c, c + 1) of rows theta outside the theta and bg blocks that all three rows hold as an aligned pair
template <int N>
__host__ __device__ constexpr bool th_pair_cols(int c)
{
    return c >= 9 && (c % 2 == 0) && c + 1 < N && !(c >= 12 && c < 15) && !(c + 1 >= 12 && c + 1 < 15)
        && is_pair<N>(6, c) && is_pair<N>(7, c) && is_pair<N>(8, c);
}
// rows theta:  E_theta on the remaining columns, + Fi Q Fi' (not scaled by dt; ImuUpdate.m:70-73 ; filter.cpp:609-610)
template <typename T, int N, int PK = FBUS_X_PACK>
__device__ __forceinline__ void cov_stage_th(T* P, const PredictCoef<T>& k, const T* qd)
{
    const T dt = k.dt;
    const T (&Th)[9] = k.Th;
    T Xn[9], Gm[9];
    if constexpr (PackedMath<T, N>::on && (PK & 4)) {
        // columns ba and g of rows theta, two neighbouring columns at a time where the storage holds them as a pair in all the
        // rows read (columns (10,11) of rows theta; (16,17) of rows theta and bg)
#pragma unroll
        for (int c = 9; c < N; ++c) {
            if (c >= 12 && c < 15) continue;
            if ((c & 1) && th_pair_cols<N>(c - 1)) continue;                 // done with its left neighbour
            if (th_pair_cols<N>(c)) {
                const f32x2 o0 = ld_pair<N>(P, 6, c), o1 = ld_pair<N>(P, 7, c), o2 = ld_pair<N>(P, 8, c);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    f32x2 r = Th[3 * i] * o0 + Th[3 * i + 1] * o1 + Th[3 * i + 2] * o2;
                    if (is_pair<N>(12 + i, c)) r -= dt * ld_pair<N>(P, 12 + i, c);
                    else { r.x -= dt * PS(12 + i, c); r.y -= dt * PS(12 + i, c + 1); }
                    st_pair<N>(P, 6 + i, c, r);
                }
            } else {
                const T o0 = PS(6, c), o1 = PS(7, c), o2 = PS(8, c);
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    PS(6 + i, c) = Th[3 * i] * o0 + Th[3 * i + 1] * o1 + Th[3 * i + 2] * o2 - dt * PS(12 + i, c);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // Theta P(theta,theta) and Theta P(theta,bg) - dt P(bg,bg): rows (0, 1) of Theta on the halves, row 2 alone
        f32x2 T01[3];
        coef_rows(Th, 0, 1, T01);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32x2 g01 = T01[0] * PS(6, 6 + j) + T01[1] * PS(7, 6 + j) + T01[2] * PS(8, 6 + j);
            Gm[j] = g01.x; Gm[3 + j] = g01.y;
            Gm[6 + j] = Th[6] * PS(6, 6 + j) + Th[7] * PS(7, 6 + j) + Th[8] * PS(8, 6 + j);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32x2 x01 = T01[0] * PS(6, 12 + j) + T01[1] * PS(7, 12 + j) + T01[2] * PS(8, 12 + j);
            Xn[j] = x01.x - dt * PS(12, 12 + j); Xn[3 + j] = x01.y - dt * PS(13, 12 + j);
            Xn[6 + j] = Th[6] * PS(6, 12 + j) + Th[7] * PS(7, 12 + j) + Th[8] * PS(8, 12 + j) - dt * PS(14, 12 + j);
        }
    } else {
#pragma unroll
    for (int c = 9; c < N; ++c) {
        if (c >= 12 && c < 15) continue;
        const T o0 = PS(6, c), o1 = PS(7, c), o2 = PS(8, c);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            PS(6 + i, c) = Th[3 * i] * o0 + Th[3 * i + 1] * o1 + Th[3 * i + 2] * o2 - dt * PS(12 + i, c);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Gm[3 * i + j] = Th[3 * i] * PS(6, 6 + j) + Th[3 * i + 1] * PS(7, 6 + j) + Th[3 * i + 2] * PS(8, 6 + j);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Xn[3 * i + j] = Th[3 * i] * PS(6, 12 + j) + Th[3 * i + 1] * PS(7, 12 + j) + Th[3 * i + 2] * PS(8, 12 + j)
                          - dt * PS(12 + i, 12 + j);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            const T t1 = Gm[3 * i] * Th[3 * j] + Gm[3 * i + 1] * Th[3 * j + 1] + Gm[3 * i + 2] * Th[3 * j + 2];
            PS(6 + i, 6 + j) = t1 - dt * (Xn[3 * i + j] + Xn[3 * j + i]) - dt * dt * PS(12 + i, 12 + j);
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) PS(6 + i, 12 + j) = Xn[3 * i + j];
#pragma unroll
    for (int i = 6; i < 15; ++i) PS(i, i) += qd[(i - 3) / 3];
}
#undef PS

// Nominal part of one ImuUpdate: nom = the 28 nominal + rotation elements (record order).  Fills the coefficient
// blocks from the PRE-step (carried) rotation (filter.cpp:510 runs UpdateCovariance before UpdateNominalState).
template <typename T, int N, int DIALECT, int PK = FBUS_X_PACK>
__device__ __forceinline__ void predict_nominal(T* nom, const T* accel, const T* gyro, T dt, PredictCoef<T>& k)
{
    using L = Lay<N>;
    T* p = nom + L::OFF_P3; T* v = nom + L::OFF_V; T* q = nom + L::OFF_Q;
    const T* ba = nom + L::OFF_BA; const T* bg = nom + L::OFF_BG; const T* g = nom + L::OFF_G;
    T* R = nom + L::OFF_R;
    T (&Th)[9] = k.Th;
    k.dt = dt;

    const T a[3] = { accel[0] - ba[0], accel[1] - ba[1], accel[2] - ba[2] };   // ImuUpdate.m:37-38
    const T w[3] = { gyro[0] - bg[0], gyro[1] - bg[1], gyro[2] - bg[2] };
    // fp32: |w| and 1 / |w| from one reciprocal square root (|w| = |w|^2 / |w|); a rate of exactly zero gives 0, 0
    const T wn2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    T wn, iwn;
    if constexpr (sizeof(T) == 4) { iwn = (wn2 >= T(FB_RSQRT_MIN)) ? fb_rsqrt(wn2) : T(0); wn = wn2 * iwn; }
    else { wn = fb_sqrt(wn2); iwn = (wn > T(0)) ? T(1) / wn : T(0); }          // fp64: the reference's operations

#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T r0 = R[3 * i], r1 = R[3 * i + 1], r2 = R[3 * i + 2];
        k.A[3 * i + 0] = -dt * (r1 * a[2] - r2 * a[1]);
        k.A[3 * i + 1] = -dt * (r2 * a[0] - r0 * a[2]);
        k.A[3 * i + 2] = -dt * (r0 * a[1] - r1 * a[0]);
        k.Bm[3 * i + 0] = -dt * r0; k.Bm[3 * i + 1] = -dt * r1; k.Bm[3 * i + 2] = -dt * r2;
    }

    // ---- rotation increments (kept small: n, sin/cos) and F(theta,theta) ----------
    T qT[4], R0[9], RT[9], kv2[3], kv4[3];
    if constexpr (sizeof(T) == 4 && (PK & 1)) {
        // fp32: the full increment (T) and the half one (H) are the same operations on two angles -- sin/cos polynomials, quaternion
        // product, rotation matrix, acceleration in the world frame -- and run as the halves of register pairs
        T n[3];
        f32x2 s24, c24;                                             // (s2, s4), (c2, c4)
        bool small_rate = false;
        if (DIALECT == DIALECT_MATLAB) {
            n[0] = w[0] * iwn; n[1] = w[1] * iwn; n[2] = w[2] * iwn;
            const T dth = wn * fb_abs(dt);
            fb_sincos_x_halfx_pk(dth * T(0.5), s24, c24);
            const T s2 = s24.x, c2 = c24.x;
            const T sa = (dt < T(0) ? -T(2) : T(2)) * s2 * c2, sb = T(2) * s2 * s2;
            Th[0] = T(1) - sb + sb * n[0] * n[0]; Th[1] = sb * n[0] * n[1] + sa * n[2]; Th[2] = sb * n[0] * n[2] - sa * n[1];
            Th[3] = sb * n[1] * n[0] - sa * n[2]; Th[4] = T(1) - sb + sb * n[1] * n[1]; Th[5] = sb * n[1] * n[2] + sa * n[0];
            Th[6] = sb * n[2] * n[0] + sa * n[1]; Th[7] = sb * n[2] * n[1] - sa * n[0]; Th[8] = T(1) - sb + sb * n[2] * n[2];
        } else {
            small_rate = !(wn > T(10e-5));
            const T inv = small_rate ? T(0) : iwn;
            n[0] = w[0] * inv; n[1] = w[1] * inv; n[2] = w[2] * inv;
            fb_sincos_x_halfx_pk(wn * dt * T(0.5), s24, c24);
            Th[0] = T(1);        Th[1] = w[2] * dt;   Th[2] = -w[1] * dt;
            Th[3] = -w[2] * dt;  Th[4] = T(1);        Th[5] = w[0] * dt;
            Th[6] = w[1] * dt;   Th[7] = -w[0] * dt;  Th[8] = T(1);
        }
        f32x2 dq[4], qq[4], RR[9];
        if (DIALECT == DIALECT_MATLAB || !small_rate) {
            dq[0] = c24; dq[1] = n[0] * s24; dq[2] = n[1] * s24; dq[3] = n[2] * s24;
        } else {                                                    // filter.cpp:553-560
            const f32x2 hq = { T(0.5), T(0.25) };
            dq[0] = f32x2{ T(1), T(1) }; dq[1] = hq * dt * w[0]; dq[2] = hq * dt * w[1]; dq[3] = hq * dt * w[2];
        }
        quat_mul_pk(q, dq, qq);
        if (DIALECT == DIALECT_MATLAB) {
#pragma unroll
            for (int i = 0; i < 9; ++i) R0[i] = R[i];             // carried, possibly stale (:46)
            quat_to_rotmat_m_pk(qq, RR);
        } else {
            quat_to_rotmat_e(q, R0);                               // fresh (filter.cpp:542)
            quat_normalize_pk(qq);
            quat_to_rotmat_e_pk(qq, RR);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x2 kv = RR[3 * i] * a[0] + RR[3 * i + 1] * a[1] + RR[3 * i + 2] * a[2] + g[i];
            kv4[i] = kv.x; kv2[i] = kv.y;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) qT[i] = qq[i].x;
#pragma unroll
        for (int i = 0; i < 9; ++i) RT[i] = RR[i].x;
    } else {
    T n[3], s2, c2, s4, c4;
    bool small_rate = false;
    if (DIALECT == DIALECT_MATLAB) {
        // ImuUpdate.m:41-43,68.  axis/|axis| is NaN at w == 0 in the reference; guarded here
        // (identity rotation), bit-identical away from zero.
        const T inv = iwn;
        n[0] = w[0] * inv; n[1] = w[1] * inv; n[2] = w[2] * inv;
        const T dth = wn * fb_abs(dt);
        fb_sincos_x_halfx(dth * T(0.5), s2, c2, s4, c4);
        // expm(-[w]x dt) in closed form: I - sin(phi)[n]x + (1-cos(phi))[n]x^2 with
        // sin(phi) = 2 s2 c2 and 1-cos(phi) = 2 s2^2 (no cancellation in fp32).
        const T sa = (dt < T(0) ? -T(2) : T(2)) * s2 * c2, sb = T(2) * s2 * s2;
        Th[0] = T(1) - sb + sb * n[0] * n[0]; Th[1] = sb * n[0] * n[1] + sa * n[2]; Th[2] = sb * n[0] * n[2] - sa * n[1];
        Th[3] = sb * n[1] * n[0] - sa * n[2]; Th[4] = T(1) - sb + sb * n[1] * n[1]; Th[5] = sb * n[1] * n[2] + sa * n[0];
        Th[6] = sb * n[2] * n[0] + sa * n[1]; Th[7] = sb * n[2] * n[1] - sa * n[0]; Th[8] = T(1) - sb + sb * n[2] * n[2];
    } else {
        // filter.cpp:544-561,603
        small_rate = !(wn > T(10e-5));
        const T inv = small_rate ? T(0) : iwn;
        n[0] = w[0] * inv; n[1] = w[1] * inv; n[2] = w[2] * inv;
        fb_sincos_x_halfx(wn * dt * T(0.5), s2, c2, s4, c4);
        Th[0] = T(1);        Th[1] = w[2] * dt;   Th[2] = -w[1] * dt;
        Th[3] = -w[2] * dt;  Th[4] = T(1);        Th[5] = w[0] * dt;
        Th[6] = w[1] * dt;   Th[7] = -w[0] * dt;  Th[8] = T(1);
    }

    // ---- quaternion, velocity, position   ImuUpdate.m:42-60 ; filter.cpp:539-581 -----
    T qH[4], RH[9];
    if (DIALECT == DIALECT_MATLAB) {
        const T dqT[4] = { c2, n[0] * s2, n[1] * s2, n[2] * s2 };
        const T dqH[4] = { c4, n[0] * s4, n[1] * s4, n[2] * s4 };
        quat_mul(q, dqT, qT);
        quat_mul(q, dqH, qH);
#pragma unroll
        for (int i = 0; i < 9; ++i) R0[i] = R[i];                 // carried, possibly stale (:46)
        quat_to_rotmat_m(qH, RH);
        quat_to_rotmat_m(qT, RT);
    } else {
        quat_to_rotmat_e(q, R0);                                   // fresh (filter.cpp:542)
        T dqT[4], dqH[4];
        if (!small_rate) {
            dqT[0] = c2; dqT[1] = n[0] * s2; dqT[2] = n[1] * s2; dqT[3] = n[2] * s2;
            dqH[0] = c4; dqH[1] = n[0] * s4; dqH[2] = n[1] * s4; dqH[3] = n[2] * s4;
        } else {                                                   // filter.cpp:553-560
            dqT[0] = T(1); dqT[1] = T(0.5) * dt * w[0]; dqT[2] = T(0.5) * dt * w[1]; dqT[3] = T(0.5) * dt * w[2];
            dqH[0] = T(1); dqH[1] = T(0.25) * dt * w[0]; dqH[2] = T(0.25) * dt * w[1]; dqH[3] = T(0.25) * dt * w[2];
        }
        quat_mul(q, dqT, qT);
        quat_mul(q, dqH, qH);
        quat_normalize(qH);
        quat_normalize(qT);
        quat_to_rotmat_e(qH, RH);
        quat_to_rotmat_e(qT, RT);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        kv2[i] = RH[3 * i] * a[0] + RH[3 * i + 1] * a[1] + RH[3 * i + 2] * a[2] + g[i];
        kv4[i] = RT[3 * i] * a[0] + RT[3 * i + 1] * a[1] + RT[3 * i + 2] * a[2] + g[i];
    }
    }
    // RK4-style   ImuUpdate.m:49-60 ; filter.cpp:567-581  (dt / 6: fp32 multiplies by the rounded 1/6 -- 1 ulp, no division sequence)
    const T dt6 = (sizeof(T) == 4) ? dt * T(1.0 / 6.0) : dt / 6;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T kv1 = R0[3 * i] * a[0] + R0[3 * i + 1] * a[1] + R0[3 * i + 2] * a[2] + g[i];
        const T kv3 = kv2[i];
        const T v0 = v[i];
        v[i] = v0 + dt6 * (kv1 + 2 * kv2[i] + 2 * kv3 + kv4[i]);
        const T kp2 = v0 + kv1 * dt / 2, kp3 = v0 + kv2[i] * dt / 2, kp4 = v0 + kv3 * dt / 2;   // dt/2 sic
        p[i] = p[i] + dt6 * (v0 + 2 * kp2 + 2 * kp3 + kp4);
    }
    if (DIALECT == DIALECT_MATLAB) quat_normalize(qT);               // ImuUpdate.m:76
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = qT[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = RT[i];                        // ImuUpdate.m:77 ; filter.cpp:564
}

// One ImuUpdate with everything resident: nom = the 28 nominal + rotation elements, P = packed covariance.
template <typename T, int N, int DIALECT, int PK = FBUS_X_PACK>
__device__ __forceinline__ void predict_step(T* nom, T* P, const T* accel, const T* gyro, T dt, const T* qd)
{
    PredictCoef<T> k;
    predict_nominal<T, N, DIALECT, PK>(nom, accel, gyro, dt, k);
    cov_stage_p<T, N, PK>(P, k);
    cov_stage_v<T, N, PK>(P, k, qd);
    cov_stage_th<T, N, PK>(P, k, qd);
}

// ================================================================================
// correct
// ================================================================================
// One scalar measurement row h (non-zeros hA in columns 0..2, hB in columns 6..8),
// residual rk, noise Rk, applied to P and accumulated into dx.
// fp32, even N, simple covariance form: the same update with the row-wise work on aligned pairs
// (v_pk_fma_f32: two FMAs per issue slot; at one wave per SIMD the correct kernel is VALU-issue bound).
template <int N, bool HAS_A>
__device__ __forceinline__ void scalar_update_packed(float* P, float* dx, const float* hA, const float* hB, float rk,
                                                     float Rk)
{
#define PS(i, j) P[pidx<N>((i), (j))]
#define LD2(r, c) f32x2{ P[pidx<N>((r), (c))], P[pidx<N>((r), (c)) + 1] }
    float Ph[N];
#pragma unroll
    for (int c = 0; c < N; c += 2) {
        float lo, hi;
        if (is_pair<N>(6, c) && is_pair<N>(7, c) && is_pair<N>(8, c)) {
            const f32x2 v = hB[0] * LD2(6, c) + hB[1] * LD2(7, c) + hB[2] * LD2(8, c);
            lo = v.x; hi = v.y;
        } else {
            lo = hB[0] * PS(c, 6) + hB[1] * PS(c, 7) + hB[2] * PS(c, 8);
            hi = hB[0] * PS(c + 1, 6) + hB[1] * PS(c + 1, 7) + hB[2] * PS(c + 1, 8);
        }
        if (HAS_A) {
            if (is_pair<N>(0, c) && is_pair<N>(1, c) && is_pair<N>(2, c)) {
                const f32x2 v = hA[0] * LD2(0, c) + hA[1] * LD2(1, c) + hA[2] * LD2(2, c);
                lo += v.x; hi += v.y;
            } else {
                lo += hA[0] * PS(c, 0) + hA[1] * PS(c, 1) + hA[2] * PS(c, 2);
                hi += hA[0] * PS(c + 1, 0) + hA[1] * PS(c + 1, 1) + hA[2] * PS(c + 1, 2);
            }
        }
        Ph[c] = lo; Ph[c + 1] = hi;
    }
    float s = Rk + hB[0] * Ph[6] + hB[1] * Ph[7] + hB[2] * Ph[8];
    float inn = rk - (hB[0] * dx[6] + hB[1] * dx[7] + hB[2] * dx[8]);
    if (HAS_A) {
        s += hA[0] * Ph[0] + hA[1] * Ph[1] + hA[2] * Ph[2];
        inn -= hA[0] * dx[0] + hA[1] * dx[1] + hA[2] * dx[2];
    }
    // 1/s: hardware reciprocal (1 ulp) + one Newton step instead of the ~10-instruction IEEE division sequence
    // (A/B in one run: correct -0.7 %, fused frame +1.5 %); s = h P h' + r >= r > 0, never denormal
    float is = __builtin_amdgcn_rcpf(s);
    is = is * (2.0f - s * is);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float ki = Ph[i] * is;
        dx[i] += ki * inn;
#pragma unroll
        for (int c = i & ~1; c < N; c += 2) {          // even-aligned column pairs covering j >= i
            if (c >= i && is_pair<N>(i, c)) {
                const int o = pidx<N>(i, c);
                const f32x2 v = f32x2{ P[o], P[o + 1] } - ki * f32x2{ Ph[c], Ph[c + 1] };
                P[o] = v.x; P[o + 1] = v.y;
            } else {
                if (c >= i) PS(i, c) -= ki * Ph[c];
                if (c + 1 >= i && c + 1 < N) PS(i, c + 1) -= ki * Ph[c + 1];
            }
        }
    }
#undef LD2
#undef PS
}

template <typename T, int N, bool HAS_A, int COV>
__device__ __forceinline__ void scalar_update(T* P, T* dx, const T* hA, const T* hB, T rk, T Rk)
{
    if constexpr (PackedMath<T, N>::on && COV == COV_SIMPLE) {
        scalar_update_packed<N, HAS_A>(P, dx, hA, hB, rk, Rk);
        return;
    }
#define PS(i, j) P[pidx<N>((i), (j))]
    T Ph[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T acc = hB[0] * PS(i, 6) + hB[1] * PS(i, 7) + hB[2] * PS(i, 8);
        if (HAS_A) acc += hA[0] * PS(i, 0) + hA[1] * PS(i, 1) + hA[2] * PS(i, 2);
        Ph[i] = acc;
    }
    T s = Rk + hB[0] * Ph[6] + hB[1] * Ph[7] + hB[2] * Ph[8];
    T inn = rk - (hB[0] * dx[6] + hB[1] * dx[7] + hB[2] * dx[8]);
    if (HAS_A) {
        s += hA[0] * Ph[0] + hA[1] * Ph[1] + hA[2] * Ph[2];
        inn -= hA[0] * dx[0] + hA[1] * dx[1] + hA[2] * dx[2];
    }
    const T is = T(1) / s;
    // gain entries K_i = Ph_i / s are formed row by row, never stored
    if (COV == COV_JOSEPH) {
        // P - K Ph' - Ph K' + s K K'  ==  P - K Ph'  for the optimal gain; kept as an option
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const T ki = Ph[i] * is;
            dx[i] += ki * inn;
#pragma unroll
            for (int j = i; j < N; ++j) {
                const T kj = Ph[j] * is;
                PS(i, j) += s * ki * kj - ki * Ph[j] - Ph[i] * kj;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const T ki = Ph[i] * is;
            dx[i] += ki * inn;
#pragma unroll
            for (int j = i; j < N; ++j) PS(i, j) -= ki * Ph[j];
        }
    }
#undef PS
}

// What the rows of EVERY marker share at one linearisation point: H(1:3,1:3) = -R_IL R', R P_IL, the constant-times-Lq(q)
// factor of the quaternion Jacobian and Q_IL (x) q*.  Built once per correct step; with M markers per frame the rows of
// the second and later markers are ~75 instructions shorter each (the compiler does not merge them across the per-marker
// branches by itself).  The same operations in the same order as before: results are bit for bit unchanged.
template <typename T, int N>
struct MarkerCommon {
    T Hpp[9], RP[3], M1[12], tq[4];
    __device__ __forceinline__ void build(const T* pqr, const DevConst<T>& dc)
    {
        using L = Lay<N>;
        const T* q = pqr + L::OFF_Q; const T* R = pqr + L::OFF_R;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const T l0 = dc.R_IL[3 * i], l1 = dc.R_IL[3 * i + 1], l2 = dc.R_IL[3 * i + 2];
#pragma unroll
            for (int j = 0; j < 3; ++j) Hpp[3 * i + j] = -(l0 * R[3 * j] + l1 * R[3 * j + 1] + l2 * R[3 * j + 2]);
            RP[i] = R[3 * i] * dc.P_IL[0] + R[3 * i + 1] * dc.P_IL[1] + R[3 * i + 2] * dc.P_IL[2];
        }
        const T qc[4] = { q[0], -q[1], -q[2], -q[3] };
        quat_mul(dc.Q_IL, qc, tq);
        const T w = q[0], x = q[1], y = q[2], z = q[3];
        const T LL[12] = { -x, -y, -z,   w, -z, y,   z, w, -x,   -y, x, w };   // Lq(q)(:,2:4)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                M1[3 * i + j] = dc.CL[4 * i] * LL[j] + dc.CL[4 * i + 1] * LL[3 + j] + dc.CL[4 * i + 2] * LL[6 + j] + dc.CL[4 * i + 3] * LL[9 + j];
    }
};

// The 7 rows of one marker (map slot constants mk), linearised at the record's nominal state (which is not
// modified until inject()): Jacobian blocks Hpp = H(1:3, p) (in mc), Hpt = H(1:3, theta), Hq = H(4:7, theta) (all other
// columns are zero) and the residuals rp (position rows), rq (quaternion rows).
template <typename T, int N, int DIALECT>
__device__ __forceinline__ void marker_rows(const T* pqr, const DevConst<T>& dc, const MarkerCommon<T, N>& mc,
                                            const T* __restrict__ mk, const T* yp, const T* yq, T (&Hpt)[9], T (&rp)[3],
                                            T (&Hq)[12], T (&rq)[4])
{
    using L = Lay<N>;
    const T* p = pqr + L::OFF_P3; const T* R = pqr + L::OFF_R;
    T Pm[3], Qm[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) Pm[i] = mk[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) Qm[i] = mk[3 + i];

    // hp = R_IL R' (Pm - p - R P_IL)          MeasureUpdate.m:67 ; filter.cpp:684-685
    T d[3], u[3], t[3], hp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        u[i] = Pm[i] - p[i];
        d[i] = u[i] - mc.RP[i];
    }
    T ru[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        t[i] = R[i] * d[0] + R[3 + i] * d[1] + R[6 + i] * d[2];
        ru[i] = R[i] * u[0] + R[3 + i] * u[1] + R[6 + i] * u[2];      // R'(Pm - p)
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) hp[i] = dc.R_IL[3 * i] * t[0] + dc.R_IL[3 * i + 1] * t[1] + dc.R_IL[3 * i + 2] * t[2];
    // hq = Q_IL (x) q* (x) Qm                  MeasureUpdate.m:68 ; filter.cpp:686
    T hq[4];
    quat_mul(mc.tq, Qm, hq);

    // H(1:3,1:3) = -R_IL R' (mc.Hpp) ; H(1:3,7:9) = R_IL [R'(Pm-p)]x      MeasureUpdate.m:72-73 ; filter.cpp:691-692
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T l0 = dc.R_IL[3 * i], l1 = dc.R_IL[3 * i + 1], l2 = dc.R_IL[3 * i + 2];
        Hpt[3 * i + 0] = l1 * ru[2] - l2 * ru[1];
        Hpt[3 * i + 1] = l2 * ru[0] - l0 * ru[2];
        Hpt[3 * i + 2] = l0 * ru[1] - l1 * ru[0];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) rp[k] = yp[k] - hp[k];

    // H(4:7,7:9) = Rq(Qm) [Lq(Q_IL) L2] [Lq(q) L1]   MeasureUpdate.m:74-75 ; filter.cpp:693-694
    // evaluated right to left: M1 = CL * Lq(q)(:,2:4) with the wave-uniform constant CL (mc.M1), then Rq(Qm) * M1 -- only
    // the marker's quaternion is read per marker (a per-marker 4x4 table cost 16 dependent loads: correct -7 %)
    const T (&M1)[12] = mc.M1;
    const T RqM[16] = { Qm[0], -Qm[1], -Qm[2], -Qm[3],   Qm[1], Qm[0], Qm[3], -Qm[2],
                        Qm[2], -Qm[3], Qm[0], Qm[1],     Qm[3], Qm[2], -Qm[1], Qm[0] };
    // sign unification                          MeasureUpdate.m:77-81 ; filter.cpp:698-706
    T k1 = T(0), k2 = T(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        k1 += (yq[i] - hq[i]) * (yq[i] - hq[i]);
        k2 += (yq[i] + hq[i]) * (yq[i] + hq[i]);
    }
    const T sg = (k1 > k2) ? T(-0.5) : T(0.5);                    // 0.5 = L1
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Hq[3 * i + j] = sg * (RqM[4 * i] * M1[j] + RqM[4 * i + 1] * M1[3 + j] + RqM[4 * i + 2] * M1[6 + j] + RqM[4 * i + 3] * M1[9 + j]);
    const T sq = (k1 > k2) ? T(-1) : T(1);
    // Matlab zeroes the quaternion residual (MeasureUpdate.m:88); C++ uses it (filter.cpp:718-721)
#pragma unroll
    for (int k = 0; k < 4; ++k) rq[k] = (DIALECT == DIALECT_CPP) ? (yq[k] - sq * hq[k]) : T(0);
}

// One marker applied row by row (the reference's 7-row update, algebraically K = P H'(H P H' + R)^-1 with the
// diagonal R): 7 sequential scalar updates at one linearisation point.
template <typename T, int N, int DIALECT, int COV>
__device__ __forceinline__ void marker_update(T* P, T* dx, const T* pqr, const DevConst<T>& dc, const MarkerCommon<T, N>& mc,
                                              const T* __restrict__ mk, const T* yp, const T* yq)
{
    T Hpt[9], rp[3], Hq[12], rq[4];
    marker_rows<T, N, DIALECT>(pqr, dc, mc, mk, yp, yq, Hpt, rp, Hq, rq);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        scalar_update<T, N, true, COV>(P, dx, mc.Hpp + 3 * k, Hpt + 3 * k, rp[k], dc.r_pos);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        scalar_update<T, N, false, COV>(P, dx, Hq + 3 * k, Hq + 3 * k, rq[k], dc.r_quat);
}

// --------------------------------------------------------------------------------
// Stacked mode: the information-compressed joint update.
// Every row of every marker has its non-zeros in the same six columns J = (p, theta), so the stacked 7M-row
// measurement enters the Kalman update only through the 6x6 information matrix Lam = sum h_J h_J' / r and the
// 6-vector b = sum h_J res / r  (K = P H'(H P H' + R)^-1 depends on H, R, res through them alone).  Lam = L D L'
// gives six equivalent scalar measurements -- row a = column a of the unit lower-triangular L (non-zeros in J-positions
// a..5), information d_a, residual-times-information beta_a = (L^-1 b)_a -- which are applied as six sequential
// scalar updates: 6 rank-1 passes over P instead of 7M, the same posterior.  A direction without information
// (d_a = 0) is a no-op, nothing divides by d.
// --------------------------------------------------------------------------------
__host__ __device__ constexpr int jcol(int k) { return k < 3 ? k : k + 3; }            // J-position -> state column
__host__ __device__ constexpr int lidx(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }   // i <= j < 6

template <typename T>
struct InfoAcc {
    T Lam[21];      // upper triangle of the 6x6 information matrix, lidx order
    T b[6];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < 21; ++i) Lam[i] = T(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] = T(0);
    }
    // a row with non-zeros hA (columns p) and hB (columns theta), residual res, weight w = 1 / noise
    __device__ __forceinline__ void add6(const T* hA, const T* hB, T res, T w)
    {
        const T h[6] = { hA[0], hA[1], hA[2], hB[0], hB[1], hB[2] };
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const T wi = w * h[i];
            b[i] += wi * res;
#pragma unroll
            for (int j = i; j < 6; ++j) Lam[lidx(i, j)] += wi * h[j];
        }
    }
    // a row with non-zeros hB in the theta columns only
    __device__ __forceinline__ void add3(const T* hB, T res, T w)
    {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const T wi = w * hB[i];
            b[3 + i] += wi * res;
#pragma unroll
            for (int j = i; j < 3; ++j) Lam[lidx(3 + i, 3 + j)] += wi * hB[j];
        }
    }
};

// the 7 rows of one marker into the accumulator
template <typename T, int N, int DIALECT>
__device__ __forceinline__ void marker_info(InfoAcc<T>& acc, const T* pqr, const DevConst<T>& dc, const MarkerCommon<T, N>& mc,
                                            const T* __restrict__ mk, const T* yp, const T* yq, T w_pos, T w_quat)
{
    T Hpt[9], rp[3], Hq[12], rq[4];
    marker_rows<T, N, DIALECT>(pqr, dc, mc, mk, yp, yq, Hpt, rp, Hq, rq);
#pragma unroll
    for (int k = 0; k < 3; ++k) acc.add6(mc.Hpp + 3 * k, Hpt + 3 * k, rp[k], w_pos);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc.add3(Hq + 3 * k, rq[k], w_quat);
}

// --------------------------------------------------------------------------------
// The fold of the 7 pose rows of M markers, with what the rows of ALL markers share taken out of the per-marker work
// (round 3; the same sums as marker_info row by row, regrouped -- exact algebra, results equal to rounding):
//   * position rows: H(1:3, p) = Hpp = -R_IL R' is the same matrix for every marker, so
//         Lam_pp = w n Hpp' Hpp          Lam_pt = w Hpp' (sum_m Hpt_m)          b_p = w Hpp' (sum_m rp_m)
//     and only  sum Hpt_m,  sum rp_m,  sum Hpt_m' Hpt_m  and  sum Hpt_m' rp_m  are accumulated per marker;
//   * quaternion rows: Hq = s Rq(Qm) [Lq(Q_IL) L2] Lq(q)(:,2:4), s = +-1/2, and Rq(Q)'Rq(Q) = |Q|^2 I, Lq(Q)'Lq(Q) = |Q|^2 I,
//     L2'L2 = I, the last three columns of Lq(q) orthogonal with norm |q|: Hq'Hq = 1/4 |Qm|^2 |Q_IL|^2 |q|^2 I_3 EXACTLY,
//     whatever the marker and the sign -- the four quaternion rows of a marker add an isotropic c_m |q|^2 w_quat to the theta
//     diagonal (c_m = 1/4 |Q_IL|^2 |Qm|^2 sits in the map slot's 8th entry), and in the Matlab dialect (quaternion residual
//     zeroed, MeasureUpdate.m:88) that is ALL they do: no Q_IL (x) q* (x) Qm, no sign test, no 4 x 3 Jacobian per marker.
//     The C++ dialect still needs the residual: b_theta += w s M1' (Rq(Qm)' rq).
// Per marker ~75 (Matlab) / ~135 (C++) instructions instead of ~275.
// --------------------------------------------------------------------------------
template <typename T, int N, int DIALECT>
struct PoseFold {
    T sH[9], sr[3], Ltt[6], bt[3], csum, cnt, btq[3];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < 9; ++i) sH[i] = T(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) Ltt[i] = T(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) { sr[i] = T(0); bt[i] = T(0); btq[i] = T(0); }
        csum = T(0); cnt = T(0);
    }
    static constexpr int NVAL = 26;                       // for exchanges: sH 9, sr 3, Ltt 6, bt 3, csum, cnt, btq 3
    __device__ __forceinline__ T& at(int k)
    {
        return k < 9 ? sH[k] : (k < 12 ? sr[k - 9] : (k < 18 ? Ltt[k - 12] : (k < 21 ? bt[k - 18] : (k == 21 ? csum : (k == 22 ? cnt : btq[k - 23])))));
    }
    // three position-type rows of ONE world point Pw seen at y in the left camera (a marker's origin, or one of its corners):
    // h = R_IL R'(Pw - p - R P_IL), H(:, p) = Hpp, H(:, theta) = R_IL [R'(Pw - p)]x
    // (MeasureUpdate.m:67,72-73 ; filter.cpp:684-685,691-692)
    __device__ __forceinline__ void add_point(const T* pqr, const DevConst<T>& dc, const MarkerCommon<T, N>& mc, const T* Pw, const T* y)
    {
        using L = Lay<N>;
        const T* p = pqr + L::OFF_P3; const T* R = pqr + L::OFF_R;
        T u[3], d[3], t[3], ru[3], Hpt[9], rp[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { u[i] = Pw[i] - p[i]; d[i] = u[i] - mc.RP[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            t[i] = R[i] * d[0] + R[3 + i] * d[1] + R[6 + i] * d[2];
            ru[i] = R[i] * u[0] + R[3 + i] * u[1] + R[6 + i] * u[2];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const T l0 = dc.R_IL[3 * i], l1 = dc.R_IL[3 * i + 1], l2 = dc.R_IL[3 * i + 2];
            rp[i] = y[i] - (l0 * t[0] + l1 * t[1] + l2 * t[2]);
            Hpt[3 * i + 0] = l1 * ru[2] - l2 * ru[1];
            Hpt[3 * i + 1] = l2 * ru[0] - l0 * ru[2];
            Hpt[3 * i + 2] = l0 * ru[1] - l1 * ru[0];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) sH[i] += Hpt[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sr[i] += rp[i];
            bt[i] += Hpt[i] * rp[0] + Hpt[3 + i] * rp[1] + Hpt[6 + i] * rp[2];
        }
        int o = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) Ltt[o++] += Hpt[i] * Hpt[j] + Hpt[3 + i] * Hpt[3 + j] + Hpt[6 + i] * Hpt[6 + j];
        cnt += T(1);
    }
    // the four triangulated corners C[12] of one marker as 12 position-type rows (north-star extension, no reference counterpart):
    // corner k of the marker frame of vision.cpp:736-759, c_k = {(0,0,0),(0,s,0),(s,s,0),(s,0,0)}, in the world: P_m + R_m c_k
    __device__ __forceinline__ void add_corners(const T* pqr, const DevConst<T>& dc, const MarkerCommon<T, N>& mc, const T* __restrict__ mk,
                                                const T* C, T size)
    {
        T Rm[9];
        const T Qm[4] = { mk[3], mk[4], mk[5], mk[6] };
        quat_to_rotmat_m(Qm, Rm);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T cx = (k >= 2) ? size : T(0), cy = (k == 1 || k == 2) ? size : T(0);
            const T Pw[3] = { mk[0] + Rm[0] * cx + Rm[1] * cy, mk[1] + Rm[3] * cx + Rm[4] * cy, mk[2] + Rm[6] * cx + Rm[7] * cy };
            add_point(pqr, dc, mc, Pw, C + 3 * k);
        }
    }
    // one marker (map slot constants mk: position 3, quaternion 4, c_m), measured pose yp, yq: 3 position rows + 4 quaternion rows
    __device__ __forceinline__ void add(const T* pqr, const DevConst<T>& dc, const MarkerCommon<T, N>& mc, const T* __restrict__ mk,
                                        const T* yp, const T* yq)
    {
        add_point(pqr, dc, mc, mk, yp);
        csum += mk[7];
        if constexpr (DIALECT == DIALECT_CPP) {
            // the quaternion residual is used (filter.cpp:698-706,718-721): b_theta += w s M1' (Rq(Qm)' rq)
            const T Qm[4] = { mk[3], mk[4], mk[5], mk[6] };
            T hq[4];
            quat_mul(mc.tq, Qm, hq);                                 // Q_IL (x) q* (x) Qm
            T k1 = T(0), k2 = T(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { k1 += (yq[i] - hq[i]) * (yq[i] - hq[i]); k2 += (yq[i] + hq[i]) * (yq[i] + hq[i]); }
            const T sq = (k1 > k2) ? T(-1) : T(1);
            T rq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rq[i] = yq[i] - sq * hq[i];
            // v = Rq(Qm)' rq   (Rq as in marker_rows: rows (w -x -y -z ; x w z -y ; y -z w x ; z y -x w))
            const T v[4] = { Qm[0] * rq[0] + Qm[1] * rq[1] + Qm[2] * rq[2] + Qm[3] * rq[3],
                             -Qm[1] * rq[0] + Qm[0] * rq[1] - Qm[3] * rq[2] + Qm[2] * rq[3],
                             -Qm[2] * rq[0] + Qm[3] * rq[1] + Qm[0] * rq[2] - Qm[1] * rq[3],
                             -Qm[3] * rq[0] - Qm[2] * rq[1] + Qm[1] * rq[2] + Qm[0] * rq[3] };
            const T sg = T(0.5) * sq;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                btq[j] += sg * (mc.M1[j] * v[0] + mc.M1[3 + j] * v[1] + mc.M1[6 + j] * v[2] + mc.M1[9 + j] * v[3]);
        }
    }
    // -> the 6 x 6 information matrix and vector of the stacked rows (what marker_info accumulates row by row)
    __device__ __forceinline__ void finish(InfoAcc<T>& acc, const T* pqr, const DevConst<T>& dc, const MarkerCommon<T, N>& mc) const
    {
        using L = Lay<N>;
        const T* q = pqr + L::OFF_Q;
        const T wp = fb_rcp1(dc.r_pos), wq = fb_rcp1(dc.r_quat);
        const T (&H)[9] = mc.Hpp;
        const T wn = wp * cnt;
        // Lam_pp = w n Hpp' Hpp
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) acc.Lam[lidx(i, j)] = wn * (H[i] * H[j] + H[3 + i] * H[3 + j] + H[6 + i] * H[6 + j]);
        // Lam_pt = w Hpp' sum Hpt ;  b_p = w Hpp' sum rp
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc.Lam[lidx(i, 3 + j)] = wp * (H[i] * sH[j] + H[3 + i] * sH[3 + j] + H[6 + i] * sH[6 + j]);
            acc.b[i] = wp * (H[i] * sr[0] + H[3 + i] * sr[1] + H[6 + i] * sr[2]);
        }
        // Lam_tt = w_pos sum Hpt'Hpt + w_quat |q|^2 (sum c_m) I ;  b_t = w_pos sum Hpt' rp (+ w_quat sum Hq' rq)
        const T iso = wq * (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) * csum;
        int o = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = i; j < 3; ++j) { acc.Lam[lidx(3 + i, 3 + j)] = wp * Ltt[o] + (i == j ? iso : T(0)); ++o; }
            acc.b[3 + i] = wp * bt[i] + (DIALECT == DIALECT_CPP ? wq * btq[i] : T(0));
        }
    }
};

// One equivalent scalar measurement: row = (0,..,0, 1, l[A+1..5]) in the J columns, information d, beta.
//   s' = 1 + d h P h' ; dx += P h' (beta - d h dx) / s' ; P -= (d / s') (P h')(P h')'
template <typename T, int N, int A, int COV, typename HOOK = NoRowHook>
__device__ __forceinline__ void scalar_update_info(T* P, T* dx, const T* l, T d, T beta, const HOOK& hook = HOOK())
{
#define PS(i, j) P[pidx<N>((i), (j))]
    T Ph[N];
    if constexpr (PackedMath<T, N>::on && COV == COV_SIMPLE) {
#define LD2(r, c) f32x2{ P[pidx<N>((r), (c))], P[pidx<N>((r), (c)) + 1] }
#pragma unroll
        for (int c = 0; c < N; c += 2) {
            f32x2 v = { 0.f, 0.f };
            float lo = 0.f, hi = 0.f;
            bool vset = false, sset = false;
#pragma unroll
            for (int k = A; k < 6; ++k) {
                const int r = jcol(k);
                if (is_pair<N>(r, c)) {
                    const f32x2 t = LD2(r, c);
                    v = !vset ? ((k == A) ? t : l[k] * t) : ((k == A) ? v + t : v + l[k] * t);
                    vset = true;
                } else {
                    const float t0 = PS(c, r), t1 = PS(c + 1, r);
                    lo = !sset ? ((k == A) ? t0 : l[k] * t0) : ((k == A) ? lo + t0 : lo + l[k] * t0);
                    hi = !sset ? ((k == A) ? t1 : l[k] * t1) : ((k == A) ? hi + t1 : hi + l[k] * t1);
                    sset = true;
                }
            }
            Ph[c] = vset ? (sset ? v.x + lo : v.x) : lo;
            Ph[c + 1] = vset ? (sset ? v.y + hi : v.y) : hi;
        }
#undef LD2
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = PS(i, jcol(A));
#pragma unroll
            for (int k = A + 1; k < 6; ++k) acc += l[k] * PS(i, jcol(k));
            Ph[i] = acc;
        }
    }
    T hPh = Ph[jcol(A)], hdx = dx[jcol(A)];
#pragma unroll
    for (int k = A + 1; k < 6; ++k) { hPh += l[k] * Ph[jcol(k)]; hdx += l[k] * dx[jcol(k)]; }
    const T sp = T(1) + d * hPh;                 // >= 1
    T is;
    if constexpr (sizeof(T) == 4) {
        is = __builtin_amdgcn_rcpf(sp);
        is = is * (2.0f - sp * is);
    } else {
        is = T(1) / sp;
    }
    const T g = (beta - d * hdx) * is, dk = d * is;
    if constexpr (PackedMath<T, N>::on && COV == COV_SIMPLE) {
        static_for<0, N>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const float ki = Ph[i] * dk;
            dx[i] += Ph[i] * g;
#pragma unroll
            for (int c = i & ~1; c < N; c += 2) {
                if (c >= i && is_pair<N>(i, c)) {
                    const int o = pidx<N>(i, c);
                    const f32x2 v = f32x2{ P[o], P[o + 1] } - ki * f32x2{ Ph[c], Ph[c + 1] };
                    P[o] = v.x; P[o + 1] = v.y;
                } else {
                    if (c >= i) PS(i, c) -= ki * Ph[c];
                    if (c + 1 >= i && c + 1 < N) PS(i, c + 1) -= ki * Ph[c + 1];
                }
            }
            hook.template row_done<i>();
        });
    } else if constexpr (COV == COV_JOSEPH) {
        // P - K Ph' - Ph K' + s K K' with s K_i K_j = Ph_i Ph_j d / s' (finite for d -> 0)
        static_for<0, N>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const T ki = Ph[i] * dk;
            dx[i] += Ph[i] * g;
#pragma unroll
            for (int j = i; j < N; ++j) {
                const T kj = Ph[j] * dk;
                PS(i, j) += (Ph[i] * Ph[j]) * dk - ki * Ph[j] - Ph[i] * kj;
            }
            hook.template row_done<i>();
        });
    } else if constexpr (std::is_same<HOOK, NoRowHook>::value) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const T ki = Ph[i] * dk;
            dx[i] += Ph[i] * g;
#pragma unroll
            for (int j = i; j < N; ++j) PS(i, j) -= ki * Ph[j];
        }
    } else {
        static_for<0, N>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const T ki = Ph[i] * dk;
            dx[i] += Ph[i] * g;
#pragma unroll
            for (int j = i; j < N; ++j) PS(i, j) -= ki * Ph[j];
            hook.template row_done<i>();
        });
    }
#undef PS
}

// Lam = L D L' in place (no pivoting: Lam is positive semi-definite; a pivot that is not clearly positive relative
// to its original diagonal carries no information and is dropped), beta = L^-1 b, then the six scalar updates.
// Where the factors of Lam = L D L' live between the factorisation and the six passes: NoPark keeps them in registers
// (fp32); LdsPark writes the 27 values to LDS (lane-strided: value k of lane l at base[k * 64 + l]) and every pass
// reads back the 7 it needs.  The fp64 kernels need it: 171 covariance doubles are 342 of the 512 registers, dx and
// P h' another 72, and with the factors (and the compiler's habit of keeping all of them live) the six passes spilled
// 600-1600 bytes per lane to scratch.
struct NoPark {
    static constexpr bool on = false;
};
template <typename T>
struct LdsPark {
    static constexpr bool on = true;
    static constexpr int NVAL = 27;             // l (15), d (6), beta (6)
    T* base;                                    // &lds[lane]
    __device__ __forceinline__ void put(int k, T v) const { base[k * 64] = v; }
    __device__ __forceinline__ T get(int k) const { return base[k * 64]; }
};
__host__ __device__ constexpr int park_l(int a, int i) { return a * 5 - (a * (a - 1)) / 2 + (i - a - 1); }   // a < i < 6 -> 0..14

// Lam = L D L' in place (no pivoting: Lam is positive semi-definite; a pivot that is not clearly positive relative
// to its original diagonal carries no information and is dropped), beta = L^-1 b, then the six scalar updates.
template <typename T>
struct InfoFactors {            // unit lower-triangular L (row a = l[a][a+1..5]), D, beta = L^-1 b
    T d[6], l[6][6], bt[6];
};

// Step 1: factorise (registers only: the covariance does not have to be resident yet).
template <typename T, typename PARK = NoPark>
__device__ __forceinline__ void joint_factor(InfoAcc<T>& acc, InfoFactors<T>& f, const PARK& park = PARK())
{
    T (&A)[21] = acc.Lam;
    T (&bt)[6] = acc.b;
    T dg0[6];
    const T tiny = (sizeof(T) == 4) ? T(2e-6) : T(4e-15);
#pragma unroll
    for (int a = 0; a < 6; ++a) dg0[a] = A[lidx(a, a)];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const T piv = A[lidx(a, a)];
        const bool ok = piv > tiny * dg0[a];
        const T inv = ok ? fb_rcp1(piv) : T(0);
        f.d[a] = ok ? piv : T(0);
        if (!ok) bt[a] = T(0);
#pragma unroll
        for (int i = a + 1; i < 6; ++i) f.l[a][i] = A[lidx(a, i)] * inv;
#pragma unroll
        for (int i = a + 1; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) A[lidx(i, j)] -= f.l[a][i] * A[lidx(a, j)];
            bt[i] -= f.l[a][i] * bt[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) f.bt[a] = bt[a];
    if constexpr (PARK::on) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int i = a + 1; i < 6; ++i) park.put(park_l(a, i), f.l[a][i]);
            park.put(15 + a, f.d[a]);
            park.put(21 + a, f.bt[a]);
        }
        // the compiler must not forward the stored values into the passes (that is the register pressure this avoids)
        asm volatile("" ::: "memory");
    }
}

// Step 2: the six scalar updates.
template <typename T, int N, int COV, typename HOOK = NoRowHook, typename PARK = NoPark>
__device__ __forceinline__ void joint_apply(T* P, T* dx, const InfoFactors<T>& f, const HOOK& hook = HOOK(), const PARK& park = PARK())
{
    if constexpr (PARK::on) {
        static_for<0, 6>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            T la[6];
#pragma unroll
            for (int i = a + 1; i < 6; ++i) la[i] = park.get(park_l(a, i));
            const T da = park.get(15 + a), ba = park.get(21 + a);
            if constexpr (a == 5) scalar_update_info<T, N, a, COV, HOOK>(P, dx, la, da, ba, hook);
            else scalar_update_info<T, N, a, COV>(P, dx, la, da, ba);
            asm volatile("" ::: "memory");
        });
    } else {
        scalar_update_info<T, N, 0, COV>(P, dx, f.l[0], f.d[0], f.bt[0]);
        scalar_update_info<T, N, 1, COV>(P, dx, f.l[1], f.d[1], f.bt[1]);
        scalar_update_info<T, N, 2, COV>(P, dx, f.l[2], f.d[2], f.bt[2]);
        scalar_update_info<T, N, 3, COV>(P, dx, f.l[3], f.d[3], f.bt[3]);
        scalar_update_info<T, N, 4, COV>(P, dx, f.l[4], f.d[4], f.bt[4]);
        scalar_update_info<T, N, 5, COV, HOOK>(P, dx, f.l[5], f.d[5], f.bt[5], hook);     // rows become final one by one
    }
}

template <typename T, int N, int COV, typename HOOK = NoRowHook, typename PARK = NoPark>
__device__ __forceinline__ void joint_update(T* P, T* dx, InfoAcc<T>& acc, const HOOK& hook = HOOK(), const PARK& park = PARK())
{
    InfoFactors<T> f;
    joint_factor<T, PARK>(acc, f, park);
    joint_apply<T, N, COV, HOOK, PARK>(P, dx, f, hook, park);
}

// --------------------------------------------------------------------------------
// Row-split form of the six passes (the fp64 kernels): the same arithmetic, element by element and in the same order,
// with at most the first RS rows of the covariance resident at a time.
//   P h' of every pass reads the p and theta COLUMNS of P only, i.e. storage rows 0..8 (P(i, c) with i > c lives in row c),
//   so the six passes can run on rows 0..RS-1 alone (RS = 9) while rows RS..N-1 are not even loaded; each pass leaves the
//   part of P h' that the late rows will need (N - RS values), its gain scale d / s' and its state factor in LDS.  The late
//   rows then arrive into the registers the early rows have vacated and take the six rank-1 terms in the same order.
// 171 covariance doubles are 342 of the 512 registers; with dx, P h' and the factors on top the unsplit passes spilled
// 150-1600 bytes per lane to scratch.  Split: 132 early doubles + 36 + 36 + 54 = 390 registers at the peak.
// --------------------------------------------------------------------------------
template <typename T, int N, int RS>
struct LateStash {
    static constexpr int PER = (N - RS) + 2;         // P h' for rows RS..N-1, d / s', (beta - d h dx) / s'
    static constexpr int NVAL = 6 * PER;
    T* base;                                         // &lds[lane], lane-strided
    __device__ __forceinline__ void put(int a, int k, T v) const { base[(a * PER + k) * 64] = v; }
    __device__ __forceinline__ T get(int a, int k) const { return base[(a * PER + k) * 64]; }
};

template <typename T, int N, int A, int COV, int RS, typename HOOK>
__device__ __forceinline__ void scalar_update_info_early(T* P, T* dx, const T* l, T d, T beta, const HOOK& hook,
                                                         const LateStash<T, N, RS>& st)
{
#define PS(i, j) P[pidx<N>((i), (j))]
    static_assert(RS > 8, "P h' needs the p and theta columns: storage rows 0..8 must be early rows");
    T Ph[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T acc = PS(i, jcol(A));
#pragma unroll
        for (int k = A + 1; k < 6; ++k) acc += l[k] * PS(i, jcol(k));
        Ph[i] = acc;
    }
    T hPh = Ph[jcol(A)], hdx = dx[jcol(A)];
#pragma unroll
    for (int k = A + 1; k < 6; ++k) { hPh += l[k] * Ph[jcol(k)]; hdx += l[k] * dx[jcol(k)]; }
    const T sp = T(1) + d * hPh;
    T is;
    if constexpr (sizeof(T) == 4) { is = __builtin_amdgcn_rcpf(sp); is = is * (2.0f - sp * is); } else { is = T(1) / sp; }
    const T g = (beta - d * hdx) * is, dk = d * is;
#pragma unroll
    for (int i = RS; i < N; ++i) st.put(A, i - RS, Ph[i]);
    st.put(A, N - RS, dk);
    st.put(A, N - RS + 1, g);
    static_for<0, RS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const T ki = Ph[i] * dk;
        dx[i] += Ph[i] * g;
#pragma unroll
        for (int j = i; j < N; ++j) {
            if constexpr (COV == COV_JOSEPH) { const T kj = Ph[j] * dk; PS(i, j) += (Ph[i] * Ph[j]) * dk - ki * Ph[j] - Ph[i] * kj; }
            else PS(i, j) -= ki * Ph[j];
        }
        if constexpr (A == 5) hook.template row_done<i>();
    });
#undef PS
}

template <typename T, int N, int COV, int RS, typename HOOK = NoRowHook>
__device__ __forceinline__ void joint_apply_early(T* P, T* dx, const InfoFactors<T>& f, const HOOK& hook,
                                                  const LateStash<T, N, RS>& st)
{
    scalar_update_info_early<T, N, 0, COV, RS>(P, dx, f.l[0], f.d[0], f.bt[0], hook, st);
    scalar_update_info_early<T, N, 1, COV, RS>(P, dx, f.l[1], f.d[1], f.bt[1], hook, st);
    scalar_update_info_early<T, N, 2, COV, RS>(P, dx, f.l[2], f.d[2], f.bt[2], hook, st);
    scalar_update_info_early<T, N, 3, COV, RS>(P, dx, f.l[3], f.d[3], f.bt[3], hook, st);
    scalar_update_info_early<T, N, 4, COV, RS>(P, dx, f.l[4], f.d[4], f.bt[4], hook, st);
    scalar_update_info_early<T, N, 5, COV, RS>(P, dx, f.l[5], f.d[5], f.bt[5], hook, st);
    asm volatile("" ::: "memory");               // the late phase reads the stash back: no forwarding of the stored values
}

// rows RS..N-1 of P and of dx: the six rank-1 terms, pass by pass (the order the unsplit passes apply them in)
template <typename T, int N, int COV, int RS>
__device__ __forceinline__ void joint_apply_late(T* P, T* dx, const LateStash<T, N, RS>& st)
{
#define PS(i, j) P[pidx<N>((i), (j))]
    static_for<0, 6>([&](auto ac) {
        constexpr int a = decltype(ac)::value;
        T Ph[N];
#pragma unroll
        for (int i = RS; i < N; ++i) Ph[i] = st.get(a, i - RS);
        const T dk = st.get(a, N - RS), g = st.get(a, N - RS + 1);
#pragma unroll
        for (int i = RS; i < N; ++i) {
            const T ki = Ph[i] * dk;
            dx[i] += Ph[i] * g;
#pragma unroll
            for (int j = i; j < N; ++j) {
                if constexpr (COV == COV_JOSEPH) { const T kj = Ph[j] * dk; PS(i, j) += (Ph[i] * Ph[j]) * dk - ki * Ph[j] - Ph[i] * kj; }
                else PS(i, j) -= ki * Ph[j];
            }
        }
    });
#undef PS
}

// State injection   MeasureUpdate.m:92-98 ; filter.cpp:726-733.  R is NOT refreshed.
template <typename T, int N>
__device__ __forceinline__ void inject(T* rec /* the 28 nominal + rotation elements */, const T* dx)
{
    using L = Lay<N>;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rec[L::OFF_P3 + i] += dx[i];
        rec[L::OFF_V + i] += dx[3 + i];
        rec[L::OFF_BA + i] += dx[9 + i];
        rec[L::OFF_BG + i] += dx[12 + i];
        if (N == 18) rec[L::OFF_G + i] += dx[15 + i];
    }
    const T n2 = dx[6] * dx[6] + dx[7] * dx[7] + dx[8] * dx[8];
    T nn, inn;                                                     // |dtheta| and its reciprocal (fp32: one refined rsq, see predict_nominal)
    if constexpr (sizeof(T) == 4) { inn = (n2 >= T(FB_RSQRT_MIN)) ? fb_rsqrt(n2) : T(0); nn = n2 * inn; }
    else { nn = fb_sqrt(n2); inn = (nn > T(0)) ? T(1) / nn : T(0); }
    T s, c;
    fb_sincos(nn * T(0.5), s, c);
    const T k = (sizeof(T) == 4) ? s * inn : ((nn > T(0)) ? s / nn : T(0));   // guard for the reference's 0/0
    const T dq[4] = { c, dx[6] * k, dx[7] * k, dx[8] * k };
    T qn[4];
    quat_mul(rec + L::OFF_Q, dq, qn);
    quat_normalize(qn);
#pragma unroll
    for (int i = 0; i < 4; ++i) rec[L::OFF_Q + i] = qn[i];
}

}  // namespace fbus
