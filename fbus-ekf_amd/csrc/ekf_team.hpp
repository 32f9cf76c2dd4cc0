// ekf_team.hpp -- "team" kernels: several waves share ONE 64-filter tile (fp32, simple covariance form).
//
// Every other kernel of this library maps one filter to one lane and one 64-filter tile to one wave: a launch of B
// filters has B / 64 waves.  Below 65 536 filters that leaves SIMDs idle (BASELINE config 2: 64 waves, config 3: 256,
// config 4's per-GPU share: 512 on 1024 SIMDs) and what is left is one long dependent instruction stream per wave
// (~1050 VALU instructions per ImuUpdate, ~3300 per stacked MeasureUpdate, 4+ cycles each).  Splitting a filter over
// LANES of one wave does not help here: the two halves of the work are different instruction streams (a wave executes
// both, masked), and the symmetric packed covariance has no partition that is uniform across lanes.  Splitting it over
// WAVES does: each wave ("role") of a workgroup runs its own instruction stream over the same 64 filters, lane l of every
// role working on filter l of the tile, every HBM access still a full-width 1 KiB piece of the unchanged record layout.
//
//   predict  (ImuUpdate.m:63-81 ; filter.cpp:588-616)  F P F' is three in-place congruences -- rows p, rows v, rows theta
//            (cov_stage_p / _v / _th).  Each of them READS ONLY PRE-STEP VALUES of the rows it does not own (that is why the
//            one-wave kernel may run them in this order in place), so three roles can run them side by side on private
//            copies of the rows they read, with no exchange of data -- only one barrier between the loads and the first store
//            (a role must not store rows another role has yet to read) -- and store disjoint parts of the record.  Same device
//            functions on the same inputs: the results equal the one-wave kernel's up to the compiler's FMA contraction
//            (which product of an a*b + c*d gets fused differs between the specialised kernels: measured 1 ulp on 5 of the
//            171 covariance elements of 17 of 311 filters, 1 ulp on single components of the nominal state).
//   predict_n  the same roles over K samples; between two steps role v hands its new rows to role p, role theta its new
//            rows to role v, and the nominal role its coefficient blocks to everybody through LDS (one barrier per step).
//   correct  (MeasureUpdate.m:84-102 ; filter.cpp:709-739)  every row of every marker has its non-zeros in the six
//            columns J = (p, theta), so with Lam = sum h h'/r (6 x 6) and V = P(:, J)
//                P+ = P - V (Lam^-1 + P_JJ)^-1 V' = P - W W',   W = V Z,  Z = Lc C^-T,  Lam = Lc Lc',  C C' = I + Lc' P_JJ Lc
//                dx = W gamma,  gamma = C^-1 Lc^-1 b
//            (the same posterior as the six sequential scalar updates of joint_update; one-shot instead of sequential so
//            that the rows of W and the elements of P can be divided among the roles).  Roles fold different markers,
//            exchange the 27 partial sums, each solves the 6 x 6 problem (redundantly, while its covariance chunks are
//            still on their way in), computes its rows of W, exchanges them, updates and stores its chunks of P.
//
// The launcher (kernels_tu.hip) picks these kernels when a launch would otherwise have too few waves to fill the chip;
// fbus_ekf_set_team() / FBUS_TEAM_PREDICT / FBUS_TEAM_CORRECT override the choice.  gfx950 only.
#pragma once
#include "ekf_kernels.hpp"

namespace {

using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

// ---------------------------------------------------------------------------------
// storage index -> (row, column) of the packed covariance, as compile-time tables
// ---------------------------------------------------------------------------------
template <int N>
struct CovMap {
    static constexpr int NP = N * (N + 1) / 2;
    int row[NP], col[NP];
    constexpr CovMap() : row{}, col{}
    {
        for (int i = 0; i < N; ++i)
            for (int j = i; j < N; ++j) {
                row[pidx<N>(i, j)] = i;
                col[pidx<N>(i, j)] = j;
            }
    }
};
template <int N> struct CovTab { static constexpr CovMap<N> m{}; };
template <int N> constexpr int cov_row(int e) { return CovTab<N>::m.row[e]; }
template <int N> constexpr int cov_col(int e) { return CovTab<N>::m.col[e]; }

// which stage of ImuUpdate writes element (i, j), i <= j: 0 rows p, 1 rows v, 2 rows theta and the + Q diagonals of ba
// and bg, 3 nothing (invariant under predict)
constexpr int stage_of(int i, int j) { return i < 3 ? 0 : (i < 6 ? 1 : (i < 9 ? 2 : ((i == j && i < 15) ? 2 : 3))); }
// does stage S read element (i, j), i <= j?  (supersets are fine: they only cost a load)
constexpr bool stage_reads(int S, int i, int j)
{
    if (S == 0) return i < 6;                                       // rows p and the pre-step rows v
    if (S == 1) return i >= 3;                                      // rows v, theta, ba, g (and what lies between)
    return (i >= 6 && i < 9) || (i >= 9 && i < 12 && j >= 12 && j < 15) || (i >= 12 && i < 15) || (i == j && i >= 9 && i < 15);
}
enum { JOB_NOM = 1, JOB_P = 2, JOB_V = 4, JOB_TH = 8 };
constexpr int job_stage_mask(int jobs) { return ((jobs & JOB_P) ? 1 : 0) | ((jobs & JOB_V) ? 2 : 0) | ((jobs & JOB_TH) ? 4 : 0); }

template <typename T, int N>
struct TeamRec {
    using RC = Rec<T, N>;
    static constexpr int NP = N * (N + 1) / 2;
    static constexpr int NCC = RC::NCOVP / RC::EPC;                 // covariance chunks (incl. prev id / padding)
    static_assert(RC::EPC == 4, "team kernels are fp32 only");
    // 4-bit mask of the elements of covariance chunk cc that stage S writes
    static constexpr int write_mask(int S, int cc)
    {
        int m = 0;
        for (int k = 0; k < 4; ++k) {
            const int e = 4 * cc + k;
            if (e < NP && stage_of(cov_row<N>(e), cov_col<N>(e)) == S) m |= 1 << k;
        }
        return m;
    }
    // does a role with the stage set `stages` (bit S) read anything in covariance chunk cc?
    static constexpr bool reads_chunk(int stages, int cc)
    {
        for (int k = 0; k < 4; ++k) {
            const int e = 4 * cc + k;
            if (e >= NP) continue;
            for (int S = 0; S < 3; ++S)
                if ((stages >> S & 1) && stage_reads(S, cov_row<N>(e), cov_col<N>(e))) return true;
        }
        return false;
    }
};

// byte offset of element k of covariance chunk cc inside the lane's tile (see store_chunks: group offset in the VGPR part)
template <typename T, int N>
__device__ __forceinline__ unsigned cov_chunk_off(unsigned lane, int cc)
{
    const int c = Rec<T, N>::CH_NOM + cc;
    return lane * 16u + (unsigned)((c >> 2) * 4096 + (c & 3) * 1024);
}

// the elements of stage S, chunk by chunk: whole chunks as one 16-byte store, mixed chunks (a chunk that holds the end of
// one row group and the start of the next, or the collected diagonals) as 8- and 4-byte stores of exactly the elements
// this stage owns -- another role stores the other bytes of the same chunk
template <typename T, int N, int S, int AUX>
__device__ __forceinline__ void store_stage(__amdgpu_buffer_rsrc_t rs, unsigned lane, const T* P)
{
    using TR = TeamRec<T, N>;
    static_for<0, TR::NCC>([&](auto cc_) {
        constexpr int cc = decltype(cc_)::value;
        constexpr int m = TR::write_mask(S, cc);
        if constexpr (m != 0) {
            const unsigned off = cov_chunk_off<T, N>(lane, cc);
            const unsigned* w = reinterpret_cast<const unsigned*>(P + 4 * cc);
            if constexpr (m == 0xF) {
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{ w[0], w[1], w[2], w[3] }, rs, off, 0, AUX);
            } else {
                if constexpr ((m & 3) == 3) __builtin_amdgcn_raw_buffer_store_b64(u32x2{ w[0], w[1] }, rs, off, 0, AUX);
                else {
                    if constexpr ((m & 1) != 0) __builtin_amdgcn_raw_buffer_store_b32(w[0], rs, off, 0, AUX);
                    if constexpr ((m & 2) != 0) __builtin_amdgcn_raw_buffer_store_b32(w[1], rs, off + 4u, 0, AUX);
                }
                if constexpr ((m & 12) == 12) __builtin_amdgcn_raw_buffer_store_b64(u32x2{ w[2], w[3] }, rs, off + 8u, 0, AUX);
                else {
                    if constexpr ((m & 4) != 0) __builtin_amdgcn_raw_buffer_store_b32(w[2], rs, off + 8u, 0, AUX);
                    if constexpr ((m & 8) != 0) __builtin_amdgcn_raw_buffer_store_b32(w[3], rs, off + 12u, 0, AUX);
                }
            }
        }
    });
}

// the covariance chunks a role with the stage set STAGES reads -> P (storage order, unread elements stay untouched)
template <typename T, int N, int STAGES, int AUX>
__device__ __forceinline__ void load_stage_chunks(__amdgpu_buffer_rsrc_t rs, unsigned lane, T* P)
{
    using TR = TeamRec<T, N>;
    constexpr int CN = Rec<T, N>::CH_NOM;
    static_for<0, TR::NCC>([&](auto cc_) {
        constexpr int cc = decltype(cc_)::value;
        if constexpr (TR::reads_chunk(STAGES, cc)) load_chunks<T, N, CN + cc, CN + cc + 1, AUX>(rs, lane, P + 4 * cc);
    });
}

// LDS exchange points of a team: the writes of this wave are done (lgkmcnt) and every wave of the workgroup has arrived.
// NOT __syncthreads(): that also drains vmcnt, i.e. waits for every global load in flight -- the record loads these
// kernels deliberately keep in flight across their exchanges.
__device__ __forceinline__ void team_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// =================================================================================
// predict, one step per launch: independent stage roles, no exchange of data (one barrier between loads and stores)
// =================================================================================
//   NR = 2:  { rows v }  { nominal state, rows p, rows theta }
//   NR = 3:  { rows v }  { rows p, rows theta }  { nominal state }
//   NR = 4:  { rows v }  { rows p }  { rows theta }  { nominal state }
// Every role evaluates predict_nominal on the same inputs for the coefficient blocks A, Bm, Th (what it does not use is
// dead code); only the nominal role stores the nominal state.
template <int NR> constexpr int predict_jobs(int role)
{
    if (NR == 2) return role == 0 ? JOB_V : (JOB_NOM | JOB_P | JOB_TH);
    if (NR == 3) return role == 0 ? JOB_V : (role == 1 ? (JOB_P | JOB_TH) : JOB_NOM);
    return role == 0 ? JOB_V : (role == 1 ? JOB_P : (role == 2 ? JOB_TH : JOB_NOM));
}

template <typename T, int N, int DIALECT, int NR, int LD, int ST>
__global__ void __launch_bounds__(64 * NR)
predict_team_kernel(T* __restrict__ recs, int B, const T* __restrict__ accel, const T* __restrict__ gyro,
                    const T* __restrict__ dt, int dt_stride, DevConst<T> dc)
{
    using RC = Rec<T, N>;
    const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned tile = blockIdx.x;
    const int b0 = (int)(tile * 64u + lane);
    const bool live = b0 < B;
    const int b = live ? b0 : (int)(tile * 64u);       // lanes past B run along on the tile's first filter and store nothing
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, tile);
    const size_t o = (size_t)b * 3;
    const T a[3] = { ld_once(accel + o), ld_once(accel + o + 1), ld_once(accel + o + 2) };
    const T w[3] = { ld_once(gyro + o), ld_once(gyro + o + 1), ld_once(gyro + o + 2) };
    const T h = dt_stride ? ld_once(dt + b) : dt[0];
    T nom[Lay<N>::NNOM], P[RC::NCOVP];
    load_chunks<T, N, 0, RC::CH_NOM, LD>(rs, lane, nom);
    auto run = [&](auto jobs_) {
        constexpr int JOBS = decltype(jobs_)::value;
        load_stage_chunks<T, N, job_stage_mask(JOBS), LD>(rs, lane, P);
        PredictCoef<T> k;
        predict_nominal<T, N, DIALECT, FBUS_X_PACK_TEAM>(nom, a, w, h, k);
        // "No exchange" does not mean "no ordering": a stage reads the PRE-step values of rows another role owns, so no role may
        // store before every role's loads have landed.  One barrier behind the loads (vmcnt(0): the data is in registers);
        // without it the kernel was right at 4096 / 16 384 filters and wrong at 32 768 (block-wise covariance error 3e-2:
        // role p's loads of rows v, queued behind a busy memory system, came back with what role v had already stored).
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr ((JOBS & JOB_NOM) != 0) {
            if (live) store_chunks<T, N, 0, RC::CH_KIN, ST>(rs, lane, nom);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr ((JOBS & JOB_P) != 0) {
            cov_stage_p<T, N, FBUS_X_PACK_TEAM>(P, k);
            if (live) store_stage<T, N, 0, ST>(rs, lane, P);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr ((JOBS & JOB_V) != 0) {
            cov_stage_v<T, N, FBUS_X_PACK_TEAM>(P, k, dc.qd);
            if (live) store_stage<T, N, 1, ST>(rs, lane, P);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr ((JOBS & JOB_TH) != 0) {
            cov_stage_th<T, N, FBUS_X_PACK_TEAM>(P, k, dc.qd);
            if (live) store_stage<T, N, 2, ST>(rs, lane, P);
        }
    };
    if (role == 0) run(std::integral_constant<int, predict_jobs<NR>(0)>{});
    else if (role == 1) run(std::integral_constant<int, predict_jobs<NR>(1)>{});
    else if (NR > 2 && role == 2) run(std::integral_constant<int, predict_jobs<NR>(NR > 2 ? 2 : 0)>{});
    else if (NR > 3) run(std::integral_constant<int, predict_jobs<NR>(NR > 3 ? 3 : 0)>{});
}

// =================================================================================
// predict_n: K steps per launch, four roles, rows handed on through LDS between steps
// =================================================================================
// All three covariance roles work on the same step in the same iteration; what a stage reads of ANOTHER stage's rows are
// their values at the start of the step, i.e. what that role produced in the previous iteration:
//     role theta  --(rows theta: 33 values)-->  role v  --(rows v: 42 values)-->  role p
// and the nominal role runs one step ahead, handing the 28 coefficients (A, Bm, Th, dt) of step k to everybody.
// Iteration t = 0 .. K: nominal computes the coefficients of step t, the covariance roles run step t - 1; one barrier
// per iteration, double-buffered LDS.  Everything else a stage reads is either its own rows or invariant under
// ImuUpdate (rows ba, bg, g) except the ba diagonal's + Q, which role v applies to its private copy as well.
// Same device functions, same operands as the one-wave predict_n (equal up to FMA contraction, see above).
template <typename T, int N>
struct StepXch {
    static constexpr int QC = 7;                             // the 28 coefficients (A, Bm, Th, dt): 7 x 16 bytes
    static constexpr int E_V0 = cov_final_before_row<N>(3) / 4 * 4, E_V1 = (cov_final_before_row<N>(6) + 3) / 4 * 4;   // chunk-aligned cover of rows v
    static constexpr int E_T0 = cov_final_before_row<N>(6) / 4 * 4, E_T1 = (cov_final_before_row<N>(9) + 3) / 4 * 4;   // ... of rows theta
    static constexpr int QV = (E_V1 - E_V0) / 4, QT = (E_T1 - E_T0) / 4;
    // even N: the odd-row diagonals (3,3), (5,5) / (7,7) live behind the rows and travel in one more 16-byte cell each
    static constexpr bool DIAG_APART = (N % 2 == 0);
    static constexpr int QVX = QV + (DIAG_APART ? 1 : 0), QTX = QT + (DIAG_APART ? 1 : 0);
    static constexpr int PER_SLOT = QC + QVX + QTX;          // 16-byte cells per buffer and lane
    u32x4* mem;                                              // [2][PER_SLOT][64], this lane's column
    __device__ __forceinline__ u32x4* cell(int buf, int q) const { return mem + (buf * PER_SLOT + q) * 64; }
    __device__ __forceinline__ void put4(int buf, int q, const T* src) const
    {
        u32x4 v;
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = src[i];
        *cell(buf, q) = v;
    }
    __device__ __forceinline__ void get4(int buf, int q, T* dst) const
    {
        const u32x4 v = *cell(buf, q);
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = e[i];
    }
    // the 28 coefficients of a step: nominal role -> everybody
    __device__ __forceinline__ void put_coef(int buf, const PredictCoef<T>& k) const
    {
        T c[28];
#pragma unroll
        for (int i = 0; i < 9; ++i) { c[i] = k.A[i]; c[9 + i] = k.Bm[i]; c[18 + i] = k.Th[i]; }
        c[27] = k.dt;
#pragma unroll
        for (int q = 0; q < QC; ++q) put4(buf, q, c + 4 * q);
    }
    __device__ __forceinline__ void get_coef(int buf, PredictCoef<T>& k) const
    {
        T c[28];
#pragma unroll
        for (int q = 0; q < QC; ++q) get4(buf, q, c + 4 * q);
#pragma unroll
        for (int i = 0; i < 9; ++i) { k.A[i] = c[i]; k.Bm[i] = c[9 + i]; k.Th[i] = c[18 + i]; }
        k.dt = c[27];
    }
    // iteration t (1 .. K) of the three covariance roles: step t - 1 on this role's rows; what the next role reads of them goes to
    // buffer t & 1 (not behind the last step).  The caller puts the barrier behind it.
    __device__ __forceinline__ void step_theta(T* P, PredictCoef<T>& k, const T* qd, int t, int K) const
    {
        get_coef((t - 1) & 1, k);
        cov_stage_th<T, N, FBUS_X_PACK_TEAM>(P, k, qd);
        if (t < K) {
#pragma unroll
            for (int q = 0; q < QT; ++q) put4(t & 1, QC + QVX + q, P + E_T0 + 4 * q);
            if constexpr (DIAG_APART) { const T d[4] = { P[pidx<N>(7, 7)], T(0), T(0), T(0) }; put4(t & 1, QC + QVX + QT, d); }
        }
    }
    __device__ __forceinline__ void step_v(T* P, PredictCoef<T>& k, const T* qd, int t, int K) const
    {
        get_coef((t - 1) & 1, k);
        if (t > 1) {
            // rows theta as role theta left them after step t - 2 (the cover [E_T0, E_T1) starts with the last elements of rows v
            // and may end with non-theta elements of the same chunk: only theta elements are taken)
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                T v4[4];
                get4((t - 1) & 1, QC + QVX + q, v4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = E_T0 + 4 * q + i;
                    if (e < N * (N + 1) / 2 && cov_row<N>(e) >= 6 && cov_row<N>(e) < 9) P[e] = v4[i];
                }
            }
            if constexpr (DIAG_APART) { T d[4]; get4((t - 1) & 1, QC + QVX + QT, d); P[pidx<N>(7, 7)] = d[0]; }
            // the ba diagonal's + Q of step t - 2 (role theta owns and stores it; this is the private copy stage v reads)
#pragma unroll
            for (int i = 9; i < 12; ++i) P[pidx<N>(i, i)] += qd[2];
        }
        cov_stage_v<T, N, FBUS_X_PACK_TEAM>(P, k, qd);
        if (t < K) {
#pragma unroll
            for (int q = 0; q < QV; ++q) put4(t & 1, QC + q, P + E_V0 + 4 * q);
            if constexpr (DIAG_APART) { const T d[4] = { P[pidx<N>(3, 3)], P[pidx<N>(5, 5)], T(0), T(0) }; put4(t & 1, QC + QV, d); }
        }
    }
    __device__ __forceinline__ void step_p(T* P, PredictCoef<T>& k, int t) const
    {
        get_coef((t - 1) & 1, k);
        if (t > 1) {
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                T v4[4];
                get4((t - 1) & 1, QC + q, v4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = E_V0 + 4 * q + i;
                    if (e < N * (N + 1) / 2 && cov_row<N>(e) >= 3 && cov_row<N>(e) < 6) P[e] = v4[i];
                }
            }
            if constexpr (DIAG_APART) { T d[4]; get4((t - 1) & 1, QC + QV, d); P[pidx<N>(3, 3)] = d[0]; P[pidx<N>(5, 5)] = d[1]; }
        }
        cov_stage_p<T, N, FBUS_X_PACK_TEAM>(P, k);
    }
};

template <typename T, int N, int DIALECT>
__global__ void __launch_bounds__(256)
predict_n_team_kernel(T* __restrict__ recs, int B, int K, const T* __restrict__ accel, const T* __restrict__ gyro,
                      const T* __restrict__ dt, int dt_stride, DevConst<T> dc)
{
    using RC = Rec<T, N>;
    using X = StepXch<T, N>;
    constexpr int CN = RC::CH_NOM;
    const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned tile = blockIdx.x;
    const int b0 = (int)(tile * 64u + lane);
    const bool live = b0 < B;
    const int b = live ? b0 : (int)(tile * 64u);          // lanes past B run along on the tile's first filter, store nothing
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, tile);
    __shared__ u32x4 xmem[2 * X::PER_SLOT * 64];
    const X xch{ xmem + lane };
    T nom[Lay<N>::NNOM], P[RC::NCOVP];
    PredictCoef<T> k;
    if (role == 3) {
        // ---- nominal role: one step ahead of the covariance roles -------------------------------------------------
        load_chunks<T, N, 0, CN, AUX_NT>(rs, lane, nom);
        ImuSample<T> cur;
        if (K > 0) cur.load(accel, gyro, dt, dt_stride, 0, B, b);
#pragma unroll 1
        for (int t = 0; t <= K; ++t) {
            if (t < K) {
                predict_nominal<T, N, DIALECT, FBUS_X_PACK_TEAM>(nom, cur.a, cur.w, cur.h, k);
                if (t + 1 < K) cur.load(accel, gyro, dt, dt_stride, t + 1, B, b);
                xch.put_coef(t & 1, k);
            }
            team_barrier();
        }
        if (live) store_chunks<T, N, 0, RC::CH_KIN>(rs, lane, nom);
    } else if (role == 2) {
        // ---- rows theta (+ Q diagonals): self-contained, publishes its rows for role v ------------------------------
        load_stage_chunks<T, N, 4, AUX_NT>(rs, lane, P);
        team_barrier();                                                  // iteration 0
#pragma unroll 1
        for (int t = 1; t <= K; ++t) {
            xch.step_theta(P, k, dc.qd, t, K);
            team_barrier();
        }
        if (live && K > 0) store_stage<T, N, 2, AUX_DEFAULT>(rs, lane, P);
    } else if (role == 0) {
        // ---- rows v: reads role theta's rows of the previous step, publishes its own for role p ----------------------
        load_stage_chunks<T, N, 2, AUX_NT>(rs, lane, P);
        team_barrier();
#pragma unroll 1
        for (int t = 1; t <= K; ++t) {
            xch.step_v(P, k, dc.qd, t, K);
            team_barrier();
        }
        if (live && K > 0) store_stage<T, N, 1, AUX_DEFAULT>(rs, lane, P);
    } else {
        // ---- rows p: reads role v's rows of the previous step ---------------------------------------------------------
        load_stage_chunks<T, N, 1, AUX_NT>(rs, lane, P);
        team_barrier();
#pragma unroll 1
        for (int t = 1; t <= K; ++t) {
            xch.step_p(P, k, t);
            team_barrier();
        }
        if (live && K > 0) store_stage<T, N, 0, AUX_DEFAULT>(rs, lane, P);
    }
}

// =================================================================================
// correct: one-shot information form, roles share the fold, the rows of W and the elements of P
// =================================================================================
// Lam, b -> Z (6 x 6, row-major) and gamma (6):  Lam = Lc Lc' (a pivot that is not clearly positive relative to its
// original diagonal carries no information and is dropped, as in joint_factor), G = Lc' P_JJ Lc, C C' = I + G,
// Z = Lc C^-T, gamma = C^-1 Lc^-1 b.  I + G has its eigenvalues >= 1: the second factorisation never meets a small pivot.
// PJ = P_JJ in lidx order (J-positions 0..5 = state columns jcol(0..5)).
__device__ __forceinline__ float team_rsq(float x)
{
    float r = __builtin_amdgcn_rsqf(x);
    return r * (1.5f - 0.5f * x * r * r);                      // one Newton step on the 1-ulp hardware value
}
template <typename T>
__device__ __forceinline__ void info_gain(InfoAcc<T>& acc, const T* PJ, T (&Z)[36], T (&gam)[6])
{
    T (&A)[21] = acc.Lam;
    T (&bb)[6] = acc.b;
    T Lc[21], bt[6], dg0[6];                                     // Lc(i, a), i >= a, at lidx(a, i)
    const T tiny = T(2e-6);
#pragma unroll
    for (int a = 0; a < 6; ++a) dg0[a] = A[lidx(a, a)];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const T piv = A[lidx(a, a)];
        const bool ok = piv > tiny * dg0[a];
        const T r = ok ? team_rsq(piv) : T(0);
        Lc[lidx(a, a)] = piv * r;
#pragma unroll
        for (int i = a + 1; i < 6; ++i) Lc[lidx(a, i)] = A[lidx(a, i)] * r;
        bt[a] = bb[a] * r;
#pragma unroll
        for (int i = a + 1; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) A[lidx(i, j)] -= Lc[lidx(a, i)] * Lc[lidx(a, j)];
            bb[i] -= Lc[lidx(a, i)] * bt[a];
        }
    }
    auto pj = [&](int i, int k) { return PJ[i <= k ? lidx(i, k) : lidx(k, i)]; };
    T T1[36];                                                    // P_JJ Lc
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            T s = pj(i, a) * Lc[lidx(a, a)];
#pragma unroll
            for (int k2 = a + 1; k2 < 6; ++k2) s += pj(i, k2) * Lc[lidx(a, k2)];
            T1[6 * i + a] = s;
        }
    T G[21];                                                     // I + Lc' P_JJ Lc, upper triangle
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = a; c < 6; ++c) {
            T s = Lc[lidx(a, a)] * T1[6 * a + c];
#pragma unroll
            for (int k2 = a + 1; k2 < 6; ++k2) s += Lc[lidx(a, k2)] * T1[6 * k2 + c];
            G[lidx(a, c)] = (a == c) ? s + T(1) : s;
        }
    T Cf[21], rc[6];                                             // C(i, a), i >= a, at lidx(a, i)
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        T s = G[lidx(a, a)];
#pragma unroll
        for (int m = 0; m < a; ++m) s -= Cf[lidx(m, a)] * Cf[lidx(m, a)];
        rc[a] = team_rsq(s);
        Cf[lidx(a, a)] = s * rc[a];
#pragma unroll
        for (int i = a + 1; i < 6; ++i) {
            T v = G[lidx(a, i)];
#pragma unroll
            for (int m = 0; m < a; ++m) v -= Cf[lidx(m, i)] * Cf[lidx(m, a)];
            Cf[lidx(a, i)] = v * rc[a];
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < 6; ++k2)                               // Z C' = Lc, row by row
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            T v = (a <= k2) ? Lc[lidx(a, k2)] : T(0);
#pragma unroll
            for (int m = 0; m < a; ++m) v -= Z[6 * k2 + m] * Cf[lidx(m, a)];
            Z[6 * k2 + a] = v * rc[a];
        }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        T v = bt[a];
#pragma unroll
        for (int m = 0; m < a; ++m) v -= Cf[lidx(m, a)] * gam[m];
        gam[a] = v * rc[a];
    }
}

// who does what in a correct team of NR roles
template <int N, int NR>
struct CorrectPlan {
    static constexpr int NN = (N + 1) / 2 * 2;                              // rows of W padded to even
    static constexpr int NCC = (Rec<float, N>::NCOVP) / 4;
    // rows [row0(r), row0(r + 1)) of W, even boundaries (role 0 also computes dx and injects: fewer elements of P)
    static constexpr int row0(int r)
    {
        if (r <= 0) return 0;
        if (r >= NR) return N;
        if (N == 18) {
            if (NR == 2) return 8;
            if (NR == 3) return r == 1 ? 6 : 12;
            return r == 1 ? 6 : (r == 2 ? 10 : 14);
        }
        if (NR == 2) return 8;
        if (NR == 3) return r == 1 ? 6 : 10;
        return 4 * r;
    }
    // covariance chunks [ch0(r), ch0(r + 1)) whose elements role r updates and stores
    static constexpr int ch0(int r)
    {
        if (r <= 0) return 0;
        if (r >= NR) return NCC;
        const int w43 = (NR == 2) ? 16 : (NR == 3 ? (r == 1 ? 9 : 26) : (r == 1 ? 6 : (r == 2 ? 18 : 31)));
        return (w43 * NCC + 21) / 43;
    }
    // is (i, j), i <= j, one of V(row, J) for a row of role r?   V(i, c) = P(min(i, c), max(i, c)), c in J = {0,1,2,6,7,8}
    static constexpr bool is_v_elem(int r, int i, int j)
    {
        const bool ij = (i < 3) || (i >= 6 && i < 9), jj = (j < 3) || (j >= 6 && j < 9);
        return (ij && j >= row0(r) && j < row0(r + 1)) || (jj && i >= row0(r) && i < row0(r + 1));
    }
    static constexpr bool loads_chunk(int r, int cc)
    {
        if (cc >= ch0(r) && cc < ch0(r + 1)) return true;
        for (int k = 0; k < 4; ++k) {
            const int e = 4 * cc + k;
            if (e < N * (N + 1) / 2 && is_v_elem(r, cov_row<N>(e), cov_col<N>(e))) return true;
        }
        return false;
    }
};

// LDS of a correct team (per lane columns, 64 lanes)
template <int N, int NR>
struct CorrectXch {
    static constexpr int NN = CorrectPlan<N, NR>::NN;
    static constexpr int QW = (6 * NN + 3) / 4;                 // W, a-major [a][i]: 16-byte cells
    static constexpr int NPART = 26;                            // PoseFold::NVAL: the partial sums of one role's markers
    u32x4* w;                                                   // [QW][64]
    float* part;                                                // [NR][NPART][64]
    float* pjj;                                                 // [21][64]
};

// (correct_team_kernel -- the pose rows' one-shot update P - W W' divided over 2-4 waves per tile -- was removed in round 4: parity-green
// since round 3 but slower than the one-wave kernel at every batch size (4096 filters 6.4 -> 7.6 us, 16 384: 7.2 -> 9.2, 32 768: 10.0 -> 17.9),
// and never selected.  Its building blocks live on in frames_team_kernel below.)


// =================================================================================
// frame window: F x { K_f ImuUpdates, one MeasureUpdate } per launch -- the predict pipeline of predict_n_team_kernel and the
// one-shot correct of correct_team_kernel, the covariance handed from phase to phase through LDS
// =================================================================================
// The fused frame / frame window kernels of ekf_kernels.hpp keep one filter's whole record in one lane from the first load
// to the last store; below 1024 tiles most SIMDs idle while each wave walks through ~1150 instructions per ImuUpdate and
// ~2100 per MeasureUpdate.  Here (FBUS_EKF.m:151-210 ; filter.cpp:229-235):
//   predict phase   the four-role pipeline of predict_n_team_kernel.  Behind the last step of a frame every covariance role
//                   writes the rows it owns to the LDS "image" of the covariance (43 cells of 16 bytes per lane); role v adds
//                   the predict-invariant part (rows ba, bg, g off the diagonal, the previous marker id), of which it holds
//                   a current copy anyway.
//   correct phase   the nominal role finished its last predict_nominal one iteration earlier and has chosen the marker(s)
//                   and folded their rows (PoseFold -> Lam, b: 27 values, left in LDS) while the others were on the last
//                   step.  Every role reads the image, solves the 6 x 6 problem (info_gain, redundantly), computes its rows
//                   of W = V Z, exchanges them, applies P -= W W' to its chunks (CorrectPlan<N, 4>) and writes them back
//                   to the image; the nominal role also forms dx = W gamma and injects it.
//   next frame      every role reads from the image what its predict stage reads; after the last frame the roles store
//                   their chunks of the record.
// K_f + 4 barriers per frame.  The posterior is that of the sequential passes (see correct_team_kernel).  LDS: the image
// and W are laid over the exchange buffers of the predict pipeline -- 77 cells per lane + the marker map = 80 KiB, and at
// most 256 registers: two workgroups per CU.
template <typename T, int N>
struct FrameImage {
    using X = StepXch<T, N>;
    static constexpr int NCC = Rec<T, N>::NCOVP / 4;
    static constexpr int QW = CorrectXch<N, 4>::QW;
    static constexpr int C_W = 0;                                 // W (correct phase); the head of exchange buffer 0 otherwise
    static constexpr int C_IMG = QW;                              // [NCC] the covariance between the phases
    // Lam (21), b (6), the new previous-marker id: written while the exchange buffers are in use and read while W and the
    // image are: behind all three
    static constexpr int C_LAM = (2 * X::PER_SLOT > QW + NCC) ? 2 * X::PER_SLOT : QW + NCC;
    static constexpr int CELLS = C_LAM + 7;
    // the first coefficients of the next frame (buffer 0, cells [0, QC)) are written while the image is being read
    static_assert(X::QC <= QW, "the coefficient cells of buffer 0 must lie inside the W area");
    static constexpr size_t bytes() { return (size_t)CELLS * 64 * 16 + sizeof(MarkerLDS<T>); }
    // the elements role v leaves in the image behind a frame's last step: its own (stage 1), the predict-invariant ones
    // (stage 3) and the padding behind the packed covariance (the previous marker id lives there)
    // ... which is why role v reads the padding chunk as well (N = 15: a chunk of its own that no stage reads)
    static constexpr bool pad_chunk(int cc) { return 4 * cc + 3 >= N * (N + 1) / 2; }
    static constexpr int v_mask(int cc)
    {
        int m = TeamRec<T, N>::write_mask(1, cc) | TeamRec<T, N>::write_mask(3, cc);
        for (int q = 0; q < 4; ++q)
            if (4 * cc + q >= N * (N + 1) / 2) m |= 1 << q;
        return m;
    }
};

template <typename T, int N, int DIALECT>
__global__ void __launch_bounds__(256, 2)
frames_team_kernel(T* __restrict__ recs, int B, int F, FrameCounts kc, const T* __restrict__ accel, const T* __restrict__ gyro,
                    const T* __restrict__ dt, int dt_stride, int M, const int* __restrict__ ids, const T* __restrict__ pos,
                    const T* __restrict__ quat, int mode, const unsigned char* __restrict__ skip,
                    unsigned char* __restrict__ applied, DevConst<T> dc)
{
    using L = Lay<N>;
    using RC = Rec<T, N>;
    using X = StepXch<T, N>;
    using FI = FrameImage<T, N>;
    using PL = CorrectPlan<N, 4>;
    constexpr int CN = RC::CH_NOM, NN = PL::NN, NP = N * (N + 1) / 2, NCC = FI::NCC;
    const unsigned role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned tile = blockIdx.x;
    const int b0 = (int)(tile * 64u + lane);
    const bool live = b0 < B;
    const int b = live ? b0 : (int)(tile * 64u);
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, tile);
    extern __shared__ u32x4 dyn_lds[];
    u32x4* const xmem = dyn_lds;
    MarkerLDS<T>& tbl = *reinterpret_cast<MarkerLDS<T>*>(dyn_lds + FI::CELLS * 64);
    const X xch{ xmem + lane };
    u32x4* const img = xmem + FI::C_IMG * 64 + lane;              // cell cc of this lane: img[cc * 64]
    float* const lam = reinterpret_cast<float*>(xmem + FI::C_LAM * 64 + lane);   // value i of this lane: cell i / 4, slot i % 4
    u32x4* const wx = xmem + FI::C_W * 64 + lane;                 // W, a-major, QW cells
    T nom[L::NNOM], P[RC::NCOVP];
    PredictCoef<T> k;
    auto lam_at = [&](int i) -> float& { return lam[(i / 4) * 256 + (i % 4)]; };
    // the elements a role owns behind the predict phase -> image (whole chunks as one 16-byte write); S = its stage
    auto img_put_stage = [&](auto s_) {
        constexpr int S = decltype(s_)::value;
        using TR = TeamRec<T, N>;
        static_for<0, TR::NCC>([&](auto cc_) {
            constexpr int cc = decltype(cc_)::value;
            constexpr int m = (S == 1) ? FI::v_mask(cc) : TR::write_mask(S, cc);
            if constexpr (m != 0) {
                const unsigned* w = reinterpret_cast<const unsigned*>(P + 4 * cc);
                if constexpr (m == 0xF) img[cc * 64] = u32x4{ w[0], w[1], w[2], w[3] };
                else {
                    unsigned* e = reinterpret_cast<unsigned*>(img + cc * 64);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((m >> q) & 1) e[q] = w[q];
                }
            }
        });
    };
    auto img_get_stage_chunks = [&](auto st_) {
        constexpr int STAGES = decltype(st_)::value;
        using TR = TeamRec<T, N>;
        static_for<0, TR::NCC>([&](auto cc_) {
            constexpr int cc = decltype(cc_)::value;
            if constexpr (TR::reads_chunk(STAGES, cc) || (STAGES == 2 && FI::pad_chunk(cc))) {
                const u32x4 v = img[cc * 64];
                const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
                for (int q = 0; q < 4; ++q) P[4 * cc + q] = e[q];
            }
        });
    };
    auto img_get_all = [&]() {
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            const u32x4 v = img[cc * 64];
            const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int q = 0; q < 4; ++q) P[4 * cc + q] = e[q];
        }
    };
    // ---- the correct phase of one role (CR = its part in CorrectPlan<N, 4>; part 0 is the nominal role) --------------------
    // returns with this role's chunks [ch0(CR), ch0(CR + 1)) of P updated in registers; dx for part 0
    auto correct_part = [&](auto cr_, T* dx) {
        constexpr int CR = decltype(cr_)::value;
        img_get_all();
        InfoAcc<T> acc;
#pragma unroll
        for (int i = 0; i < 21; ++i) acc.Lam[i] = lam_at(i);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc.b[i] = lam_at(21 + i);
        const T np = lam_at(27);
        T PJ[21];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int c = i; c < 6; ++c) PJ[lidx(i, c)] = P[pidx<N>(jcol(i), jcol(c))];
        T Z[36], gam[6];
        info_gain<T>(acc, PJ, Z, gam);
        constexpr int I0 = PL::row0(CR), I1 = PL::row0(CR + 1);
        static_for<0, 6>([&](auto a_) {
            constexpr int a = decltype(a_)::value;
            T wa[NN];
            if constexpr (PackedMath<T, N>::on) {
#pragma unroll
                for (int i = I0; i < I1; i += 2) {
                    PairAcc<N> pa;
#pragma unroll
                    for (int k2 = 0; k2 < 6; ++k2) pa.add(P, Z[6 * k2 + a], jcol(k2), i);
                    const f32x2 v = pa.get();
                    wa[i] = v.x; wa[i + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int i = I0; i < I1; ++i) {
                    T s2 = T(0);
#pragma unroll
                    for (int k2 = 0; k2 < 6; ++k2) s2 += P[pidx<N>(i, jcol(k2))] * Z[6 * k2 + a];
                    wa[i] = s2;
                }
                if constexpr ((I1 & 1) != 0) wa[I1] = T(0);
            }
            float* wl = reinterpret_cast<float*>(wx);
            constexpr int I1P = (I1 + 1) / 2 * 2;
#pragma unroll
            for (int i = I0; i < I1P; i += 2) {
                const int e = a * NN + i;
                *reinterpret_cast<f32x2*>(wl + (e / 4) * 256 + (e % 4)) = f32x2{ wa[i], wa[i + 1] };
            }
        });
        team_barrier();                                             // all of W is in LDS
        T Wt[FI::QW * 4];
#pragma unroll
        for (int q = 0; q < FI::QW; ++q) {
            const u32x4 v = wx[q * 64];
            const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int i = 0; i < 4; ++i) Wt[4 * q + i] = e[i];
        }
        constexpr int C0 = PL::ch0(CR), C1 = PL::ch0(CR + 1);
        static_for<C0, C1>([&](auto cc_) {
            constexpr int cc = decltype(cc_)::value;
            static_for<0, 2>([&](auto h_) {
                constexpr int e = 4 * cc + 2 * decltype(h_)::value;
                if constexpr (e + 1 < NP && PackedMath<T, N>::on && cov_row<N>(e) == cov_row<N>(e + 1) && cov_col<N>(e + 1) == cov_col<N>(e) + 1 &&
                              cov_col<N>(e) % 2 == 0) {
                    constexpr int i = cov_row<N>(e), c = cov_col<N>(e);
                    f32x2 v = f32x2{ P[e], P[e + 1] };
#pragma unroll
                    for (int a = 0; a < 6; ++a) v -= Wt[a * NN + i] * f32x2{ Wt[a * NN + c], Wt[a * NN + c + 1] };
                    P[e] = v.x; P[e + 1] = v.y;
                } else {
                    static_for<0, 2>([&](auto s_) {
                        constexpr int e1 = e + decltype(s_)::value;
                        if constexpr (e1 < NP) {
                            constexpr int i = cov_row<N>(e1), c = cov_col<N>(e1);
                            T v = P[e1];
#pragma unroll
                            for (int a = 0; a < 6; ++a) v -= Wt[a * NN + i] * Wt[a * NN + c];
                            P[e1] = v;
                        }
                    });
                }
            });
        });
        if constexpr (C1 == NCC) {
            if (np >= T(0)) P[L::OFF_PREV - L::OFF_COV] = np;       // filter.cpp:675
        }
        if constexpr (CR == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                T s2 = Wt[i] * gam[0];
#pragma unroll
                for (int a = 1; a < 6; ++a) s2 += Wt[a * NN + i] * gam[a];
                dx[i] = s2;
            }
        }
    };
    auto img_put_part = [&](auto cr_) {
        constexpr int CR = decltype(cr_)::value;
        constexpr int C0 = PL::ch0(CR), C1 = PL::ch0(CR + 1);
#pragma unroll
        for (int cc = C0; cc < C1; ++cc) {
            const unsigned* w = reinterpret_cast<const unsigned*>(P + 4 * cc);
            img[cc * 64] = u32x4{ w[0], w[1], w[2], w[3] };
        }
    };
    auto store_part = [&](auto cr_, bool from_image) {
        constexpr int CR = decltype(cr_)::value;
        constexpr int C0 = PL::ch0(CR), C1 = PL::ch0(CR + 1);
        if (from_image) {
#pragma unroll
            for (int cc = C0; cc < C1; ++cc) {
                const u32x4 v = img[cc * 64];
                const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
                for (int q = 0; q < 4; ++q) P[4 * cc + q] = e[q];
            }
        }
        if (live) store_chunks<T, N, CN + C0, CN + C1, FBUS_X_FRAME_ST>(rs, lane, P + 4 * C0);
    };
    // what a covariance role does behind the last predict step of a frame (CR: its part of the correct); true = leave the kernel
    auto frame_tail = [&](auto cr_, auto stage_, auto reads_, int f) -> bool {
        img_put_stage(stage_);
        team_barrier();                                             // the predicted covariance is in the image
        const bool last_frame = f + 1 == F;
        if (M > 0) {
            T dx_unused[1];
            correct_part(cr_, dx_unused);
            if (last_frame) { store_part(cr_, false); return true; }
            img_put_part(cr_);
            team_barrier();                                         // the posterior is in the image
        } else if (last_frame) {
            store_part(cr_, true);
            return true;
        }
        img_get_stage_chunks(reads_);
        return false;
    };
    using I0_ = std::integral_constant<int, 0>; using I1_ = std::integral_constant<int, 1>; using I2_ = std::integral_constant<int, 2>;
    using I3_ = std::integral_constant<int, 3>; using I4_ = std::integral_constant<int, 4>;
    if (role == 3) {
        // ---- nominal state, the fold of the measurements, part 0 of the correct ------------------------------------------
        {
            constexpr int NI = (int)sizeof(short) * (FBUS_MAX_MARKER_ID + 1) / 16, NM = (int)sizeof(T) * FBUS_MAX_MARKERS * MK_STRIDE / 16;
            constexpr int PI = (NI + 63) / 64, PM = (NM + 63) / 64;
            const u32x4* si = reinterpret_cast<const u32x4*>(dc.id2slot);
            const u32x4* sm = reinterpret_cast<const u32x4*>(dc.mk);
            u32x4 vi[PI], vm[PM];
#pragma unroll
            for (int q = 0; q < PI; ++q) { const int i = (int)lane + q * 64; vi[q] = si[i < NI ? i : 0]; }
#pragma unroll
            for (int q = 0; q < PM; ++q) { const int i = (int)lane + q * 64; vm[q] = sm[i < NM ? i : 0]; }
            order_fence();
            load_chunks<T, N, 0, CN, AUX_NT>(rs, lane, nom);
            constexpr int C_PREV = CN + (L::OFF_PREV - L::OFF_COV) / 4;          // the chunk of the previous marker id
            load_chunks<T, N, C_PREV, C_PREV + 1, AUX_NT>(rs, lane, P + 4 * (C_PREV - CN));
            order_fence();
            u32x4* di = reinterpret_cast<u32x4*>(tbl.id2slot);
            u32x4* dm = reinterpret_cast<u32x4*>(tbl.mk);
#pragma unroll
            for (int q = 0; q < PI; ++q) { const int i = (int)lane + q * 64; if (i < NI) di[i] = vi[q]; }
#pragma unroll
            for (int q = 0; q < PM; ++q) { const int i = (int)lane + q * 64; if (i < NM) dm[i] = vm[q]; }
            order_fence();
        }
        T prev_val = P[L::OFF_PREV - L::OFF_COV];                    // the C++ dialect's previous marker id, carried here
        int k0 = 0, last_used = 0;
#pragma unroll 1
        for (int f = 0; f < F; ++f) {
            const int K = kc.k[f];
            const T* fa = accel + (size_t)k0 * B * 3;
            const T* fg = gyro + (size_t)k0 * B * 3;
            const T* fd = dt + (size_t)k0 * (dt_stride ? B : 1);
            k0 += K;
            ImuSample<T> cur;
            if (K > 0) cur.load(fa, fg, fd, dt_stride, 0, B, b);
            const size_t fo = (size_t)f * B + b;
            int first = 0, last = (M > 0 && live && !(skip && skip[fo])) ? M : 0;
            int new_prev = -1, used = 0;
            const int* my_ids = ids + fo * M;
            const T* my_pos = pos + fo * M * 3;
            const T* my_quat = quat + fo * M * 4;
#pragma unroll 1
            for (int t = 0; t <= K; ++t) {
                if (t < K) {
                    predict_nominal<T, N, DIALECT, FBUS_X_PACK_TEAM>(nom, cur.a, cur.w, cur.h, k);
                    if (t + 1 < K) cur.load(fa, fg, fd, dt_stride, t + 1, B, b);
                    xch.put_coef(t & 1, k);
                } else if (M > 0) {
                    // the nominal state is final: marker choice (MeasureUpdate.m:51-60 ; filter.cpp:639-664) and the fold of the
                    // rows -> Lam, b in LDS, while the covariance roles run the frame's last step
                    if (last > 0 && mode == MODE_NEAREST) {
                        const int prev_id = (DIALECT == DIALECT_CPP) ? (int)prev_val : 0;
                        int min_i = -1, prev_i = -1;
                        T min_d = T(10), prev_d = T(0);
                        for (int i = 0; i < M; ++i) {
                            const int id = my_ids[i];
                            if (id < 0) continue;
                            const T x = my_pos[3 * i], y = my_pos[3 * i + 1], z = my_pos[3 * i + 2];
                            const T dist = fb_sqrt(x * x + y * y + z * z);
                            if (dist < min_d) { min_d = dist; min_i = i; }
                            if (DIALECT == DIALECT_CPP && id == prev_id) { prev_d = dist; prev_i = i; }
                        }
                        if (min_i >= 0 && DIALECT == DIALECT_CPP && fb_abs(prev_d - min_d) < dc.switch_thres && prev_d != T(0))
                            min_i = prev_i;
                        int slot = -1, id = -1;
                        if (min_i >= 0) {
                            id = my_ids[min_i];
                            slot = (id >= 0 && id <= FBUS_MAX_MARKER_ID) ? (int)tbl.id2slot[id] : -1;
                        }
                        if (slot < 0) { first = last = 0; }
                        else {
                            if (DIALECT == DIALECT_CPP) new_prev = id;
                            first = min_i; last = min_i + 1;
                        }
                    }
                    InfoAcc<T> acc;
                    PoseFold<T, N, DIALECT> fold;
                    fold.clear();
                    MarkerCommon<T, N> mc;
                    mc.build(nom, dc);
                    for (int i0 = first; i0 < last; i0 += FBUS_MARKER_GROUP) {
                        MarkerGroup<T, FBUS_MARKER_GROUP> mg;
                        mg.fetch(my_ids, my_pos, my_quat, i0, last);
                        mg.resolve(tbl);
#pragma unroll
                        for (int g = 0; g < FBUS_MARKER_GROUP; ++g) {
                            if (mg.slot[g] < 0) continue;
                            fold.add(nom, dc, mc, mg.mk[g], mg.yp[g], mg.yq[g]);
                            ++used;
                        }
                    }
                    fold.finish(acc, nom, dc, mc);
                    if (used == 0) { acc.clear(); new_prev = -1; }
#pragma unroll
                    for (int i = 0; i < 21; ++i) lam_at(i) = acc.Lam[i];
#pragma unroll
                    for (int i = 0; i < 6; ++i) lam_at(21 + i) = acc.b[i];
                    lam_at(27) = (T)new_prev;
                    if (new_prev >= 0) prev_val = (T)new_prev;
                }
                team_barrier();
            }
            team_barrier();                                          // the predicted covariance is in the image
            const bool last_frame = f + 1 == F;
            last_used = used;
            if (M > 0) {
                T dx[N];
                correct_part(I0_{}, dx);
                if (used > 0) inject<T, N>(nom, dx);
                if (last_frame) { store_part(I0_{}, false); break; }
                img_put_part(I0_{});
                team_barrier();                                      // the posterior is in the image
            } else if (last_frame) {
                store_part(I0_{}, true);
                break;
            }
        }
        if (live) {
            if (M > 0 && F > 0) applied[b0] = last_used > 0 ? 1 : 0;
            store_chunks<T, N, 0, CN, FBUS_X_FRAME_ST>(rs, lane, nom);
        }
    } else if (role == 2) {
        // ---- rows theta (+ Q diagonals); part 3 of the correct -----------------------------------------------------------------
        load_stage_chunks<T, N, 4, AUX_NT>(rs, lane, P);
#pragma unroll 1
        for (int f = 0; f < F; ++f) {
            const int K = kc.k[f];
            team_barrier();                                                  // iteration 0
#pragma unroll 1
            for (int t = 1; t <= K; ++t) {
                xch.step_theta(P, k, dc.qd, t, K);
                team_barrier();
            }
            if (frame_tail(I3_{}, I2_{}, I4_{}, f)) break;
        }
    } else if (role == 0) {
        // ---- rows v; part 1 of the correct -----------------------------------------------------------------------------------------
        load_stage_chunks<T, N, 2, AUX_NT>(rs, lane, P);
        static_for<0, NCC>([&](auto cc_) {
            constexpr int cc = decltype(cc_)::value;
            if constexpr (FI::pad_chunk(cc) && !TeamRec<T, N>::reads_chunk(2, cc)) load_chunks<T, N, CN + cc, CN + cc + 1, AUX_NT>(rs, lane, P + 4 * cc);
        });
#pragma unroll 1
        for (int f = 0; f < F; ++f) {
            const int K = kc.k[f];
            team_barrier();
#pragma unroll 1
            for (int t = 1; t <= K; ++t) {
                xch.step_v(P, k, dc.qd, t, K);
                team_barrier();
            }
            if (frame_tail(I1_{}, I1_{}, I2_{}, f)) break;
        }
    } else {
        // ---- rows p; part 2 of the correct -----------------------------------------------------------------------------------------
        load_stage_chunks<T, N, 1, AUX_NT>(rs, lane, P);
#pragma unroll 1
        for (int f = 0; f < F; ++f) {
            const int K = kc.k[f];
            team_barrier();
#pragma unroll 1
            for (int t = 1; t <= K; ++t) {
                xch.step_p(P, k, t);
                team_barrier();
            }
            if (frame_tail(I2_{}, I0_{}, I1_{}, f)) break;
        }
    }
}


// (correct from stereo corners / from corner pixels with the markers divided among the roles: csrc/ekf_meas.hpp since round 4.)

}  // namespace
