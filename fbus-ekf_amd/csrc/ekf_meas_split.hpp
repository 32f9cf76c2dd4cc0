// ekf_meas_split.hpp -- correct() from corner pixels with the UPDATE divided between the waves of a tile (round 5).
//
// Reference algebra (paths relative to the upstream repository): the rows of matlab/MeasureUpdate.m:67,72-73 with a marker corner in
// place of the marker origin, through the flat-port model of C++/src/vision.cpp:496-599 run forward (ekf_meas.hpp); the update
// matlab/MeasureUpdate.m:84-102 ; filter.cpp:709-739 in the one-shot form of ekf_meas.hpp::direct_update.
//
// Why.  correct_pixels2_kernel (ekf_meas.hpp) divides the FOLD of a filter's markers over NR waves ("roles") when a launch has fewer
// tiles than the chip has SIMDs, but its tail -- sums -> 6 x 6 stage -> update of 171 covariance elements -> injection -> 50 chunk stores
// -- stays with role 0 while the others have left.  Here the tail is divided too:
//   fold      role r folds markers r, r + NR, ...  (pixel_fold_marker, four corners in lock step, as in the one-tail kernel)
//   exchange  roles >= 1 leave their 27 sums in LDS; barrier
//   SOLVER    (role 0) has requested the chunks that hold P(J, J) in front of the exchange; it adds the sums in role order, forms Lam, b,
//             solves the 6 x 6 stage in double, leaves G, S^-1, m (63 values) in LDS; barrier; then updates and stores the LATE part of
//             the covariance -- the chunks that hold only elements P(a, c), a, c >= 9 (N = 18: storage [128, 172); all of type "outside
//             J": P(a, c) -= x_a' S^-1 x_c), for which it has requested the x_c = P(J, c >= 9) behind the 6 x 6 stage
//   UPDATER   (role 1) has requested the EARLY part (rows 0..8 and the collected diagonals: every x_c = P(J, c) lives there) and the
//             nominal state while the solver worked; behind the second barrier it reads the 63 coefficients, forms dx = P(:, J) m,
//             updates the early part (the J rows as the product G P(J, :), the rest by subtraction), injects dx, stores
//   roles >= 2 (four waves per tile) fold their share and leave behind the second barrier.
// Per element the operations are those of direct_update in the same order (direct_update_part, ekf_meas.hpp): the posterior equals the
// one-tail kernel's with the same number of roles bit for bit (the sums are added in role order there as here).
// Measured (alternating A/B, HIP-event bracket per launch, profiles/r05_meas_split.txt): 16 384 filters x 4 slots, four waves per tile
// 17.6 -> 16.7 us; 32 768 x 4, two waves per tile 23.3 -> 22.1 us; 32 768 x 16: 41.1 -> 39.8 us.  Chosen up to half a chip of tiles.
// What was also built and LOST (commit b55c4d4, same file): two waves per tile at FULL chip (65 536 filters) with a 256-register fold,
// corner by corner, so that two waves share a SIMD -- 85.6 us against 67.9 (16 slots), 47.3 against 33.2 (4 slots): one corner's single
// dependent chain issues an fp64 instruction every ~10 cycles where four corners in lock step reach 5.3, and a second wave per SIMD
// (at best 4.45 cycles per instruction for the pair) does not win that back.  Round 4's two-wave form lost the same way.
// Square port (normal = (0, 0, 1), the reference's configuration) and fp32 records only: everything else keeps correct_pixels2_kernel.
#pragma once
#include "ekf_meas.hpp"

namespace {

// (late_start, chunk_sel, load_cov_chunks, direct_update_part: ekf_meas.hpp -- the fp64 tail there runs the same two parts in one wave)

// LDS image of the 63 coefficients + the verdict (64 values of T per lane)
constexpr int SPLIT_NCOEF = 64;

// =================================================================================
// correct() from corner pixels, update divided between the waves of a tile (see the head of this file).
// NR = 2 (launches of a quarter to half a chip of tiles) or 4 (smaller ones): at most one wave per SIMD either way.
// =================================================================================
template <typename T, int N, int NR>
__global__ void __launch_bounds__(64 * NR)
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
            {
                const double wgt = slot >= 0 ? 1.0 : 0.0;            // (every slot is folded: see pixel_fold_marker)
                const int sl = slot >= 0 ? slot : 0;
                double mk[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) mk[q] = tbl.mkc[sl * MKC_STRIDE + q];
                T yl_[8], yr_[8];                                     // (a slot without a marker of the map: zeros for its image points)
#pragma unroll
                for (int k = 0; k < 8; ++k) { yl_[k] = slot >= 0 ? cur.l[k] : T(0); yr_[k] = slot >= 0 ? cur.r[k] : T(0); }
                if (stereo) pixel_fold_marker_stereo_halves<T, true>(acc, pd, Rd, pil, mc, mk, yl_, yr_, size, wgt);
                else pixel_fold_marker<1, T, true>(acc, pd, Rd, pil, mc, mk, yl_, yl_, size, wgt);
                nfold += wgt;
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
            order_fence();                                             // (the x_c chunks FIRST: barrier (2) waits for them by count)
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
        // (2) the coefficients are in LDS.  (round 6, advisor) vmcnt(0) in front of it: the x_c chunks requested above (SEL_XL) are the
        // very chunks the updater overwrites and stores behind this barrier (its J rows become G P(J, :)); a barrier that waits for LDS
        // only would leave "this wave's loads are served before that wave's later stores" to the ~1.5 k instructions the updater runs
        // first -- the class of failure ekf_team.hpp:200-204 documents.
        // Vector memory returns in order: once at most the LATE chunk loads (requested behind them, and stored by nobody else) are
        // outstanding, the x_c loads have been served -- the late chunks keep flying across the barrier (a full vmcnt(0) here cost
        // 1.6-2 us at config 3's size: 18.6 / 21.4 us left / stereo, against 15.9-16.1 / 18.2-18.7 with this count; tools/run_configs.py).
        if constexpr (HAS_LATE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RC::NCH - (CN + C_E)) : "memory");
        meas_barrier();
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
