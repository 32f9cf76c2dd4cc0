// ekf_kernels.hpp -- the __global__ kernels of libfbus_ekf.so and their record <-> register helpers.
// Included by fbus_ekf.hip (launchers + C ABI) and by the single-kernel experiments under tools/ (which
// instantiate one kernel and read its ISA / time it in isolation).  gfx950 only.
#pragma once
#include "../../include/fbus_ekf.h"
#include "ekf_device.hpp"
#include "vision_device.hpp"

#include <hip/hip_runtime.h>

using namespace fbus;

namespace {

#ifndef FBUS_BLOCK
#define FBUS_BLOCK 64
#endif
constexpr int BLOCK = FBUS_BLOCK;       // 64 = one wave per workgroup: B/64 workgroups over 256 CUs x 4 SIMDs
__device__ __forceinline__ unsigned my_tile() { return __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + threadIdx.x) >> 6); }
__device__ __forceinline__ unsigned my_lane() { return threadIdx.x & 63u; }

// Stagger of the four SIMDs of a CU (round 5).  A launch of one wave per SIMD runs every wave through the same phases at the same time:
// all four waves of a CU request their records together (and store them together at the end), sharing the CU's one path to memory
// (~12.8 B/clk per CU: 4 x 51 KB take ~6.7 us, one wave alone ~1.7 us) while its VALUs idle; then all four compute while that path idles.
// In the VALU-bound resident kernels (fused frames, the pixel / corner updates) a wave on SIMD k therefore sleeps k x FBUS_X_SIMD_STAGGER
// x 64 clocks before it asks for its record: the four load (and store) phases of a CU fall one behind the other, under the other SIMDs'
// arithmetic.  (Round 2 staggered alternate WORKGROUPS of the memory-bound correct and found nothing: neighbouring workgroups sit on
// different CUs, the four waves of one CU stayed in phase -- and a memory-bound kernel has no arithmetic to hide a load under.)
// HW_ID (hwreg 4): SIMD_ID = bits 5:4.
// Measured (65 536 filters, same box, profiles/r05_stagger.txt; units of 64 clocks per SIMD id, fused / per-call):
//   0 / 0      fused pose frame 1.446e10 steps/s, fused pixel frame (M = 4) 9.36e9, correct_pixels M = 4 31.8 us, stereo 41.4 us
//   48 / 24    1.545e10, 9.82e9, 30.9 / 41.3 us          64 / 32   1.564e10 (+8 %), 9.82e9 (+5 %), 30.6 / 39.6 us
//   96 / 16    1.501e10, 9.38e9, 30.3 / 41.0 us          127 / 48  1.398e10, 9.40e9, 31.2 / 40.6 us
// (the frame window, the per-call predict and the pose-form correct do not change: one is all arithmetic, the others all memory)
#ifndef FBUS_X_STAGGER_FRAME
#define FBUS_X_STAGGER_FRAME 64     // the fused frames (frame_kernel, frame_meas_kernel): 64 x 64 clocks = 1.7 us per SIMD id
#endif
#ifndef FBUS_X_STAGGER_MEAS
#define FBUS_X_STAGGER_MEAS 32      // the per-call pixel / corner updates with one wave per tile
#endif
// (also tried: predict_n K = 7 at 64 units 30.0 -> 29.3 us, frame2_kernel<double> at 128 units 85 -> 88-90 us: not adopted)
template <int UNITS>
__device__ __forceinline__ void simd_stagger()
{
    if constexpr (UNITS > 0) {
        const unsigned simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);      // 2 bits at offset 4 of HW_ID
        for (unsigned i = 0; i < simd; ++i) {
#pragma unroll
            for (int j = 0; j < UNITS / 127; ++j) __builtin_amdgcn_s_sleep(127);
            if constexpr (UNITS % 127 > 0) __builtin_amdgcn_s_sleep(UNITS % 127);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------
// record <-> registers
// ---------------------------------------------------------------------------------
// A tile = the records of 64 consecutive filters = NCH pieces of 1 KiB; piece c holds
// chunk c (16 bytes) of each of the 64 filters, so one wave moves a piece with one
// buffer_load_dwordx4 / buffer_store_dwordx4.  The descriptor covers exactly one tile
// (wave-uniform base), the lane contributes a 32-bit offset, the piece index goes
// into soffset/imm -- no per-piece 64-bit address lives in VGPRs.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
// cache policy (aux) of a record access: 0 = default, 2 = nt (non-temporal).  Measured at B = 65 536:
// nt on the once-read record stream of predict 18.05 -> 17.30 us; nt on lines that are re-read +9 %; nt STORES win in
// the streamed predict kernel (the lines leave L2 while the launch still reads: 14.9 -> 14.0 us) and lose in correct
// (its default-policy stores are what puts the records back into the Infinity Cache for the predicts that follow).
constexpr int AUX_DEFAULT = 0, AUX_NT = 2, AUX_SC1 = 16;
// Per-step inputs (IMU samples, marker measurements) are read once: nt as well.  Measured: with default-policy
// loads of 1.5 MB of fresh IMU data per launch the nt-streamed records lose their Infinity Cache residency and a
// predict launch takes 13.8 us instead of 11.8 us (tools/exp_predict_timeline.hip).
template <typename T>
__device__ __forceinline__ T ld_once(const T* p) { return __builtin_nontemporal_load(p); }
// experiment knobs (tools/ab_bench.sh builds variants with -D...)
#ifndef FBUS_X_CORRECT_ST
// record stores of the per-call correct: sc1 = write-through.  The lines reach the Infinity Cache at once instead of sitting
// dirty in the XCD's L2 until something evicts them, and the predict that follows streams them with the same non-temporal
// loads as every other predict: correct 20.9 -> 19.0 us (HIP events), headline 4.94e9 -> 5.03e9 at 65 536 filters, +0.8 % /
// +0.6 % at 131 072 / 262 144 (round 1's answer to the same problem, default-policy loads in the first predict behind a
// correct, is no longer used behind correct; sc1 + nt stores lose: profiles/logs/r02_sc1.log)
#define FBUS_X_CORRECT_ST AUX_SC1
#endif
#ifndef FBUS_X_SPLIT
#define FBUS_X_SPLIT RC::CH_VAR_END
#endif
#ifndef FBUS_X_MEAS_NT
#define FBUS_X_MEAS_NT 0
#endif
#ifndef FBUS_X_STREAM_ST
#define FBUS_X_STREAM_ST 1      // stacked correct / fused frame: the last rank-1 pass stores each chunk when its rows are final
#endif
#ifndef FBUS_X_FRAME_ST
#define FBUS_X_FRAME_ST AUX_DEFAULT    // record stores of the fused frame / frame window kernels
#endif
#ifndef FBUS_X_FMEAS_ST
// ... of the fused frame with the pixel / corner update (frame_meas_kernel): write-through, as the per-call updates store.  Same-box A/B
// (profiles/r05_stagger.txt): M = 4 left 9.60e9 / 9.98e9 -> 1.005e10 / 1.012e10 steps/s, stereo 8.25 / 8.36 -> 8.61 / 8.38e9; the pose frame
// and the frame window do not care (1.58-1.60e10 / 2.48-2.5e10 either way) and keep the default policy
#define FBUS_X_FMEAS_ST AUX_SC1
#endif
#ifndef FBUS_X_PREDICT_LD
#define FBUS_X_PREDICT_LD AUX_NT       // record-load policy of the streamed per-call predict (records that fit the Infinity Cache)
#endif
#ifndef FBUS_X_PREDICT_LD_WARM
#define FBUS_X_PREDICT_LD_WARM AUX_DEFAULT     // ... of the first predict behind a kernel that left the records in L2 (fused frame, corners, pixels)
#endif
#ifndef FBUS_X_PREDICT_LD_BIG
#define FBUS_X_PREDICT_LD_BIG AUX_DEFAULT      // ... and of larger batches (records > 56 MB, see fbus_ekf.hip::launch_predict_t)
#endif
#ifndef FBUS_X_PREDICT_ST_BIG
#define FBUS_X_PREDICT_ST_BIG AUX_DEFAULT
#endif
#ifndef FBUS_X_PREDICT_ST
#define FBUS_X_PREDICT_ST AUX_NT       // store policy of the streamed per-call predict
#endif
#ifndef FBUS_X_CORRECT_LD
// load policy of the covariance in the correct kernels.  With the write-through stores the default policy here makes the predicts
// behind a correct 0.17 us faster and correct itself 1.0 us slower (16.6 -> 17.6 us by rocprofv3): + 0.5 % on the headline in a
// same-box A/B (5.025e9-5.035e9 -> 5.043e9-5.064e9), less than boxes differ from each other; nt keeps correct at 0.84 of peak
#define FBUS_X_CORRECT_LD AUX_NT
#endif
#ifndef FBUS_X_CORRECT_STAGGER_BIT
#define FBUS_X_CORRECT_STAGGER_BIT 3
#endif
#ifndef FBUS_X_CORRECT_STAGGER
#define FBUS_X_CORRECT_STAGGER 0   // experiment: every other wave of an XCD sleeps this many x 3.9 us before it requests its record
#endif
#ifndef FBUS_X_FRAME_WAVES
#define FBUS_X_FRAME_WAVES 1    // __launch_bounds__ waves per SIMD of the fused frame kernel (2 = at most 256 registers)
#endif
#ifndef FBUS_X_CORRECT_WAVES
#define FBUS_X_CORRECT_WAVES 1
#endif
#ifndef FBUS_X_IMU_PREFETCH
#define FBUS_X_IMU_PREFETCH 1   // predict_n / fused frame: the IMU sample of step k + 1 is requested in the middle of step k
#endif
template <typename T>
__device__ __forceinline__ T ld_meas(const T* p) { return FBUS_X_MEAS_NT ? __builtin_nontemporal_load(p) : *p; }

template <typename T, int N>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const T* recs, unsigned tile)
{
    constexpr unsigned TILE_BYTES = Rec<T, N>::NCH * 1024u;
    char* tb = const_cast<char*>(reinterpret_cast<const char*>(recs)) + (size_t)tile * TILE_BYTES;
    return __builtin_amdgcn_make_buffer_rsrc(tb, 0, (int)TILE_BYTES, 0x00020000);
}

// chunks [C0, C1) of the lane's record -> dst[0 .. (C1-C0)*EPC)
template <typename T, int N, int C0, int C1, int AUX = AUX_DEFAULT>
__device__ __forceinline__ void load_chunks(__amdgpu_buffer_rsrc_t rs, unsigned lane, T* dst)
{
    constexpr int EPC = Rec<T, N>::EPC;
    const unsigned off = lane * 16u;
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (c & 3) * 1024u, (c >> 2) * 4096, AUX);
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int k = 0; k < EPC; ++k) dst[(c - C0) * EPC + k] = e[k];
    }
}

template <typename T, int N, int C0, int C1, int AUX = AUX_DEFAULT>
__device__ __forceinline__ void store_chunks(__amdgpu_buffer_rsrc_t rs, unsigned lane, const T* src)
{
    constexpr int EPC = Rec<T, N>::EPC;
    const unsigned off = lane * 16u;
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        u32x4 v;
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int k = 0; k < EPC; ++k) e[k] = src[(c - C0) * EPC + k];
        // HAZARD (measured on gfx950, round 2): a buffer_store_dwordx4 reads its four data registers over several cycles,
        // and a VALU write to one of them in the very next issue slot (the register allocator's v_accvgpr_read into the
        // fourth data register right behind a store in the middle of a kernel) reached memory on lanes 12-15 of every
        // 16 instead of the stored value.  LLVM's hazard recognizer pads this case only when soffset is NOT a register
        // (GCNHazardRecognizer::createsVALUHazard) -- with the 4 KiB group offset in an SGPR, as the loads have it, it
        // emits nothing.  Stores therefore carry the group offset in the VGPR offset and a constant-zero soffset: the
        // recognizer then inserts the wait state itself, exactly where a data register is overwritten too early.
#ifdef FBUS_X_STORE_SGPR_SOFFSET      // experiment only: UNSAFE (see above)
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + (c & 3) * 1024u, (c >> 2) * 4096, AUX);
#else
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + (c >> 2) * 4096u + (c & 3) * 1024u, 0, AUX);
#endif
    }
}

// element e of filter b inside the tiled record storage (pack/unpack helpers)
template <typename T, int N>
__device__ __forceinline__ size_t elem_index(size_t b, int e)
{
    constexpr int EPC = Rec<T, N>::EPC;
    return ((b >> 6) * Rec<T, N>::NCH + (size_t)(e / EPC)) * (64 * EPC) + (b & 63) * EPC + (e % EPC);
}

// Keeps memory instructions in source order within a block: an empty asm with a memory clobber plus a sched_barrier
// for the machine scheduler.  It does NOT stop LLVM from sinking a buffer load into a later block that is its only
// user -- the correct kernels avoid that structurally (no early exit in front of the loads, first marker group folded
// unconditionally).
__device__ __forceinline__ void order_fence()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Row hook of the last rank-1 pass (NoRowHook's counterpart): when row I of the covariance is final, every 16-byte
// chunk whose elements all belong to rows <= I goes out -- the stores of the first 30 of the 43 covariance chunks are
// issued while the pass is still running instead of in one burst behind it.
template <typename T, int N, int AUX>
struct RowStore {
    __amdgpu_buffer_rsrc_t rs;
    unsigned lane;
    const T* P;
    static constexpr int fin(int r)          // chunks [CH_NOM, fin(r)) are final once rows < r are
    {
        return r >= N ? Rec<T, N>::NCH : Rec<T, N>::CH_NOM + cov_final_before_row<N>(r) / Rec<T, N>::EPC;
    }
    template <int I>
    __device__ __forceinline__ void row_done() const
    {
        constexpr int C0 = fin(I), C1 = fin(I + 1);
        if constexpr (C1 > C0 && I + 1 < N) {            // the last group goes out with the rest of the record
            __builtin_amdgcn_sched_barrier(0);
            store_chunks<T, N, C0, C1, AUX>(rs, lane, P + (C0 - Rec<T, N>::CH_NOM) * Rec<T, N>::EPC);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static constexpr int streamed_end() { return fin(N - 1); }     // chunks [CH_NOM, streamed_end()) were stored by the hook
};

// Marker map in LDS.  A measurement names its marker by ArUco id: id -> map slot -> map constants are two dependent
// lookups.  As vector loads they queue behind the 50 KiB of record loads the wave already has in flight (vector loads
// return in issue order), so the rows of the first marker could not be built before the whole covariance had landed
// plus two more round trips (tools/exp_correct_timeline.hip: rows folded 10.9 us after entry).  The tables are small
// (2 KiB + 1 KiB), so every workgroup copies them to LDS with its first loads and looks them up there: LDS reads
// count on lgkmcnt and do not wait for the record stream.
template <typename T>
struct alignas(16) MarkerLDS {
    short id2slot[FBUS_MAX_MARKER_ID + 1];
    T mk[FBUS_MAX_MARKERS * MK_STRIDE];
};
template <typename T>
struct MarkerTableRegs {            // the global -> register half of the copy (issued before the record loads)
    static constexpr int NI = (int)sizeof(short) * (FBUS_MAX_MARKER_ID + 1) / 16, NM = (int)sizeof(T) * FBUS_MAX_MARKERS * MK_STRIDE / 16;
    static constexpr int PI = (NI + BLOCK - 1) / BLOCK, PM = (NM + BLOCK - 1) / BLOCK;
    u32x4 vi[PI], vm[PM];
    __device__ __forceinline__ void load(const DevConst<T>& dc)
    {
        const u32x4* si = reinterpret_cast<const u32x4*>(dc.id2slot);
        const u32x4* sm = reinterpret_cast<const u32x4*>(dc.mk);
#pragma unroll
        for (int k = 0; k < PI; ++k) { const int i = threadIdx.x + k * BLOCK; vi[k] = si[(NI % BLOCK == 0 || i < NI) ? i : 0]; }
#pragma unroll
        for (int k = 0; k < PM; ++k) { const int i = threadIdx.x + k * BLOCK; vm[k] = sm[(NM % BLOCK == 0 || i < NM) ? i : 0]; }
    }
    __device__ __forceinline__ void to_lds(MarkerLDS<T>& t) const
    {
        u32x4* di = reinterpret_cast<u32x4*>(t.id2slot);
        u32x4* dm = reinterpret_cast<u32x4*>(t.mk);
        // no branches here (straight-line code keeps the compiler's s_waitcnt bookkeeping exact: the writes must wait
        // for the map pieces only, not for the record loads issued behind them)
#pragma unroll
        for (int k = 0; k < PI; ++k) { const int i = threadIdx.x + k * BLOCK; if (NI % BLOCK == 0 || i < NI) di[i] = vi[k]; }
#pragma unroll
        for (int k = 0; k < PM; ++k) { const int i = threadIdx.x + k * BLOCK; if (NM % BLOCK == 0 || i < NM) dm[i] = vm[k]; }
        if constexpr (BLOCK > 64) __syncthreads();      // one wave per workgroup needs no barrier: LDS ops are in order
    }
};

// The measurements of up to G marker slots [i0, i0 + G) of one filter: ids and poses in ONE round of loads (fetch),
// the map lookups from LDS (resolve).  Slots past `last` and markers that are invisible / not in the map come back
// with slot = -1.
#ifndef FBUS_MARKER_GROUP
#define FBUS_MARKER_GROUP 4
#endif
template <typename T, int G>
struct MarkerGroup {
    int id[G], slot[G], n;          // n = slots of this group that exist (the loaded values are not touched in fetch)
    T yp[G][3], yq[G][4], mk[G][MK_STRIDE];
    // vec (wave-uniform; G == 4, M % 4 == 0 and 16-byte aligned arrays, checked by the launcher): the group's 4 ids, 12 position
    // and 16 quaternion values as 8 16-byte loads per lane instead of 32 4-byte ones (fp64: 15 instead of 32)
    __device__ __forceinline__ void fetch(const int* my_ids, const T* my_pos, const T* my_quat, int i0, int last, bool vec = false)
    {
        n = last - i0;
        if (G == 4 && vec) {
            constexpr int EP = 16 / (int)sizeof(T);
            const u32x4 vi = *reinterpret_cast<const u32x4*>(my_ids + i0);
            const u32x4* pp = reinterpret_cast<const u32x4*>(my_pos + 3 * i0);
            const u32x4* pq = reinterpret_cast<const u32x4*>(my_quat + 4 * i0);
            T bp[12], bq[16];
#pragma unroll
            for (int c = 0; c < 12 / EP; ++c) {
                const u32x4 v = pp[c];
                const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
                for (int k = 0; k < EP; ++k) bp[c * EP + k] = e[k];
            }
#pragma unroll
            for (int c = 0; c < 16 / EP; ++c) {
                const u32x4 v = pq[c];
                const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
                for (int k = 0; k < EP; ++k) bq[c * EP + k] = e[k];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                id[g] = (int)vi[g];
#pragma unroll
                for (int k = 0; k < 3; ++k) yp[g][k] = bp[3 * g + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) yq[g][k] = bq[4 * g + k];
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = (i0 + g < last) ? i0 + g : last - 1;
            id[g] = my_ids[i];
#pragma unroll
            for (int k = 0; k < 3; ++k) yp[g][k] = ld_meas(my_pos + 3 * i + k);
#pragma unroll
            for (int k = 0; k < 4; ++k) yq[g][k] = ld_meas(my_quat + 4 * i + k);
        }
    }
    __device__ __forceinline__ void resolve(const MarkerLDS<T>& t)
    {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool ok = g < n && id[g] >= 0 && id[g] <= FBUS_MAX_MARKER_ID;
            const int s_ = t.id2slot[ok ? id[g] : 0];
            slot[g] = ok ? s_ : -1;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const T* m = t.mk + (slot[g] < 0 ? 0 : slot[g]) * MK_STRIDE;
#pragma unroll
            for (int k = 0; k < 8; ++k) mk[g][k] = m[k];
        }
    }
};

// one IMU sample of one filter (accel/gyro: [K][B][3]; dt: [K] or [K][B]), read once: non-temporal
template <typename T>
struct ImuSample {
    T a[3], w[3], h;
    __device__ __forceinline__ void load(const T* accel, const T* gyro, const T* dt, int dt_stride, int k, int B, int b)
    {
        const size_t o = ((size_t)k * B + b) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) { a[i] = ld_once(accel + o + i); w[i] = ld_once(gyro + o + i); }
        h = dt_stride ? ld_once(dt + (size_t)k * B + b) : dt[k];
    }
};

// LDS parking for the multi-step predict loop (two waves per SIMD = at most 256 registers).  Inside the loop the whole
// record is live (172 + 28) beside a step's own 28 coefficients, 36 + 9 products of the v stage and the prefetched IMU
// sample: ~330 registers.  But rows p of the covariance (storage [0, E_P): final when stage p has run, not touched by
// stages v and theta) are only needed by stage p, and the nominal state only by predict_nominal.  They wait in LDS in
// between: 13 + NOMCH ds_write_b128 / ds_read_b128 per step (conflict-free: lane-consecutive 16-byte slots) against
// ~1400 VALU instructions.
struct NoStepPark {
    template <typename T> __device__ __forceinline__ void rows_out(const T*) const {}
    template <typename T> __device__ __forceinline__ void rows_in(T*) const {}
    template <typename T> __device__ __forceinline__ void nom_out(const T*) const {}
    template <typename T> __device__ __forceinline__ void nom_in(T*) const {}
};
template <typename T, int N, int NOMCH>
struct StepPark {
    using RC = Rec<T, N>;
    static constexpr int EPC = RC::EPC;
    static constexpr int E_P = cov_final_before_row<N>(3);               // rows p = storage [0, E_P) (+ odd-row diagonals kept in registers)
    static constexpr int PCH = (E_P + EPC - 1) / EPC;
    static constexpr int NCHUNK = PCH + NOMCH;
    static_assert(NOMCH * EPC <= Lay<N>::NNOM, "nominal chunks");
    u32x4* mem;                                                          // [NCHUNK][BLOCK], this lane's column
    __device__ __forceinline__ void put(int c, const T* src, int n) const
    {
        u32x4 v = { 0u, 0u, 0u, 0u };
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int k = 0; k < EPC; ++k) if (k < n) e[k] = src[k];
        mem[c * BLOCK] = v;
    }
    __device__ __forceinline__ void get(int c, T* dst, int n) const
    {
        const u32x4 v = mem[c * BLOCK];
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int k = 0; k < EPC; ++k) if (k < n) dst[k] = e[k];
    }
    __device__ __forceinline__ void rows_out(const T* P) const
    {
#pragma unroll
        for (int c = 0; c < PCH; ++c) put(c, P + c * EPC, E_P - c * EPC);
    }
    __device__ __forceinline__ void rows_in(T* P) const
    {
#pragma unroll
        for (int c = 0; c < PCH; ++c) get(c, P + c * EPC, E_P - c * EPC);
    }
    __device__ __forceinline__ void nom_out(const T* nom) const
    {
#pragma unroll
        for (int c = 0; c < NOMCH; ++c) put(PCH + c, nom + c * EPC, EPC);
    }
    __device__ __forceinline__ void nom_in(T* nom) const
    {
#pragma unroll
        for (int c = 0; c < NOMCH; ++c) get(PCH + c, nom + c * EPC, EPC);
    }
};

// K ImuUpdates with the record resident in registers (predict_n, fused frame)
struct NoMidHook { __device__ __forceinline__ void operator()() const {} };
template <typename T, int N, int DIALECT, typename MID = NoMidHook, typename PARK = NoStepPark, int PK = FBUS_X_PACK>
__device__ __forceinline__ void predict_steps(T* nom, T* P, int K, const T* accel, const T* gyro, const T* dt, int dt_stride,
                                              int B, int b, const T* qd, const MID& mid_last = MID(), const PARK& park = PARK())
{
#if FBUS_X_IMU_PREFETCH
    // The sample of step k + 1 is requested as soon as step k's kinematics have consumed sample k, i.e. ~700 VALU
    // instructions (the three covariance stages) before it is needed.  Requesting it at the top of the iteration into a
    // second buffer does not work: the compiler's s_waitcnt placement then drains vmcnt to 0 right behind the new
    // loads (to release the previous sample across the loop back-edge) and the full latency stays exposed -- that is
    // what the first version of this prefetch measured (no gain).
    if (K <= 0) { mid_last(); return; }
    ImuSample<T> cur;
    cur.load(accel, gyro, dt, dt_stride, 0, B, b);
    constexpr bool PARKED = !std::is_same<PARK, NoStepPark>::value;
    if constexpr (PARKED) { park.rows_out(P); park.nom_out(nom); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        PredictCoef<T> c;
        if constexpr (PARKED) { park.nom_in(nom); }
        predict_nominal<T, N, DIALECT, PK>(nom, cur.a, cur.w, cur.h, c);
        if constexpr (PARKED) { park.nom_out(nom); }
        __builtin_amdgcn_sched_barrier(0);
#ifdef FBUS_X_IMU_ONCE      // experiment: every step on the first sample -- no memory access inside the loop (what does the sample's latency cost?)
        if (k + 1 >= K) mid_last();
#else
        if (k + 1 < K) cur.load(accel, gyro, dt, dt_stride, k + 1, B, b);
        else mid_last();                           // last step: the caller's loads for what follows the predicts
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PARKED) { park.rows_in(P); }
        cov_stage_p<T, N, PK>(P, c);
        if constexpr (PARKED) { park.rows_out(P); __builtin_amdgcn_sched_barrier(0); }
        cov_stage_v<T, N, PK>(P, c, qd);
        cov_stage_th<T, N, PK>(P, c, qd);
        if constexpr (PARKED) { __builtin_amdgcn_sched_barrier(0); }
    }
    if constexpr (PARKED) { park.rows_in(P); park.nom_in(nom); }
#else
    for (int k = 0; k < K; ++k) {
        const size_t o = ((size_t)k * B + b) * 3;
        const T a[3] = { ld_once(accel + o), ld_once(accel + o + 1), ld_once(accel + o + 2) };
        const T w[3] = { ld_once(gyro + o), ld_once(gyro + o + 1), ld_once(gyro + o + 2) };
        const T h = dt_stride ? ld_once(dt + (size_t)k * B + b) : dt[k];
        predict_step<T, N, DIALECT, PK>(nom, P, a, w, h, qd);
    }
    mid_last();
#endif
}

// ---------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------
// K consecutive ImuUpdates per launch (K = 1 is the per-call API).
// accel/gyro: [K][B][3]; dt: [K] (dt_stride 0) or [K][B] (dt_stride 1).
//
// K = 1 (MULTI = false) is written as a stream: at one wave per SIMD nothing else hides this wave's latency, and
// all waves of a launch run in lock step, so a load-everything / compute / store-everything body leaves HBM idle
// while the chip computes and the VALUs idle while it loads.  Here every load is issued up front in the order the
// stages need the data (IMU sample, nominal state, the separately stored diagonals, covariance rows p and v, rows
// theta, then the predict-invariant ba/bg/g part), the arithmetic follows in that order (s_waitcnt vmcnt(n) lets a
// stage start when ITS chunks have landed), and each stage's chunks are stored as soon as they are final.
// ba, bg, g and the covariance elements among ba, bg, g (off-diagonal) are not written by ImuUpdate: their chunks
// stay as they are in HBM (N = 18: 33 of the 43 covariance chunks are stored).
//
// ST = cache policy of the record stores, LD = cache policy of the record loads.  A batch whose records do not fit the
// 256 MB Infinity Cache -- in practice every batch beyond one round of 1024 waves (records > 56 MB) -- runs with the default policy on both: measured at
// 262 144 filters 65.4 us (nt / nt) -> 59.8 (default loads) -> 55.7 us (default loads and stores) = 6.75 TB/s moved; at
// 131 072 and below the non-temporal forms win (records stay resident in the Infinity Cache between launches).
// LD = cache policy of the record loads.  Non-temporal in a run of predicts (each line is read once per launch); the
// FIRST predict after a correct reads with the default policy: correct stores with the default policy, its lines are
// still in the XCD L2s, and nt loads of such lines were measured slow (first predict after a correct 15.8 us with nt
// loads, 14.2 us with default loads, and the following launches reach their steady 12.6 us two launches earlier:
// -5 us per camera frame; default-policy STORES there, or default loads for a second launch, lose again --
// tools/exp_gap_after_correct.py under rocprofv3, reduced by tools/trace_positions.py).
// PARK (MULTI, fp32): the K-step loop with rows p of the covariance and PARK_NOM_CHUNKS chunks of the nominal state parked
// in LDS between their uses (StepPark): 246 registers instead of 335, no scratch, 17 KiB of LDS -- two waves per SIMD.
// The launcher picks it from 2048 waves on: predict_n K = 7 at 131 072 filters 64.5 -> 53.4 us, at 262 144 filters
// 122.1 -> 97.6 us (91 % of the VALU bound of its 1600 instructions per step); with one wave per SIMD (65 536 filters) the
// parking is pure overhead, 31.4 -> 32.4 us.  Parking rows p alone leaves 16-32 bytes of scratch, all 7 nominal chunks
// cost 20 KiB of LDS for nothing (profiles/logs/r02_park.log).
constexpr int PARK_NOM_CHUNKS = 4;
// fp64 (round 4: the reference's own arithmetic as a resident K-step loop): 171 covariance doubles are 342 of a lane's 512 registers,
// so ALL of the nominal state waits in LDS between its uses (14 chunks of two doubles)
#ifndef FBUS_X_PARK_NOM_F64
#define FBUS_X_PARK_NOM_F64 14
#endif
#ifndef FBUS_X_PREDICT_TWO
#define FBUS_X_PREDICT_TWO 0
#endif
template <typename T> constexpr int park_nom_chunks() { return sizeof(T) == 8 ? FBUS_X_PARK_NOM_F64 : PARK_NOM_CHUNKS; }
template <typename T, int N, int DIALECT, bool MULTI, int LD = AUX_NT, int ST = FBUS_X_PREDICT_ST, bool PARK = false>
__global__ void __launch_bounds__(BLOCK, (sizeof(T) == 4 && (PARK || (!MULTI && FBUS_X_PREDICT_TWO))) ? 2 : 1)
predict_kernel(T* __restrict__ recs, int B, int K, const T* __restrict__ accel, const T* __restrict__ gyro,
               const T* __restrict__ dt, int dt_stride, DevConst<T> dc)
{
    // (round 5, measured and NOT kept: the tiles beyond the last whole round of waves as sub-tile waves of 16 / 32 active lanes spread
    // over all CUs -- no gain at any size, one extra tile costs +3.4 us however it is cut: profiles/r05_tail_split.txt, commit b55c4d4)
    const int b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= B) return;
    using RC = Rec<T, N>;
    constexpr int EPC = RC::EPC, CN = RC::CH_NOM;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, my_tile());
    T nom[Lay<N>::NNOM], P[RC::NCOVP];
    if (MULTI) {
        load_chunks<T, N, 0, CN, AUX_NT>(rs, my_lane(), nom);
        load_chunks<T, N, CN, RC::NCH, AUX_NT>(rs, my_lane(), P);
        if constexpr (PARK) {
            using Park = StepPark<T, N, park_nom_chunks<T>()>;
            __shared__ u32x4 park_mem[Park::NCHUNK * BLOCK];
            predict_steps<T, N, DIALECT, NoMidHook, Park>(nom, P, K, accel, gyro, dt, dt_stride, B, b, dc.qd, NoMidHook(),
                                                          Park{ park_mem + threadIdx.x });
        } else
        predict_steps<T, N, DIALECT>(nom, P, K, accel, gyro, dt, dt_stride, B, b, dc.qd);
        store_chunks<T, N, 0, RC::CH_KIN>(rs, my_lane(), nom);
        store_chunks<T, N, CN, RC::CH_VAR_END>(rs, my_lane(), P);
    } else {
        // chunk boundaries of the stages (whole chunks whose elements all belong to finished rows)
        constexpr int C_P = CN + cov_final_before_row<N>(3) / EPC;       // rows p final
        constexpr int C_V = CN + cov_final_before_row<N>(6) / EPC;       // rows p, v final
        constexpr int C_PV_IN = CN + (cov_final_before_row<N>(6) + EPC - 1) / EPC;   // chunks holding rows p, v
        // N = 18 keeps the odd-row and ba/bg diagonals behind rows 0..8: bring them in first
        constexpr int C_DG0 = (N == 18) ? CN + 122 / EPC : RC::CH_VAR_END;
        constexpr int C_DG1 = RC::CH_VAR_END;
        static_assert(C_PV_IN <= C_DG0, "rows p, v must precede the collected diagonals");

        const size_t o = (size_t)b * 3;
        const T a[3] = { ld_once(accel + o), ld_once(accel + o + 1), ld_once(accel + o + 2) };
        const T w[3] = { ld_once(gyro + o), ld_once(gyro + o + 1), ld_once(gyro + o + 2) };
        constexpr int LDP = LD, STP = ST;
        const T h = dt_stride ? ld_once(dt + b) : dt[0];
        load_chunks<T, N, 0, CN, LDP>(rs, my_lane(), nom);
        load_chunks<T, N, C_DG0, C_DG1, LDP>(rs, my_lane(), P + (C_DG0 - CN) * EPC);
        load_chunks<T, N, CN, C_PV_IN, LDP>(rs, my_lane(), P);
        load_chunks<T, N, C_PV_IN, C_DG0, LDP>(rs, my_lane(), P + (C_PV_IN - CN) * EPC);
        load_chunks<T, N, C_DG1, RC::NCH, LDP>(rs, my_lane(), P + (C_DG1 - CN) * EPC);

        // stores are non-temporal: the lines leave the XCD's L2 while the launch is still reading (reads and writes
        // overlap) instead of piling up dirty until the end-of-kernel write-back; the sched_barriers keep the
        // compiler from sinking a stage's stores behind the next stage's arithmetic
        PredictCoef<T> k;
        predict_nominal<T, N, DIALECT>(nom, a, w, h, k);
        store_chunks<T, N, 0, RC::CH_KIN, STP>(rs, my_lane(), nom);
        __builtin_amdgcn_sched_barrier(0);
        cov_stage_p<T, N>(P, k);
        store_chunks<T, N, CN, C_P, STP>(rs, my_lane(), P);
        __builtin_amdgcn_sched_barrier(0);
        cov_stage_v<T, N>(P, k, dc.qd);
        store_chunks<T, N, C_P, C_V, STP>(rs, my_lane(), P + (C_P - CN) * EPC);
        __builtin_amdgcn_sched_barrier(0);
        cov_stage_th<T, N>(P, k, dc.qd);
        store_chunks<T, N, C_V, RC::CH_VAR_END, STP>(rs, my_lane(), P + (C_V - CN) * EPC);
    }
}

// JOINT = stacked mode (all visible markers at one linearisation point) through the information-compressed joint
// update; !JOINT = the reference's behaviour, the nearest marker (C++ dialect: with hysteresis) applied row by row.
//
// Both paths share the prologue.  Vector loads return in issue order, so it issues them in need order -- the marker map
// (every lane carries a piece of it to LDS), the first group's measurements, the nominal state, then the covariance
// -- and keeps that order with order_fence(); the arithmetic that needs only the early loads (the rows of the markers)
// then runs while the covariance is still on its way in.  Lanes that are skipped or past B run along with no markers
// and leave before the stores: an early exit in front of the loads would let the compiler sink loads into the live
// branch, behind the covariance stream.
template <typename T, int N, int DIALECT, int COV, bool JOINT, bool SPLIT = (sizeof(T) == 8)>
__global__ void __launch_bounds__(BLOCK, FBUS_X_CORRECT_WAVES)
correct_kernel(T* __restrict__ recs, int B, int M, const int* __restrict__ ids, const T* __restrict__ pos,
               const T* __restrict__ quat, int mode, const unsigned char* __restrict__ skip,
               unsigned char* __restrict__ applied, DevConst<T> dc)
{
    using L = Lay<N>;
    using RC = Rec<T, N>;
    constexpr int G = FBUS_MARKER_GROUP;
    const bool vec = (mode & MODE_MEAS_VEC) != 0;     // set by the launcher: 16-byte loads of the measurement inputs are legal
    mode &= ~MODE_MEAS_VEC;
    const int b = blockIdx.x * BLOCK + threadIdx.x;
    const bool live = b < B && !(skip && skip[b < B ? b : 0]);
    const int bc = live ? b : 0;
    const int* my_ids = ids + (size_t)bc * M;
    const T* my_pos = pos + (size_t)bc * M * 3;
    const T* my_quat = quat + (size_t)bc * M * 4;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, my_tile());
    T P[RC::NCOVP], nom[L::NNOM];
    __shared__ MarkerLDS<T> tbl;
    MarkerGroup<T, G> mg;
    if (FBUS_X_CORRECT_STAGGER > 0 && ((blockIdx.x >> FBUS_X_CORRECT_STAGGER_BIT) & 1)) {
#pragma unroll
        for (int i = 0; i < FBUS_X_CORRECT_STAGGER / 4; ++i) __builtin_amdgcn_s_sleep(127);
        if (FBUS_X_CORRECT_STAGGER % 4) __builtin_amdgcn_s_sleep(32 * (FBUS_X_CORRECT_STAGGER % 4) - 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the stacked path asks for the predict-invariant covariance tail behind the fold: fewer registers are tied up
    // while the rows are built, and the first scalar update only needs it for its last rows
    constexpr int C_SPLIT = JOINT ? FBUS_X_SPLIT : RC::NCH;
#ifndef FBUS_X_NEAREST_INFO
#define FBUS_X_NEAREST_INFO 1     // the reference mode (one marker, 7 rows) through the information form too: 6 passes, streamed stores
#endif
    // fp32 reference mode: the 7 rows of the chosen marker are folded like the rows of the stacked mode and applied as six
    // rank-1 passes whose last one streams the stores (instead of 7 row-by-row updates and a store phase behind them); the
    // same posterior -- the reference itself solves the 7 x 7 system at once (inv / LDLT), neither form is its operation order
    constexpr bool NEAREST_INFO = !JOINT && FBUS_X_NEAREST_INFO && COV == COV_SIMPLE;
    constexpr bool STREAM_ST = (JOINT || NEAREST_INFO) && FBUS_X_STREAM_ST;
    // fp64 (LEAN): 171 covariance doubles are 342 of the 512 registers.  The six passes run in the row-split form
    // (joint_apply_early / joint_apply_late, ekf_device.hpp): factorise first, then bring in storage rows 0..8 only, run
    // the passes on them (the last one streams them out), then bring in rows 9..17 and give them their six rank-1
    // terms from the LDS stash.  The nominal state is read a second time behind the passes instead of being held
    // across them, and the reference mode (one marker, 7 rows) goes through the same information form as the stacked
    // mode (6 passes, the same posterior) instead of holding the 37 Jacobian entries of its 7 rows live.
    // fp32 (SPLIT chosen by the launcher for launches of >= 2048 waves): the same form needs 194 registers instead of
    // 355, so two waves share a SIMD and overlap each other's load / pass / store phases: 44.4 -> 41.1 us at 131 072
    // filters, 73.0 -> 71.9 us at 262 144; with one wave per SIMD (65 536 filters) it is slower, 21.9 -> 25.4 us
    // (the covariance is requested behind the fold instead of under it) -- profiles/logs/r02_ab8.log.
    constexpr bool LEAN = SPLIT;
    constexpr int RS = 9;
    using Stash = LateStash<T, N, RS>;
    using Hook = RowStore<T, N, FBUS_X_CORRECT_ST>;
    __shared__ T stash_mem[LEAN ? Stash::NVAL * BLOCK : 1];
    const Stash stash{ stash_mem + threadIdx.x };
    constexpr int E_END = cov_final_before_row<N>(RS);                       // first storage index of a late row
    constexpr int C_E = RC::CH_NOM + (E_END + RC::EPC - 1) / RC::EPC;        // early chunks: [CH_NOM, C_E)
    T prev_raw = T(0);
    {
        MarkerTableRegs<T> treg;
        treg.load(dc);
        order_fence();
        if (!JOINT && DIALECT == DIALECT_CPP) prev_raw = recs[elem_index<T, N>(bc, L::OFF_PREV)];
        if (M > 0) mg.fetch(my_ids, my_pos, my_quat, 0, M, vec);
        order_fence();
        // the whole nominal state up front (p, q, R for the rows; v, ba, bg, g only for the injection -- 12 registers
        // that save a dependent reload between the last update and the stores)
        load_chunks<T, N, 0, RC::CH_NOM>(rs, my_lane(), nom);
        order_fence();
        // fp64 (LEAN): the covariance is requested behind the factorisation instead (342 registers of load targets)
        if constexpr (!LEAN) load_chunks<T, N, RC::CH_NOM, C_SPLIT, FBUS_X_CORRECT_LD>(rs, my_lane(), P);
        order_fence();
        treg.to_lds(tbl);
        order_fence();
    }
    const int last = live ? M : 0;
    if (last == 0) mg.n = 0;
    T dx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) dx[i] = T(0);
    int used = 0, new_prev = -1;
    auto lean_passes = [&](InfoFactors<T>& fac, bool go) {
        order_fence();
        load_chunks<T, N, RC::CH_NOM, C_E, AUX_NT>(rs, my_lane(), P);
        if (go) joint_apply_early<T, N, COV, RS, Hook>(P, dx, fac, Hook{ rs, my_lane(), P }, stash);
        order_fence();
        load_chunks<T, N, C_E, RC::NCH, AUX_NT>(rs, my_lane(), P + (C_E - RC::CH_NOM) * RC::EPC);
        if (go) joint_apply_late<T, N, COV, RS>(P, dx, stash);
    };

    if constexpr (JOINT) {
        // all visible markers: their rows are folded into the 6x6 information matrix while the covariance is still
        // on its way in, then applied as six scalar updates (joint_update)
        InfoAcc<T> acc;
        PoseFold<T, N, DIALECT> fold;
        fold.clear();
        MarkerCommon<T, N> mc;
        mc.build(nom, dc);
        auto fold_group = [&]() {
            mg.resolve(tbl);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (mg.slot[g] < 0) continue;
                fold.add(nom, dc, mc, mg.mk[g], mg.yp[g], mg.yq[g]);
                ++used;
            }
        };
        // the first group (fetched in the prologue) unconditionally -- a lane without markers has n = 0 and folds
        // nothing -- so that the nominal loads it needs stay where they were issued; further groups in a loop
        fold_group();
        for (int i0 = G; i0 < last; i0 += G) {
            mg.fetch(my_ids, my_pos, my_quat, i0, last, vec);
            fold_group();
        }
        fold.finish(acc, nom, dc, mc);
        if constexpr (LEAN) {
            InfoFactors<T> fac;
            if (used > 0) joint_factor<T>(acc, fac);
            lean_passes(fac, used > 0);
        } else {
            order_fence();
            load_chunks<T, N, C_SPLIT, RC::NCH, FBUS_X_CORRECT_LD>(rs, my_lane(), P + (C_SPLIT - RC::CH_NOM) * RC::EPC);
            // the last of the six passes stores every covariance chunk as soon as its rows are final (FBUS_X_STREAM_ST)
            if (used > 0) {
                if constexpr (STREAM_ST) joint_update<T, N, COV>(P, dx, acc, RowStore<T, N, FBUS_X_CORRECT_ST>{ rs, my_lane(), P });
                else joint_update<T, N, COV>(P, dx, acc);
            }
        }
    } else {
        // nearest visible marker, start threshold 10   MeasureUpdate.m:51-60 ; filter.cpp:639-664.
        // The candidates keep their measurement with them: no reload behind the covariance stream.
        const int prev_id = (int)prev_raw;
        int min_id = -1, pv_id = -1;
        T min_d = T(10), prev_d = T(0);
        T min_y[7], pv_y[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) { min_y[k] = T(0); pv_y[k] = T(0); }
        auto scan_group = [&]() {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int id = mg.id[g];
                if (g >= mg.n || id < 0) continue;
                const T x = mg.yp[g][0], y = mg.yp[g][1], z = mg.yp[g][2];
                const T dist = fb_sqrt(x * x + y * y + z * z);
                if (dist < min_d) {
                    min_d = dist; min_id = id;
#pragma unroll
                    for (int k = 0; k < 3; ++k) min_y[k] = mg.yp[g][k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) min_y[3 + k] = mg.yq[g][k];
                }
                if (DIALECT == DIALECT_CPP && id == prev_id) {
                    prev_d = dist; pv_id = id;
#pragma unroll
                    for (int k = 0; k < 3; ++k) pv_y[k] = mg.yp[g][k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pv_y[3 + k] = mg.yq[g][k];
                }
            }
        };
        scan_group();
        for (int i0 = G; i0 < last; i0 += G) {
            mg.fetch(my_ids, my_pos, my_quat, i0, last, vec);
            scan_group();
        }
        if (min_id >= 0 && DIALECT == DIALECT_CPP && pv_id >= 0 && fb_abs(prev_d - min_d) < dc.switch_thres && prev_d != T(0)) {
            min_id = pv_id;
#pragma unroll
            for (int k = 0; k < 7; ++k) min_y[k] = pv_y[k];
        }
        const bool ok = min_id >= 0 && min_id <= FBUS_MAX_MARKER_ID;
        const int slot = ok ? (int)tbl.id2slot[ok ? min_id : 0] : -1;              // filter.cpp:671-673
        if constexpr (LEAN) {
            InfoAcc<T> acc;
            InfoFactors<T> fac;
            acc.clear();
            if (slot >= 0) {
                if (DIALECT == DIALECT_CPP) new_prev = min_id;                       // filter.cpp:675
                T mk[MK_STRIDE];
#pragma unroll
                for (int k = 0; k < 8; ++k) mk[k] = tbl.mk[slot * MK_STRIDE + k];
                MarkerCommon<T, N> mc;
                mc.build(nom, dc);
                PoseFold<T, N, DIALECT> fold;
                fold.clear();
                fold.add(nom, dc, mc, mk, min_y, min_y + 3);
                fold.finish(acc, nom, dc, mc);
                joint_factor<T>(acc, fac);
                used = 1;
            }
            lean_passes(fac, used > 0);
        } else if (slot >= 0) {
            if (DIALECT == DIALECT_CPP) new_prev = min_id;                           // filter.cpp:675
            T mk[MK_STRIDE];
#pragma unroll
            for (int k = 0; k < 8; ++k) mk[k] = tbl.mk[slot * MK_STRIDE + k];
            MarkerCommon<T, N> mc;
            mc.build(nom, dc);
            if constexpr (NEAREST_INFO) {
                InfoAcc<T> acc;
                PoseFold<T, N, DIALECT> fold;
                fold.clear();
                fold.add(nom, dc, mc, mk, min_y, min_y + 3);
                fold.finish(acc, nom, dc, mc);
                if constexpr (STREAM_ST) joint_update<T, N, COV>(P, dx, acc, RowStore<T, N, FBUS_X_CORRECT_ST>{ rs, my_lane(), P });
                else joint_update<T, N, COV>(P, dx, acc);
            } else {
                marker_update<T, N, DIALECT, COV>(P, dx, nom, dc, mc, mk, min_y, min_y + 3);
            }
            used = 1;
        }
    }
    if (used == 0) { if (b < B) applied[b] = 0; return; }
    if constexpr (LEAN) {               // the nominal state was not held across the passes: read it again (L2-hot)
        order_fence();
        load_chunks<T, N, 0, RC::CH_NOM>(rs, my_lane(), nom);
    }
    inject<T, N>(nom, dx);
    if (new_prev >= 0) P[L::OFF_PREV - L::OFF_COV] = (T)new_prev;
    // the carried rotation is NOT refreshed by MeasureUpdate: chunks holding only R are left alone
    store_chunks<T, N, 0, RC::CH_PQ, FBUS_X_CORRECT_ST>(rs, my_lane(), nom);
    store_chunks<T, N, RC::CH_PQR, RC::CH_NOM, FBUS_X_CORRECT_ST>(rs, my_lane(), nom + L::NPQR);
    constexpr int C_REST = LEAN ? Hook::fin(RS) : (STREAM_ST ? Hook::streamed_end() : RC::CH_NOM);
    store_chunks<T, N, C_REST, RC::NCH, FBUS_X_CORRECT_ST>(rs, my_lane(), P + (C_REST - RC::CH_NOM) * RC::EPC);
    applied[b] = 1;
}

// One camera frame in ONE launch: K ImuUpdates then one MeasureUpdate with the record resident in
// registers in between (the reference's BatchImuProcessing + ObservationUpdate, filter.cpp:232-235).
// Same device functions, same arithmetic as K predict launches + one correct launch; the record makes
// one HBM round trip per frame instead of one per EKF step.
template <typename T, int N, int DIALECT, int COV, bool JOINT>
__global__ void __launch_bounds__(BLOCK, FBUS_X_FRAME_WAVES)
frame_kernel(T* __restrict__ recs, int B, int K, const T* __restrict__ accel, const T* __restrict__ gyro,
             const T* __restrict__ dt, int dt_stride, int M, const int* __restrict__ ids, const T* __restrict__ pos,
             const T* __restrict__ quat, int mode, const unsigned char* __restrict__ skip,
             unsigned char* __restrict__ applied, DevConst<T> dc)
{
    using L = Lay<N>;
    using RC = Rec<T, N>;
    const int b = blockIdx.x * BLOCK + threadIdx.x;
    __shared__ MarkerLDS<T> tbl;                      // the marker map, looked up from LDS (see MarkerLDS)
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, my_tile());
    T nom[L::NNOM], P[RC::NCOVP];
    simd_stagger<FBUS_X_STAGGER_FRAME>();
    {
        // map pieces first, the record behind them, the LDS copy after both are on their way (lanes past B load
        // their existing tile too: no branch in front of the loads)
        MarkerTableRegs<T> treg;
        treg.load(dc);
        order_fence();
        load_chunks<T, N, 0, RC::CH_NOM, AUX_NT>(rs, my_lane(), nom);
        load_chunks<T, N, RC::CH_NOM, RC::NCH, AUX_NT>(rs, my_lane(), P);
        order_fence();
        treg.to_lds(tbl);
        order_fence();
    }
    if (b >= B) return;
    int first = 0, last = (M > 0 && !(skip && skip[b])) ? M : 0;
    int new_prev = -1;
    const int* my_ids = ids + (size_t)b * M;
    const T* my_pos = pos + (size_t)b * M * 3;
    const T* my_quat = quat + (size_t)b * M * 4;
    predict_steps<T, N, DIALECT>(nom, P, K, accel, gyro, dt, dt_stride, B, b, dc.qd);

    if (last > 0 && mode == MODE_NEAREST) {
        const int prev_id = (DIALECT == DIALECT_CPP) ? (int)P[L::OFF_PREV - L::OFF_COV] : 0;
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
            slot = (id <= FBUS_MAX_MARKER_ID) ? dc.id2slot[id] : -1;
        }
        if (slot < 0) { first = last = 0; }
        else {
            if (DIALECT == DIALECT_CPP) new_prev = id;
            first = min_i; last = min_i + 1;
        }
    }
    T dx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) dx[i] = T(0);
    int used = 0;
    InfoAcc<T> acc;
    PoseFold<T, N, DIALECT> fold;
    // the reference mode in the simple form goes through the information fold as well (see correct_kernel: NEAREST_INFO)
    constexpr bool joint = JOINT || COV == COV_SIMPLE;
    if (joint) fold.clear();
    MarkerCommon<T, N> mc;
    mc.build(nom, dc);
    for (int i0 = first; i0 < last; i0 += FBUS_MARKER_GROUP) {
        MarkerGroup<T, FBUS_MARKER_GROUP> mg;
        mg.fetch(my_ids, my_pos, my_quat, i0, last);
        mg.resolve(tbl);
#pragma unroll
        for (int g = 0; g < FBUS_MARKER_GROUP; ++g) {
            if (mg.slot[g] < 0) continue;
            if constexpr (joint) fold.add(nom, dc, mc, mg.mk[g], mg.yp[g], mg.yq[g]);
            else marker_update<T, N, DIALECT, COV>(P, dx, nom, dc, mc, mg.mk[g], mg.yp[g], mg.yq[g]);
            ++used;
        }
    }
    constexpr bool STREAM_ST = joint && FBUS_X_STREAM_ST;
    bool streamed = false;
    if constexpr (joint) {
        if (used > 0) {
            fold.finish(acc, nom, dc, mc);
            if constexpr (STREAM_ST) { joint_update<T, N, COV>(P, dx, acc, RowStore<T, N, FBUS_X_FRAME_ST>{ rs, my_lane(), P }); streamed = true; }
            else joint_update<T, N, COV>(P, dx, acc);
        }
    }
    if (used > 0) {
        inject<T, N>(nom, dx);
        if (new_prev >= 0) P[L::OFF_PREV - L::OFF_COV] = (T)new_prev;
    }
    if (M > 0) applied[b] = used > 0 ? 1 : 0;
    store_chunks<T, N, 0, RC::CH_NOM, FBUS_X_FRAME_ST>(rs, my_lane(), nom);
    constexpr int C_REST = RowStore<T, N, FBUS_X_FRAME_ST>::streamed_end();
    if (!streamed) store_chunks<T, N, RC::CH_NOM, C_REST, FBUS_X_FRAME_ST>(rs, my_lane(), P);
    store_chunks<T, N, C_REST, RC::NCH, FBUS_X_FRAME_ST>(rs, my_lane(), P + (C_REST - RC::CH_NOM) * RC::EPC);
}

// A WINDOW of camera frames in one launch (offline replay: the frame loop of FBUS_EKF.m:151-210 / FilterThreadFunction,
// filter.cpp:229-235, over a recorded stretch): F times { K_f ImuUpdates, one MeasureUpdate } with the record resident in
// registers from the first load to the last store.  frame_kernel pays the record's way in and out once per frame -- at one
// wave per SIMD ~15 of its 38 us are that head and tail, during which the wave cannot compute; here they are paid once per
// window.  Same device functions, same order of operations per filter as F fused frames (bit-identical results).
struct FrameCounts { unsigned char k[FBUS_MAX_WINDOW_FRAMES]; };      // IMU samples in front of each frame
template <typename T, int N, int DIALECT, int COV, bool JOINT>
__global__ void __launch_bounds__(BLOCK)
frames_kernel(T* __restrict__ recs, int B, int F, FrameCounts kc, const T* __restrict__ accel, const T* __restrict__ gyro,
              const T* __restrict__ dt, int dt_stride, int M, const int* __restrict__ ids, const T* __restrict__ pos,
              const T* __restrict__ quat, int mode, const unsigned char* __restrict__ skip,
              unsigned char* __restrict__ applied, DevConst<T> dc)
{
    using L = Lay<N>;
    using RC = Rec<T, N>;
    const int b = blockIdx.x * BLOCK + threadIdx.x;
    __shared__ MarkerLDS<T> tbl;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, my_tile());
    T nom[L::NNOM], P[RC::NCOVP];
    {
        MarkerTableRegs<T> treg;
        treg.load(dc);
        order_fence();
        load_chunks<T, N, 0, RC::CH_NOM, AUX_NT>(rs, my_lane(), nom);
        load_chunks<T, N, RC::CH_NOM, RC::NCH, AUX_NT>(rs, my_lane(), P);
        order_fence();
        treg.to_lds(tbl);
        order_fence();
    }
    if (b >= B) return;
    int k0 = 0, last_used = 0;
#pragma unroll 1
    for (int f = 0; f < F; ++f) {
        const int K = kc.k[f];
        predict_steps<T, N, DIALECT>(nom, P, K, accel + (size_t)k0 * B * 3, gyro + (size_t)k0 * B * 3,
                                     dt + (size_t)k0 * (dt_stride ? B : 1), dt_stride, B, b, dc.qd);
        k0 += K;
        const size_t fo = (size_t)f * B + b;
        int first = 0, last = (M > 0 && !(skip && skip[fo])) ? M : 0;
        int new_prev = -1;
        const int* my_ids = ids + fo * M;
        const T* my_pos = pos + fo * M * 3;
        const T* my_quat = quat + fo * M * 4;
        if (last > 0 && mode == MODE_NEAREST) {
            const int prev_id = (DIALECT == DIALECT_CPP) ? (int)P[L::OFF_PREV - L::OFF_COV] : 0;
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
        T dx[N];
#pragma unroll
        for (int i = 0; i < N; ++i) dx[i] = T(0);
        int used = 0;
        InfoAcc<T> acc;
        PoseFold<T, N, DIALECT> fold;
        constexpr bool INFO = JOINT || COV == COV_SIMPLE;      // reference mode, simple form: through the information fold too
        if (INFO) fold.clear();
        MarkerCommon<T, N> mc;
        mc.build(nom, dc);
        for (int i0 = first; i0 < last; i0 += FBUS_MARKER_GROUP) {
            MarkerGroup<T, FBUS_MARKER_GROUP> mg;
            mg.fetch(my_ids, my_pos, my_quat, i0, last);
            mg.resolve(tbl);
#pragma unroll
            for (int g = 0; g < FBUS_MARKER_GROUP; ++g) {
                if (mg.slot[g] < 0) continue;
                if constexpr (INFO) fold.add(nom, dc, mc, mg.mk[g], mg.yp[g], mg.yq[g]);
                else marker_update<T, N, DIALECT, COV>(P, dx, nom, dc, mc, mg.mk[g], mg.yp[g], mg.yq[g]);
                ++used;
            }
        }
        if constexpr (INFO) {
            if (used > 0) { fold.finish(acc, nom, dc, mc); joint_update<T, N, COV>(P, dx, acc); }
        }
        if (used > 0) {
            inject<T, N>(nom, dx);
            if (new_prev >= 0) P[L::OFF_PREV - L::OFF_COV] = (T)new_prev;
        }
        last_used = used;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (M > 0 && F > 0) applied[b] = last_used > 0 ? 1 : 0;
    store_chunks<T, N, 0, RC::CH_NOM, FBUS_X_FRAME_ST>(rs, my_lane(), nom);
    store_chunks<T, N, RC::CH_NOM, RC::NCH, FBUS_X_FRAME_ST>(rs, my_lane(), P);
}

// The fused frame for launches of >= 2048 waves: at most 256 registers, two waves per SIMD (stacked mode, simple form).
// frame_kernel holds the record in 383 registers and is VALU-bound with one wave per SIMD at 57 % issue utilisation;
// from 131 072 filters on a second wave per SIMD fills its stalls.  What makes the 256 registers possible:
//   * the predict loop parks rows p of the covariance and 16 of the 28 nominal values in LDS between their uses (StepPark);
//   * behind the loop the late covariance rows (storage rows 9..17, final as predicted) go out to their record at once --
//     they come back from L2 for the late half of the update, exactly as in the row-split correct_kernel -- and the head
//     of the early rows waits in LDS while the marker rows are folded and the 6x6 information matrix is factorised;
//   * the six rank-1 passes then run in the row-split form (joint_apply_early / joint_apply_late), their stash in the
//     same LDS area (17 KiB per wave + the 3 KiB marker map = 20 KiB: eight workgroups per CU).
// Same device functions as K predict launches + one correct launch; per filter the order of operations is that of the
// row-split correct_kernel.
template <typename T, int N, int DIALECT>
__global__ void __launch_bounds__(BLOCK, sizeof(T) == 4 ? 2 : 1)
frame2_kernel(T* __restrict__ recs, int B, int K, const T* __restrict__ accel, const T* __restrict__ gyro,
              const T* __restrict__ dt, int dt_stride, int M, const int* __restrict__ ids, const T* __restrict__ pos,
              const T* __restrict__ quat, const unsigned char* __restrict__ skip, unsigned char* __restrict__ applied,
              DevConst<T> dc)
{
    using L = Lay<N>;
    using RC = Rec<T, N>;
    constexpr int RS = 9, EPC = RC::EPC, CN = RC::CH_NOM;
    using Park = StepPark<T, N, park_nom_chunks<T>()>;
    using Stash = LateStash<T, N, RS>;
    using Hook = RowStore<T, N, AUX_DEFAULT>;
    // the stash of the passes reuses the parking area (fp32: 17 chunks hold both; fp64: the stash, 66 doubles, is the larger one)
    constexpr int LDS_CH = Park::NCHUNK * 16 >= (int)(Stash::NVAL * sizeof(T)) ? Park::NCHUNK : (int)((Stash::NVAL * sizeof(T) + 15) / 16);
    const int b = blockIdx.x * BLOCK + threadIdx.x;
    // fp64 records: the parking area alone is 39 KiB; with the 4 KiB marker map beside it a workgroup would need 43 KiB and only
    // three of them fit a CU's 160 KiB -- 1024 workgroups then run as two rounds (measured: 13.3 us per resident step instead of
    // predict_n's 6.1).  The map therefore takes the last chunks of the parking area and is copied in BEHIND the predict loop
    // (the head of the covariance parks four chunks fewer while the marker rows are folded): 39 KiB, four workgroups per CU.
    constexpr bool MAP_LATE = sizeof(T) == 8;
    constexpr int TBL_CH = MAP_LATE ? (int)((sizeof(MarkerLDS<T>) + 16 * BLOCK - 1) / (16 * BLOCK)) : 0;
    constexpr int HEAD_CH = Park::NCHUNK - TBL_CH;                        // chunks of the covariance head parked during the fold
    static_assert(!MAP_LATE || Stash::NVAL * sizeof(T) <= (size_t)(LDS_CH - TBL_CH) * 16, "stash and map overlap");
    __shared__ std::conditional_t<MAP_LATE, int, MarkerLDS<T>> tbl_own;   // (fp64: never referenced, no LDS)
    __shared__ u32x4 lds_mem[LDS_CH * BLOCK];
    MarkerLDS<T>* tblp;
    if constexpr (MAP_LATE) tblp = reinterpret_cast<MarkerLDS<T>*>(lds_mem + (LDS_CH - TBL_CH) * BLOCK);
    else tblp = &tbl_own;
    MarkerLDS<T>& tbl = *tblp;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, my_tile());
    T nom[L::NNOM], P[RC::NCOVP];
    if constexpr (MAP_LATE) {
        load_chunks<T, N, 0, CN, AUX_NT>(rs, my_lane(), nom);
        load_chunks<T, N, CN, RC::NCH, AUX_NT>(rs, my_lane(), P);
    } else {
        MarkerTableRegs<T> treg;
        treg.load(dc);
        order_fence();
        load_chunks<T, N, 0, CN, AUX_NT>(rs, my_lane(), nom);
        load_chunks<T, N, CN, RC::NCH, AUX_NT>(rs, my_lane(), P);
        order_fence();
        treg.to_lds(tbl);
        order_fence();
    }
    if (b >= B) return;
    const int last = (M > 0 && !(skip && skip[b])) ? M : 0;
    const int* my_ids = ids + (size_t)b * M;
    const T* my_pos = pos + (size_t)b * M * 3;
    const T* my_quat = quat + (size_t)b * M * 4;
    const Park park{ lds_mem + threadIdx.x };
    predict_steps<T, N, DIALECT, NoMidHook, Park, FBUS_X_PACK_2W>(nom, P, K, accel, gyro, dt, dt_stride, B, b, dc.qd, NoMidHook(), park);
    order_fence();
    // the late rows are final as predicted: out to the record now (they return from L2 for joint_apply_late)
    constexpr int E_END = cov_final_before_row<N>(RS);
    constexpr int C_E = CN + (E_END + EPC - 1) / EPC;                    // early chunks: [CN, C_E)
    store_chunks<T, N, C_E, RC::NCH>(rs, my_lane(), P + (C_E - CN) * EPC);
    int used = 0;
    InfoFactors<T> fac;
    T dx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) dx[i] = T(0);
    if constexpr (MAP_LATE) {
        // every lane of the batch takes part (also those whose frame is skipped); the lanes past the batch end have left: the copy
        // is shared by the nact lanes that remain (wave-uniform)
        const int nact = min(BLOCK, B - (int)blockIdx.x * BLOCK);
        const u32x4* si = reinterpret_cast<const u32x4*>(dc.id2slot);
        const u32x4* sm = reinterpret_cast<const u32x4*>(dc.mk);
        u32x4* di = reinterpret_cast<u32x4*>(tbl.id2slot);
        u32x4* dm = reinterpret_cast<u32x4*>(tbl.mk);
        for (int i = threadIdx.x; i < MarkerTableRegs<T>::NI; i += nact) di[i] = si[i];
        for (int i = threadIdx.x; i < MarkerTableRegs<T>::NM; i += nact) dm[i] = sm[i];
        order_fence();
    }
    if (last > 0) {
        // the head of the early rows waits in LDS while the rows of the markers are built and folded
#pragma unroll
        for (int c = 0; c < HEAD_CH; ++c) park.put(c, P + c * EPC, EPC);
        order_fence();
        InfoAcc<T> acc;
        PoseFold<T, N, DIALECT> fold;
        fold.clear();
        MarkerCommon<T, N> mc;
        mc.build(nom, dc);
        for (int i0 = 0; i0 < last; i0 += FBUS_MARKER_GROUP) {
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
        if (used > 0) { fold.finish(acc, nom, dc, mc); joint_factor<T>(acc, fac); }
        order_fence();
#pragma unroll
        for (int c = 0; c < HEAD_CH; ++c) park.get(c, P + c * EPC, EPC);
        order_fence();
    }
    if (M > 0) applied[b] = used > 0 ? 1 : 0;
    if (used > 0) {
        const Stash stash{ reinterpret_cast<T*>(lds_mem) + threadIdx.x };
        joint_apply_early<T, N, COV_SIMPLE, RS, Hook>(P, dx, fac, Hook{ rs, my_lane(), P }, stash);   // the last pass streams rows 0..8 out
        order_fence();
        load_chunks<T, N, C_E, RC::NCH>(rs, my_lane(), P + (C_E - CN) * EPC);
        joint_apply_late<T, N, COV_SIMPLE, RS>(P, dx, stash);
        inject<T, N>(nom, dx);
        store_chunks<T, N, 0, CN>(rs, my_lane(), nom);
        constexpr int C_REST = Hook::fin(RS);
        store_chunks<T, N, C_REST, RC::NCH>(rs, my_lane(), P + (C_REST - CN) * EPC);
    } else {
        // no marker for this filter: the predicted record as it is (its late chunks are out already)
        store_chunks<T, N, 0, CN>(rs, my_lane(), nom);
        store_chunks<T, N, CN, C_E>(rs, my_lane(), P);
    }
}

// (correct() from stereo corners / from corner pixels: csrc/ekf_meas.hpp.  The round-1..3 kernels that lived here -- fp32 fold, six
// sequential rank-1 passes -- were removed in round 4.)

// One marker per lane: corners (stereo pairs or 3-D) -> marker pose in the left camera frame.
// The arithmetic runs in DOUBLE whatever the array type T (round 5; the reference computes in double, vision.cpp:472-759):
// fp32 triangulation left the corner positions ~2e-6 m off, which the pose update behind it amplified to 10x the parity
// gate (tests/test_vision_gpu.py); the kernel is a front-door step (one launch per camera frame, 64 B in / 28 B out per
// marker) and fp64 FMAs cost what fp32 ones cost on this part.  Outputs are rounded to T once.
template <typename T>
__global__ void __launch_bounds__(256)
marker_pose_kernel(int n, int geometry, const T* __restrict__ left, const T* __restrict__ right,
                   T* __restrict__ pos, T* __restrict__ quat, T* __restrict__ corners3d, VisConst<double> vc)
{
    using S = double;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    S C[12];
    if (geometry == VIS_CORNERS3D) {
#pragma unroll
        for (int k = 0; k < 12; ++k) C[k] = (S)left[(size_t)i * 12 + k];
    } else {
        S l[8], r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { l[k] = (S)left[(size_t)i * 8 + k]; r[k] = (S)right[(size_t)i * 8 + k]; }
        if (geometry == VIS_REFRACTIVE) {
#pragma unroll
            for (int c = 0; c < 4; ++c) refraction_corner(vc, l[2 * c], l[2 * c + 1], r[2 * c], r[2 * c + 1], C + 3 * c);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) pinhole_corner(vc, l[2 * c], l[2 * c + 1], r[2 * c], r[2 * c + 1], C + 3 * c);
        }
    }
    S p[3], q[4];
    marker_pose(C, p, q);
#pragma unroll
    for (int k = 0; k < 3; ++k) pos[(size_t)i * 3 + k] = (T)p[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) quat[(size_t)i * 4 + k] = (T)q[k];
    if (corners3d) {
#pragma unroll
        for (int k = 0; k < 12; ++k) corners3d[(size_t)i * 12 + k] = (T)C[k];
    }
}

// ---- init / reset / front door (rows f-2, f-4): one filter per lane, element-wise record access -------
// gravity = (0, 0, -|mean accel|), gyro bias = mean gyro over T samples (accel/gyro T x B x 3).
// InitGravityAndGyrobias.m:36-40 ; FILTER::InitializeGravityAndBias filter.cpp:256-285
template <typename T, int N>
__global__ void init_gravity_bias_kernel(T* __restrict__ recs, int B, int Tn, const T* __restrict__ accel,
                                         const T* __restrict__ gyro)
{
    using L = Lay<N>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    T ma[3] = { T(0), T(0), T(0) }, mg[3] = { T(0), T(0), T(0) };
    for (int t = 0; t < Tn; ++t) {
        const size_t o = ((size_t)t * B + b) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) { ma[i] += accel[o + i]; mg[i] += gyro[o + i]; }
    }
    const T inv = T(1) / T(Tn);
#pragma unroll
    for (int i = 0; i < 3; ++i) { ma[i] *= inv; recs[elem_index<T, N>(b, L::OFF_BG + i)] = mg[i] * inv; }
    recs[elem_index<T, N>(b, L::OFF_G + 0)] = T(0);
    recs[elem_index<T, N>(b, L::OFF_G + 1)] = T(0);
    recs[elem_index<T, N>(b, L::OFF_G + 2)] = -fb_sqrt(ma[0] * ma[0] + ma[1] * ma[1] + ma[2] * ma[2]);
}

// IMU pose from the nearest marker: init / reset / vision-only (what = 0 / 1 / 2).
// InitPositionAndQuaternion.m:38-80, ResetState.m:37-80, ComputeVisionOnlyResults.m:39-79 ;
// FILTER::InitializePose filter.cpp:291-399, FILTER::ResetSystemState filter.cpp:405-477
template <typename T, int N, int DIALECT>
__global__ void pose_init_kernel(T* __restrict__ recs, int B, int M, const int* __restrict__ ids,
                                 const T* __restrict__ pos, const T* __restrict__ quat, int what, T max_dist,
                                 const unsigned char* __restrict__ mask, T* __restrict__ out7,
                                 unsigned char* __restrict__ applied, DevConst<T> dc)
{
    using L = Lay<N>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    applied[b] = 0;
    if (mask && !mask[b]) return;
    int mi = -1;
    T md = T(10);
    for (int i = 0; i < M; ++i) {
        if (ids[(size_t)b * M + i] < 0) continue;
        const T* y = pos + ((size_t)b * M + i) * 3;
        const T d = fb_sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
        if (d < md) { md = d; mi = i; }
    }
    if (mi < 0) return;
    if (DIALECT == DIALECT_CPP && max_dist > T(0) && md > max_dist) return;     // filter.cpp:343-347,432-436
    const int id = ids[(size_t)b * M + mi];
    const int slot = (id <= FBUS_MAX_MARKER_ID) ? dc.id2slot[id] : -1;
    if (slot < 0) return;                                                        // filter.cpp:355-359,444-448
    const T* mk = dc.mk + (size_t)slot * MK_STRIDE;
    const T* yp = pos + ((size_t)b * M + mi) * 3;
    const T* yq = quat + ((size_t)b * M + mi) * 4;
    const T Qm[4] = { mk[3], mk[4], mk[5], mk[6] };
    const T qc[4] = { yq[0], -yq[1], -yq[2], -yq[3] };
    T t4[4], q[4], R[9];
    quat_mul(Qm, qc, t4);
    quat_mul(t4, dc.Q_IL, q);                       // Q_IG = Q_MG (x) Q_ML* (x) Q_IL
    if (what == 2) quat_normalize(q);               // ComputeVisionOnlyResults.m:67
    if (DIALECT == DIALECT_CPP) quat_to_rotmat_e(q, R); else quat_to_rotmat_m(q, R);
    T a[3], p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = dc.R_IL[i] * yp[0] + dc.R_IL[3 + i] * yp[1] + dc.R_IL[6 + i] * yp[2];   // R_IL' P_ML
#pragma unroll
    for (int i = 0; i < 3; ++i)                     // P_IG = -R_IG R_IL' P_ML + P_MG - R_IG P_IL
        p[i] = -(R[3 * i] * a[0] + R[3 * i + 1] * a[1] + R[3 * i + 2] * a[2]) + mk[i]
               - (R[3 * i] * dc.P_IL[0] + R[3 * i + 1] * dc.P_IL[1] + R[3 * i + 2] * dc.P_IL[2]);
    applied[b] = 1;
    if (what == 2) {
#pragma unroll
        for (int i = 0; i < 3; ++i) out7[(size_t)b * 7 + i] = p[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) out7[(size_t)b * 7 + 3 + i] = q[i];
        return;
    }
    auto put = [&](int e, T v) { recs[elem_index<T, N>(b, e)] = v; };
#pragma unroll
    for (int i = 0; i < 3; ++i) put(L::OFF_P3 + i, p[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) put(L::OFF_Q + i, q[i]);
    if (what == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) put(L::OFF_R + i, R[i]);
        put(L::OFF_G, T(9.8)); put(L::OFF_G + 1, T(0)); put(L::OFF_G + 2, T(0));   // InitPositionAndQuaternion.m:79
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) { put(L::OFF_V + i, T(0)); put(L::OFF_BA + i, T(0)); }
        if (DIALECT == DIALECT_CPP) {
#pragma unroll
            for (int i = 0; i < 3; ++i) put(L::OFF_BG + i, T(0));                    // filter.cpp:470 ; rotmatI2G stays stale
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) put(L::OFF_R + i, R[i]);                     // ResetState.m:77
        }
    }
}

// y[t] = 0.9 y[t-1] + 0.1 x[t] per filter over T samples, in place (FILTER::SetImuData filter.cpp:36-47).
// accel/gyro: T x B x 3; carry: B x 6 previous filtered sample (read if have_carry, always written).
template <typename T>
__global__ void imu_ema_kernel(int B, int Tn, T* __restrict__ accel, T* __restrict__ gyro, T* __restrict__ carry,
                               int have_carry)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const T c = T(0.1);
    T prev[6];
    if (have_carry) {
#pragma unroll
        for (int i = 0; i < 6; ++i) prev[i] = carry[(size_t)b * 6 + i];
    }
    for (int t = 0; t < Tn; ++t) {
        const size_t o = ((size_t)t * B + b) * 3;
        T x[6] = { accel[o], accel[o + 1], accel[o + 2], gyro[o], gyro[o + 1], gyro[o + 2] };
        if (t > 0 || have_carry) {
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = prev[i] * (T(1) - c) + x[i] * c;
#pragma unroll
            for (int i = 0; i < 3; ++i) { accel[o + i] = x[i]; gyro[o + i] = x[3 + i]; }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) prev[i] = x[i];
    }
    if (Tn > 0 && carry) {
#pragma unroll
        for (int i = 0; i < 6; ++i) carry[(size_t)b * 6 + i] = prev[i];
    }
}

// The L0 helpers one by one (unit-test hook behind fbus_ekf_l0_eval): exactly the inline functions the kernels use.
template <typename T>
__global__ void l0_eval_kernel(int op, int n, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == FBUS_L0_QUAT_MUL) {
        quat_mul(a + 4 * i, b + 4 * i, out + 4 * i);
    } else if (op == FBUS_L0_QUAT_TO_ROTMAT_M) {
        quat_to_rotmat_m(a + 4 * i, out + 9 * i);
    } else if (op == FBUS_L0_QUAT_TO_ROTMAT_E) {
        quat_to_rotmat_e(a + 4 * i, out + 9 * i);
    } else if (op == FBUS_L0_QUAT_NORMALIZE) {
        T q[4] = { a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3] };
        quat_normalize(q);
        for (int k = 0; k < 4; ++k) out[4 * i + k] = q[k];
    } else if (op == FBUS_L0_EXPM_SO3_NEG) {
        // F(theta, theta) of the Matlab dialect through predict_nominal itself (zero biases, identity attitude)
        T nom[Lay<18>::NNOM] = {};
        nom[Lay<18>::OFF_Q] = T(1); nom[Lay<18>::OFF_R] = T(1); nom[Lay<18>::OFF_R + 4] = T(1); nom[Lay<18>::OFF_R + 8] = T(1);
        const T acc[3] = { T(0), T(0), T(0) };
        PredictCoef<T> k;
        predict_nominal<T, 18, DIALECT_MATLAB>(nom, acc, a + 3 * i, b[i], k);
        for (int j = 0; j < 9; ++j) out[9 * i + j] = k.Th[j];
    } else if (op == FBUS_L0_DTHETA_TO_QUAT) {
        T rec[Lay<18>::NNOM] = {};
        rec[Lay<18>::OFF_Q] = T(1);
        T dx[18] = {};
        dx[6] = a[3 * i]; dx[7] = a[3 * i + 1]; dx[8] = a[3 * i + 2];
        inject<T, 18>(rec, dx);                           // q = 1 (x) dq(dtheta), normalised
        for (int k2 = 0; k2 < 4; ++k2) out[4 * i + k2] = rec[Lay<18>::OFF_Q + k2];
    } else if (op == FBUS_L0_SINCOS_HALF) {
        T s_, c_, sh, ch;
        fb_sincos_x_halfx(a[i], s_, c_, sh, ch);
        out[4 * i] = s_; out[4 * i + 1] = c_; out[4 * i + 2] = sh; out[4 * i + 3] = ch;
    }
}

// AoS (API arrays) <-> records.  Not on the hot path, but the API arrays are filter-major (1296 bytes of P per filter) and the
// records lane-major, so a lane-per-filter copy touches a different cache line with every 4-byte access (the round-1 kernels:
// 0.35 of peak with 9x write amplification on the unpack side).  One wave per 64-filter tile: the tile's records pass
// through LDS ([filter][element], odd pitch), the records move as 1 KiB chunk loads / stores and the API arrays as
// lane-consecutive accesses over the tile's contiguous stretch of each array.
template <typename T, int N>
struct TileIO {
    using L = Lay<N>;
    using RC = Rec<T, N>;
    static constexpr int NR = RC::NRECP;
    static constexpr int PITCH = NR | 1;                   // odd: lanes writing their own record hit different banks
    // record element of API nominal element i (API order p v q ba bg g)
    __device__ static int nom_elem(int i)
    {
        return i < 3 ? L::OFF_P3 + i : i < 6 ? L::OFF_V + (i - 3) : i < 10 ? L::OFF_Q + (i - 6)
             : i < 13 ? L::OFF_BA + (i - 10) : i < 16 ? L::OFF_BG + (i - 13) : L::OFF_G + (i - 16);
    }
    __device__ static int cov_elem(int r)                  // r = i * N + j of the full matrix
    {
        const int i = r / N, j = r - i * N;
        return L::OFF_COV + (i <= j ? pidx<N>(i, j) : pidx<N>(j, i));
    }
};

template <typename T, int N>
__global__ void __launch_bounds__(BLOCK)
pack_kernel(T* __restrict__ recs, int B, const T* __restrict__ nominal, const T* __restrict__ rot,
            const T* __restrict__ P, const int* __restrict__ prev)
{
    using IO = TileIO<T, N>;
    using L = Lay<N>;
    using RC = Rec<T, N>;
    __shared__ T lds[BLOCK * IO::PITCH];
    __shared__ short tab[N * N];
    const int lane = threadIdx.x, tile = blockIdx.x;
    const int nvalid = min(BLOCK, B - tile * BLOCK);
    const size_t f0 = (size_t)tile * BLOCK;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, tile);
    T rec[IO::NR];
    // Only the parts of the record that the caller supplies are read, patched and written back (whatever is not supplied
    // stays as it is): the nominal chunks for nominal / rot, the covariance chunks for P, the last chunk for prev alone.
    const bool do_nom = nominal || rot, do_cov = P != nullptr, do_prev = prev != nullptr;
    constexpr int CN = RC::CH_NOM, EPC = RC::EPC, CL = RC::NCH - 1;
    if (do_nom) {
        load_chunks<T, N, 0, CN>(rs, lane, rec);
#pragma unroll
        for (int e = 0; e < CN * EPC; ++e) lds[lane * IO::PITCH + e] = rec[e];
    }
    if (do_cov) {
        load_chunks<T, N, CN, RC::NCH>(rs, lane, rec + CN * EPC);
#pragma unroll
        for (int e = CN * EPC; e < IO::NR; ++e) lds[lane * IO::PITCH + e] = rec[e];
        for (int r = lane; r < N * N; r += BLOCK) {            // tab[packed index] = i * N + j of the upper-triangle element stored there
            const int ii = r / N, jj = r - ii * N;
            if (ii <= jj) tab[IO::cov_elem(r) - L::OFF_COV] = (short)r;
        }
    } else if (do_prev) {
        load_chunks<T, N, CL, RC::NCH>(rs, lane, rec + CL * EPC);
#pragma unroll
        for (int e = CL * EPC; e < IO::NR; ++e) lds[lane * IO::PITCH + e] = rec[e];
    }
    __syncthreads();
    if (nominal)
        for (int o = lane; o < nvalid * 19; o += BLOCK) {
            const int f = o / 19, i = o - f * 19;
            lds[f * IO::PITCH + IO::nom_elem(i)] = nominal[f0 * 19 + o];
        }
    if (rot)
        for (int o = lane; o < nvalid * 9; o += BLOCK) {
            const int f = o / 9, i = o - f * 9;
            lds[f * IO::PITCH + L::OFF_R + i] = rot[f0 * 9 + o];
        }
    if (prev && lane < nvalid) lds[lane * IO::PITCH + L::OFF_PREV] = (T)prev[f0 + lane];
    if (P) {
        // the reference symmetrises every step: store the mean of the two halves.  One packed element per lane and iteration:
        // both halves are read by the same lane (the upper one nearly lane-consecutive, the lower one strided -- all inside the
        // tile's 83 KB stretch of P, which stays in cache), UB pairs of loads are requested before the first is used (three waves
        // fit a CU beside their 51 KiB of LDS: a load per loop iteration would cost its whole latency 648 times).
        constexpr int UB = 9, NN = N * N, NP = L::NP;
        const int total = nvalid * NP;
        const T* src = P + f0 * NN;
        for (int o0 = lane; o0 < total; o0 += BLOCK * UB) {
            T u[UB], l[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int o = o0 + k * BLOCK < total ? o0 + k * BLOCK : 0;
                const int f = o / NP, q = o - f * NP, r = tab[q], i = r / N, jj = r - i * N;
                u[k] = src[f * NN + r];
                l[k] = src[f * NN + jj * N + i];
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int o = o0 + k * BLOCK;
                const int f = o / NP, q = o - f * NP;
                if (o < total) lds[f * IO::PITCH + L::OFF_COV + q] = (u[k] + l[k]) / 2;
            }
        }
    }
    __syncthreads();
    if (lane >= nvalid) return;
    if (do_nom) {
#pragma unroll
        for (int e = 0; e < CN * EPC; ++e) rec[e] = lds[lane * IO::PITCH + e];
        store_chunks<T, N, 0, CN>(rs, lane, rec);
    }
    if (do_cov) {
#pragma unroll
        for (int e = CN * EPC; e < IO::NR; ++e) rec[e] = lds[lane * IO::PITCH + e];
        store_chunks<T, N, CN, RC::NCH>(rs, lane, rec + CN * EPC);
    } else if (do_prev) {
#pragma unroll
        for (int e = CL * EPC; e < IO::NR; ++e) rec[e] = lds[lane * IO::PITCH + e];
        store_chunks<T, N, CL, RC::NCH>(rs, lane, rec + CL * EPC);
    }
}

template <typename T, int N>
__global__ void __launch_bounds__(BLOCK)
unpack_kernel(const T* __restrict__ recs, int B, T* __restrict__ nominal, T* __restrict__ rot, T* __restrict__ P,
              int* __restrict__ prev)
{
    using IO = TileIO<T, N>;
    using L = Lay<N>;
    using RC = Rec<T, N>;
    __shared__ T lds[BLOCK * IO::PITCH];
    __shared__ short tab[N * N];
    const int lane = threadIdx.x, tile = blockIdx.x;
    const int nvalid = min(BLOCK, B - tile * BLOCK);
    const size_t f0 = (size_t)tile * BLOCK;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, tile);
    T rec[IO::NR];
    // only the parts that are asked for are read
    constexpr int CN = RC::CH_NOM, EPC = RC::EPC, CL = RC::NCH - 1;
    if (nominal || rot) {
        load_chunks<T, N, 0, CN>(rs, lane, rec);
#pragma unroll
        for (int e = 0; e < CN * EPC; ++e) lds[lane * IO::PITCH + e] = rec[e];
    }
    if (P) {
        load_chunks<T, N, CN, RC::NCH>(rs, lane, rec + CN * EPC);
#pragma unroll
        for (int e = CN * EPC; e < IO::NR; ++e) lds[lane * IO::PITCH + e] = rec[e];
        for (int r = lane; r < N * N; r += BLOCK) tab[r] = (short)IO::cov_elem(r);
    } else if (prev) {
        load_chunks<T, N, CL, RC::NCH>(rs, lane, rec + CL * EPC);
    }
    __syncthreads();
    if (nominal)
        for (int o = lane; o < nvalid * 19; o += BLOCK) {
            const int f = o / 19, i = o - f * 19;
            nominal[f0 * 19 + o] = lds[f * IO::PITCH + IO::nom_elem(i)];
        }
    if (rot)
        for (int o = lane; o < nvalid * 9; o += BLOCK) {
            const int f = o / 9, i = o - f * 9;
            rot[f0 * 9 + o] = lds[f * IO::PITCH + L::OFF_R + i];
        }
    if (prev && lane < nvalid) prev[f0 + lane] = (int)rec[L::OFF_PREV];
    if (P)
        for (int o = lane; o < nvalid * N * N; o += BLOCK) {
            const int f = o / (N * N), r = o - f * (N * N);
            P[f0 * N * N + o] = lds[f * IO::PITCH + tab[r]];
        }
}

template <typename T, int N>
__global__ void __launch_bounds__(BLOCK)
reset_cov_kernel(T* __restrict__ recs, int B, T d0, T d1, T d2, T d3, T d4, T d5)
{
    // P = diag(P0) for every filter: the covariance chunks are built in registers and stored whole (1 KiB per store
    // instruction); the last chunk also holds the previous-marker id, so it is read, patched and written back
    using L = Lay<N>;
    using RC = Rec<T, N>;
    constexpr int CN = RC::CH_NOM, EPC = RC::EPC, CL = RC::NCH - 1;
    const int b = blockIdx.x * BLOCK + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<T, N>(recs, my_tile());
    T cov[RC::NCOVP];
    load_chunks<T, N, CL, RC::NCH>(rs, my_lane(), cov + (CL - CN) * EPC);
    const T d[6] = { d0, d1, d2, d3, d4, d5 };
#pragma unroll
    for (int e = 0; e < L::NP; ++e) cov[e] = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) cov[pidx<N>(i, i)] = d[i / 3];
    if (b >= B) return;
    store_chunks<T, N, CN, RC::NCH>(rs, my_lane(), cov);
}

}  // namespace
