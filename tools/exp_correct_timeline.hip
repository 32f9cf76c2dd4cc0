// Experiment: where does a wave of the stacked (joint-update) correct kernel spend its time?
//
// The body of correct_kernel<float,18,MATLAB,SIMPLE,JOINT=true> (ekf_kernels.hpp) with s_memrealtime stamps
// (100 MHz, chip-wide) read by lane 0 of every wave:
//   t0 entry   t1 rows of all markers folded into the information matrix   t2 covariance landed (vmcnt(0))
//   t3 six scalar updates + injection done   t4 stores issued   t5 stores acknowledged
// and the un-instrumented library kernel timed beside it.  Synthetic but valid inputs: M markers per filter, all in the
// map, plausible poses; every record element non-zero.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I fbus-ekf_amd/csrc -I include \
//        tools/exp_correct_timeline.hip -o tools/_build/exp_correct
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ekf_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int N = 18;
using RC = Rec<float, N>;
using L = Lay<N>;
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

template <int SAUX>
__global__ void __launch_bounds__(64)
correct_timeline(float* recs, int M, const int* ids, const float* pos, const float* quat, DevConst<float> dc,
                 unsigned long long* stamps)
{
    const unsigned long long t0 = now();
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int* my_ids = ids + (size_t)b * M;
    const float* my_pos = pos + (size_t)b * M * 3;
    const float* my_quat = quat + (size_t)b * M * 4;
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc<float, N>(recs, my_tile());
    float P[RC::NCOVP], nom[L::NNOM], dx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) dx[i] = 0.f;
    __shared__ MarkerLDS<float> tbl;
    MarkerGroup<float, 4> mg;
    {
        MarkerTableRegs<float> treg;
        treg.load(dc);
        order_fence();
        mg.fetch(my_ids, my_pos, my_quat, 0, M);
        order_fence();
        load_chunks<float, N, 0, RC::CH_NOM>(rs, my_lane(), nom);
        order_fence();
        load_chunks<float, N, RC::CH_NOM, RC::NCH, AUX_NT>(rs, my_lane(), P);
        order_fence();
        treg.to_lds(tbl);
        order_fence();
    }
    InfoAcc<float> acc;
    acc.clear();
    const float w_pos = 1.f / dc.r_pos, w_quat = 1.f / dc.r_quat;
    int used = 0;
    auto fold_group = [&]() {
        mg.resolve(tbl);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (mg.slot[g] < 0) continue;
            marker_info<float, N, DIALECT_MATLAB>(acc, nom, dc, mg.mk[g], mg.yp[g], mg.yq[g], w_pos, w_quat);
            ++used;
        }
    };
    fold_group();
    for (int i0 = 4; i0 < M; i0 += 4) {
        mg.fetch(my_ids, my_pos, my_quat, i0, M);
        fold_group();
    }
    // keep the fold in front of the stamp
    asm volatile("" :: "v"(acc.Lam[0]), "v"(acc.Lam[20]), "v"(acc.b[5]));
    const unsigned long long t1 = now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = now();
    if (used > 0) joint_update<float, N, COV_SIMPLE>(P, dx, acc);
    inject<float, N>(nom, dx);
    asm volatile("" :: "v"(P[0]), "v"(P[170]), "v"(nom[0]));
    const unsigned long long t3 = now();
    store_chunks<float, N, 0, RC::CH_PQ, SAUX>(rs, my_lane(), nom);
    store_chunks<float, N, RC::CH_PQR, RC::CH_NOM, SAUX>(rs, my_lane(), nom + L::NPQR);
    store_chunks<float, N, RC::CH_NOM, RC::NCH, SAUX>(rs, my_lane(), P);
    const unsigned long long t4 = now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t5 = now();
    if (threadIdx.x == 0) {
        unsigned long long* s = stamps + (size_t)blockIdx.x * 6;
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = t4; s[5] = t5;
    }
}

struct Inputs { float* recs; int* ids; float *pos, *quat; unsigned char* applied; unsigned long long* stamps; DevConst<float> dc; std::vector<float> h; };

static void reset(Inputs& in) { CK(hipMemcpy(in.recs, in.h.data(), in.h.size() * 4, hipMemcpyHostToDevice)); }

template <int SAUX>
static void run_stamped(Inputs& in, int B, int M, const char* name)
{
    const int tiles = B / 64, reps = 30;
    reset(in);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 5; ++r) correct_timeline<SAUX><<<tiles, 64>>>(in.recs, M, in.ids, in.pos, in.quat, in.dc, in.stamps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) correct_timeline<SAUX><<<tiles, 64>>>(in.recs, M, in.ids, in.pos, in.quat, in.dc, in.stamps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> s((size_t)tiles * 6);
    CK(hipMemcpy(s.data(), in.stamps, s.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long base = ~0ull;
    for (int t = 0; t < tiles; ++t) base = std::min(base, s[t * 6]);
    printf("%s, M = %d: %.2f us per launch (back to back, stamps included)\n", name, M, ms * 1e3 / reps);
    const char* lbl[6] = { "t0 entry", "t1 rows folded", "t2 covariance landed", "t3 updates + inject done", "t4 stores issued", "t5 stores acked" };
    for (int k = 0; k < 6; ++k) {
        std::vector<double> v(tiles);
        for (int t = 0; t < tiles; ++t) v[t] = double(s[t * 6 + k] - base) * 0.01;
        std::sort(v.begin(), v.end());
        printf("   %-26s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us\n", lbl[k], v[0], v[tiles / 10], v[tiles / 2], v[tiles * 9 / 10], v[tiles - 1]);
    }
}

static void run_lib(Inputs& in, int B, int M)
{
    const int tiles = B / 64, reps = 30;
    reset(in);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&]() { correct_kernel<float, N, 0, 0, true><<<tiles, 64>>>(in.recs, B, M, in.ids, in.pos, in.quat, MODE_STACKED, nullptr, in.applied, in.dc); };
    for (int r = 0; r < 5; ++r) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) go();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("library correct_kernel<float,18,matlab,simple,joint>, M = %d: %.2f us per launch (back to back)\n", M, ms * 1e3 / reps);
}

int main()
{
    const int B = 65536, MMAX = 16;
    Inputs in;
    CK(hipMalloc(&in.recs, (size_t)B / 64 * RC::NCH * 1024));
    CK(hipMalloc(&in.ids, (size_t)B * MMAX * 4)); CK(hipMalloc(&in.pos, (size_t)B * MMAX * 12)); CK(hipMalloc(&in.quat, (size_t)B * MMAX * 16));
    CK(hipMalloc(&in.applied, B)); CK(hipMalloc(&in.stamps, (size_t)B / 64 * 6 * 8));
    unsigned rng = 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return float(rng >> 8) * (1.0f / 16777216.0f) - 0.5f; };
    in.h.assign((size_t)B / 64 * RC::NCH * 256, 0.f);
    for (int b = 0; b < B; ++b) {
        auto at = [&](int e) -> float& { return in.h[((size_t)(b >> 6) * RC::NCH + e / 4) * 256 + (b & 63) * 4 + e % 4]; };
        for (int e = 0; e < 28; ++e) at(e) = 0.05f * rnd();
        at(3) += 1.f; at(7) += 1.f; at(11) += 1.f; at(15) += 1.f; at(27) = -9.8f;
        for (int i = 0; i < 18; ++i)
            for (int j = i; j < 18; ++j) at(28 + pidx<N>(i, j)) = (i == j) ? 1e-2f * (1.f + 0.2f * rnd()) : 1e-4f * rnd();
    }
    // map: 16 markers with ids 0..15 on a plane 1 m in front, identity orientation; camera = body frame
    std::vector<float> mk(FBUS_MAX_MARKERS * MK_STRIDE, 0.f);
    std::vector<short> id2slot(FBUS_MAX_MARKER_ID + 1, (short)-1);
    for (int k = 0; k < 16; ++k) {
        mk[k * MK_STRIDE + 0] = 0.2f * float(k % 4) - 0.3f; mk[k * MK_STRIDE + 1] = 0.2f * float(k / 4) - 0.3f; mk[k * MK_STRIDE + 2] = 1.f;
        mk[k * MK_STRIDE + 3] = 1.f;
        id2slot[k] = (short)k;
    }
    float* d_mk; short* d_id2slot;
    CK(hipMalloc(&d_mk, mk.size() * 4)); CK(hipMemcpy(d_mk, mk.data(), mk.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_id2slot, id2slot.size() * 2)); CK(hipMemcpy(d_id2slot, id2slot.data(), id2slot.size() * 2, hipMemcpyHostToDevice));
    DevConst<float> dc = {};
    dc.r_pos = 0.01f; dc.r_quat = 0.01f;
    dc.R_IL[0] = dc.R_IL[4] = dc.R_IL[8] = 1.f; dc.Q_IL[0] = 1.f;
    const float CL[16] = { 1, 0, 0, 0,  0, -1, 0, 0,  0, 0, -1, 0,  0, 0, 0, -1 };
    for (int i = 0; i < 16; ++i) dc.CL[i] = CL[i];
    dc.switch_thres = 0.5f; dc.mk = d_mk; dc.id2slot = d_id2slot;
    in.dc = dc;
    std::vector<int> hid((size_t)B * MMAX);
    std::vector<float> hp((size_t)B * MMAX * 3), hq((size_t)B * MMAX * 4);
    for (size_t i = 0; i < hid.size(); ++i) {
        hid[i] = int(i % 16);
        hp[3 * i] = 0.3f * rnd(); hp[3 * i + 1] = 0.3f * rnd(); hp[3 * i + 2] = 1.f + 0.1f * rnd();
        hq[4 * i] = 1.f; hq[4 * i + 1] = 0.02f * rnd(); hq[4 * i + 2] = 0.02f * rnd(); hq[4 * i + 3] = 0.02f * rnd();
    }
    CK(hipMemcpy(in.ids, hid.data(), hid.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(in.pos, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(in.quat, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    for (int M : {4, 1, 16}) run_lib(in, B, M);
    run_stamped<0>(in, B, 4, "stamped body, plain stores");
    run_stamped<2>(in, B, 4, "stamped body, nt stores");
    run_lib(in, B, 4);
    return 0;
}
