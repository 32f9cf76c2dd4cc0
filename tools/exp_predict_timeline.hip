// Experiment: where does a wave of the per-call predict kernel spend its time?
//
// The same record layout, loads, arithmetic (ekf_device.hpp) and stores as predict_kernel<float,18,MATLAB,false>,
// with s_memrealtime (100 MHz, chip-wide) read by lane 0 of every wave at:
//   t0 kernel entry   t1 all loads landed (s_waitcnt vmcnt(0))   t2 arithmetic done   t3 stores issued
//   t4 stores acknowledged
// MODE 0 = load all / compute / store all (the round-1 kernel); MODE 1 = the staged stream body (no waits inserted
// except at the end, t1 is then "nominal stage done", t2 "rows p stored").
// Launches run back to back; the per-wave stamps of the LAST launch are reduced to percentiles relative to the
// earliest t0 of that launch.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I fbus-ekf_amd/csrc -I include \
//        tools/exp_predict_timeline.hip -o tools/_build/exp_timeline
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ekf_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int N = 18;
using RC = Rec<float, N>;
static std::vector<float>* g_h = nullptr;
static void reset_records(float* recs) { CK(hipMemcpy(recs, g_h->data(), g_h->size() * 4, hipMemcpyHostToDevice)); }

template <int C0, int C1, int AUX>
__device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t rs, unsigned lane, float* dst)
{
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u + (c & 3) * 1024u, (c >> 2) * 4096, AUX);
        const float* e = reinterpret_cast<const float*>(&v);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[(c - C0) * 4 + k] = e[k];
    }
}
template <int C0, int C1, int SAUX>
__device__ __forceinline__ void st_(__amdgpu_buffer_rsrc_t rs, unsigned lane, const float* src)
{
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        u32x4 v;
        float* e = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = src[(c - C0) * 4 + k];
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16u + (c & 3) * 1024u, (c >> 2) * 4096, SAUX);
    }
}
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

template <int MODE, int SAUX>
__global__ void __launch_bounds__(64)
timeline_kernel(float* recs, const float* accel, const float* gyro, const float* dt, float q0, float q1, float q2, float q3,
                unsigned long long* stamps, int groups, int naps, int parts = 1)
{
    // optional stagger: wave group g = (tile / 8) % groups naps g * naps * ~0.2 us before it issues its loads, so
    // that the read phase of one group overlaps the write phase of another (reads and writes travel separately)
    for (int i = ((blockIdx.x >> 3) % groups) * naps; i > 0; --i) __builtin_amdgcn_s_sleep(8);
    const unsigned long long t0 = now();
    // parts > 1: a tile is shared by `parts` workgroups of 64 / parts threads (partial waves, parts waves per SIMD)
    const unsigned tile = blockIdx.x / parts, lane = threadIdx.x + (blockIdx.x % parts) * (64 / parts);
    constexpr int CN = RC::CH_NOM;
    char* tb = reinterpret_cast<char*>(recs) + (size_t)tile * RC::NCH * 1024u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tb, 0, RC::NCH * 1024, 0x00020000);
    const float qd[4] = { q0, q1, q2, q3 };
    float nom[Lay<N>::NNOM], P[RC::NCOVP];
    const size_t o = (size_t)(tile * 64 + lane) * 3;
#ifdef IMU_NT
    const float a[3] = { __builtin_nontemporal_load(accel + o), __builtin_nontemporal_load(accel + o + 1), __builtin_nontemporal_load(accel + o + 2) };
    const float w[3] = { __builtin_nontemporal_load(gyro + o), __builtin_nontemporal_load(gyro + o + 1), __builtin_nontemporal_load(gyro + o + 2) };
#else
    const float a[3] = { accel[o], accel[o + 1], accel[o + 2] };
    const float w[3] = { gyro[o], gyro[o + 1], gyro[o + 2] };
#endif
    const float h = dt[0];
    unsigned long long t1, t2, t3;
    if (MODE == 0) {
        ld<0, CN, 2>(rs, lane, nom);
        ld<CN, RC::NCH, 2>(rs, lane, P);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = now();
        predict_step<float, N, DIALECT_MATLAB>(nom, P, a, w, h, qd);
        asm volatile("" ::: "memory");
        t2 = now();
        st_<0, RC::CH_KIN, SAUX>(rs, lane, nom);
        st_<CN, RC::CH_VAR_END, SAUX>(rs, lane, P);
        t3 = now();
    } else {
        constexpr int C_P = CN + cov_final_before_row<N>(3) / 4, C_V = CN + cov_final_before_row<N>(6) / 4;
        constexpr int C_PV_IN = CN + (cov_final_before_row<N>(6) + 3) / 4, C_DG0 = CN + 122 / 4, C_DG1 = RC::CH_VAR_END;
        ld<0, CN, 2>(rs, lane, nom);
        ld<C_DG0, C_DG1, 2>(rs, lane, P + (C_DG0 - CN) * 4);
        ld<CN, C_PV_IN, 2>(rs, lane, P);
        ld<C_PV_IN, C_DG0, 2>(rs, lane, P + (C_PV_IN - CN) * 4);
        ld<C_DG1, RC::NCH, 2>(rs, lane, P + (C_DG1 - CN) * 4);
        // IMU sample = the first three vector loads, nominal chunks = the next 7, 43 more behind them
        asm volatile("s_waitcnt vmcnt(50)" ::: "memory");
        t1 = now();
        asm volatile("s_waitcnt vmcnt(43)" ::: "memory");
        t2 = now();
        PredictCoef<float> k;
        predict_nominal<float, N, DIALECT_MATLAB>(nom, a, w, h, k);
        st_<0, RC::CH_KIN, SAUX>(rs, lane, nom);
        cov_stage_p<float, N>(P, k);
        st_<CN, C_P, SAUX>(rs, lane, P);
        cov_stage_v<float, N>(P, k, qd);
        st_<C_P, C_V, SAUX>(rs, lane, P + (C_P - CN) * 4);
        cov_stage_th<float, N>(P, k, qd);
        st_<C_V, RC::CH_VAR_END, SAUX>(rs, lane, P + (C_V - CN) * 4);
        t3 = now();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t4 = now();
    if (threadIdx.x == 0 && blockIdx.x % parts == 0) {
        unsigned long long* s = stamps + (size_t)tile * 5;
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = t4;
    }
}

template <int MODE, int SAUX>
static void run(float* recs, const float* acc, const float* gyr, const float* dt, unsigned long long* d_st, int B, const char* name, int groups = 1, int naps = 0, bool verbose = true, int pool = 1, int parts = 1)
{
    const int tiles = B / 64, reps = 30;
    reset_records(recs);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 5; ++r) timeline_kernel<MODE, SAUX><<<tiles * parts, 64 / parts>>>(recs, acc + (size_t)(r % pool) * B * 3, gyr + (size_t)(r % pool) * B * 3, dt, 1e-4f, 1e-6f, 1e-8f, 1e-10f, d_st, groups, naps, parts);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) timeline_kernel<MODE, SAUX><<<tiles * parts, 64 / parts>>>(recs, acc + (size_t)(r % pool) * B * 3, gyr + (size_t)(r % pool) * B * 3, dt, 1e-4f, 1e-6f, 1e-8f, 1e-10f, d_st, groups, naps, parts);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> s((size_t)tiles * 5);
    CK(hipMemcpy(s.data(), d_st, s.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long base = ~0ull;
    for (int t = 0; t < tiles; ++t) base = std::min(base, s[t * 5]);
    printf("%s [pool %d, %d workgroup(s) per tile]: %.2f us per launch (back to back, stamps included)\n", name, pool, parts, ms * 1e3 / reps);
    if (!verbose) return;
    const char* lbl0[5] = { "t0 entry", "t1 loads landed", "t2 arithmetic done", "t3 stores issued", "t4 stores acked" };
    const char* lbl1[5] = { "t0 entry", "t1 IMU sample landed", "t2 nominal landed", "t3 all stores issued", "t4 stores acked" };
    for (int k = 0; k < 5; ++k) {
        std::vector<double> v(tiles);
        for (int t = 0; t < tiles; ++t) v[t] = double(s[t * 5 + k] - base) * 0.01;   // 100 MHz -> us
        std::sort(v.begin(), v.end());
        printf("   %-22s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us\n", MODE ? lbl1[k] : lbl0[k], v[0], v[tiles / 10],
               v[tiles / 2], v[tiles * 9 / 10], v[tiles - 1]);
    }
    if (tiles > 1024) {
        // the workgroups past the first 1024 (one per SIMD): when do they start, when do they end?
        std::vector<double> a, b_, c;
        for (int t = 1024; t < tiles; ++t) { a.push_back(double(s[t * 5] - base) * 0.01); b_.push_back(double(s[t * 5 + 4] - base) * 0.01); }
        for (int t = 0; t < 1024; ++t) c.push_back(double(s[t * 5 + 4] - base) * 0.01);
        std::sort(a.begin(), a.end()); std::sort(b_.begin(), b_.end()); std::sort(c.begin(), c.end());
        printf("   workgroups >= 1024 (%zu): entry min %6.2f p50 %6.2f max %6.2f us; done p50 %6.2f max %6.2f us | first 1024: done p50 %6.2f max %6.2f us\n",
               a.size(), a[0], a[a.size() / 2], a.back(), b_[b_.size() / 2], b_.back(), c[512], c.back());
    }
    // per-wave phase lengths
    for (int k = 1; k < 5; ++k) {
        std::vector<double> v(tiles);
        for (int t = 0; t < tiles; ++t) v[t] = double(s[t * 5 + k] - s[t * 5 + k - 1]) * 0.01;
        std::sort(v.begin(), v.end());
        printf("   phase %d->%d             min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us\n", k - 1, k, v[0], v[tiles / 10],
               v[tiles / 2], v[tiles * 9 / 10], v[tiles - 1]);
    }
}

template <int LD, int ST>
static void run_lib_pol(float* recs, const float* acc, const float* gyr, const float* dt, int B, int pool, const char* name)
{
    DevConst<float> dc = {};
    dc.qd[0] = 1e-4f; dc.qd[1] = 1e-6f; dc.qd[2] = 1e-8f; dc.qd[3] = 1e-10f;
    const int tiles = B / 64, reps = 200;
    reset_records(recs);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&](int r) {
        const size_t o = (size_t)(r % pool) * B * 3;
        predict_kernel<float, 18, 0, false, LD, ST><<<tiles, 64>>>(recs, B, 1, acc + o, gyr + o, dt, 0, dc);
    };
    for (int r = 0; r < 5; ++r) go(r);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) go(r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("   B %6d (%4d waves) library predict_kernel %s: %.2f us per launch\n", B, tiles, name, ms * 1e3 / reps);
}

static void run_lib(float* recs, const float* acc, const float* gyr, const float* dt, int B, int pool)
{
    DevConst<float> dc = {};
    dc.qd[0] = 1e-4f; dc.qd[1] = 1e-6f; dc.qd[2] = 1e-8f; dc.qd[3] = 1e-10f;
    const int tiles = B / 64, reps = 200;
    reset_records(recs);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&](int r) {
        const size_t o = (size_t)(r % pool) * B * 3;
        predict_kernel<float, 18, 0, false><<<tiles, 64>>>(recs, B, 1, acc + o, gyr + o, dt, 0, dc);
    };
    for (int r = 0; r < 5; ++r) go(r);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) go(r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("library predict_kernel<float,18,matlab,false>, IMU pool of %d slices: %.2f us per launch\n", pool, ms * 1e3 / reps);
}

// the bench pattern: K predicts, then one pass that reads and writes every record with the DEFAULT cache policy (a
// stand-in for the correct kernel; SAUX = its store policy); events around each run of K predicts
template <int SAUX>
static void run_mixed(float* recs, const float* acc, const float* gyr, const float* dt, unsigned long long* d_st, int B, int K, const char* name)
{
    DevConst<float> dc = {};
    dc.qd[0] = 1e-4f; dc.qd[1] = 1e-6f; dc.qd[2] = 1e-8f; dc.qd[3] = 1e-10f;
    const int tiles = B / 64, frames = 40;
    reset_records(recs);
    std::vector<hipEvent_t> ev(2 * frames);
    for (auto& e : ev) CK(hipEventCreate(&e));
    int r = 0;
    for (int f = -5; f < frames; ++f) {
        if (f >= 0) CK(hipEventRecord(ev[2 * f]));
        for (int k = 0; k < K; ++k, ++r) {
            const size_t o = (size_t)(r % 96) * B * 3;
            predict_kernel<float, 18, 0, false><<<tiles, 64>>>(recs, B, 1, acc + o, gyr + o, dt, 0, dc);
        }
        if (f >= 0) CK(hipEventRecord(ev[2 * f + 1]));
        timeline_kernel<0, SAUX><<<tiles, 64>>>(recs, acc, gyr, dt, 1e-4f, 1e-6f, 1e-8f, 1e-10f, d_st, 1, 0);
    }
    CK(hipDeviceSynchronize());
    double tot = 0;
    for (int f = 0; f < frames; ++f) { float ms; CK(hipEventElapsedTime(&ms, ev[2 * f], ev[2 * f + 1])); tot += ms; }
    printf("%s: %d predicts then one default-policy pass: %.2f us per predict launch\n", name, K, tot * 1e3 / (frames * K));
}

int main()
{
    const int B = getenv("TAIL_B") ? atoi(getenv("TAIL_B")) : 65536;
    float *recs, *acc, *gyr, *dt; unsigned long long* d_st;
    CK(hipMalloc(&recs, (size_t)B / 64 * RC::NCH * 1024));
    CK(hipMalloc(&acc, (size_t)B * 12 * 96)); CK(hipMalloc(&gyr, (size_t)B * 12 * 96)); CK(hipMalloc(&dt, 4));
    CK(hipMalloc(&d_st, (size_t)B / 64 * 5 * 8));
    // a plausible state: every element non-zero (all-zero records run measurably faster: less toggling, higher clocks),
    // q = (1,0,0,0) + noise, R = I + noise, diagonally dominant covariance
    std::vector<float> h((size_t)B / 64 * RC::NCH * 256, 0.f);
    unsigned rng = 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return float(rng >> 8) * (1.0f / 16777216.0f) - 0.5f; };
    for (int b = 0; b < B; ++b) {
        auto at = [&](int e) -> float& { return h[((size_t)(b >> 6) * RC::NCH + e / 4) * 256 + (b & 63) * 4 + e % 4]; };
        for (int e = 0; e < 28; ++e) at(e) = 0.1f * rnd();
        at(3) += 1.f; at(7) += 1.f; at(11) += 1.f; at(15) += 1.f; at(27) = -9.8f;
        for (int i = 0; i < 18; ++i)
            for (int j = i; j < 18; ++j) at(28 + pidx<N>(i, j)) = (i == j) ? 1e-2f * (1.f + 0.2f * rnd()) : 1e-4f * rnd();
    }
    g_h = &h;
    std::vector<float> hv((size_t)B * 3 * 96);
    for (size_t i = 0; i < hv.size(); ++i) hv[i] = 0.01f * float(i % 7);
    CK(hipMemcpy(acc, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gyr, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    const float hdt = 0.005f; CK(hipMemcpy(dt, &hdt, 4, hipMemcpyHostToDevice));
    if (getenv("MIXED")) {
        for (int K : {7, 20, 3, 1}) {
            run_mixed<0>(recs, acc, gyr, dt, d_st, B, K, "plain-store pass");
            run_mixed<2>(recs, acc, gyr, dt, d_st, B, K, "nt-store pass   ");
        }
        return 0;
    }
    if (getenv("PARTS")) {
        // partial waves: 2 or 4 workgroups of 32 / 16 lanes per tile (2 or 4 waves per SIMD at B = 65 536)
        for (int parts : {1, 2, 4}) {
            run<1, 2>(recs, acc, gyr, dt, d_st, B, "staged stream, nt stores", 1, 0, false, 8, parts);
            run<0, 2>(recs, acc, gyr, dt, d_st, B, "load / compute / store, nt stores", 1, 0, false, 8, parts);
        }
        return 0;
    }
    if (getenv("SEQ")) {
        // Infinity Cache residency: the same launches with the IMU sample taken from a pool of 1 .. 96 slices
        // (build with -DIMU_NT for non-temporal IMU loads: the times then stay at the pool-1 level)
        for (int pool : {1, 1, 8, 32, 96, 1, 1}) run_lib(recs, acc, gyr, dt, B, pool);
        return 0;
    }
    if (getenv("TAIL_POL")) {
        run_lib_pol<AUX_NT, AUX_NT>(recs, acc, gyr, dt, B, 8, "nt loads, nt stores          ");
        run_lib_pol<AUX_DEFAULT, AUX_NT>(recs, acc, gyr, dt, B, 8, "default loads, nt stores     ");
        run_lib_pol<AUX_DEFAULT, AUX_DEFAULT>(recs, acc, gyr, dt, B, 8, "default loads, default stores");
        run_lib_pol<AUX_NT, AUX_DEFAULT>(recs, acc, gyr, dt, B, 8, "nt loads, default stores     ");
        return 0;
    }
    if (getenv("TAIL_B")) {
        run_lib(recs, acc, gyr, dt, B, 8);
        run<1, 2>(recs, acc, gyr, dt, d_st, B, "staged stream, nt stores", 1, 0, true, 8);
        return 0;
    }
    run_lib(recs, acc, gyr, dt, B, 8);
    run<0, 0>(recs, acc, gyr, dt, d_st, B, "load / compute / store, plain stores", 1, 0, true, 8);
    run<0, 2>(recs, acc, gyr, dt, d_st, B, "load / compute / store, nt stores", 1, 0, true, 8);
    run<1, 0>(recs, acc, gyr, dt, d_st, B, "staged stream, plain stores", 1, 0, true, 8);
    run<1, 2>(recs, acc, gyr, dt, d_st, B, "staged stream, nt stores", 1, 0, true, 8);
    run_lib(recs, acc, gyr, dt, B, 8);
    return 0;
}
