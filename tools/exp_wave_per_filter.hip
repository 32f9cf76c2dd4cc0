// exp_wave_per_filter.hip -- EXPERIMENT, not product code.
//
// BASELINE.json's north_star suggests "one EKF instance per wavefront, the covariance held in LDS".  The
// product maps one filter per LANE instead.  This standalone program measures the
// suggested mapping for the predict step so that the choice rests on a measurement, not only on arithmetic:
//
//   * one wave per filter, 4 waves per workgroup, each wave owns an LDS slab: the full 18 x 18 covariance,
//     the 18 x 18 intermediate X = F P, and F's non-zeros (the block structure of ImuUpdate.m:63-69 is
//     exploited: 2.8 non-zeros per row on average, column lists are static tables);
//   * records are AoS (one filter = 800 contiguous bytes: nominal 28 + packed P 171 + prev), so that the wave's
//     50 active lanes load its record with ONE coalesced dwordx4 instruction -- the layout this mapping wants;
//   * lane l owns covariance elements l, l+64, ... (6 of 324); every lane computes the nominal-state update
//     redundantly (what a wave-per-filter kernel does with scalar work);
//   * P' = F P F' + Q in two LDS-staged passes, result stored packed.
//
// Build + run (on the GPU box):  hipcc --offload-arch=gfx950 -O3 tools/exp_wave_per_filter.hip -o /tmp/wpf && /tmp/wpf
// Prints the per-launch time at B = 65 536 and the max error of P' against a host fp64 dense F P F' + Q.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int N = 18, NP = 171, REC = 200, LD = 19;     // LD: padded leading dimension of the LDS matrices
constexpr int WAVES = 4;

__host__ __device__ inline int pidx(int i, int j) { if (i > j) { int t = i; i = j; j = t; } return i * N - (i * (i - 1)) / 2 + (j - i); }

// static sparsity of F = I + N (ImuUpdate.m:63-69): columns of the non-zeros of each row, -1 = none
__constant__ signed char kFcol[N][8] = {
    { 0, 3, -1, -1, -1, -1, -1, -1 }, { 1, 4, -1, -1, -1, -1, -1, -1 }, { 2, 5, -1, -1, -1, -1, -1, -1 },
    { 3, 6, 7, 8, 9, 10, 11, 15 }, { 4, 6, 7, 8, 9, 10, 11, 16 }, { 5, 6, 7, 8, 9, 10, 11, 17 },
    { 6, 7, 8, 12, -1, -1, -1, -1 }, { 6, 7, 8, 13, -1, -1, -1, -1 }, { 6, 7, 8, 14, -1, -1, -1, -1 },
    { 9, -1, -1, -1, -1, -1, -1, -1 }, { 10, -1, -1, -1, -1, -1, -1, -1 }, { 11, -1, -1, -1, -1, -1, -1, -1 },
    { 12, -1, -1, -1, -1, -1, -1, -1 }, { 13, -1, -1, -1, -1, -1, -1, -1 }, { 14, -1, -1, -1, -1, -1, -1, -1 },
    { 15, -1, -1, -1, -1, -1, -1, -1 }, { 16, -1, -1, -1, -1, -1, -1, -1 }, { 17, -1, -1, -1, -1, -1, -1, -1 } };

struct Slab { float rec[REC]; float P[N * LD]; float X[N * LD]; float Fv[N][8]; };

__device__ inline void quat_mul(const float* p, const float* q, float* o)
{
    o[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
    o[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
    o[2] = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
    o[3] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
}
__device__ inline void q2R(const float* q, float* R)
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}

// record order: p3 v3 q4 ba3 bg3 g3 R9 | P packed row-major upper | prev
__global__ void __launch_bounds__(64 * WAVES)
predict_wpf(float* __restrict__ recs, int B, const float* __restrict__ accel, const float* __restrict__ gyro, float dt,
            float qv, float qth, float qba, float qbg)
{
    __shared__ Slab slabs[WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    Slab& s = slabs[wv];
    const int nw = gridDim.x * WAVES;
    for (int f = blockIdx.x * WAVES + wv; f < B; f += nw) {
        // ---- load the record: 50 lanes x 16 B, one coalesced instruction ----
        if (lane < REC / 4) reinterpret_cast<float4*>(s.rec)[lane] = reinterpret_cast<const float4*>(recs + (size_t)f * REC)[lane];
        __builtin_amdgcn_wave_barrier();
        // ---- unpack P into the full symmetric LDS matrix ----
        for (int e = lane; e < N * N; e += 64) { const int i = e / N, j = e % N; s.P[i * LD + j] = s.rec[28 + pidx(i, j)]; }
        // ---- nominal state, redundantly on every lane (ImuUpdate.m:37-60) ----
        const float* r = s.rec;
        float a[3], w[3];
        for (int i = 0; i < 3; ++i) { a[i] = accel[(size_t)f * 3 + i] - r[10 + i]; w[i] = gyro[(size_t)f * 3 + i] - r[13 + i]; }
        const float wn = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const float inv = wn > 0 ? 1.0f / wn : 0.0f;
        const float n[3] = { w[0] * inv, w[1] * inv, w[2] * inv };
        float s2, c2, s4, c4;
        sincosf(wn * dt * 0.5f, &s2, &c2);
        sincosf(wn * dt * 0.25f, &s4, &c4);
        const float q[4] = { r[6], r[7], r[8], r[9] };
        const float dqT[4] = { c2, n[0] * s2, n[1] * s2, n[2] * s2 }, dqH[4] = { c4, n[0] * s4, n[1] * s4, n[2] * s4 };
        float qT[4], qH[4], RH[9], RT[9], R0[9];
        quat_mul(q, dqT, qT); quat_mul(q, dqH, qH);
        q2R(qH, RH); q2R(qT, RT);
        for (int i = 0; i < 9; ++i) R0[i] = r[19 + i];
        float vnew[3], pnew[3];
        for (int i = 0; i < 3; ++i) {
            const float g = r[16 + i];
            const float kv1 = R0[3 * i] * a[0] + R0[3 * i + 1] * a[1] + R0[3 * i + 2] * a[2] + g;
            const float kv2 = RH[3 * i] * a[0] + RH[3 * i + 1] * a[1] + RH[3 * i + 2] * a[2] + g;
            const float kv4 = RT[3 * i] * a[0] + RT[3 * i + 1] * a[1] + RT[3 * i + 2] * a[2] + g;
            const float v0 = r[3 + i];
            vnew[i] = v0 + dt / 6 * (kv1 + 4 * kv2 + kv4);
            pnew[i] = r[i] + dt / 6 * (v0 + 2 * (v0 + kv1 * dt / 2) + 2 * (v0 + kv2 * dt / 2) + (v0 + kv2 * dt / 2));
        }
        const float qn = 1.0f / sqrtf(qT[0] * qT[0] + qT[1] * qT[1] + qT[2] * qT[2] + qT[3] * qT[3]);
        // ---- F's values into LDS (lanes 0..17, one row each) ----
        const float sa = 2 * s2 * c2, sb = 2 * s2 * s2;
        if (lane < N) {
            float* Fv = s.Fv[lane];
            const int i = lane;
            if (i < 3) { Fv[0] = 1; Fv[1] = dt; }
            else if (i < 6) {
                const int k = i - 3;
                const float r0 = R0[3 * k], r1 = R0[3 * k + 1], r2 = R0[3 * k + 2];
                Fv[0] = 1;
                Fv[1] = -dt * (r1 * a[2] - r2 * a[1]); Fv[2] = -dt * (r2 * a[0] - r0 * a[2]); Fv[3] = -dt * (r0 * a[1] - r1 * a[0]);
                Fv[4] = -dt * r0; Fv[5] = -dt * r1; Fv[6] = -dt * r2; Fv[7] = dt;
            } else if (i < 9) {
                const int k = i - 6;                      // Theta = I - sin(phi)[n]x + (1-cos(phi))[n]x^2
                const float nx[9] = { 0, -n[2], n[1], n[2], 0, -n[0], -n[1], n[0], 0 };
                for (int j = 0; j < 3; ++j) Fv[j] = (k == j ? 1.0f - sb : 0.0f) + sb * n[k] * n[j] - sa * nx[3 * k + j];
                Fv[3] = -dt;
            } else Fv[0] = 1;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- X = F P ----
        for (int e = lane; e < N * N; e += 64) {
            const int i = e / N, j = e % N;
            float acc = 0;
            for (int t = 0; t < 8; ++t) { const int c = kFcol[i][t]; if (c < 0) break; acc += s.Fv[i][t] * s.P[c * LD + j]; }
            s.X[i * LD + j] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- P' = X F' + Q, packed upper triangle back into the record image ----
        for (int e = lane; e < N * N; e += 64) {
            const int i = e / N, j = e % N;
            if (j < i) continue;
            float acc = 0;
            for (int t = 0; t < 8; ++t) { const int c = kFcol[j][t]; if (c < 0) break; acc += s.Fv[j][t] * s.X[i * LD + c]; }
            if (i == j && i >= 3 && i < 15) acc += (i < 6 ? qv : i < 9 ? qth : i < 12 ? qba : qbg);
            s.rec[28 + pidx(i, j)] = acc;
        }
        if (lane == 0) {
            for (int i = 0; i < 3; ++i) { s.rec[i] = pnew[i]; s.rec[3 + i] = vnew[i]; }
            for (int i = 0; i < 4; ++i) s.rec[6 + i] = qT[i] * qn;
            for (int i = 0; i < 9; ++i) s.rec[19 + i] = RT[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < REC / 4) reinterpret_cast<float4*>(recs + (size_t)f * REC)[lane] = reinterpret_cast<const float4*>(s.rec)[lane];
        __builtin_amdgcn_wave_barrier();
    }
}

int main()
{
    const int B = 65536;
    std::vector<float> h((size_t)B * REC), acc((size_t)B * 3), gyr((size_t)B * 3);
    srand(7);
    auto rnd = []() { return (float)rand() / (float)RAND_MAX * 2.0f - 1.0f; };
    for (int f = 0; f < B; ++f) {
        float* r = &h[(size_t)f * REC];
        for (int i = 0; i < 3; ++i) { r[i] = rnd(); r[3 + i] = 0.1f * rnd(); r[10 + i] = 0.05f * rnd(); r[13 + i] = 0.002f * rnd(); }
        float q[4] = { rnd(), rnd(), rnd(), rnd() };
        const float qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int i = 0; i < 4; ++i) r[6 + i] = q[i] / qn;
        r[16] = 9.8f; r[17] = 0; r[18] = 0;
        const float w = r[6], x = r[7], y = r[8], z = r[9];
        const float R[9] = { w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), w * w - x * x + y * y - z * z,
                             2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z };
        for (int i = 0; i < 9; ++i) r[19 + i] = R[i];
        const float d0[6] = { 1e-4f, 0.1f, 1e-4f, 1e-3f, 1e-3f, 100.f };
        for (int i = 0; i < N; ++i) for (int j = i; j < N; ++j)
            r[28 + pidx(i, j)] = (i == j ? d0[i / 3] : 0.02f * rnd() * std::sqrt(d0[i / 3] * d0[j / 3]));
        for (int i = 0; i < 3; ++i) { acc[(size_t)f * 3 + i] = -(R[0 + i] * 9.8f) + 0.5f * rnd(); gyr[(size_t)f * 3 + i] = 0.02f * rnd(); }
    }
    float *d, *da, *dg;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&da, acc.size() * 4)); CHECK(hipMalloc(&dg, gyr.size() * 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(da, acc.data(), acc.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dg, gyr.data(), gyr.size() * 4, hipMemcpyHostToDevice));
    const float dt = 0.005f;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // correctness of one step against a host fp64 dense F P F' + Q
    hipLaunchKernelGGL(predict_wpf, dim3(B / WAVES), dim3(64 * WAVES), 0, 0, d, B, da, dg, dt, 1e-3f, 1e-4f, 1e-3f, 1e-4f);
    std::vector<float> out(h.size());
    CHECK(hipMemcpy(out.data(), d, out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int f = 0; f < B; f += 4099) {
        const float* r = &h[(size_t)f * REC];
        double a[3], w[3], F[N][N] = {}, P[N][N], FP[N][N], Pn[N][N];
        for (int i = 0; i < 3; ++i) { a[i] = acc[(size_t)f * 3 + i] - r[10 + i]; w[i] = gyr[(size_t)f * 3 + i] - r[13 + i]; }
        for (int i = 0; i < N; ++i) { F[i][i] = 1; for (int j = 0; j < N; ++j) P[i][j] = r[28 + pidx(i, j)]; }
        const double ax[9] = { 0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0 };
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            if (i == j) { F[i][3 + j] = dt; F[3 + i][15 + j] = dt; F[6 + i][12 + j] = -dt; }
            double ra = 0; for (int k = 0; k < 3; ++k) ra += r[19 + 3 * i + k] * ax[3 * k + j];
            F[3 + i][6 + j] = -ra * dt; F[3 + i][9 + j] = -r[19 + 3 * i + j] * dt;
        }
        const double wn = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), phi = wn * dt;
        const double n[3] = { w[0] / wn, w[1] / wn, w[2] / wn }, nx[9] = { 0, -n[2], n[1], n[2], 0, -n[0], -n[1], n[0], 0 };
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            F[6 + i][6 + j] = (i == j ? std::cos(phi) : 0.0) + (1 - std::cos(phi)) * n[i] * n[j] - std::sin(phi) * nx[3 * i + j];
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { double s = 0; for (int k = 0; k < N; ++k) s += F[i][k] * P[k][j]; FP[i][j] = s; }
        const double qd[4] = { 1e-3, 1e-4, 1e-3, 1e-4 };
        double pmax = 0;
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
            double s = 0; for (int k = 0; k < N; ++k) s += FP[i][k] * F[j][k];
            if (i == j && i >= 3 && i < 15) s += qd[(i - 3) / 3];
            Pn[i][j] = s; pmax = std::fmax(pmax, std::fabs(s));
        }
        for (int i = 0; i < N; ++i) for (int j = i; j < N; ++j)
            worst = std::fmax(worst, std::fabs(out[(size_t)f * REC + 28 + pidx(i, j)] - Pn[i][j]) / pmax);
    }
    std::printf("wave-per-filter predict: covariance max rel err vs host fp64 dense = %.2e\n", worst);
    for (int grid : { B / WAVES, 2048, 1024 }) {
        for (int rep = 0; rep < 3; ++rep)
            hipLaunchKernelGGL(predict_wpf, dim3(grid), dim3(64 * WAVES), 0, 0, d, B, da, dg, dt, 1e-3f, 1e-4f, 1e-3f, 1e-4f);
        CHECK(hipEventRecord(e0, 0));
        const int reps = 20;
        for (int rep = 0; rep < reps; ++rep)
            hipLaunchKernelGGL(predict_wpf, dim3(grid), dim3(64 * WAVES), 0, 0, d, B, da, dg, dt, 1e-3f, 1e-4f, 1e-3f, 1e-4f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("wave-per-filter predict, B = %d, grid %5d x %d waves: %8.1f us per launch  (%.3g EKF steps/s, %.0f GB/s algorithmic)\n",
                    B, grid, WAVES, ms / reps * 1e3, B / (ms / reps * 1e-3), 1620.0 * B / (ms / reps * 1e-3) / 1e9);
    }
    return 0;
}
