// Experiment: what is the memory floor of the predict kernel's access pattern on gfx950?
//
// Same geometry as predict_kernel<float,18>: tiles of 64 filters, 50 pieces of 1 KiB per tile, one wave per tile,
// one buffer descriptor per tile; every wave loads all 50 chunks (nt) and stores 38 of them (chunks 0..4 and
// 7..39, what predict writes).  Variants:
//   copy          no arithmetic (value + 1)
//   fma L         a dependent chain of L FMA per lane between the loads and the stores (emulates the lock-step
//                 compute phase of predict; all waves of a launch run load -> compute -> store together)
//   split S       the batch split over S streams (S launches of B/S filters issued round-robin), so that the
//                 compute phase of one part can overlap the memory phases of another
//   nt stores     the same with non-temporal stores (what the library kernel uses)
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp_stream_floor.hip -o tools/_build/exp_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NCH = 50;

template <int L, int SAUX>
__global__ __launch_bounds__(64) void stream_kernel(float* rec, int tile0)
{
    const int tile = tile0 + blockIdx.x;
    float* base = rec + size_t(tile) * NCH * 256;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, NCH * 1024, 0x00020000);
    const int vo = threadIdx.x * 16;
    f4 v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        v[c] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, c * 1024, 2));
    if (L > 0) {
        // L dependent-ish FMA spread over the registers (4 independent chains per chunk element group)
#pragma unroll 1
        for (int it = 0; it < L / 200; ++it) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int d = (c + 1) % NCH;
                v[c].x = __builtin_fmaf(v[d].x, 0.999f, v[c].y);
                v[c].y = __builtin_fmaf(v[d].y, 0.999f, v[c].z);
                v[c].z = __builtin_fmaf(v[d].z, 0.999f, v[c].w);
                v[c].w = __builtin_fmaf(v[d].w, 0.999f, v[c].x);
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[c].x += 1.0f;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (c < 5 || (c >= 7 && c < 40))
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v[c]), rs, vo, c * 1024, SAUX);
}

template <int L, int SAUX = 0>
static double run(float* d, int B, int S, int reps, hipStream_t* st)
{
    const int tiles = B / 64, per = tiles / S;
    hipEvent_t e0, e1, fork, join[8];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int s = 0; s < S; ++s) CK(hipEventCreateWithFlags(&join[s], hipEventDisableTiming));
    auto body = [&](int n) {
        for (int r = 0; r < n; ++r)
            for (int s = 0; s < S; ++s) stream_kernel<L, SAUX><<<per, 64, 0, st[s]>>>(d, s * per);
    };
    body(20);
    for (int s = 0; s < S; ++s) CK(hipStreamSynchronize(st[s]));
    CK(hipEventRecord(e0, st[0]));
    if (S > 1) { CK(hipEventRecord(fork, st[0])); for (int s = 1; s < S; ++s) CK(hipStreamWaitEvent(st[s], fork, 0)); }
    body(reps);
    for (int s = 1; s < S; ++s) { CK(hipEventRecord(join[s], st[s])); CK(hipStreamWaitEvent(st[0], join[s], 0)); }
    CK(hipEventRecord(e1, st[0]));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main()
{
    hipStream_t st[8];
    for (int s = 0; s < 8; ++s) CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
    for (int B : {65536, 262144, 1048576}) {
        float* d; CK(hipMalloc(&d, size_t(B) * NCH * 16));
        CK(hipMemset(d, 0, size_t(B) * NCH * 16));
        const double mb = B * (NCH + 38) * 16 / 1e6;
        const int reps = B > 65536 ? 50 : 200;
        printf("B=%d  (%.1f MB moved per pass)\n", B, mb);
        {
            double t;
            t = run<0, 2>(d, B, 1, reps, st);    printf("  nt stores  copy      %7.2f us  %6.0f GB/s\n", t, mb / t * 1e3);
            t = run<800, 2>(d, B, 1, reps, st);  printf("  nt stores  fma 800   %7.2f us  %6.0f GB/s\n", t, mb / t * 1e3);
            t = run<1600, 2>(d, B, 1, reps, st); printf("  nt stores  fma 1600  %7.2f us  %6.0f GB/s\n", t, mb / t * 1e3);
        }
        for (int S : {1, 2}) {
            double t;
            t = run<0>(d, B, S, reps, st);    printf("  split %d  copy      %7.2f us  %6.0f GB/s\n", S, t, mb / t * 1e3);
            t = run<800>(d, B, S, reps, st);  printf("  split %d  fma 800   %7.2f us  %6.0f GB/s\n", S, t, mb / t * 1e3);
            t = run<1600>(d, B, S, reps, st); printf("  split %d  fma 1600  %7.2f us  %6.0f GB/s\n", S, t, mb / t * 1e3);
        }
        CK(hipFree(d));
    }
    return 0;
}
