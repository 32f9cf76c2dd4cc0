// Experiment: up to how many VGPRs do two waves share a SIMD on gfx950?  A pure-ALU kernel (one dependent v_fma_f32 chain) that
// reserves NREG VGPRs, launched as 1024 and as 2048 one-wave workgroups: co-resident pairs finish in the time of one wave,
// pairs that have to take turns in twice that.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp_two_waves.hip -o tools/_build/exp_two_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define STR2(x) #x
#define STR(x) STR2(x)
template <int NREG> __global__ __launch_bounds__(64) void k(float* out, int rounds, float seed);
#define DEF(N)                                                                                     \
    template <> __global__ __launch_bounds__(64) void k<N>(float* out, int rounds, float seed)     \
    {                                                                                              \
        float a = seed + threadIdx.x;                                                              \
        asm volatile("v_mov_b32 v" STR(N) ", 0" ::: "v" STR(N));                                   \
        for (int r = 0; r < rounds; ++r) a = __builtin_fmaf(a, 0.999f, 0.5f);                      \
        out[blockIdx.x * 64 + threadIdx.x] = a;                                                    \
    }
DEF(127) DEF(167) DEF(239) DEF(247) DEF(251) DEF(255)
template <int N> static void run(float* d)
{
    float t[2];
    for (int w = 1; w <= 2; ++w) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        k<N><<<1024 * w, 64>>>(d, 1000, 1.f); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k<N><<<1024 * w, 64>>>(d, 20000, 1.f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[w - 1], e0, e1));
    }
    printf("%3d VGPRs: 1024 workgroups %.1f us, 2048 workgroups %.1f us  -> %s\n", N + 1, t[0] * 1e3, t[1] * 1e3,
           t[1] < 1.5f * t[0] ? "two waves share a SIMD" : "one wave per SIMD at a time");
}
int main()
{
    float* d; CK(hipMalloc(&d, 2048 * 64 * 4));
    for (int i = 0; i < 50; ++i) k<127><<<4096, 64>>>(d, 20000, 1.f);
    CK(hipDeviceSynchronize());
    run<127>(d); run<167>(d); run<239>(d); run<247>(d); run<251>(d); run<255>(d);
    return 0;
}
