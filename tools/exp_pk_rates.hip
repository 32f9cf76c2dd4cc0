// Experiment (round 5): what a packed fp32 instruction costs a wave that has its SIMD to itself, by operand form and by
// dependence -- the question behind packing the ImuUpdate stages (csrc/ekf_device.hpp, FBUS_X_PACK): the static count says
// -10 % VALU slots for the frame window's step, the window measured +1.6 %.
// s_memtime around R rounds of 64 instructions written in inline assembly (the compiler does not get to rearrange them).
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp_pk_rates.hip -o tools/_build/exp_pk_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

enum { C_FMA_IND, C_FMA_DEP, C_PK_IND, C_PK_DEP, C_PK_IND_BCAST, C_PK_DEP_BCAST, C_PK_DEP2, C_PK_DEP4, C_MIX_IND, C_MIX_DEP_PK,
       C_PKMUL_IND, C_PKADD_IND, C_MOV, C_ACC_RD, C_ACC_WR, C_FMA_DEP2, C_PK_AFTER_FMA, C_FMA_AFTER_PK, NCASE };
static const char* names[NCASE] = {
    "v_fma_f32, 16 independent chains (3 VGPR sources)",
    "v_fma_f32, ONE dependent chain",
    "v_pk_fma_f32, 16 independent chains (3 VGPR-pair sources)",
    "v_pk_fma_f32, ONE dependent chain (accumulator)",
    "v_pk_fma_f32 op_sel_hi:[1,0,1] (one half on both), independent",
    "v_pk_fma_f32 op_sel_hi:[1,0,1], ONE dependent chain",
    "v_pk_fma_f32, TWO interleaved dependent chains",
    "v_pk_fma_f32, FOUR interleaved dependent chains",
    "v_fma_f32 / v_pk_fma_f32 alternating, all independent",
    "v_pk_fma_f32 chain with an independent v_fma_f32 between its links",
    "v_pk_mul_f32 independent",
    "v_pk_add_f32 independent",
    "v_mov_b32 independent",
    "v_accvgpr_read_b32",
    "v_accvgpr_write_b32",
    "v_fma_f32, TWO interleaved dependent chains",
    "v_pk_fma_f32 reading the result of the v_fma_f32 before it",
    "v_fma_f32 reading the result of the v_pk_fma_f32 before it",
};

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

template <int CASE>
__global__ __launch_bounds__(64) void pk_kernel(float* out, unsigned long long* ticks, int rounds, float seed)
{
    f2 p[16], q[16], r[16];
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed * 1e-3f + i * 1e-4f + threadIdx.x * 1e-6f;
        p[i] = f2{ a[i], a[i] + 1e-5f }; q[i] = f2{ 0.999f + a[i] * 1e-3f, 0.998f }; r[i] = f2{ 1e-3f, 2e-3f };
        asm volatile("" : "+v"(p[i]), "+v"(q[i]), "+v"(r[i]), "+v"(a[i]));
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < rounds; ++it) {
        if (CASE == C_FMA_IND) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(q[i].x), "v"(r[i].x));
        } else if (CASE == C_FMA_DEP) {
            asm volatile(R64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a[0]) : "v"(q[0].x), "v"(r[0].x));
        } else if (CASE == C_FMA_DEP2) {
            asm volatile(R16(R4("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")) : "+v"(a[0]), "+v"(a[1]) : "v"(q[0].x), "v"(r[0].x));
        } else if (CASE == C_PK_IND) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q[i]), "v"(r[i]));
        } else if (CASE == C_PK_DEP) {
            asm volatile(R64("v_pk_fma_f32 %0, %1, %2, %0\n") : "+v"(p[0]) : "v"(q[0]), "v"(r[0]));
        } else if (CASE == C_PK_IND_BCAST) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(q[i]), "v"(r[i]));
        } else if (CASE == C_PK_DEP_BCAST) {
            asm volatile(R64("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n") : "+v"(p[0]) : "v"(q[0]), "v"(r[0]));
        } else if (CASE == C_PK_DEP2) {
            asm volatile(R16(R4("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n")) : "+v"(p[0]), "+v"(p[1]) : "v"(q[0]), "v"(r[0]));
        } else if (CASE == C_PK_DEP4) {
            asm volatile(R16("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n")
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(q[0]), "v"(r[0]));
        } else if (CASE == C_MIX_IND) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(q[i].x), "v"(r[i].x));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q[i]), "v"(r[i]));
                }
        } else if (CASE == C_MIX_DEP_PK) {
            asm volatile(R16(R4("v_pk_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %1, %4, %5\n")) : "+v"(p[0]), "+v"(a[1]) : "v"(q[0]), "v"(r[0]), "v"(q[1].x), "v"(r[1].x));
        } else if (CASE == C_PKMUL_IND) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
        } else if (CASE == C_PKADD_IND) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(r[i]));
        } else if (CASE == C_MOV) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(q[i].x));
        } else if (CASE == C_ACC_RD) {
            asm volatile(R64("v_accvgpr_read_b32 %0, a0\n") : "=v"(a[0]) : : "a0");
        } else if (CASE == C_ACC_WR) {
            asm volatile(R64("v_accvgpr_write_b32 a0, %0\n") : : "v"(a[0]) : "a0");
        } else if (CASE == C_PK_AFTER_FMA) {
            // scalar result feeds the pair source of the next packed instruction, whose low half feeds the next scalar one
            asm volatile(R16(R4("v_fma_f32 v100, v100, %0, %1\n v_pk_fma_f32 v[100:101], v[100:101], %2, v[100:101]\n")) : : "v"(q[0].x), "v"(r[0].x), "v"(q[1]) : "v100", "v101");
        } else if (CASE == C_FMA_AFTER_PK) {
            asm volatile(R16(R4("v_pk_fma_f32 v[100:101], v[100:101], %0, %1\n v_fma_f32 v101, v101, %2, %3\n")) : : "v"(q[0]), "v"(r[0]), "v"(q[1].x), "v"(r[1].x) : "v100", "v101");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static float* d_out; static unsigned long long* d_ticks;

template <int CASE>
static void run(int waves_per_simd)
{
    const int rounds = 512, blocks = 1024 * waves_per_simd;
    pk_kernel<CASE><<<blocks, 64>>>(d_out, d_ticks, rounds, 1.0f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    pk_kernel<CASE><<<blocks, 64>>>(d_out, d_ticks, rounds, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(blocks);
    CK(hipMemcpy(t.data(), d_ticks, blocks * 8, hipMemcpyDeviceToHost));
    std::sort(t.begin(), t.end());
    const double n = double(rounds) * 64;
    printf("%-68s %d wave(s)/SIMD: %6.2f ticks per instruction (median wave); launch %7.1f us = %6.3f ns per instruction per SIMD\n",
           names[CASE], waves_per_simd, t[blocks / 2] / n, ms * 1e3, ms * 1e6 / (n * waves_per_simd));
}

template <int C> static void run_all(int w) { if constexpr (C < NCASE) { run<C>(w); run_all<C + 1>(w); } }

int main()
{
    CK(hipMalloc(&d_out, 1024 * 4 * 64 * sizeof(float)));
    CK(hipMalloc(&d_ticks, 1024 * 4 * 8));
    for (int i = 0; i < 200; ++i) pk_kernel<C_FMA_IND><<<4096, 64>>>(d_out, d_ticks, 512, 1.0f);    // clocks up
    CK(hipDeviceSynchronize());
    for (int w : { 1, 2 }) run_all<0>(w);
    return 0;
}
