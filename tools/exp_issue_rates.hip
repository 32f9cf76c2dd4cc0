// Experiment: issue cost of the vector instructions the reprojection-row fold uses, measured in shader clocks by the
// waves themselves (s_memtime around R rounds of 64 independent instructions = 16 chains x 4), for 1, 2 and 4 waves per
// SIMD, plus where the dispatcher puts the waves of a 1024-workgroup launch (HW_ID: SIMD / CU / SE / XCC of every wave).
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp_issue_rates.hip -o tools/_build/exp_issue_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

enum { OP_FMA32, OP_PKFMA32, OP_FMA64, OP_ADD64, OP_CVT_64_32, OP_CVT_32_64, OP_RSQ32, OP_RCP32, OP_RSQ64, OP_RCP64,
       OP_DPP_MOV, OP_BPERM, OP_DIV32, OP_SQRT32_IEEE, OP_DIV64, OP_FMA32_DEP, OP_FMA64_DEP, OP_FMA64_3V, OP_FMA32_3V, OP_MUL64_2V, OP_ADD64_2V, OP_CNDMASK64, NOPS };
static const char* names[NOPS] = { "v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_add_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_rsq_f32",
                                   "v_rcp_f32", "v_rsq_f64", "v_rcp_f64", "v_mov_b32 dpp row_shr", "ds_bpermute_b32",
                                   "a / b fp32 IEEE", "sqrtf IEEE", "a / b fp64 IEEE", "v_fma_f32 ONE dependent chain", "v_fma_f64 ONE dependent chain", "v_fma_f64 three VGPR-pair sources", "v_fma_f32 three VGPR sources", "v_mul_f64 two VGPR-pair sources", "v_add_f64 two VGPR-pair sources", "select on a double (2 x v_cndmask)" };
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NC = 16;

template <int OP>
__global__ __launch_bounds__(64) void rate_kernel(float* out, unsigned long long* ticks, unsigned* hwid, int rounds, float seed)
{
    float a[NC]; double d[NC]; f2 p[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; d[i] = a[i]; p[i] = f2{ a[i], a[i] + 1 }; }
    const float k = 0.999f + seed * 1e-6f; const double kd = k;
    double e2[NC], g2[NC]; float e1[NC], g1[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) { e2[i] = 0.999 + threadIdx.x * 1e-9 + i * 1e-7; g2[i] = 0.5 + threadIdx.x * 1e-6; e1[i] = (float)e2[i]; g1[i] = (float)g2[i];
        asm volatile("" : "+v"(e2[i]), "+v"(g2[i]), "+v"(e1[i]), "+v"(g1[i])); }
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (OP == OP_FMA32) a[i] = __builtin_fmaf(a[i], k, 0.5f);
            if (OP == OP_FMA32_DEP) a[0] = __builtin_fmaf(a[0], k, 0.5f);
            if (OP == OP_FMA64_DEP) d[0] = __builtin_fma(d[0], kd, 0.5);
            if (OP == OP_FMA64) d[i] = __builtin_fma(d[i], kd, 0.5);
            if (OP == OP_ADD64) d[i] = d[i] + kd;
            if (OP == OP_CVT_64_32) { d[i] = (double)a[i]; asm volatile("" : "+v"(d[i])); }
            if (OP == OP_CVT_32_64) { a[i] = (float)d[i]; asm volatile("" : "+v"(a[i])); }
            if (OP == OP_RSQ32) a[i] = __builtin_amdgcn_rsqf(a[i]);
            if (OP == OP_RCP32) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == OP_RSQ64) d[i] = __builtin_amdgcn_rsq(d[i]);
            if (OP == OP_RCP64) d[i] = __builtin_amdgcn_rcp(d[i]);
            if (OP == OP_PKFMA32) p[i] = p[i] * f2{ k, k } + f2{ 0.5f, 0.25f };
            if (OP == OP_DPP_MOV) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x111, 0xf, 0xf, false));
            if (OP == OP_BPERM) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 16) << 2, __builtin_bit_cast(int, a[i])));
            if (OP == OP_DIV32) a[i] = k / a[i];
            if (OP == OP_SQRT32_IEEE) a[i] = sqrtf(a[i]);
            if (OP == OP_DIV64) d[i] = kd / d[i];
            if (OP == OP_FMA64_3V) d[i] = __builtin_fma(d[i], e2[i], g2[i]);
            if (OP == OP_FMA32_3V) a[i] = __builtin_fmaf(a[i], e1[i], g1[i]);
            if (OP == OP_MUL64_2V) d[i] = d[i] * e2[i];
            if (OP == OP_ADD64_2V) d[i] = d[i] + g2[i];
            if (OP == OP_CNDMASK64) d[i] = (a[i] > 0.5f) ? e2[i] : d[i];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) s += a[i] + (float)d[i] + p[i].x + p[i].y + (float)e2[i] + (float)g2[i] + e1[i] + g1[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x] = t1 - t0;
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hwid[2 * blockIdx.x] = id; hwid[2 * blockIdx.x + 1] = xcc;
    }
}

static float* d_out; static unsigned long long* d_ticks; static unsigned* d_hw;

template <int OP>
static void run(int waves_per_simd, bool placement = false)
{
    const int rounds = 512, blocks = 1024 * waves_per_simd;
    rate_kernel<OP><<<blocks, 64>>>(d_out, d_ticks, d_hw, rounds, 1.0f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, 64>>>(d_out, d_ticks, d_hw, rounds, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(blocks);
    CK(hipMemcpy(t.data(), d_ticks, blocks * 8, hipMemcpyDeviceToHost));
    std::sort(t.begin(), t.end());
    const double n = double(rounds) * 64;
    printf("%-30s %d wave(s)/SIMD: per wave-instruction %6.2f ticks (median wave; min %6.2f max %6.2f); launch %7.1f us = %6.3f ns per instruction per SIMD\n",
           names[OP], waves_per_simd, t[blocks / 2] / n, t[0] / n, t[blocks - 1] / n, ms * 1e3, ms * 1e6 / (n * waves_per_simd));
    if (placement) {
        std::vector<unsigned> h(2 * blocks);
        CK(hipMemcpy(h.data(), d_hw, 2 * blocks * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, int> per_simd;
        for (int b = 0; b < blocks; ++b) {
            const unsigned id = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            per_simd[(xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd]++;
        }
        std::map<int, int> hist;
        for (auto& kv : per_simd) hist[kv.second]++;
        printf("   placement of %d one-wave workgroups: %zu distinct (xcc, se, sh, cu, simd) slots;", blocks, per_simd.size());
        for (auto& kv : hist) printf("  %d slot(s) hold %d wave(s)", kv.second, kv.first);
        printf("\n");
    }
}

int main()
{
    CK(hipMalloc(&d_out, 1024 * 8 * 64 * sizeof(float)));
    CK(hipMalloc(&d_ticks, 1024 * 8 * 8));
    CK(hipMalloc(&d_hw, 1024 * 8 * 8));
    for (int i = 0; i < 200; ++i) rate_kernel<OP_FMA32><<<4096, 64>>>(d_out, d_ticks, d_hw, 512, 1.0f);    // clocks up
    CK(hipDeviceSynchronize());
    for (int w : { 1, 2, 4 }) {
        run<OP_FMA32>(w, true); run<OP_FMA32_DEP>(w); run<OP_PKFMA32>(w); run<OP_FMA64>(w); run<OP_FMA64_DEP>(w); run<OP_ADD64>(w);
        run<OP_CVT_64_32>(w); run<OP_CVT_32_64>(w); run<OP_RSQ32>(w); run<OP_RCP32>(w); run<OP_RSQ64>(w); run<OP_RCP64>(w);
        run<OP_DPP_MOV>(w); run<OP_BPERM>(w); run<OP_DIV32>(w); run<OP_SQRT32_IEEE>(w); run<OP_DIV64>(w);
        run<OP_FMA64_3V>(w); run<OP_FMA32_3V>(w); run<OP_MUL64_2V>(w); run<OP_ADD64_2V>(w); run<OP_CNDMASK64>(w);
    }
    return 0;
}
