// Direct-to-LDS 16-byte buffer loads on gfx950 (buffer_load_dwordx4 ... lds): where does lane i's data land?
//   hipcc --offload-arch=gfx950 -O3 tools/exp_lds_dma.hip -o tools/_build/exp_lds_dma && tools/_build/exp_lds_dma
// Expectation (CDNA ISA: LDS_addr = M0 base + inst_offset + TID * size): chunk c of a tile at [c * 1024, (c + 1) * 1024), lane i at + 16 i --
// the layout the record tiles have in memory, so a wave can prefetch covariance chunks into LDS with no VGPRs (round 6, item 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) k(const unsigned* src, unsigned* dst, int nch)
{
    __shared__ u32x4 park[16 * 64];
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src) + (size_t)blockIdx.x * 16 * 256, 0, 16 * 1024, 0x00020000);
    const unsigned lane = threadIdx.x;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(park + c * 64), 16, lane * 16u + (c & 3) * 1024u, (c >> 2) * 4096, 0, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int c = 0; c < 16; ++c) {
        const u32x4 v = park[c * 64 + lane];
        unsigned* o = dst + ((size_t)blockIdx.x * 16 + c) * 256 + lane * 4;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
}
int main()
{
    const int tiles = 2048, n = tiles * 16 * 256;
    std::vector<unsigned> h(n), out(n);
    for (int i = 0; i < n; ++i) h[i] = 0x9e3779b9u * (unsigned)i + 7u;
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(o, 0, n * 4);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(64), 0, 0, d, o, 16);
    hipMemcpy(out.data(), o, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) bad += out[i] != h[i];
    std::printf("direct-to-LDS dwordx4 loads: %ld of %d words differ from the lane-major expectation (%s)\n", bad, n, bad ? "MISMATCH" : "layout confirmed");
    return bad != 0;
}
