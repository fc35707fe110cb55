// fill_probe.hip — how fast can one CU pull L2-resident 128-byte lines into LDS?  (gfx950)
//   mode 0: direct LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction, no VGPRs) -- what the GEMM family's loaders use
//   mode 1: global_load_dwordx4 into VGPRs, then ds_write_b128 (software-pipelined: DEPTH instructions in flight per wave)
// One workgroup per CU (96 KB of dynamic LDS), LW loader waves, each streaming its share of 32 KB "chunks" from a buffer every workgroup
// shares (2 MB: resident in every XCD's L2).  No compute, no consumers: an upper bound of the fill rate beside which the MFMAs would run.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/fill_probe.hip -o tools/probe/fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int LW, int DEPTH>
__global__ __launch_bounds__(64 * LW) void fill_kernel(const unsigned char* __restrict__ src, size_t src_bytes, int chunks, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = 32 / LW;  // 1 KB pieces of a 32 KB chunk per wave
    const size_t base = ((size_t)blockIdx.x * 7919 * 1024) % src_bytes;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        for (int c = 0; c < chunks; ++c) {
            unsigned char* stage = smem + (c % 3) * 32768;
            const size_t off = (base + (size_t)c * 32768) % src_bytes;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int piece = j * LW + wave;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off + piece * 1024 + lane * 16), (lds_ptr_t)(stage + piece * 1024), 16, 0, 0);
            }
            if (c >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");  // two chunks stay in flight, as in the GEMM ring
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        f32x4 r[DEPTH][PER];
        auto issue = [&](int c, int slot) {
            const size_t off = (base + (size_t)c * 32768) % src_bytes;
#pragma unroll
            for (int j = 0; j < PER; ++j) r[slot][j] = *reinterpret_cast<const f32x4*>(src + off + (j * LW + wave) * 1024 + lane * 16);
        };
#pragma unroll
        for (int p = 0; p < DEPTH - 1; ++p) issue(p, p);
        for (int c = 0; c < chunks; c += DEPTH) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                const int cc = c + s;
                if (cc + DEPTH - 1 < chunks) issue(cc + DEPTH - 1, (s + DEPTH - 1) % DEPTH);
                unsigned char* stage = smem + (cc % 3) * 32768;
#pragma unroll
                for (int j = 0; j < PER; ++j) *reinterpret_cast<f32x4*>(stage + (j * LW + wave) * 1024 + lane * 16) = r[s][j];
            }
        }
    }
    __syncthreads();
    acc = *reinterpret_cast<f32x4*>(smem + threadIdx.x * 16);
    if (acc[0] == 12345.678f) sink[0] = acc[1];
}

template <int MODE, int LW, int DEPTH>
void run(const unsigned char* src, size_t bytes, float* sink, const char* name) {
    const int chunks = 3000;
    auto k = fill_kernel<MODE, LW, DEPTH>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 2; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * LW), 98304, 0, src, bytes, chunks, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double per_cu = (double)chunks * 32768 / (ms * 1e-3) / 1e9;
    printf("%-44s %7.1f GB/s per CU  %6.2f TB/s chip  (%.3f ms)\n", name, per_cu, per_cu * 256 / 1e3, ms);
}

int main() {
    const size_t bytes = 2u << 20;
    unsigned char* src; float* sink;
    CK(hipMalloc(&src, bytes + 65536)); CK(hipMemset(src, 1, bytes + 65536)); CK(hipMalloc(&sink, 16));
    run<0, 2, 1>(src, bytes, sink, "LDS-DMA, 2 loader waves");
    run<0, 4, 1>(src, bytes, sink, "LDS-DMA, 4 loader waves");
    run<0, 8, 1>(src, bytes, sink, "LDS-DMA, 8 loader waves");
    run<1, 4, 2>(src, bytes, sink, "load->VGPR->ds_write, 4 waves, depth 2");
    run<1, 4, 3>(src, bytes, sink, "load->VGPR->ds_write, 4 waves, depth 3");
    run<1, 8, 2>(src, bytes, sink, "load->VGPR->ds_write, 8 waves, depth 2");
    run<1, 8, 4>(src, bytes, sink, "load->VGPR->ds_write, 8 waves, depth 4");
    run<1, 16, 2>(src, bytes, sink, "load->VGPR->ds_write, 16 waves, depth 2");
    return 0;
}
