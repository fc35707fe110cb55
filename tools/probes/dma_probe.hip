// Developer probe: LDS-DMA fill rate per CU against the number of chunks in flight (ring depth), chunk size and source (one L2-resident tile that
// every workgroup re-reads, or a private stream per workgroup from HBM).  One workgroup per CU (160 KB of dynamic LDS), LW loader waves, no consumers.
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe tools/probes/dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// STAGE_KB per chunk, NST stages (NST - 1 chunks in flight), LW loader waves; each wave issues PW = STAGE_KB / LW pieces of 1 KB per chunk
template <int STAGE_KB, int NST, int LW>
__global__ __launch_bounds__(64 * LW) void probe(const unsigned char* __restrict__ src, size_t wg_stride, size_t span, int nchunks, long long* out) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int PW = STAGE_KB / LW;
    static_assert(PW * (NST - 1) <= 60, "vmcnt");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * wg_stride;
    size_t off = 0;
    auto issue = [&](int stage) {
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const size_t o = (off + (size_t)(wave * PW + p) * 1024) & (span - 1);  // span is a power of two
            glds16(base + o + lane * 16, smem + stage * STAGE_KB * 1024 + (wave * PW + p) * 1024);
        }
        off += (size_t)STAGE_KB * 1024;
    };
    const long long t0 = wall_clock64();
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) issue(p);
    int is = NST - 1;
    for (int i = 0; i < nchunks; ++i) {
        wait_vm<PW * (NST - 2)>();  // chunk i has landed
        __builtin_amdgcn_s_barrier();
        issue(is);
        is = is + 1 == NST ? 0 : is + 1;
    }
    wait_vm<0>();
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int STAGE_KB, int NST, int LW>
static void run(const char* what, const unsigned char* src, size_t wg_stride, size_t span, long long* out, int cus) {
    const int nchunks = 400;
    const int lds = STAGE_KB * 1024 * NST;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<STAGE_KB, NST, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<STAGE_KB, NST, LW>), dim3(cus), dim3(64 * LW), lds, 0, src, wg_stride, span, nchunks, out);
    hipDeviceSynchronize();
    std::vector<long long> h(cus);
    hipMemcpy(h.data(), out, sizeof(long long) * cus, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < cus; ++i) s += (double)h[i];
    const double us = s / cus * 0.01;  // 100 MHz wall clock
    const double bytes = (double)(nchunks + NST - 1) * STAGE_KB * 1024;
    printf("%-8s stage %2d KB  ring %d (%d in flight = %3d KB)  loaders %d : %6.3f us per chunk, %6.1f GB/s per CU, %5.2f TB/s chip\n", what, STAGE_KB, NST, NST - 1,
           (NST - 1) * STAGE_KB, LW, us / (nchunks + NST - 1), bytes / us / 1e3, bytes / us / 1e3 * cus / 1e3);
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t total = (size_t)cus * (8u << 20);  // 8 MB private stream per workgroup
    unsigned char* src = nullptr;
    long long* out = nullptr;
    hipMalloc(&src, total);
    hipMemset(src, 1, total);
    hipMalloc(&out, sizeof(long long) * cus);
    printf("CUs %d\n", cus);
    // every workgroup re-reads one 256 KB tile (L2-resident)
#define L2RUN(S, N, L) run<S, N, L>("L2", src, 0, 256u << 10, out, cus)
#define HBMRUN(S, N, L) run<S, N, L>("HBM", src, 8u << 20, 8u << 20, out, cus)
    L2RUN(32, 2, 4); L2RUN(32, 3, 4); L2RUN(32, 4, 4); L2RUN(32, 5, 4);
    L2RUN(16, 3, 4); L2RUN(16, 5, 4); L2RUN(16, 9, 4);
    L2RUN(32, 3, 2); L2RUN(32, 4, 2); L2RUN(16, 7, 2); L2RUN(16, 4, 1);
    HBMRUN(32, 3, 4); HBMRUN(32, 4, 4); HBMRUN(32, 5, 4); HBMRUN(16, 9, 4);
    return 0;
}
