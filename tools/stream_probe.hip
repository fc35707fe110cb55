// developer aid (not part of libfcl_hip.so): how fast can EVERY CU pull the same few MB of weights through LDS-DMA, cyclically, from L2 —
// the feasibility question behind a persistent row-tile decoder kernel (each workgroup keeps 32 rows' h0 / h1 / prenet state in LDS and streams
// all the decoder's weights past them once per decoder step).  Skeleton of that kernel: 8 consumer waves + NL loader waves, a ring of NS 16 KB
// slots (one slot = one 16-column weight tile x one 32-k chunk per consumer wave, hi | lo planes), one s_barrier per slot.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned char u8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int SLOT = 16384, STATE = 96 * 1024;

template <int NL, int TM, int NS, int MODE>  // MODE 0: loaders only (consumers just meet the barrier); 1: + fragment reads + MFMAs
__global__ __launch_bounds__(64 * (8 + NL)) void probe(const u8* __restrict__ W, int slots_per_step, int steps, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(128))) u8 smem[];
    u8* ring = smem + STATE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = slots_per_step * steps;
    constexpr int PPL = 16 / NL;  // 1 KB pieces per loader wave and slot
    if (wave >= 8) {
        const int lw = wave - 8;
        const u8* src = W + lw * PPL * 1024 + lane * 16;
        int sstep = 0;  // slot index inside the step of the next slot to issue
        auto issue = [&](int i) {
            u8* dst = ring + (i % NS) * SLOT + lw * PPL * 1024;
            const u8* s = src + (size_t)sstep * SLOT;
#pragma unroll
            for (int j = 0; j < PPL; ++j) glds16(s + j * 1024, dst + j * 1024);
            sstep = sstep + 1 == slots_per_step ? 0 : sstep + 1;
        };
        for (int p = 0; p < NS - 1; ++p) issue(p);
        for (int i = 0; i < total; ++i) {
            const int left = total - 1 - i;
            if (left >= NS - 2) wait_vm<(NS - 2) * PPL>();
            else wait_vm<0>();
            asm volatile("s_barrier" ::: "memory");
            if (i + NS - 1 < total) issue(i + NS - 1);
        }
        return;
    }
    const int r16 = lane & 15, kq = lane >> 4, sw = r16 >> 1;
    const int w_hi = wave * 2048 + r16 * 128 + ((kq ^ sw) << 4), w_lo = wave * 2048 + r16 * 128 + (((4 + kq) ^ sw) << 4);
    f32x4 acc[4][TM];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) acc[g][tm] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (int c = 0; c < total / 4; ++c) {
        s16x8 ah[TM], al[TM];
        const u8* ab = smem + (c & 15) * (32 * 128);  // chunk-major state: 32 rows x 128 B per chunk
#pragma unroll
        for (int g = 0; g < 4; ++g, ++i) {
            asm volatile("s_barrier" ::: "memory");
            if (MODE == 0) continue;
            if (g == 0) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    ah[tm] = *reinterpret_cast<const s16x8*>(ab + ((tm & 1) * 16 + r16) * 128 + ((kq ^ sw) << 4));
                    al[tm] = *reinterpret_cast<const s16x8*>(ab + ((tm & 1) * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4));
                }
            }
            const u8* sb = ring + (i % NS) * SLOT;
            const s16x8 wh = *reinterpret_cast<const s16x8*>(sb + w_hi), wl = *reinterpret_cast<const s16x8*>(sb + w_lo);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[g][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[tm], acc[g][tm], 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[g][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[tm], acc[g][tm], 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[g][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[tm], acc[g][tm], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) s += acc[g][tm][0] + acc[g][tm][1] + acc[g][tm][2] + acc[g][tm][3];
    if (s == 12345.678f) out[blockIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NL, int TM, int NS, int MODE>
static void run(const u8* W, float* out, int grid, size_t wbytes, int steps) {
    const int slots = (int)(wbytes / SLOT) / 4 * 4;
    const int lds = STATE + NS * SLOT;
    auto k = probe<NL, TM, NS, MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * (8 + NL)), lds, 0, W, slots, 2, out);  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * (8 + NL)), lds, 0, W, slots, steps, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_step = 1e3 * ms / steps, bytes = (double)slots * SLOT;
    const int rounds = (grid + 255) / 256;
    printf("NL=%d TM=%d NS=%d mode=%d grid=%4d W=%5.2f MB : %7.1f us/step  %6.1f GB/s per WG  %5.1f B/clk per WG @2.4GHz  chip %5.1f TB/s  (LDS %d KB, %d round%s)\n", NL, TM, NS, MODE, grid,
           bytes / 1e6, us_step, bytes / us_step / 1e3, bytes / us_step / 2400.0, bytes * grid / us_step / 1e6, lds / 1024, rounds, rounds > 1 ? "s" : "");
}

int main(int argc, char** argv) {
    const size_t cap = 32u << 20;
    u8* W; float* out;
    CK(hipMalloc(&W, cap)); CK(hipMalloc(&out, 4096 * 4));
    std::vector<unsigned short> h(cap / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)(i * 2654435761u >> 25);
    CK(hipMemcpy(W, h.data(), cap, hipMemcpyHostToDevice));
    const int steps = 20;
    for (int grid : {64, 256, 300, 512}) {
        run<4, 2, 4, 0>(W, out, grid, 4600000, steps);
        run<4, 2, 4, 1>(W, out, grid, 4600000, steps);
    }
    run<2, 2, 4, 1>(W, out, 256, 4600000, steps);
    run<8, 2, 4, 1>(W, out, 256, 4600000, steps);
    run<4, 2, 3, 1>(W, out, 256, 4600000, steps);
    run<4, 3, 4, 1>(W, out, 256, 4600000, steps);
    run<4, 4, 4, 1>(W, out, 256, 4600000, steps);
    for (size_t wb : {1000000u, 2300000u, 9200000u, 18400000u}) run<4, 2, 4, 1>(W, out, 256, wb, steps);
    return 0;
}
