// clock_probe.hip — shader clock under fp32-MFMA load (developer probe).
// Each wave issues N independent-accumulator v_mfma_f32_16x16x4_f32 and stamps s_memtime (shader clock) and
// wall_clock64 (100 MHz): clock = dcycles / dwall; MFMA issue interval = dcycles / N.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(long long* out, int iters, float seed) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = seed + threadIdx.x, y = seed * 0.5f;
    long long w0 = wall_clock64();
    long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    long long c1 = clock64();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = c1 - c0;
        out[blockIdx.x * 4 + 1] = w1 - w0;
    }
    if (a0[0] + a1[0] + a2[0] + a3[0] == 12345.f) out[0] = 0;
}

int main() {
    long long* d;
    hipMalloc(&d, 4096 * 4 * 8);
    long long h[8];
    int grids[] = {1, 64, 256, 1024, 2048};
    int blocks[] = {64, 256};
    for (int bi = 0; bi < 2; ++bi)
        for (int gi = 0; gi < 5; ++gi)
            for (int iters : {500, 5000, 50000}) {
                hipLaunchKernelGGL(probe, dim3(grids[gi]), dim3(blocks[bi]), 0, 0, d, iters, 1.0f);
                hipDeviceSynchronize();
                hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                double cyc = (double)h[0], wall_us = h[1] / 100.0;
                printf("grid %5d block %3d iters %6d: %.2f us, s_memtime/wall = %.0f MHz, cycles per MFMA (this wave) = %.1f\n", grids[gi], blocks[bi], iters,
                       wall_us, cyc / wall_us, cyc / (4.0 * iters));
            }
    return 0;
}
