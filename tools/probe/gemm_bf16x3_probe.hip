// gemm_bf16x3_probe.hip — prototype: fp32-accurate GEMM on the bf16 MFMA pipes by operand splitting.
//   x = hi + lo (+2^-17 |x|),  hi = bf16_rn(x),  lo = bf16_rn(x - hi);   A.W^T ~= Ah.Wh + Ah.Wl + Al.Wh  (fp32 accumulate)
// Inputs are PRE-SPLIT planes (producers write them once): A_hi/A_lo [M,K] bf16, W_hi/W_lo [N,K] bf16; C fp32.
// Tile 128x128, 4 waves (2x2), wave tile 64x64 = 4x4 MFMA 16x16x32 tiles, BK = 32, register-staged, padded LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_LD = BK + 8;  // bf16 elements per LDS row: 80 B stride -> conflict-free b128 rows

__global__ __launch_bounds__(256) void gemm_bf16x3(const u16* __restrict__ Ah, const u16* __restrict__ Al, const u16* __restrict__ Wh,
                                                   const u16* __restrict__ Wl, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) u16 lds[2][4][BM * LDS_LD];  // [buf][Ah,Al,Wh,Wl]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r16 = lane & 15, kq = lane >> 4;
    // loader: 128 rows x 32 bf16 = 128 x 64 B = 512 x 16 B per plane -> 2 x b128 per thread per plane
    const int lrow = tid >> 2, lc = (tid & 3) * 8;  // rows 0..63 (+64), 8 bf16 per thread
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s16x8 st[4][2];
    auto fetch = [&](int k0) {
        const u16* src[4] = {Ah, Al, Wh, Wl};
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = lrow + h * 64;
                const int g = (p < 2 ? m0 : n0) + row;
                const int lim = p < 2 ? M : N;
                const int gc = g < lim ? g : lim - 1;
                st[p][h] = *reinterpret_cast<const s16x8*>(src[p] + (size_t)gc * K + k0 + lc);
            }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) *reinterpret_cast<s16x8*>(&lds[buf][p][(lrow + h * 64) * LDS_LD + lc]) = st[p][h];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) fetch(k0 + BK);
        s16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *reinterpret_cast<const s16x8*>(&lds[buf][0][(wm * 64 + i * 16 + r16) * LDS_LD + kq * 8]);
            al[i] = *reinterpret_cast<const s16x8*>(&lds[buf][1][(wm * 64 + i * 16 + r16) * LDS_LD + kq * 8]);
            bh[i] = *reinterpret_cast<const s16x8*>(&lds[buf][2][(wn * 64 + i * 16 + r16) * LDS_LD + kq * 8]);
            bl[i] = *reinterpret_cast<const s16x8*>(&lds[buf][3][(wn * 64 + i * 16 + r16) * LDS_LD + kq * 8]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
        if (more) {
            stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + rq * 4 + r, n = n0 + wn * 64 + j * 16 + col;
                if (m < M && n < N) C[(size_t)m * N + n] = acc[i][j][r];
            }
}

static u16 bf16_rn(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (u16)(u >> 16);
}
static float bf16_f(u16 h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main() {
    const int shapes[][3] = {{2500, 1024, 512}, {2048, 1024, 512}, {1100, 1024, 512}, {25000, 128, 640}, {3200, 384, 768}, {8192, 1024, 512}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        std::vector<float> A((size_t)M * K), W((size_t)N * K);
        for (auto& v : A) v = (float)rand() / RAND_MAX * 2 - 1;
        for (auto& v : W) v = ((float)rand() / RAND_MAX * 2 - 1) / sqrtf((float)K);
        std::vector<u16> ah(A.size()), al(A.size()), wh(W.size()), wl(W.size());
        for (size_t i = 0; i < A.size(); ++i) { ah[i] = bf16_rn(A[i]); al[i] = bf16_rn(A[i] - bf16_f(ah[i])); }
        for (size_t i = 0; i < W.size(); ++i) { wh[i] = bf16_rn(W[i]); wl[i] = bf16_rn(W[i] - bf16_f(wh[i])); }
        u16 *dah, *dal, *dwh, *dwl;
        float* dc;
        hipMalloc(&dah, ah.size() * 2); hipMalloc(&dal, al.size() * 2); hipMalloc(&dwh, wh.size() * 2); hipMalloc(&dwl, wl.size() * 2);
        hipMalloc(&dc, (size_t)M * N * 4);
        hipMemcpy(dah, ah.data(), ah.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dal, al.data(), al.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dwh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dwl, wl.data(), wl.size() * 2, hipMemcpyHostToDevice);
        dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_bf16x3, grid, dim3(256), 0, 0, dah, dal, dwh, dwl, dc, M, N, K);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_bf16x3, grid, dim3(256), 0, 0, dah, dal, dwh, dwl, dc, M, N, K);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<float> C((size_t)M * N);
        hipMemcpy(C.data(), dc, C.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int t = 0; t < 2000; ++t) {
            const int m = rand() % M, n = rand() % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * W[(size_t)n * K + k];
            maxerr = fmax(maxerr, fabs(ref - C[(size_t)m * N + n]));
            maxref = fmax(maxref, fabs(ref));
        }
        const double us = ms / reps * 1e3;
        printf("M=%5d N=%4d K=%4d grid %4d: %7.1f us  %6.1f TF fp32-equivalent (%6.1f TF bf16 executed)  max err %.2e (|ref| max %.2f)\n", M, N, K,
               grid.x * grid.y, us, 2.0 * M * N * K / us / 1e6, 6.0 * M * N * K / us / 1e6, maxerr, maxref);
        hipFree(dah); hipFree(dal); hipFree(dwh); hipFree(dwl); hipFree(dc);
    }
    return 0;
}
