// mainloop_probe.hip — which part of the pre-split GEMM's main loop costs the chunk time?  (gfx950)
// The loader-specialised main loop of gemm_planes.hip (128 x 128 tile, 8 compute + LW loader waves, NST-stage LDS ring, one s_barrier per 32-k
// chunk) with parts switched off:  EXP 0 full | 1 no MFMAs (fragment reads only) | 2 no fragment reads (MFMAs on registers) | 3 no LDS-DMA
// (loaders only join the barriers) | 4 LDS-DMA only (compute waves only join the barriers) | 5 MFMAs only (no reads, no DMA).
// Operands are zero planes of the right size (timing only).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/mainloop_probe.hip -o tools/probe/mainloop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned short u16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EXP, int LW, int NST, int GROUP>
__global__ __launch_bounds__(64 * (8 + LW)) void probe_kernel(const unsigned char* Ap, const unsigned char* Wp, int M, int N, int K, float* Y) {
    constexpr int WM = 4, WN = 2, TM = 2, TN = 4, BM = 128, BN = 128, NW = 8;
    constexpr int GA = BM / 8 / LW, GB = BN / 8 / LW, GPW = GA + GB, STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nx = gridDim.x, ny = gridDim.y, nwg = nx * ny, orig = blockIdx.y * nx + blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    int bx, by;
    if (GROUP <= 1) { by = t / nx; bx = t - by * nx; }
    else { const int per = GROUP * nx, grp = t / per, first = grp * GROUP, gsz = min(ny - first, GROUP), rem = t - grp * per; bx = rem / gsz; by = first + (rem - bx * gsz); }
    const int m0 = by * BM, n0 = bx * BN, nchunks = K >> 5;
    const size_t ld = (size_t)(K >> 5) * 128;
    if (wave >= NW) {
        const int lw = wave - NW;
        const unsigned coff = (unsigned)(((lane & 7) ^ (((lw & 1) << 2) | (lane >> 4))) * 16);
        const unsigned char *pa[GA], *pb[GB];
#pragma unroll
        for (int j = 0; j < GA; ++j) pa[j] = Ap + (size_t)min(m0 + (j * LW + lw) * 8 + (lane >> 3), M - 1) * ld + coff;
#pragma unroll
        for (int j = 0; j < GB; ++j) pb[j] = Wp + (size_t)min(n0 + (j * LW + lw) * 8 + (lane >> 3), N - 1) * ld + coff;
        auto issue = [&](int stage) {
            if (EXP == 3 || EXP == 5) return;
            unsigned char* sbase = smem + stage * STAGE + lw * 1024;
#pragma unroll
            for (int j = 0; j < GA; ++j) { glds16(pa[j], sbase + j * LW * 1024); pa[j] += 128; }
#pragma unroll
            for (int j = 0; j < GB; ++j) { glds16(pb[j], sbase + BM * 128 + j * LW * 1024); pb[j] += 128; }
        };
        for (int p = 0; p < NST - 1; ++p) issue(p);
        int is = NST - 1;
        for (int i = 0; i < nchunks; ++i) {
            const int left = nchunks - 1 - i;
            if (NST == 2) wait_vm<0>();  // two stages: only the chunk consumed next is in flight
            else if (NST >= 4 && left >= 2) wait_vm<2 * GPW>();
            else if (left >= 1) wait_vm<GPW>();
            else wait_vm<0>();
            asm volatile("s_barrier" ::: "memory");
            if (left >= NST - 1) { issue(is); is = is + 1 == NST ? 0 : is + 1; }
        }
        return;
    }
    const int wm = wave / WN, wn = wave % WN, r16 = lane & 15, kq = lane >> 4, sw = r16 >> 1;
    const int a_hi = (wm * TM * 16 + r16) * 128 + ((kq ^ sw) << 4), a_lo = (wm * TM * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
    const int b_hi = BM * 128 + (wn * TN * 16 + r16) * 128 + ((kq ^ sw) << 4), b_lo = BM * 128 + (wn * TN * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
    f32x4 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s16x8 ah[TM], al[TM], bh[TN], bl[TN];
    for (int i = 0; i < TM; ++i) { ah[i] = (s16x8){1, 2, 3, 4, 5, 6, 7, (short)lane}; al[i] = ah[i]; }
    for (int j = 0; j < TN; ++j) { bh[j] = (s16x8){1, 2, 3, 4, 5, 6, 7, (short)lane}; bl[j] = bh[j]; }
    if (EXP == 8) {  // fragment reads of chunk i+1 issued BEFORE the MFMAs of chunk i (register double buffer); lgkmcnt(0) before every barrier
        s16x8 ah2[TM], al2[TM], bh2[TN], bl2[TN];
        auto rd = [&](const unsigned char* sb, s16x8 (&xah)[TM], s16x8 (&xal)[TM], s16x8 (&xbh)[TN], s16x8 (&xbl)[TN]) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) { xah[tm] = *reinterpret_cast<const s16x8*>(sb + a_hi + tm * 2048); xal[tm] = *reinterpret_cast<const s16x8*>(sb + a_lo + tm * 2048); }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) { xbh[tn] = *reinterpret_cast<const s16x8*>(sb + b_hi + tn * 2048); xbl[tn] = *reinterpret_cast<const s16x8*>(sb + b_lo + tn * 2048); }
        };
        auto mm = [&](s16x8 (&xah)[TM], s16x8 (&xal)[TM], s16x8 (&xbh)[TN], s16x8 (&xbl)[TN]) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xal[tm], xbh[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xah[tm], xbl[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xah[tm], xbh[tn], acc[tm][tn], 0, 0, 0);
            }
        };
        asm volatile("s_barrier" ::: "memory");
        rd(smem, ah, al, bh, bl);
        int i = 0, st = 0;
        while (true) {
            if (i + 1 < nchunks) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                st = st + 1 == NST ? 0 : st + 1;
                rd(smem + st * STAGE, ah2, al2, bh2, bl2);
            }
            mm(ah, al, bh, bl);
            if (++i >= nchunks) break;
            if (i + 1 < nchunks) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                st = st + 1 == NST ? 0 : st + 1;
                rd(smem + st * STAGE, ah, al, bh, bl);
            }
            mm(ah2, al2, bh2, bl2);
            if (++i >= nchunks) break;
        }
        float s8 = 0.f;
        for (int ii = 0; ii < TM; ++ii) for (int j = 0; j < TN; ++j) s8 += acc[ii][j][0] + acc[ii][j][1] + acc[ii][j][2] + acc[ii][j][3];
        if (s8 == 12345.678f) Y[0] = s8;
        return;
    }
    int cs = 0;
    for (int i = 0; i < nchunks; ++i) {
        asm volatile("s_barrier" ::: "memory");
        const unsigned char* sb = smem + cs * STAGE;
        if (EXP != 2 && EXP != 4 && EXP != 5) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) { ah[tm] = *reinterpret_cast<const s16x8*>(sb + a_hi + tm * 2048); al[tm] = *reinterpret_cast<const s16x8*>(sb + a_lo + tm * 2048); }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) { bh[tn] = *reinterpret_cast<const s16x8*>(sb + b_hi + tn * 2048); bl[tn] = *reinterpret_cast<const s16x8*>(sb + b_lo + tn * 2048); }
        }
        if (EXP == 1) {  // keep the reads alive without MFMAs
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[tm][0][0] += (float)(ah[tm][0] + al[tm][1]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[0][tn][1] += (float)(bh[tn][0] + bl[tn][1]);
        } else if (EXP != 4) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            }
        }
        cs = cs + 1 == NST ? 0 : cs + 1;
    }
    if (EXP >= 6) {  // the library's epilogue: accumulators -> fp32 tile in the (idle) ring -> row-wise 16-byte stores
        constexpr int LDT = BN + 4;
        float* tile = reinterpret_cast<float*>(smem);
        const int col = lane & 15, rq = lane >> 4;
        __syncthreads();
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) tile[((wm * TM + tm) * 16 + rq * 4 + rr) * LDT + (wn * TN + tn) * 16 + col] = acc[tm][tn][rr];
        __syncthreads();
        const int rows = min(BM, M - m0);
        if (EXP == 6) {
            for (int i = threadIdx.x; i < rows * (BN / 4); i += 512) {
                const int rm = i / (BN / 4), c4 = (i - rm * (BN / 4)) * 4;
                *reinterpret_cast<f32x4*>(Y + (size_t)(m0 + rm) * N + n0 + c4) = *reinterpret_cast<const f32x4*>(tile + rm * LDT + c4);
            }
        } else {  // EXP 7: the same stores, non-temporal
            for (int i = threadIdx.x; i < rows * (BN / 4); i += 512) {
                const int rm = i / (BN / 4), c4 = (i - rm * (BN / 4)) * 4;
                __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(tile + rm * LDT + c4), reinterpret_cast<f32x4*>(Y + (size_t)(m0 + rm) * N + n0 + c4));
            }
        }
        return;
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) Y[0] = s;
}

template <int EXP, int LW, int NST, int GROUP>
void run(const unsigned char* A, const unsigned char* W, int M, int N, int K, float* Y, const char* name) {
    auto k = probe_kernel<EXP, LW, NST, GROUP>;
    const int lds = NST * 256 * 128;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(64 * (8 + LW)), lds, 0, A, W, M, N, K, Y);
    CK(hipEventRecord(a));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(64 * (8 + LW)), lds, 0, A, W, M, N, K, Y);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters, rounds = (double)grid.x * grid.y / 256.0;
    printf("  %-46s %8.1f us   %6.3f us per chunk per workgroup-round   (%5.1f TFLOP/s fp32-eq if it were the GEMM)\n", name, us, us / rounds / (K / 32), 2.0 * M * N * K / us / 1e6);
}

template <int LW, int NST, int GROUP>
void suite(const unsigned char* A, const unsigned char* W, int M, int N, int K, float* Y) {
    printf("M=%d N=%d K=%d, %d loader waves, %d stages, tile-order group %d: %d workgroups = %.2f rounds of 256\n", M, N, K, LW, NST, GROUP, ((N + 127) / 128) * ((M + 127) / 128),
           ((N + 127) / 128) * ((M + 127) / 128) / 256.0);
    run<0, LW, NST, GROUP>(A, W, M, N, K, Y, "full main loop");
    run<1, LW, NST, GROUP>(A, W, M, N, K, Y, "no MFMAs (DMA + fragment reads)");
    run<2, LW, NST, GROUP>(A, W, M, N, K, Y, "no fragment reads (DMA + MFMAs)");
    run<3, LW, NST, GROUP>(A, W, M, N, K, Y, "no DMA (fragment reads + MFMAs)");
    run<4, LW, NST, GROUP>(A, W, M, N, K, Y, "DMA only");
    run<5, LW, NST, GROUP>(A, W, M, N, K, Y, "MFMAs only");
    run<8, LW, NST, GROUP>(A, W, M, N, K, Y, "full main loop, fragment reads one chunk ahead");
    run<6, LW, NST, GROUP>(A, W, M, N, K, Y, "full main loop + epilogue (stage, 16-byte stores)");
    run<7, LW, NST, GROUP>(A, W, M, N, K, Y, "full main loop + epilogue, non-temporal stores");
}

int main() {
    const int Mmax = 12800, Nmax = 4096, Kmax = 2048;
    unsigned char *A, *W; float* Y;
    CK(hipMalloc(&A, (size_t)Mmax * Kmax * 4 + 4096)); CK(hipMemset(A, 0, (size_t)Mmax * Kmax * 4 + 4096));
    CK(hipMalloc(&W, (size_t)Nmax * Kmax * 4 + 4096)); CK(hipMemset(W, 0, (size_t)Nmax * Kmax * 4 + 4096));
    CK(hipMalloc(&Y, (size_t)25600 * 4096 * 4));
    suite<4, 3, 8>(A, W, 2560, 4096, 2048, Y);   // FCL-taco2-T LSTM step
    suite<4, 3, 8>(A, W, 12800, 4096, 512, Y);   // frame-sized training GEMM
    suite<4, 3, 8>(A, W, 24320, 1024, 256, Y);   // KD projection over every frame
    suite<4, 3, 8>(A, W, 24320, 512, 128, Y);
    suite<4, 3, 1>(A, W, 2560, 1024, 512, Y);    // FCL-taco2-S LSTM step (160 workgroups)
    printf("---- two ring stages (64 KB: TWO workgroups per CU)\n");
    suite<4, 2, 8>(A, W, 2560, 4096, 2048, Y);
    suite<4, 2, 8>(A, W, 12800, 4096, 512, Y);
    suite<4, 2, 8>(A, W, 24320, 1024, 256, Y);
    suite<4, 2, 1>(A, W, 2560, 1024, 512, Y);
    return 0;
}
