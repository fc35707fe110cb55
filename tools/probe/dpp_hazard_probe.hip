// dpp_hazard_probe.hip -- standalone reproducer attempt for the round-5 finding (csrc/bilstm.hip, ks_exclusive): a wave64 DPP reduce-scatter right after a
// chain of v_pk_fma_f32 returns wrong values in lanes 48 - 63 when a THIRD, foreign wave shares the SIMD.  Kernel `probe` mimics the lane-split BiLSTM step
// (512 threads, 128 weight registers per lane, 16 LDS reads, 64 packed FMAs, 3 DPP exchanges, 4 more DPP broadcasts after a transcendental) and checks every
// step against the same sums exchanged through LDS.  `noise` is a foreign kernel (64 VGPRs, 256 threads) on a second stream.
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/probe/dpp_hazard_probe.hip -o tools/probe/dpp_hazard_probe ; run: ./dpp_hazard_probe  -> a pass / fail table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CTRL, int NOPS> __device__ __forceinline__ float dpp(float v) {
    if (NOPS > 0) asm volatile("s_nop %1" : "+v"(v) : "n"(NOPS - 1));
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <bool EXCL, int NOPS> __global__ __launch_bounds__(512) void probe(const float* __restrict__ w0, int steps, unsigned* __restrict__ bad, float* __restrict__ sink, const float4* __restrict__ gsrc) {
    __shared__ __attribute__((aligned(16))) float h_s[2][128];
    __shared__ float xch[512][9];
    if (EXCL) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const int j = threadIdx.x, q = j & 7;
    f32x2 w[4][16];
    for (int p = 0; p < 4; ++p) for (int k = 0; k < 16; ++k) w[p][k] = f32x2{w0[(j * 4 + p) * 32 + 2 * k], w0[(j * 4 + p) * 32 + 2 * k + 1]};
    if (j < 256) (&h_s[0][0])[j] = 0.01f * (j % 37);
    __syncthreads();
    unsigned nbad = 0; float keep = 0.f;
    float4 gq = gsrc[(size_t)blockIdx.x * 512 + j];  // (a dwordx4 load per step, prefetched one step ahead as the real kernels prefetch their gate pre-activations)
    for (int s = 0; s < steps; ++s) {
        const float4 gcur = gq;
        gq = gsrc[((size_t)(s + 1) * 128 + blockIdx.x) * 512 + j];
        const float gexp = (float)((((size_t)s * 128 + blockIdx.x) * 512 + j) & 1023);
        if (gcur.x != gexp || gcur.y != gexp + 0.25f || gcur.z != gexp + 0.5f || gcur.w != gexp + 0.75f) ++nbad;
        f32x4 hv[4];
        for (int e = 0; e < 4; ++e) hv[e] = *reinterpret_cast<const f32x4*>(&h_s[s & 1][16 * q + 4 * e]);
        f32x2 acc[4] = {f32x2{0, 0}, f32x2{0, 0}, f32x2{0, 0}, f32x2{0, 0}};
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float hk = hv[k >> 2][k & 3];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p] = __builtin_elementwise_fma(w[p][k], f32x2{hk, hk}, acc[p]); }
        float a8[8]; for (int p = 0; p < 4; ++p) { a8[2 * p] = acc[p][0]; a8[2 * p + 1] = acc[p][1]; }
        const bool l1 = q < 4, l2 = (q & 2) == 0, l3 = (q & 1) == 0;
        float k4[4], k2[2];
        for (int x = 0; x < 4; ++x) k4[x] = (l1 ? a8[x] : a8[4 + x]) + dpp<0x141, NOPS>(l1 ? a8[4 + x] : a8[x]);
        for (int x = 0; x < 2; ++x) k2[x] = (l2 ? k4[x] : k4[2 + x]) + dpp<0x4E, NOPS>(l2 ? k4[2 + x] : k4[x]);
        const float z = (l3 ? k2[0] : k2[1]) + dpp<0xB1, NOPS>(l3 ? k2[1] : k2[0]);
        const float a = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        const float hq = __builtin_fmaf(dpp<0x00, NOPS>(a), dpp<0x55, NOPS>(a), dpp<0xAA, NOPS>(a) * dpp<0xFF, NOPS>(a));
        for (int x = 0; x < 8; ++x) xch[j][x] = a8[x];  // the same reduction through LDS, in the DPP order of additions
        xch[j][8] = a;
        __syncthreads();
        const int g0 = j & ~7;
        auto at = [&](int lane, int x) { return xch[g0 + lane][x]; };
        const int m1 = 7 - q, m2 = q ^ 2, m3 = q ^ 1;
        auto K4 = [&](int lane, int x) { const bool L = lane < 4; return at(lane, L ? x : 4 + x) + at(7 - lane, L ? x : 4 + x); };
        auto K2 = [&](int lane, int x) { const bool L = (lane & 2) == 0; return K4(lane, L ? x : 2 + x) + K4(lane ^ 2, L ? x : 2 + x); };
        const float zr = K2(q, l3 ? 0 : 1) + K2(m3, l3 ? 0 : 1);
        const int q0 = j & ~3;
        const float hr = __builtin_fmaf(xch[q0][8], xch[q0 + 1][8], xch[q0 + 2][8] * xch[q0 + 3][8]);
        (void)m1; (void)m2;
        if (z != zr || hq != hr) ++nbad;
        if ((j & 3) == 0) h_s[(s & 1) ^ 1][j >> 2] = 0.5f * hq;
        keep += z;
        __syncthreads();
    }
    if (nbad) atomicAdd(bad + (j & 63), nbad);
    if (keep == 123.456f) sink[0] = keep;
}
template <bool TRANS> __global__ __launch_bounds__(256) void noise(float* x, int n) {  // TRANS: the foreign waves issue transcendentals (v_exp_f32 / v_rcp_f32) too
    float v[48]; for (int i = 0; i < 48; ++i) v[i] = x[(threadIdx.x + i) & 255];
    for (int it = 0; it < n; ++it) for (int i = 0; i < 48; ++i)
        v[i] = TRANS ? __builtin_amdgcn_rcpf(1.0f + __expf(-v[(i + 1) % 48])) + v[i] * 1e-3f : __builtin_fmaf(v[i], 1.0001f, v[(i + 1) % 48] * 1e-6f);
    float s = 0; for (int i = 0; i < 48; ++i) s += v[i];
    if (s == 1.2345f) x[0] = s;
}
__global__ __launch_bounds__(256) void noise_mem(float4* __restrict__ dst, const float4* __restrict__ src, long long n, int reps) {  // foreign waves that stream memory
    for (int r = 0; r < reps; ++r)
        for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) { float4 v = src[i]; v.x += 1.f; dst[i] = v; }
}
typedef short s16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void noise_mfma(float* x, int n) {  // foreign waves that keep the SIMD's matrix pipe busy (v_mfma_f32_16x16x32_bf16 back to back)
    s16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    for (int it = 0; it < n; ++it)
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
    float s = 0; for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][3];
    if (s == 1.2345f) x[0] = s;
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__global__ __launch_bounds__(256) void noise_ldsdma(const float4* __restrict__ src, float* x, int n) {  // foreign waves that stream global memory into LDS by LDS-DMA (the GEMM kernels' loaders)
    __shared__ __attribute__((aligned(16))) unsigned char ring[4][4096];
    const unsigned char* g = reinterpret_cast<const unsigned char*>(src) + (size_t)blockIdx.x * 65536 + threadIdx.x * 16;
    for (int it = 0; it < n; ++it) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + (size_t)(it & 15) * 4096), (lds_ptr_t)(&ring[it & 3][(threadIdx.x >> 6) * 1024]), 16, 0, 0);
        if ((it & 3) == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<float*>(ring)[threadIdx.x] == 1.2345f) x[0] = 1.f;
}
static float4 *g_src, *g_dst, *g_pat;
template <bool EXCL, int NOPS> static void run(const char* name, float* w, unsigned* bad, float* sink, float* nx, hipStream_t s1, hipStream_t s2) {
    for (int beside = 0; beside < 7; ++beside) {
        int bad_launches = 0; unsigned lanes_hi = 0, lanes_lo = 0;
        for (int rep = 0; rep < 100; ++rep) {
            hipMemsetAsync(bad, 0, 64 * 4, s1); hipStreamSynchronize(s1);
            if (beside == 1) for (int k = 0; k < 6; ++k) noise<false><<<1024, 256, 0, s2>>>(nx, 3000);
            if (beside == 6) for (int k = 0; k < 4; ++k) noise_ldsdma<<<1024, 256, 0, s2>>>(g_src, nx, 4000);
            if (beside == 5) for (int k = 0; k < 300; ++k) {  // the runtime's own fill / copy kernels (hipMemsetAsync / hipMemcpyAsync device to device) as the foreign waves
                hipMemsetAsync(nx, 0, 1024, s2); hipMemcpyAsync(g_dst, g_src, 8192, hipMemcpyDeviceToDevice, s2); }
            if (beside == 4) for (int k = 0; k < 4; ++k) noise_mfma<<<2048, 256, 0, s2>>>(nx, 20000);
            if (beside == 3) for (int k = 0; k < 4; ++k) noise_mem<<<2048, 256, 0, s2>>>(g_dst, g_src, 1LL << 22, 6);
            if (beside == 2) for (int k = 0; k < 6; ++k) noise<true><<<1024, 256, 0, s2>>>(nx, 600);
            probe<EXCL, NOPS><<<128, 512, 0, s1>>>(w, 200, bad, sink, g_pat);
            hipDeviceSynchronize();
            unsigned h[64]; hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
            unsigned lo = 0, hi = 0; for (int i = 0; i < 64; ++i) (i < 48 ? lo : hi) += h[i];
            bad_launches += (lo + hi) != 0; lanes_lo += lo; lanes_hi += hi;
        }
        printf("%-34s %-22s bad launches %3d / 100   mismatches lanes 0-47: %u  lanes 48-63: %u\n", name, beside == 0 ? "alone" : beside == 1 ? "beside FMA waves" : beside == 2 ? "beside exp/rcp waves" : beside == 3 ? "beside streaming waves" : beside == 4 ? "beside MFMA waves" : beside == 5 ? "beside runtime fill/copy" : "beside LDS-DMA waves", bad_launches, lanes_lo, lanes_hi);
    }
}
int main() {
    float *w, *sink, *nx; unsigned* bad; hipStream_t s1, s2;
    hipMalloc(&w, 512 * 4 * 32 * 4); hipMalloc(&sink, 4); hipMalloc(&nx, 1024); hipMalloc(&bad, 256); hipMemset(nx, 0, 1024);
    std::vector<float> hw(512 * 4 * 32); for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * ((int)(i * 2654435761u >> 20) % 200 - 100);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&g_src, 64 << 20); hipMalloc(&g_dst, 64 << 20); hipMemset(g_src, 0, 64 << 20);
    { const size_t n = (size_t)202 * 128 * 512; std::vector<float4> hp(n); for (size_t i = 0; i < n; ++i) { const float v = (float)(i & 1023); hp[i] = make_float4(v, v + 0.25f, v + 0.5f, v + 0.75f); }
      hipMalloc(&g_pat, n * 16); hipMemcpy(g_pat, hp.data(), n * 16, hipMemcpyHostToDevice); }
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    run<false, 0>("compiler wait states only", w, bad, sink, nx, s1, s2);
    run<false, 4>("+ s_nop 3 before every DPP", w, bad, sink, nx, s1, s2);
    run<true, 0>("whole register file (v255)", w, bad, sink, nx, s1, s2);
    run<true, 4>("whole register file + s_nop 3", w, bad, sink, nx, s1, s2);
    return 0;
}
