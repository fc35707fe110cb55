// decoder_step.hip — latency-tuned kernels for the decoder's time loop (H6-H8), gfx950.
//
// The loop is a chain of small dependent contractions (SURVEY.md §7 "sequential dependence"): once the
// live row count drops, a 64-row LDS-staged GEMM tile spends its time in per-chunk barriers, not MFMAs.
// Both kernels here use 16-row tiles whose A operand (the rows' activations, full K) sits in LDS once,
// while every wave streams its own W rows straight from L2 into MFMA B fragments (no sharing between
// waves => no LDS round trip, guide §5 "GEMV / small-M: load straight to VGPRs"):
//   * feat_prenet_kernel : feat_out(t-1) -> [H10 scatter] -> prenet layer 0 -> prenet layer 1 of step t,
//                          three row-local GEMMs chained through LDS in ONE launch (was three).
//   * lstm_small_kernel  : one wave per gate (16 rows x 16 units each), gates exchanged through LDS, the
//                          LSTMCell + zoneout epilogue one (row, unit) per thread.
#include "fcl_common.h"

namespace fcl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigm_(float x) { return 1.0f / (1.0f + expf(-x)); }

// acc += A_l[16 x K] . W[16 rows x K]^T ; A_l row stride lda_l floats (LDS), wrow = &W[(n0 + r16) * ldw] or null.
__device__ __forceinline__ f32x4 rowtile_mma(const float* A_l, int lda_l, const float* wrow, int K, int r16, int kq) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* ap = A_l + r16 * lda_l + kq * 4;
    const float* wp = wrow ? wrow + kq * 4 : nullptr;
    int k = 0;
    for (; k + 64 <= K; k += 64) {  // 4 W fragments in flight per lane
        f32x4 b[4], a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b[i] = wp ? *reinterpret_cast<const f32x4*>(wp + k + i * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
            a[i] = *reinterpret_cast<const f32x4*>(ap + k + i * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][0], b[i][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][1], b[i][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][2], b[i][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][3], b[i][3], acc1, 0, 0, 0);
        }
    }
    for (; k < K; k += 16) {
        const bool in = k + kq * 4 < K;  // K % 4 == 0
        f32x4 b = {0.f, 0.f, 0.f, 0.f}, a = {0.f, 0.f, 0.f, 0.f};
        if (in) {
            if (wp) b = *reinterpret_cast<const f32x4*>(wp + k);
            a = *reinterpret_cast<const f32x4*>(ap + k);
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

// cooperative load of a [16 x K] row tile (rows m0.., zero beyond M) into LDS with row stride ld_l
__device__ __forceinline__ void load_rowtile(float* dst, int ld_l, const float* src, int ld, int K, int m0, int M) {
    const int per_row = K >> 2;
    for (int i = threadIdx.x; i < 16 * per_row; i += blockDim.x) {
        const int r = i / per_row, c = (i - r * per_row) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + r < M) v = *reinterpret_cast<const f32x4*>(src + (size_t)(m0 + r) * ld + c);
        *reinterpret_cast<f32x4*>(dst + r * ld_l + c) = v;
    }
}

__device__ __forceinline__ float drop_apply(float v, int mode, const uint8_t* keep, int ldkeep, int m, int n, int N,
                                            float scale, float p, unsigned int seed) {
    if (mode == 1) return keep[(size_t)m * ldkeep + n] ? v * scale : 0.f;
    if (mode == 2) {
        const unsigned int h = hash_u32(((unsigned int)m * (unsigned int)N + (unsigned int)n) ^ seed);
        return ((h >> 8) * (1.0f / 16777216.0f) >= p) ? v * scale : 0.f;
    }
    return v;
}

__global__ __launch_bounds__(512) void feat_prenet_kernel(const FeatPrenetArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ldU = a.U + 4, ldO = a.O + 4, ldP = a.P + 4;
    float* A1 = smem;            // [16, U]  h1 tile
    float* A2 = A1 + 16 * ldU;   // [16, O]  feat_out / prenet input
    float* A3 = A2 + 16 * ldO;   // [16, P]  prenet layer-0 output
    const int m0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int col = lane & 15, rq = lane >> 4;
    const unsigned int sbump = a.seed_dev ? *a.seed_dev * 0x9E3779B9u : 0u;
    const unsigned int seed0 = hash_u32(a.seed0 + sbump), seed1 = hash_u32(a.seed1 + sbump);

    // ---- H8 feat_out of the previous step (+ H10 scatter) ------------------------------------------
    if (a.h1) {
        load_rowtile(A1, ldU, a.h1, a.U, a.U, m0, a.M_feat);
        __syncthreads();
        for (int tile = wave; tile * 16 < a.O; tile += nwaves) {
            const int n = tile * 16 + r16;
            const f32x4 acc = rowtile_mma(A1, ldU, n < a.O ? a.wf_h + (size_t)n * a.U : nullptr, a.U, r16, kq);
            const int nc = tile * 16 + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rq * 4 + r, m = m0 + row;
                float v = 0.f;
                if (m < a.M_feat && nc < a.O) {
                    v = acc[r] + a.F0[(size_t)m * a.O + nc];
                    a.before[(size_t)(a.frame_off[m] + a.t_prev) * a.O + nc] = v;
                }
                if (nc < a.O) A2[row * ldO + nc] = v;
            }
        }
    } else {
        for (int i = threadIdx.x; i < 16 * ldO; i += blockDim.x) A2[i] = 0.f;  // prev_out = 0 at t = 0
    }
    if (!a.w0 || m0 >= a.M_pre) return;  // last step: feat only / rows that just finished
    __syncthreads();
    if (a.teacher_in) {  // teacher forcing: prenet input is y_{t-1}, not the decoder's own output
        load_rowtile(A2, ldO, a.teacher_in, a.teacher_ld, a.O, m0, a.M_pre);
        __syncthreads();
    }
    // ---- H6 prenet layer 0 ---------------------------------------------------------------------------
    for (int tile = wave; tile * 16 < a.P; tile += nwaves) {
        const int n = tile * 16 + r16;
        const f32x4 acc = rowtile_mma(A2, ldO, n < a.P ? a.w0 + (size_t)n * a.O : nullptr, a.O, r16, kq);
        const int nc = tile * 16 + col;
        if (nc < a.P) {
            const float bn = a.b0[nc];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rq * 4 + r, m = m0 + row;
                float v = fmaxf(acc[r] + bn, 0.f);
                if (m < a.M_pre) v = drop_apply(v, a.drop_mode, a.keep0, a.P, m, nc, a.P, a.keep_scale, a.drop_p, seed0);
                A3[row * ldP + nc] = v;
            }
        }
    }
    __syncthreads();
    // ---- H6 prenet layer 1 -> global (+ KD tap) --------------------------------------------------------
    for (int tile = wave; tile * 16 < a.P; tile += nwaves) {
        const int n = tile * 16 + r16;
        const f32x4 acc = rowtile_mma(A3, ldP, n < a.P ? a.w1 + (size_t)n * a.P : nullptr, a.P, r16, kq);
        const int nc = tile * 16 + col;
        if (nc < a.P) {
            const float bn = a.b1[nc];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + rq * 4 + r;
                if (m >= a.M_pre) continue;
                float v = fmaxf(acc[r] + bn, 0.f);
                v = drop_apply(v, a.drop_mode, a.keep1, a.P, m, nc, a.P, a.keep_scale, a.drop_p, seed1);
                a.pre_out[(size_t)m * a.P + nc] = v;
                if (a.tap_prenet) a.tap_prenet[(size_t)(a.frame_off[m] + a.t_cur) * a.P + nc] = v;
            }
        }
    }
}

// ---- small-M LSTM step: wave g owns gate g of 16 units x 16 rows ------------------------------------
constexpr int SMALL_KC = 512;  // K chunk resident in LDS

__global__ __launch_bounds__(256) void lstm_small_kernel(const LstmStepArgs a) {
    __shared__ __attribute__((aligned(16))) float A_l[16 * (SMALL_KC + 4)];
    __shared__ float g_l[4][16][17];
    const int m0 = blockIdx.y * 16, u0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int u = u0 + r16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < a.nterms; ++t) {
        const GemmTerm T = a.term[t];
        for (int k0 = 0; k0 < T.K; k0 += SMALL_KC) {
            const int kc = min(SMALL_KC, T.K - k0);
            __syncthreads();  // previous chunk fully consumed
            load_rowtile(A_l, kc + 4, T.A + k0, T.lda, kc, m0, a.M);
            __syncthreads();
            const float* wrow = u < a.U ? T.W + (size_t)(g * a.U + u) * T.ldw + k0 : nullptr;
            acc += rowtile_mma(A_l, kc + 4, wrow, kc, r16, kq);
        }
    }
    {
        const int col = lane & 15, rq = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) g_l[g][rq * 4 + r][col] = acc[r];
    }
    __syncthreads();
    // cell epilogue: thread -> (row, unit)
    const int row = threadIdx.x >> 4, uc = threadIdx.x & 15;
    const int m = m0 + row, uu = u0 + uc;
    if (m >= a.M || uu >= a.U) return;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pre[q] = g_l[q][row][uc] + (a.bias ? a.bias[q * a.U + uu] : 0.f);
    if (a.G) {
        const float* gr = a.G + (size_t)((long long)m * a.g_row_mul + a.g_row_add) * (4 * a.U);
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] += gr[q * a.U + uu];
    }
    if (a.rank1_w) {
        const float pos = (float)a.step / (float)a.dur[m];
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] += pos * a.rank1_w[q * a.U + uu];
    }
    const size_t off = (size_t)m * a.U + uu;
    const float h_old = a.h_in[off], c_old = a.c[off];
    const float ig = sigm_(pre[0]), fg = sigm_(pre[1]), gg = tanhf(pre[2]), og = sigm_(pre[3]);
    const float c_new = fg * c_old + ig * gg;
    const float h_new = og * tanhf(c_new);
    float h_o, c_o;
    if (a.zone_keep_h) {
        h_o = a.zone_keep_h[off] ? h_old : h_new;
        c_o = a.zone_keep_c[off] ? c_old : c_new;
    } else {
        h_o = a.zoneout * h_old + (1.0f - a.zoneout) * h_new;
        c_o = a.zoneout * c_old + (1.0f - a.zoneout) * c_new;
    }
    bool live = true;
    if (a.row_len) live = a.step < a.row_len[m];
    a.h_out[off] = live ? h_o : h_old;
    a.c[off] = live ? c_o : c_old;
    if (a.out2) {
        const long long orow = (a.out2_row_base ? (long long)a.out2_row_base[m] : (long long)m * a.out2_row_mul) + a.out2_row_add;
        a.out2[(size_t)orow * a.ld2 + a.out2_col_off + uu] = live ? h_o : 0.f;
    }
}

int launch_lstm_small(const LstmStepArgs& a, hipStream_t s) {
    double ksum = 0;
    for (int i = 0; i < a.nterms; ++i) ksum += a.term[i].K;
    ProfScope ps("lstm_small_kernel", 2.0 * a.M * 4.0 * a.U * ksum, a.M, s);
    dim3 grid((a.U + 15) / 16, (a.M + 15) / 16);
    hipLaunchKernelGGL(lstm_small_kernel, grid, dim3(256), 0, s, a);
    return check_hip(hipGetLastError(), "lstm_small launch");
}

int launch_feat_prenet(const FeatPrenetArgs& a, hipStream_t s) {
    const int rows = a.h1 ? a.M_feat : a.M_pre;
    if (rows <= 0) return 0;
    FCL_REQUIRE(!(a.U & 3) && !(a.O & 3) && !(a.P & 3), FCL_ERR_SHAPE, "feat_prenet: U/O/P must be multiples of 4");
    const size_t lds = sizeof(float) * 16 * ((size_t)(a.U + 4) + (a.O + 4) + (a.P + 4));
    FCL_REQUIRE(lds <= 160 * 1024, FCL_ERR_SHAPE, "feat_prenet: tile does not fit LDS (%zu B)", lds);
    static bool attr_set = false;
    if (!attr_set) {
        FCL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(feat_prenet_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    double fl = 0;
    if (a.h1) fl += 2.0 * a.M_feat * a.O * a.U;
    if (a.w0) fl += 2.0 * a.M_pre * ((double)a.P * a.O + (double)a.P * a.P);
    ProfScope ps("feat_prenet_kernel", fl, rows, s);
    hipLaunchKernelGGL(feat_prenet_kernel, dim3((rows + 15) / 16), dim3(512), lds, s, a);
    return check_hip(hipGetLastError(), "feat_prenet launch");
}

}  // namespace fcl
