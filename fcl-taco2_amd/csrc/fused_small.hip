// fused_small.hip — round 6: the small-launch tail of the training update, gfx950.
//
// The KD update (tts_distill.py:143-182) is a chain of ~260 dependent launches on the student's stream beside the frozen teacher's forward; a kernel
// trace shows a median of 8.8 us between two launches of that chain (each waits for a compute unit the teacher's LSTM-step workgroups hold), so a
// launch that only re-reads and re-writes a tensor costs its bytes AND a slot wait.  The entries here remove such launches:
//   * fcl_loss_terms_batch   : every element-wise loss term of a step (..._kd_student.py:759-802: mel L1 + MSE, duration / pitch / energy MSE, the
//                              output-KD and prosody-KD terms on the same tensors) in ONE launch driven by a term table; a term may carry two
//                              targets (ground truth and the teacher's output), so the gradient is written once instead of read-modify-written;
//   * fcl_sum_rows           : dst = row_valid * (src0 + ... + src5) (+ planes): the chains of fcl_add2d that assemble a gradient from its sources;
//   * fcl_bn_bwd_sums        : activation / dropout backward fused into the column sums of train-mode BatchNorm's backward (dz written once);
//   * fcl_act_bwd_sum        : fcl_act_bwd of dy + dy2;
//   * fcl_gather_rows_sum_fwd: fcl_gather_rows_fwd of src + src2 + src3.
// All exact fp32 with fp64 reductions, as the kernels they replace (backward.hip, pointwise.hip).
#include <algorithm>
#include "fcl_common.h"

namespace fcl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LossBatch {
    fcl_loss_term_t t[FCL_LOSS_MAX_TERMS];
    int first_block[FCL_LOSS_MAX_TERMS + 1];
    int n;
};

// (loss_grad1: fcl_common.h)

// one workgroup works on ONE term (first_block[] maps block -> term); the arithmetic per element is l1_mse_loss_grad_kernel's (backward.hip), the second
// target's gradient is added to the first's exactly as the accumulate pass did (g1 + g2 in fp32)
__global__ __launch_bounds__(256) void loss_terms_kernel(const LossBatch b) {
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.first_block[k + 1]) ++k;
    const fcl_loss_term_t& t = b.t[k];
    const int blk = blockIdx.x - b.first_block[k], nblk = b.first_block[k + 1] - b.first_block[k];
    const float* __restrict__ a = t.a;
    const float* __restrict__ tb = t.b;
    const float* __restrict__ tb2 = t.b2;
    const uint8_t* __restrict__ v1 = t.valid;
    const uint8_t* __restrict__ v2 = t.valid2;
    float* __restrict__ da = t.da;
    const int C = t.c, M = t.m;
    const float ic1 = (float)(1.0 / t.count), ic2 = tb2 ? (float)(1.0 / t.count2) : 0.f;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0, u1 = 0.0, u2 = 0.0, cnt2 = 0.0;
    if ((C & 3) == 0) {
        const int c4 = C >> 2;
        const long long total = (long long)M * c4;
        for (long long i = blk * 256LL + threadIdx.x; i < total; i += nblk * 256LL) {
            const int r = (int)(i / c4);
            const bool ok1 = !v1 || v1[r], ok2 = tb2 && (!v2 || v2[r]);
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (ok1 || ok2) {
                const f32x4 av = reinterpret_cast<const f32x4*>(a)[i];
                if (ok1) {
                    f32x4 bv = reinterpret_cast<const f32x4*>(tb)[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (t.b_log) bv[e] = logf(bv[e] + t.b_log_offset);
                        const float d = av[e] - bv[e];
                        s1 += fabsf(d);
                        s2 += (double)d * d;
                        g[e] = loss_grad1(d, t.w_l1, t.w_mse, ic1);
                    }
                    cnt += 4.0;
                }
                if (ok2) {
                    const f32x4 bv = reinterpret_cast<const f32x4*>(tb2)[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = av[e] - bv[e];
                        u1 += fabsf(d);
                        u2 += (double)d * d;
                        g[e] += loss_grad1(d, t.w_l1_2, t.w_mse_2, ic2);
                    }
                    cnt2 += 4.0;
                }
            }
            reinterpret_cast<f32x4*>(da)[i] = g;
            if (t.da_planes) {
                const int n = (int)(i - (long long)r * c4) * 4;
                uint2 hi, lo;
                split4(g, hi, lo);
                unsigned short* line = t.da_planes + ((size_t)r * (C >> 5) + (n >> 5)) * 64 + (n & 31);
                *reinterpret_cast<uint2*>(line) = hi;
                *reinterpret_cast<uint2*>(line + 32) = lo;
            }
        }
    } else {  // the scalar heads (C = 1) and any width that is not a multiple of 4
        const long long total = (long long)M * C;
        for (long long i = blk * 256LL + threadIdx.x; i < total; i += nblk * 256LL) {
            const int r = (int)(i / C);
            const bool ok1 = !v1 || v1[r], ok2 = tb2 && (!v2 || v2[r]);
            float g = 0.f;
            const float av = (ok1 || ok2) ? a[i] : 0.f;
            if (ok1) {
                float bv = tb[i];
                if (t.b_log) bv = logf(bv + t.b_log_offset);
                const float d = av - bv;
                s1 += fabsf(d);
                s2 += (double)d * d;
                cnt += 1.0;
                g = loss_grad1(d, t.w_l1, t.w_mse, ic1);
            }
            if (ok2) {
                const float d = av - tb2[i];
                u1 += fabsf(d);
                u2 += (double)d * d;
                cnt2 += 1.0;
                g += loss_grad1(d, t.w_l1_2, t.w_mse_2, ic2);
            }
            da[i] = g;
        }
    }
    double v[6] = {s1, s2, cnt, u1, u2, cnt2};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o);
    __shared__ double part[4][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 6; ++q) part[wave][q] = v[q];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int q = threadIdx.x;
        const double s = (part[0][q] + part[1][q]) + (part[2][q] + part[3][q]);
        double* dst = q < 3 ? t.sums : t.sums2;
        if (dst && s != 0.0) atomicAdd(dst + (q < 3 ? q : q - 3), s);
    }
}

struct SumRows {
    const float* src[FCL_SUM_ROWS_MAX];
    int n;
};

// dst[r, :] = row_valid[r] ? sum_k src[k][r, :] : 0 (dense [rows, cols], cols % 4 == 0); dst may alias one of the sources (each element is read, then written, by one thread)
__global__ void sum_rows_kernel(const SumRows s, const uint8_t* __restrict__ row_valid, float* dst, unsigned short* __restrict__ dst_p, int rows, int cols) {
    const int c4 = cols >> 2;
    const long long total = (long long)rows * c4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (!row_valid || row_valid[r]) {
            acc = reinterpret_cast<const f32x4*>(s.src[0])[i];
#pragma unroll
            for (int k = 1; k < FCL_SUM_ROWS_MAX; ++k)
                if (k < s.n) {
                    const f32x4 v = reinterpret_cast<const f32x4*>(s.src[k])[i];
                    acc += v;
                }
        }
        if (dst) reinterpret_cast<f32x4*>(dst)[i] = acc;
        if (dst_p) {
            const int n = (int)(i - (long long)r * c4) * 4;
            uint2 hi, lo;
            split4(acc, hi, lo);
            unsigned short* line = dst_p + ((size_t)r * (cols >> 5) + (n >> 5)) * 64 + (n & 31);
            *reinterpret_cast<uint2*>(line) = hi;
            *reinterpret_cast<uint2*>(line + 32) = lo;
        }
    }
}

__device__ __forceinline__ float act_bwd1(float g, float v, int act) {
    if (act == FCL_ACT_RELU) return v > 0.f ? g : 0.f;
    if (act == FCL_ACT_TANH) return g * (1.0f - v * v);
    if (act == FCL_ACT_SIGMOID) return g * v * (1.0f - v);
    return g;
}

// colsum_kernel (backward.hip) mode 3 with fcl_act_bwd as its prologue: dz = (dy + dy2) * keep * scale * act'(y) is formed on the fly, WRITTEN (the
// BatchNorm backward proper reads it next) and summed: dbeta[c] += sum_m dz, dgamma[c] += sum_m dz * (z - mean[c]) * invstd[c].  Same block shape,
// same eight rows in flight, same fp64 accumulation as colsum_kernel.
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ dy, const float* __restrict__ dy2, const float* __restrict__ y,
                                                          const uint8_t* __restrict__ keep, float scale, int act, const float* __restrict__ z,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dz,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C, int rows_per_block) {
    __shared__ double part[4][64], part_x[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
    double s = 0.0, sx = 0.0;
    if (c < C) {
        const float mu = mean[c], is = invstd[c];
        for (int m = m_lo + ty; m < m_hi; m += 32) {
            float g[8], w[8], yv[8];
            unsigned char kp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t o = (size_t)min(m + 4 * j, m_hi - 1) * C + c;
                g[j] = dy[o];
                if (dy2) g[j] += dy2[o];
                w[j] = z[o];
                yv[j] = act != FCL_ACT_NONE ? y[o] : 0.f;
                kp[j] = keep ? keep[o] : 1;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = keep ? (kp[j] ? g[j] * scale : 0.f) : g[j];
                t = act_bwd1(t, yv[j], act);
                if (m + 4 * j < m_hi) {
                    dz[(size_t)(m + 4 * j) * C + c] = t;
                    s += (double)(t * ((w[j] - mu) * is));
                    sx += (double)t;
                }
            }
        }
    }
    part[ty][tx] = s;
    part_x[ty][tx] = sx;
    __syncthreads();
    if (ty == 0 && c < C) {
        atomicAdd(dgamma + c, (float)((part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx])));
        atomicAdd(dbeta + c, (float)((part_x[0][tx] + part_x[1][tx]) + (part_x[2][tx] + part_x[3][tx])));
    }
}

// act_bwd_kernel (backward.hip) on dy + dy2
__global__ void act_bwd_sum_kernel(const float* __restrict__ dy, const float* __restrict__ dy2, const float* __restrict__ y, const uint8_t* __restrict__ keep,
                                   float scale, float* __restrict__ dz, long long n, int act, unsigned short* __restrict__ dzp, int cols) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float g = dy[i] + dy2[i];
        if (keep) g = keep[i] ? g * scale : 0.f;
        g = act_bwd1(g, act != FCL_ACT_NONE ? y[i] : 0.f, act);
        if (dz) dz[i] = g;
        if (dzp) store_p32(dzp, cols >> 5, (int)(i / cols), (int)(i % cols), g);
    }
}

// gather_rows_kernel (pointwise.hip) of src + src2 + src3 (dense [*, c], c % 4 == 0): one wave per destination row
__global__ void gather_rows_sum_kernel(const float* __restrict__ src, const float* __restrict__ src2, const float* __restrict__ src3, const int* __restrict__ idx,
                                       float* __restrict__ dst, int n, int c, unsigned short* __restrict__ dst_p) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    const long long r = (long long)idx[wave];
    const bool ok = r >= 0;
    const int ldp = (c + 31) >> 5;
    for (int j = lane * 4; j < ldp * 32; j += 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok && j < c) {
            v = *reinterpret_cast<const f32x4*>(src + (size_t)r * c + j);
            if (src2) v += *reinterpret_cast<const f32x4*>(src2 + (size_t)r * c + j);
            if (src3) v += *reinterpret_cast<const f32x4*>(src3 + (size_t)r * c + j);
        }
        if (dst && j < c) *reinterpret_cast<f32x4*>(dst + (size_t)wave * c + j) = v;
        if (dst_p) {
            uint2 hi, lo;
            split4(v, hi, lo);
            unsigned short* line = dst_p + ((size_t)wave * ldp + (j >> 5)) * 64 + (j & 31);
            *reinterpret_cast<uint2*>(line) = hi;
            *reinterpret_cast<uint2*>(line + 32) = lo;
        }
    }
}

static inline int grid1d(long long total, int block) {
    long long g = (total + block - 1) / block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace fcl

using namespace fcl;

extern "C" {

int fcl_loss_terms_batch(const fcl_loss_term_t* terms, int n_terms, fcl_stream_t stream) {
    FCL_REQUIRE(terms && n_terms >= 1 && n_terms <= FCL_LOSS_MAX_TERMS, FCL_ERR_INVALID, "loss_terms_batch: 1 .. %d terms", FCL_LOSS_MAX_TERMS);
    LossBatch b;
    int nb = 0;
    b.n = 0;
    for (int k = 0; k < n_terms; ++k) {
        const fcl_loss_term_t& t = terms[k];
        FCL_REQUIRE(t.m >= 0 && t.c > 0 && t.count > 0 && t.sums, FCL_ERR_INVALID, "loss_terms_batch: bad term %d", k);
        if (t.m == 0) continue;
        FCL_REQUIRE(t.a && t.b && t.da, FCL_ERR_INVALID, "loss_terms_batch: term %d: null operand", k);
        FCL_REQUIRE(!t.b2 || (t.sums2 && t.count2 > 0), FCL_ERR_INVALID, "loss_terms_batch: term %d: a second target needs sums2 and count2", k);
        FCL_REQUIRE(!t.da_planes || ((t.c & 31) == 0 && (reinterpret_cast<uintptr_t>(t.da_planes) & 127u) == 0), FCL_ERR_SHAPE,
                    "loss_terms_batch: term %d: planes need C %% 32 == 0 and a 128-byte aligned buffer", k);
        FCL_REQUIRE((t.c & 3) || (aligned16(t.a) && aligned16(t.b) && aligned16(t.da) && (!t.b2 || aligned16(t.b2))), FCL_ERR_ALIGN,
                    "loss_terms_batch: term %d: 16-byte aligned operands required when C %% 4 == 0", k);
        const long long work = (t.c & 3) ? (long long)t.m * t.c : (long long)t.m * (t.c >> 2);
        b.t[b.n] = t;
        b.first_block[b.n] = nb;
        nb += std::min(grid1d(work, 256), 1024);
        ++b.n;
    }
    if (b.n == 0) return 0;
    b.first_block[b.n] = nb;
    hipLaunchKernelGGL(loss_terms_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, b);
    return check_hip(hipGetLastError(), "loss_terms_batch");
}

int fcl_sum_rows(const float* const* srcs, int n_src, const uint8_t* row_valid, float* dst, uint16_t* dst_p, int rows, int cols, fcl_stream_t stream) {
    FCL_REQUIRE(srcs && n_src >= 1 && n_src <= FCL_SUM_ROWS_MAX && (dst || dst_p) && rows >= 0 && cols > 0, FCL_ERR_INVALID, "sum_rows: bad arguments");
    FCL_REQUIRE((cols & 3) == 0 && (!dst || aligned16(dst)), FCL_ERR_ALIGN, "sum_rows: cols %% 4 == 0 and 16-byte aligned operands required");
    FCL_REQUIRE(!dst_p || ((cols & 31) == 0 && (reinterpret_cast<uintptr_t>(dst_p) & 127u) == 0), FCL_ERR_SHAPE,
                "sum_rows: planes need cols %% 32 == 0 and a 128-byte aligned buffer");
    SumRows s;
    s.n = n_src;
    for (int k = 0; k < FCL_SUM_ROWS_MAX; ++k) {
        s.src[k] = k < n_src ? srcs[k] : nullptr;
        FCL_REQUIRE(k >= n_src || (srcs[k] && aligned16(srcs[k])), FCL_ERR_ALIGN, "sum_rows: source %d null or not 16-byte aligned", k);
    }
    if (rows == 0) return 0;
    hipLaunchKernelGGL(sum_rows_kernel, dim3(grid1d((long long)rows * (cols >> 2), 256)), dim3(256), 0, (hipStream_t)stream, s, row_valid, dst, dst_p, rows, cols);
    return check_hip(hipGetLastError(), "sum_rows");
}

int fcl_bn_bwd_sums(const float* dy, const float* dy2, const float* y_act, const uint8_t* keep, float keep_scale, int act, const float* z, const float* mean,
                    const float* invstd, float* dz, float* dgamma, float* dbeta, int m, int c, fcl_stream_t stream) {
    FCL_REQUIRE(dy && z && mean && invstd && dz && dgamma && dbeta && m >= 0 && c > 0, FCL_ERR_INVALID, "bn_bwd_sums: bad arguments");
    FCL_REQUIRE(act >= FCL_ACT_NONE && act <= FCL_ACT_SIGMOID && (y_act || act == FCL_ACT_NONE), FCL_ERR_INVALID, "bn_bwd_sums: bad act / missing activation output");
    if (m == 0) return 0;
    const int rpb = m >= 8192 ? 128 : 64;
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3((c + 63) / 64, (m + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, dy, dy2, y_act, keep, keep_scale, act, z,
                       mean, invstd, dz, dgamma, dbeta, m, c, rpb);
    return check_hip(hipGetLastError(), "bn_bwd_sums");
}

int fcl_act_bwd_sum(const float* dy, const float* dy2, const float* y, const uint8_t* keep, float keep_scale, float* dz, uint16_t* dzp, int cols, size_t n, int act,
                    fcl_stream_t stream) {
    FCL_REQUIRE(dy && dy2 && (dz || dzp) && (y || act == FCL_ACT_NONE) && act >= FCL_ACT_NONE && act <= FCL_ACT_SIGMOID, FCL_ERR_INVALID, "act_bwd_sum: bad arguments");
    FCL_REQUIRE(!dzp || (cols > 0 && (cols & 31) == 0 && n % (size_t)cols == 0 && (reinterpret_cast<uintptr_t>(dzp) & 127u) == 0), FCL_ERR_SHAPE,
                "act_bwd_sum: planes need cols %% 32 == 0, n %% cols == 0 and a 128-byte aligned buffer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_bwd_sum_kernel, dim3(grid1d((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, dy, dy2, y, keep, keep_scale, dz, (long long)n, act,
                       dzp, cols);
    return check_hip(hipGetLastError(), "act_bwd_sum");
}

int fcl_gather_rows_sum_fwd(const float* src, const float* src2, const float* src3, const int32_t* idx, float* dst, uint16_t* dst_p, int n, int c,
                            fcl_stream_t stream) {
    FCL_REQUIRE(src && idx && (dst || dst_p) && n >= 0 && c > 0, FCL_ERR_INVALID, "gather_rows_sum_fwd: bad arguments");
    FCL_REQUIRE((c & 3) == 0 && aligned16(src) && aligned16(dst) && aligned16(src2) && aligned16(src3), FCL_ERR_ALIGN,
                "gather_rows_sum_fwd: c %% 4 == 0 and 16-byte aligned operands required");
    FCL_REQUIRE((reinterpret_cast<uintptr_t>(dst_p) & 127u) == 0, FCL_ERR_ALIGN, "gather_rows_sum_fwd: planes must be 128-byte aligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gather_rows_sum_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, src2, src3, idx, dst, n, c, dst_p);
    return check_hip(hipGetLastError(), "gather_rows_sum_fwd");
}

}  // extern "C"
