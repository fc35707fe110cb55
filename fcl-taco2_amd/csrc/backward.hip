// backward.hip — gradient primitives of the teacher-forced training step, gfx950.
//
// dX of every Linear / Conv1d reuses the forward multi-term GEMM (gemm_f32.hip) on transposed weights; what is new here:
//   * gemm_tn_kernel   : dW[n, k] += sum_m dY[m, n] * X[row(m) + shift, k]  (contraction over ROWS, optional conv tap shift with
//                        segment-bounded zero fill) — split over M across workgroups, fp32 atomics into the gradient buffer;
//   * colsum           : bias / beta gradients and BatchNorm-fold parameter gradients (column reductions over rows);
//   * act_bwd          : dz = dy * act'(y) (* dropout keep * scale);
//   * l1_mse_grad      : gradient of the masked-mean L1 + MSE losses;
//   * layernorm_bwd    : channel LayerNorm backward (+ the predictor's scalar head);
//   * lstm_cell_bwd    : LSTMCell + zoneout backward for one step (gate pre-activation gradients);
//   * scatter-add      : embedding gradient;  adam / sumsq: the optimizer.
// Exact fp32 everywhere (gradients are accumulated over up to 25 k rows; the split-bf16 trick is not used here).
#include <algorithm>
#include "fcl_common.h"
#include "lstm_epilogue.h"

namespace fcl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// C[n, k] += sum_{m in slice} A[m, n] * B[m + shift, k].  Tile 64(n) x 64(k) per workgroup, M walked in 32-row chunks;
// both operands are staged TRANSPOSED into LDS ([col][m], m contiguous) so MFMA fragments are float4 reads along m.
constexpr int TN_BM = 32;
constexpr int TN_LD = TN_BM + 4;

// PREC 0: exact fp32 MFMAs on fp32 tiles staged transposed ([col][m]) in LDS.
// PREC 1: bf16x3.  The transposition comes for free from the loader: lane = column, wave w loads rows 8w..8w+7 of the chunk (coalesced 256-byte row
// segments), so every lane ends up with 8 consecutive m of ITS column in registers, splits them once into bf16 hi / lo and stores one 16-byte
// vector per plane; the MFMA phase reads ready-made bf16 fragments (one 32-row MFMA step per chunk, three MFMAs per product).
// Both: the next chunk's global loads are issued before the MFMAs of the current one and consumed after the next barrier.
constexpr int TN_LDK = TN_BM + 8;  // bf16 elements per plane row: 80 B stride, conflict-free b128 reads and writes

template <int PREC>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                      float* __restrict__ C, int ldc, int M, int N, int K, int shift0,
                                                      const int* __restrict__ seg_lo, const int* __restrict__ seg_hi, int rows_per_slice, int ntaps,
                                                      long long c_tap_stride, const int hi_only) {
    constexpr int LDS_BYTES = PREC == 0 ? 2 * 64 * TN_LD * 4 : 4 * 64 * TN_LDK * 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    // XCD-aware order (as in gemm_f32.hip): the ids one XCD receives walk a contiguous range of (slice, tap, k tile, n tile), so the tiles that
    // read the same M slice of A and B meet in one L2 instead of all eight
    const int nx = gridDim.x, nxy = gridDim.x * gridDim.y, nwg = nxy * gridDim.z;
    const int orig = (blockIdx.z * gridDim.y + blockIdx.y) * nx + blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tl = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    const int bz = tl / nxy, by = (tl - bz * nxy) / nx, bx = tl - bz * nxy - by * nx;
    const int n0 = bx * 64, k0 = by * 64;
    const int tap = bz % ntaps, slice = bz / ntaps;  // conv weight gradients: all taps of a layer in one launch
    const int shift = shift0 + tap;
    C += (size_t)tap * c_tap_stride;
    const int m_lo = slice * rows_per_slice, m_hi = min(M, m_lo + rows_per_slice);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    // wave w owns C rows (n) [w*16, w*16+16) x all 64 k columns: acc[j] = 16x16 tile j
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (PREC == 0) {
        float* At = reinterpret_cast<float*>(lds_raw);  // [n][m]
        float* Bt = At + 64 * TN_LD;                    // [k][m]
        // loader: chunk = 32 rows x 64 cols per operand = 512 float4; thread handles float4 #tid and #tid+256
        f32x4 va[2], vb[2];
        auto fetch = [&](int mc) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = tid + h * 256;
                const int mr = idx >> 4, c4 = (idx & 15) * 4;  // row within chunk, first of 4 columns
                const int m = mc + mr;
                va[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
                vb[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (m < m_hi) {
                    if (n0 + c4 < N) va[h] = *reinterpret_cast<const f32x4*>(A + (size_t)m * lda + n0 + c4);  // N % 4 == 0
                    const int src = m + shift;
                    bool ok = k0 + c4 < K;
                    if (seg_lo) ok = ok && src >= seg_lo[m] && src < seg_hi[m];
                    else ok = ok && src >= 0 && src < M;
                    if (ok) vb[h] = *reinterpret_cast<const f32x4*>(B + (size_t)src * ldb + k0 + c4);
                }
            }
        };
        if (m_lo < m_hi) fetch(m_lo);
        for (int mc = m_lo; mc < m_hi; mc += TN_BM) {
            __syncthreads();  // every wave is done with the previous chunk
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = tid + h * 256;
                const int mr = idx >> 4, c4 = (idx & 15) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    At[(c4 + e) * TN_LD + mr] = va[h][e];
                    Bt[(c4 + e) * TN_LD + mr] = vb[h][e];
                }
            }
            __syncthreads();
            if (mc + TN_BM < m_hi) fetch(mc + TN_BM);  // in flight under the MFMAs below
#pragma unroll
            for (int s = 0; s < TN_BM / 16; ++s) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(At + (wave * 16 + r16) * TN_LD + s * 16 + kq * 4);
                f32x4 bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bt + (j * 16 + r16) * TN_LD + s * 16 + kq * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[j][e], acc[j], 0, 0, 0);
            }
        }
    } else {
        static_assert(TN_BM == 32, "one 32-row bf16 MFMA step per chunk; 4 waves x 8 rows");
        unsigned short* Ath = reinterpret_cast<unsigned short*>(lds_raw);  // [n][m] planes
        unsigned short *Atl = Ath + 64 * TN_LDK, *Bth = Atl + 64 * TN_LDK, *Btl = Bth + 64 * TN_LDK;
        const bool a_col = n0 + lane < N, b_col = k0 + lane < K;
        const float* ap = A + n0 + (a_col ? lane : 0);
        const float* bp = B + k0 + (b_col ? lane : 0);
        float ra[8], rb[8];
        unsigned okb = 0;  // validity bits of the 8 B rows of the stage in flight (applied at the split, so no load result is consumed early)
        auto fetch = [&](int mc) {
            okb = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = mc + wave * 8 + e;
                const int mcl = min(m, M - 1);
                ra[e] = ap[(size_t)mcl * lda];
                const int src = m + shift;
                bool ok = m < m_hi;
                if (seg_lo) ok = ok && src >= seg_lo[mcl] && src < seg_hi[mcl];
                else ok = ok && src >= 0 && src < M;
                rb[e] = bp[(size_t)(ok ? src : mcl) * ldb];
                okb |= ok ? (1u << e) : 0u;
            }
        };
        auto stash = [&](int mc) {
            f32x4 a0, a1, b0, b1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] = (a_col && mc + wave * 8 + e < m_hi) ? ra[e] : 0.f;
                a1[e] = (a_col && mc + wave * 8 + 4 + e < m_hi) ? ra[4 + e] : 0.f;
                b0[e] = (b_col && ((okb >> e) & 1u)) ? rb[e] : 0.f;
                b1[e] = (b_col && ((okb >> (4 + e)) & 1u)) ? rb[4 + e] : 0.f;
            }
            uint2 h0, l0, h1, l1;
            split4(a0, h0, l0);
            split4(a1, h1, l1);
            *reinterpret_cast<uint4*>(Ath + lane * TN_LDK + wave * 8) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4*>(Atl + lane * TN_LDK + wave * 8) = make_uint4(l0.x, l0.y, l1.x, l1.y);
            split4(b0, h0, l0);
            split4(b1, h1, l1);
            *reinterpret_cast<uint4*>(Bth + lane * TN_LDK + wave * 8) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4*>(Btl + lane * TN_LDK + wave * 8) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        };
        if (m_lo < m_hi) fetch(m_lo);
        for (int mc = m_lo; mc < m_hi; mc += TN_BM) {
            __syncthreads();  // every wave is done with the previous chunk
            stash(mc);
            __syncthreads();
            if (mc + TN_BM < m_hi) fetch(mc + TN_BM);  // in flight under the MFMAs below
            const int ao = (wave * 16 + r16) * TN_LDK + kq * 8;
            const s16x8 ah = *reinterpret_cast<const s16x8*>(Ath + ao), al = *reinterpret_cast<const s16x8*>(Atl + ao);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bo = (j * 16 + r16) * TN_LDK + kq * 8;
                const s16x8 bh = *reinterpret_cast<const s16x8*>(Bth + bo), bl = *reinterpret_cast<const s16x8*>(Btl + bo);
                if (!hi_only) {  // FCL_GEMM_BF16: bf16-rounded operands, the hi.hi product alone
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[j], 0, 0, 0);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[j], 0, 0, 0);
            }
        }
    }
    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + j * 16 + col;
        if (k >= K) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wave * 16 + rq * 4 + r;
            if (n < N) atomicAdd(C + (size_t)n * ldc + k, acc[j][r]);
        }
    }
}

// ---- bias / BatchNorm-affine gradients: column sums over rows --------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g,
                                                     const float* __restrict__ b, float* __restrict__ out, int M, int C, int mode, int rows_per_block,
                                                     float* __restrict__ out_x) {
    // fp64 accumulation (the kernel is HBM-bound either way): these sums are the dbeta / dgamma that train-mode BatchNorm's backward SUBTRACTS from
    // dy, i.e. operands of a cancellation; 7e-8 instead of 3e-7 relative at 1 600 rows.
    // 64 columns x rows_per_block rows per block; a wave reads 64 consecutive columns of one row (256-byte coalesced) and keeps EIGHT rows in
    // flight per thread: with one load per iteration the kernel ran at the latency of its row loop (r3 trace: 55 us for 16 MB).
    // out_x (optional): the plain column sums of x from the same pass (train-mode BatchNorm: dbeta beside dgamma, dy read once)
    __shared__ double part[4][64], part_x[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
    double s = 0.0, sx = 0.0;
    if (c < C) {
        const float bc = mode >= 2 ? b[c] : 0.f, gc = mode == 2 ? 1.0f / g[c] : (mode == 3 ? g[c] : 1.f);
        for (int m = m_lo + ty; m < m_hi; m += 32) {
            float v[8], w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t o = (size_t)min(m + 4 * j, m_hi - 1) * C + c;
                v[j] = x[o];
                w[j] = mode >= 1 ? y[o] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = v[j];
                if (mode == 1) t *= w[j];
                else if (mode >= 2) t *= (w[j] - bc) * gc;
                if (m + 4 * j < m_hi) {
                    s += (double)t;
                    sx += (double)v[j];
                }
            }
        }
    }
    part[ty][tx] = s;
    part_x[ty][tx] = sx;
    __syncthreads();
    if (ty == 0 && c < C) {
        atomicAdd(out + c, (float)((part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx])));
        if (out_x) atomicAdd(out_x + c, (float)((part_x[0][tx] + part_x[1][tx]) + (part_x[2][tx] + part_x[3][tx])));
    }
}

// ---- weight / bias gradients of a Conv1d with ONE input channel (the pitch / energy embeddings, ..._sa.py:435-443: Conv1d(1 -> C, k = 9)):
// dw[c, j] += sum_m dy[m, c] * x[m + j - pad] (positions inside row m's utterance), db[c] += sum_m dy[m, c].  The signal value of a tap is the same
// for the 64 columns a wave owns, so the inner loop is one dy load and k FMAs per row; rounds 1-2 issued k skinny TN GEMMs + k adds per embedding.
template <int KMAX>
__global__ __launch_bounds__(256) void conv1d_in1_dw_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x, const int* __restrict__ seg_lo,
                                                            const int* __restrict__ seg_hi, float* __restrict__ dw, float* __restrict__ db, int M, int C,
                                                            int k, int rows_per_block) {
    __shared__ float part[4][KMAX + 1][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx, cc = min(c, C - 1);
    const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block), pad = (k - 1) / 2;
    float acc[KMAX + 1];
#pragma unroll
    for (int j = 0; j <= KMAX; ++j) acc[j] = 0.f;
    for (int m = m_lo + ty; m < m_hi; m += 4) {
        const float v = dy[(size_t)m * ldy + cc];
        const int lo = seg_lo ? seg_lo[m] : 0, hi = seg_hi ? seg_hi[m] : M;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            const int src = m + j - pad;
            const float xv = (j < k && src >= lo && src < hi) ? x[src] : 0.f;
            acc[j] = fmaf(v, xv, acc[j]);
        }
        acc[KMAX] += v;
    }
#pragma unroll
    for (int j = 0; j <= KMAX; ++j) part[ty][j][tx] = acc[j];
    __syncthreads();
    if (ty == 0 && c < C) {
        for (int j = 0; j < k; ++j) atomicAdd(dw + (size_t)c * k + j, (part[0][j][tx] + part[1][j][tx]) + (part[2][j][tx] + part[3][j][tx]));
        if (db) atomicAdd(db + c, (part[0][KMAX][tx] + part[1][KMAX][tx]) + (part[2][KMAX][tx] + part[3][KMAX][tx]));
    }
}

// dz = dy * act'(y) * (keep ? keep*scale : 1)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const uint8_t* __restrict__ keep, float scale,
                               float* __restrict__ dz, long long n, int act, unsigned short* __restrict__ dzp, int cols) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float g = dy[i];
        if (keep) g = keep[i] ? g * scale : 0.f;
        const float v = y[i];  // for dropout sites y is the PRE-dropout activation
        if (act == FCL_ACT_RELU) g = v > 0.f ? g : 0.f;
        else if (act == FCL_ACT_TANH) g *= 1.0f - v * v;
        else if (act == FCL_ACT_SIGMOID) g *= v * (1.0f - v);
        dz[i] = g;
        if (dzp) store_p32(dzp, cols >> 5, (int)(i / cols), (int)(i % cols), g);  // pre-split operand of the input-gradient GEMM
    }
}

// yp (optional): the result as P32 planes of a [n / cols, cols] matrix (cols % 32 == 0) for the GEMM that consumes it
__global__ void act_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ keep, float scale, float* __restrict__ y, long long n, int act,
                               unsigned short* __restrict__ yp, int cols) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = x[i];
        v = act_apply(v, act);
        if (keep) v = keep[i] ? v * scale : 0.f;
        if (y) y[i] = v;
        if (yp) store_p32(yp, cols >> 5, (int)(i / cols), (int)(i % cols), v);
    }
}

__global__ void unpack_conv_grad_kernel(const float* __restrict__ dwp, const float* __restrict__ scale, float* __restrict__ dw, int cout, int cin, int k) {
    const long long total = (long long)k * cout * cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin), co = (int)((i / cin) % cout), j = (int)(i / ((long long)cin * cout));
        dw[((size_t)co * cin + ci) * k + j] += dwp[i] * (scale ? scale[co] : 1.0f);
    }
}

// grad of  w1 * mean_valid|a - b'| + w2 * mean_valid (a - b')^2  w.r.t. a:   (w1*sign(d) + 2*w2*d) / count   on valid rows (+= when accumulate)
__global__ void l1_mse_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, const uint8_t* __restrict__ valid, int M, int C,
                                   int b_log, float off, float w1, float w2, float inv_count, float* __restrict__ da, int accumulate) {
    const long long total = (long long)M * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C);
        float g = 0.f;
        if (!valid || valid[r]) {
            float bv = b[i];
            if (b_log) bv = logf(bv + off);
            const float d = a[i] - bv;
            g = (w1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + 2.f * w2 * d) * inv_count;
        }
        da[i] = accumulate ? da[i] + g : g;
    }
}

// the loss sums of fcl_masked_l1_mse_fwd and the gradient of fcl_l1_mse_grad in one pass over a and b (the KD terms read 100 MB teacher taps)
__global__ __launch_bounds__(256) void l1_mse_loss_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, const uint8_t* __restrict__ valid,
                                                               int M, int C, int b_log, float off, float w1, float w2, float inv_count,
                                                               float* __restrict__ da, int accumulate, double* __restrict__ sums,
                                                               unsigned short* __restrict__ dap) {
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    const int c4 = C >> 2;  // C % 4 == 0: one float4 per thread-iteration
    const long long total = (long long)M * c4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4);
        const bool ok = !valid || valid[r];
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const f32x4 av = reinterpret_cast<const f32x4*>(a)[i];
            f32x4 bv = reinterpret_cast<const f32x4*>(b)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (b_log) bv[e] = logf(bv[e] + off);
                const float d = av[e] - bv[e];
                s1 += fabsf(d);
                s2 += (double)d * d;
                g[e] = (w1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + 2.f * w2 * d) * inv_count;
            }
            cnt += 4.0;
        }
        f32x4* o = reinterpret_cast<f32x4*>(da) + i;
        if (accumulate) {
            const f32x4 old = *o;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += old[e];
        }
        *o = g;
        if (dap) {  // the gradient as P32 planes too (C % 32 == 0): the pre-split operand of the input-gradient GEMM that consumes it
            const int n = (int)(i - (long long)r * c4) * 4;
            uint2 hi, lo;
            split4(g, hi, lo);
            unsigned short* line = dap + ((size_t)r * (C >> 5) + (n >> 5)) * 64 + (n & 31);
            *reinterpret_cast<uint2*>(line) = hi;
            *reinterpret_cast<uint2*>(line + 32) = lo;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
        cnt += __shfl_xor(cnt, o);
    }
    __shared__ double part[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[wave][0] = s1; part[wave][1] = s2; part[wave][2] = cnt; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (v != 0.0) atomicAdd(sums + threadIdx.x, v);
    }
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// LayerNorm backward, one wave per row.  Forward: xh = (x - mean) * rstd ; y = xh*gamma + beta ; (optional head) s = y.w + b0, masked.
// Inputs: x (pre-LN), dy (grad of y, may be null when only the head contributes), ds (grad of the scalar head, may be null).
template <int MAXPER>
__global__ void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     const float* __restrict__ dy, const float* __restrict__ lin_w, const float* __restrict__ ds,
                                     const uint8_t* __restrict__ pad_mask, const uint8_t* __restrict__ keep, float keep_scale,
                                     float* __restrict__ dx, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, float* __restrict__ dlin_w, float* __restrict__ dlin_b, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    // parameter gradients of this wave's rows stay in registers; one atomic per (wave, column) at the end instead of one per (row, column)
    float gam[MAXPER], bet[MAXPER], lw[MAXPER], acc_g[MAXPER], acc_b[MAXPER], acc_w[MAXPER];
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) {
        const int j = lane + i * 64;
        gam[i] = j < C ? gamma[j] : 0.f;
        bet[i] = j < C ? beta[j] : 0.f;
        lw[i] = (ds && j < C) ? lin_w[j] : 0.f;
        acc_g[i] = acc_b[i] = acc_w[i] = 0.f;
    }
    float acc_lb = 0.f;
    for (int row = wave0; row < M; row += nwaves) {
        float v[MAXPER], gy[MAXPER];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXPER; ++i) {
            const int j = lane + i * 64;
            v[i] = j < C ? x[(size_t)row * C + j] : 0.f;
            s += v[i];
        }
        const float mean = wsum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXPER; ++i) {
            const int j = lane + i * 64;
            const float d = j < C ? v[i] - mean : 0.f;
            q += d * d;
        }
        const float rstd = 1.0f / sqrtf(wsum(q) / (float)C + eps);
        float dsr = 0.f;
        if (ds) dsr = (pad_mask && pad_mask[row]) ? 0.f : ds[row];
        acc_lb += dsr;
        float a1 = 0.f, a2 = 0.f;  // sum(g*gamma), sum(g*gamma*xh)
#pragma unroll
        for (int i = 0; i < MAXPER; ++i) {
            const int j = lane + i * 64;
            gy[i] = 0.f;
            if (j < C) {
                const float xh = (v[i] - mean) * rstd;
                float g = dy ? dy[(size_t)row * C + j] : 0.f;
                const float ks = keep ? (keep[(size_t)row * C + j] ? keep_scale : 0.f) : 1.f;  // dropout between the LayerNorm and its consumers
                if (ds) {
                    g += dsr * lw[i];
                    acc_w[i] += dsr * (xh * gam[i] + bet[i]) * ks;
                }
                g *= ks;
                gy[i] = g;
                acc_g[i] += g * xh;
                acc_b[i] += g;
                a1 += g * gam[i];
                a2 += g * gam[i] * xh;
            }
        }
        a1 = wsum(a1) / (float)C;
        a2 = wsum(a2) / (float)C;
#pragma unroll
        for (int i = 0; i < MAXPER; ++i) {
            const int j = lane + i * 64;
            if (j < C) {
                const float xh = (v[i] - mean) * rstd;
                dx[(size_t)row * C + j] = rstd * (gy[i] * gam[i] - a1 - xh * a2);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) {
        const int j = lane + i * 64;
        if (j < C) {
            atomicAdd(dgamma + j, acc_g[i]);
            atomicAdd(dbeta + j, acc_b[i]);
            if (ds && dlin_w) atomicAdd(dlin_w + j, acc_w[i]);
        }
    }
    if (ds && dlin_b && lane == 0) atomicAdd(dlin_b, acc_lb);
}

// LSTMCell + zoneout backward for one step.  Saved from forward: gates act [M,4U] (i,f,g,o after their nonlinearity), c_old, c_new (raw cell, before
// zoneout) -- see LstmStepArgs.save_*.  In: dh_out, dc_out (gradients w.r.t. the zoneout-ed outputs).  Out: dgates [M,4U] (pre-activation),
// dh_old_direct / dc_old (the zoneout "keep old" path and the f-gate path), to which the caller adds dgates . W_hh.
// dh_out and dh_old have their own row strides (ld_dh, ld_dho) and MAY BE THE SAME BUFFER (an element is read before it is written by the same
// thread): the decoder's reverse pass keeps both layers' hidden-state carries side by side in one [N, 2U] array.
__device__ __forceinline__ void lstm_cell_bwd_body(const CellBwdArgs& a) {
    const float* __restrict__ gates = a.gates;
    const float* __restrict__ c_old = a.c_old;
    const float* __restrict__ c_new = a.c_new;
    const float* dh_out = a.dh_out;
    const float* __restrict__ dh_out2 = a.dh_out2;
    const float* __restrict__ dc_out = a.dc_out;
    const uint8_t* __restrict__ zk_h = a.zk_h;
    const uint8_t* __restrict__ zk_c = a.zk_c;
    const int* __restrict__ row_len = a.row_len;
    float* __restrict__ dgates = a.dgates;
    float* dh_old = a.dh_old;
    float* __restrict__ dc_old_out = a.dc_old;
    unsigned short* __restrict__ dgates_p = a.dgates_p;
    const int ld_dh = a.ld_dh, ld_dho = a.ld_dho, ld_dh2 = a.ld_dh2, step = a.step, M = a.m, U = a.u;
    const float zoneout = a.zoneout;
    const long long total = (long long)M * U;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / U), u = (int)(idx - (long long)m * U);
        const float dho = dh_out[(size_t)m * ld_dh + u] + (dh_out2 ? dh_out2[(size_t)m * ld_dh2 + u] : 0.f), dco = dc_out ? dc_out[idx] : 0.f;
        const bool live = row_len ? (step < row_len[m]) : true;
        float dh_new, dc_new_z, dh_keep, dc_keep;
        if (!live) {  // state passed through untouched
            dh_new = 0.f; dc_new_z = 0.f; dh_keep = dho; dc_keep = dco;
        } else if (zk_h) {
            dh_new = zk_h[idx] ? 0.f : dho; dh_keep = zk_h[idx] ? dho : 0.f;
            dc_new_z = zk_c[idx] ? 0.f : dco; dc_keep = zk_c[idx] ? dco : 0.f;
        } else {
            dh_new = (1.0f - zoneout) * dho; dh_keep = zoneout * dho;
            dc_new_z = (1.0f - zoneout) * dco; dc_keep = zoneout * dco;
        }
        const float* gr = gates + (size_t)m * 4 * U;
        const float ig = gr[u], fg = gr[U + u], gg = gr[2 * U + u], og = gr[3 * U + u];
        const float tc = tanh_f(c_new[idx]);
        const float dc_new = dc_new_z + dh_new * og * (1.0f - tc * tc);
        float* dg = dgates + (size_t)m * 4 * U;
        const float d0 = dc_new * gg * ig * (1.0f - ig), d1 = dc_new * c_old[idx] * fg * (1.0f - fg);
        const float d2 = dc_new * ig * (1.0f - gg * gg), d3 = dh_new * tc * og * (1.0f - og);
        dg[u] = d0;
        dg[U + u] = d1;
        dg[2 * U + u] = d2;
        dg[3 * U + u] = d3;
        if (dgates_p) {  // the same gate gradients as the pre-split operand of the recurrence / input-gradient GEMMs (4U % 32 == 0)
            const int ld = (4 * U) >> 5;
            store_p32(dgates_p, ld, m, u, d0);
            store_p32(dgates_p, ld, m, U + u, d1);
            store_p32(dgates_p, ld, m, 2 * U + u, d2);
            store_p32(dgates_p, ld, m, 3 * U + u, d3);
        }
        dh_old[(size_t)m * ld_dho + u] = dh_keep;
        dc_old_out[idx] = dc_new * fg + dc_keep;
    }
}

__global__ void lstm_cell_bwd_kernel(const CellBwdArgs a) { lstm_cell_bwd_body(a); }
// two independent cell problems in one launch (blockIdx.y): layer 0 of step t beside layer 1 of step t - 1 in the decoder's reverse pass
__global__ void lstm_cell_bwd_pair_kernel(const CellBwdArgs a0, const CellBwdArgs a1) {
    if (blockIdx.y == 0) lstm_cell_bwd_body(a0);
    else lstm_cell_bwd_body(a1);
}

// dst[idx[m], :] += src[m, :]   (embedding gradient; rows with idx == skip contribute nothing: padding_idx)
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ dst, int M, int C,
                                        long long skip) {
    const long long total = (long long)M * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / C), c = (int)(i - (long long)m * C);
        const long long r = idx[m];
        if (r != skip) atomicAdd(dst + (size_t)r * C + c, src[i]);
    }
}

// dst[r, 0:cols] += alpha * src[r, 0:cols] for rows with row_valid[r] != 0 (all rows when row_valid is null); strided on both sides
__global__ void add2d_kernel(float* __restrict__ dst, int ld_dst, const float* __restrict__ src, int ld_src, int rows, int cols, float alpha,
                             const uint8_t* __restrict__ row_valid) {
    const long long total = (long long)rows * cols;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
        if (row_valid && !row_valid[r]) continue;
        dst[(size_t)r * ld_dst + c] += alpha * src[(size_t)r * ld_src + c];
    }
}

// ---- train-mode BatchNorm1d over rows (the reference normalises over ALL B x T positions of the padded batch, padding included) ----------------
// One launch: per-column sum and sum of squares in fp64 (ws[0:C], ws[C:2C], zero on entry), 64 columns x rows_per_block rows per block with eight
// rows in flight per thread; the LAST block of a column group to finish (a ticket per group, zero on entry) turns the sums into mean,
// 1/sqrt(biased var + eps) and the running statistics as torch (momentum m: r = (1-m) r + m stat, variance unbiased), and leaves the sums and its
// ticket zero again -- the workspace of one call is ready for the next (rounds 1-2: memset + statistics + finalize = 3 launches per BatchNorm).
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, int M, int C, int rows_per_block, double* __restrict__ ws,
                                                       unsigned int* __restrict__ tickets, float eps, float momentum, float* __restrict__ mean,
                                                       float* __restrict__ invstd, float* __restrict__ running_mean, float* __restrict__ running_var) {
    __shared__ double ps[4][64], pq[4][64];
    __shared__ int last;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
    double s = 0.0, q = 0.0;
    if (c < C) {
        for (int m = m_lo + ty; m < m_hi; m += 32) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = z[(size_t)min(m + 4 * j, m_hi - 1) * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double d = m + 4 * j < m_hi ? (double)v[j] : 0.0;
                s += d;
                q += d * d;
            }
        }
    }
    ps[ty][tx] = s;
    pq[ty][tx] = q;
    __syncthreads();
    if (ty == 0 && c < C) {
        // RETURNING atomics: the wave waits for the old values, i.e. both adds have been performed (at the device's coherence point) before this
        // block draws its ticket -- the ordering a __threadfence() would give, without its L2 write-back (buffer_wbl2: 157 us per call at 243 blocks)
        const double o1 = atomicAdd(ws + c, (ps[0][tx] + ps[1][tx]) + (ps[2][tx] + ps[3][tx]));
        const double o2 = atomicAdd(ws + C + c, (pq[0][tx] + pq[1][tx]) + (pq[2][tx] + pq[3][tx]));
        asm volatile("" ::"v"(o1), "v"(o2));
    }
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(tickets + blockIdx.x, 1u) == gridDim.y - 1;
    __syncthreads();
    if (!last || ty != 0 || c >= C) return;
    const double S = atomicAdd(ws + c, 0.0), Q = atomicAdd(ws + C + c, 0.0);  // (atomic reads: served by L2, where the other blocks' atomics landed)
    const double mu = S / M;
    double var = Q / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
    ws[c] = 0.0;
    ws[C + c] = 0.0;
    if (tx == 0) tickets[blockIdx.x] = 0u;
}


// y_act = act(gamma * (z - mean) * invstd + beta) ; y_drop = y_act * keep * scale (optional second output)
__global__ void bn_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const uint8_t* __restrict__ keep, float scale, float* __restrict__ y_act,
                                  float* __restrict__ y_drop, long long total, int C, int act, unsigned short* __restrict__ yp) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = (z[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        if (act == FCL_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == FCL_ACT_TANH) v = tanh_f(v);
        if (y_act) y_act[i] = v;  // (null: a forward that keeps nothing for a backward -- the frozen KD teacher -- writes the dropped output alone)
        const float vd = keep ? (keep[i] ? v * scale : 0.f) : v;
        if (y_drop) y_drop[i] = vd;
        if (yp) store_p32(yp, C >> 5, (int)(i / C), c, vd);  // the block's output (after dropout) as the next conv's pre-split operand
    }
}

// dz = gamma * invstd * (dy - dbeta/M - zhat * dgamma/M), dbeta = sum dy, dgamma = sum dy*zhat (this batch's sums)
__global__ void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd,
                              const float* __restrict__ gamma, const float* __restrict__ dbeta, const float* __restrict__ dgamma, float* __restrict__ dz,
                              long long total, int C, float inv_m, unsigned short* __restrict__ dzp, float* __restrict__ acc_dbeta,
                              float* __restrict__ acc_dgamma) {
    if (acc_dbeta && blockIdx.x == 0) {  // this batch's affine gradients join the accumulated ones (g += d: what the two fcl_add2d launches did)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            acc_dbeta[c] += dbeta[c];
            acc_dgamma[c] += dgamma[c];
        }
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float zh = (z[i] - mean[c]) * invstd[c];
        const float g = gamma[c] * invstd[c] * (dy[i] - dbeta[c] * inv_m - zh * dgamma[c] * inv_m);
        dz[i] = g;
        if (dzp) store_p32(dzp, C >> 5, (int)(i / C), c, g);
    }
}

// ---- round 6: the same three element-wise maps on FOUR consecutive channels per thread (C % 4 == 0, 16-byte aligned operands): float4 loads / stores, one 4-byte
// mask load, the planes as two 8-byte stores (the scalar forms above issue two 2-byte stores per element).  Per element the arithmetic is the scalar kernels':
// bit-identical results.  The frozen teacher's five postnet blocks move 170 MB each through bn_act_fwd: 0.37 ms of its stream per KD update with the scalar form.
__device__ __forceinline__ void planes4(unsigned short* __restrict__ p, int lines, long long row, int n, const f32x4 v) {
    uint2 hi, lo;
    split4(v, hi, lo);
    unsigned short* line = p + ((size_t)row * lines + (n >> 5)) * 64 + (n & 31);
    *reinterpret_cast<uint2*>(line) = hi;
    *reinterpret_cast<uint2*>(line + 32) = lo;
}
__device__ __forceinline__ f32x4 keep4(const f32x4 v, const uint8_t* __restrict__ keep, long long i4, float scale) {
    const unsigned int k = reinterpret_cast<const unsigned int*>(keep)[i4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ((k >> (8 * e)) & 0xFFu) ? v[e] * scale : 0.f;
    return o;
}

__global__ void bn_act_fwd4_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const uint8_t* __restrict__ keep, float scale, float* __restrict__ y_act,
                                   float* __restrict__ y_drop, long long total4, int C, int act, unsigned short* __restrict__ yp) {
    const int c4 = C >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4;
        const int c = (int)(i - row * c4) * 4;
        const f32x4 zv = reinterpret_cast<const f32x4*>(z)[i], mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c),
                    ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = (zv[e] - mu[e]) * is[e] * ga[e] + be[e];
            if (act == FCL_ACT_RELU) t = fmaxf(t, 0.f);
            else if (act == FCL_ACT_TANH) t = tanh_f(t);
            v[e] = t;
        }
        if (y_act) reinterpret_cast<f32x4*>(y_act)[i] = v;
        const f32x4 vd = keep ? keep4(v, keep, i, scale) : v;
        if (y_drop) reinterpret_cast<f32x4*>(y_drop)[i] = vd;
        if (yp) planes4(yp, C >> 5, row, c, vd);
    }
}

__global__ void act_fwd4_kernel(const float* __restrict__ x, const uint8_t* __restrict__ keep, float scale, float* __restrict__ y, long long total4, int act,
                                unsigned short* __restrict__ yp, int cols) {
    const int c4 = cols > 0 ? cols >> 2 : 1;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], act);
        if (keep) v = keep4(v, keep, i, scale);
        if (y) reinterpret_cast<f32x4*>(y)[i] = v;
        if (yp) {
            const long long row = i / c4;
            planes4(yp, cols >> 5, row, (int)(i - row * c4) * 4, v);
        }
    }
}

__global__ void bn_bwd4_kernel(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd,
                               const float* __restrict__ gamma, const float* __restrict__ dbeta, const float* __restrict__ dgamma, float* __restrict__ dz,
                               long long total4, int C, float inv_m, unsigned short* __restrict__ dzp, float* __restrict__ acc_dbeta,
                               float* __restrict__ acc_dgamma) {
    if (acc_dbeta && blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            acc_dbeta[c] += dbeta[c];
            acc_dgamma[c] += dgamma[c];
        }
    }
    const int c4 = C >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4;
        const int c = (int)(i - row * c4) * 4;
        const f32x4 dv = reinterpret_cast<const f32x4*>(dy)[i], zv = reinterpret_cast<const f32x4*>(z)[i], mu = *reinterpret_cast<const f32x4*>(mean + c),
                    is = *reinterpret_cast<const f32x4*>(invstd + c), ga = *reinterpret_cast<const f32x4*>(gamma + c), db = *reinterpret_cast<const f32x4*>(dbeta + c),
                    dg = *reinterpret_cast<const f32x4*>(dgamma + c);
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float zh = (zv[e] - mu[e]) * is[e];
            g[e] = ga[e] * is[e] * (dv[e] - db[e] * inv_m - zh * dg[e] * inv_m);
        }
        reinterpret_cast<f32x4*>(dz)[i] = g;
        if (dzp) planes4(dzp, C >> 5, row, c, g);
    }
}

// keep[i] = 1 with probability p_one (counter hash of (seed, i): the production source of dropout / zoneout masks in training)
__global__ void bernoulli_u8_kernel(uint8_t* __restrict__ out, long long n, unsigned int thresh, unsigned int seed, const unsigned int* __restrict__ seed_dev) {
    const unsigned int s = hash_u32(seed + (seed_dev ? seed_dev[0] : 0u));
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned int h = hash_u32(hash_u32((unsigned int)i ^ s) + (unsigned int)(i >> 32) * 0x9e3779b9U);
        out[i] = (h >> 8) < thresh ? 1 : 0;
    }
}

// fcl_bernoulli_batch: the sites travel as kernel arguments; a workgroup draws 4 096 consecutive bytes of one site
struct BernBatch {
    int n;
    unsigned char* out[FCL_BERNOULLI_MAX_SITES];
    long long len[FCL_BERNOULLI_MAX_SITES];
    unsigned int thresh[FCL_BERNOULLI_MAX_SITES], seed[FCL_BERNOULLI_MAX_SITES];
    int first_block[FCL_BERNOULLI_MAX_SITES + 1];
};

__global__ __launch_bounds__(256) void bernoulli_batch_kernel(const BernBatch b) {
    int k = 0;
    while (k + 1 < b.n && b.first_block[k + 1] <= (int)blockIdx.x) ++k;
    const long long i0 = (long long)((int)blockIdx.x - b.first_block[k]) * 4096;
    const unsigned int s = hash_u32(b.seed[k]);
    unsigned char* out = b.out[k];
    const long long n = b.len[k];
    const unsigned int thresh = b.thresh[k];
    if ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) {  // round 6: 16 consecutive bytes per thread, ONE 16-byte store (byte stores ran at 0.1 TB/s: 63 us for 7 MB)
        const long long i = i0 + threadIdx.x * 16;
        if (i + 16 <= n) {
            unsigned int w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned int v = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long ii = i + q * 4 + e;
                    const unsigned int h = hash_u32(hash_u32((unsigned int)ii ^ s) + (unsigned int)(ii >> 32) * 0x9e3779b9U);
                    v |= ((h >> 8) < thresh ? 1u : 0u) << (8 * e);
                }
                w[q] = v;
            }
            *reinterpret_cast<uint4*>(out + i) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (long long ii = i; ii < n; ++ii) {
                const unsigned int h = hash_u32(hash_u32((unsigned int)ii ^ s) + (unsigned int)(ii >> 32) * 0x9e3779b9U);
                out[ii] = (h >> 8) < thresh ? 1 : 0;
            }
        }
        return;
    }
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const long long i = i0 + j * 256 + threadIdx.x;
        if (i < n) {
            const unsigned int h = hash_u32(hash_u32((unsigned int)i ^ s) + (unsigned int)(i >> 32) * 0x9e3779b9U);
            out[i] = (h >> 8) < thresh ? 1 : 0;
        }
    }
}

__global__ void scale_kernel(float* __restrict__ x, long long n, float alpha) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= alpha;
}

__global__ void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = bx + j, c = by + threadIdx.x;  // dst is [cols, rows]
        if (r < cols && c < rows) dst[(size_t)r * rows + c] = tile[threadIdx.x][j];
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, double* __restrict__ out) {
    // one fp64 atomic per BLOCK of a capped grid (one per wave of an uncapped grid was 100 000 serialised atomics on one address: 200 us for 26 MB)
    __shared__ double part[4];
    double s = 0.0;
    const long long n4 = (reinterpret_cast<uintptr_t>(x) & 15u) == 0 ? n >> 2 : 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    for (long long i = 4 * n4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += (double)x[i] * x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

// Adam (torch.optim.Adam semantics, weight_decay 0, amsgrad off) with the clip coefficient and the NaN guard read from device memory:
// ctrl[0] = total grad-norm^2 (double).  clip = min(1, max_norm / (norm + 1e-6)); non-finite norm => no update (tts.py:173-179).
// skip = NaN / inf gradient norm (tts.py:173-179 skips optimizer.step() on NaN; inf: see include/fcl_hip.h) or a non-zero status word
__device__ __forceinline__ bool adam_skip(const double* ctrl, const unsigned int* status) {
    const double norm = sqrt(ctrl[0]);
    return !(norm == norm) || norm > 1e300 || (status && *status != 0u);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                            const double* __restrict__ ctrl, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                            const int* __restrict__ step_dev, const unsigned int* __restrict__ status) {
    if (adam_skip(ctrl, status)) return;
    const double norm = sqrt(ctrl[0]);
    const float t = (float)(*step_dev + 1);  // the counter is advanced by adam_commit_kernel, after every block has read it
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    const float clip = max_norm > 0.f ? fminf(1.0f, max_norm / ((float)norm + 1e-6f)) : 1.0f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        // torch.optim.Adam(weight_decay): L2 term added to the (already clipped: clip_grad_norm_ ran on .grad before step()) gradient
        const float gi = g[i] * clip + weight_decay * p[i];
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr * (mi / bc1) / (sqrtf(vi) / sqrtf(bc2) + eps);
    }
}

__global__ void adam_commit_kernel(const double* __restrict__ ctrl, int* __restrict__ step_dev, const unsigned int* __restrict__ status) {
    if (!adam_skip(ctrl, status)) *step_dev += 1;
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

int fcl_gemm_tn_taps_fwd(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, int shift0, int ntaps,
                         size_t c_tap_stride, const int32_t* seg_lo, const int32_t* seg_hi, fcl_stream_t stream) {
    FCL_REQUIRE(a && b && c && m >= 0 && n > 0 && k > 0 && ntaps >= 1, FCL_ERR_INVALID, "gemm_tn_fwd: bad arguments");
    FCL_REQUIRE(!(n & 3) && !(k & 3) && !(lda & 3) && !(ldb & 3) && aligned16(a) && aligned16(b), FCL_ERR_ALIGN,
                "gemm_tn_fwd: N, K, lda, ldb must be multiples of 4 and operands 16-byte aligned");
    FCL_REQUIRE((seg_lo == nullptr) == (seg_hi == nullptr), FCL_ERR_INVALID, "gemm_tn_fwd: seg_lo/seg_hi come in pairs");
    if (m == 0) return 0;
    static const int prec0 = tunable("PRECISION", 1);
    // bf16x3 / bf16 arithmetic: the 128 x 128 kernel with the transposition fused into its LDS reads (dw_gemm.hip) takes every shape it covers
    if (prec0 && launch_dw_mfma(a, lda, b, ldb, c, ldc, m, n, k, shift0, ntaps, c_tap_stride, seg_lo, seg_hi, fcl::gemm_mode() == FCL_GEMM_BF16, (hipStream_t)stream))
        return check_hip(hipGetLastError(), "gemm_tn_fwd");
    const int tiles = ((n + 63) / 64) * ((k + 63) / 64) * ntaps;
    static const int tn_wgs = tunable("TN_WORKGROUPS", 1024);
    int slices = (tn_wgs + tiles - 1) / tiles;  // ~1024 workgroups in flight, but at least 128 rows each: every slice ends in 4096 atomics per tile
    int rps = ((m + slices - 1) / slices + TN_BM - 1) / TN_BM * TN_BM;
    if (rps < 4 * TN_BM) rps = 4 * TN_BM;
    slices = (m + rps - 1) / rps;
    dim3 grid((n + 63) / 64, (k + 63) / 64, slices * ntaps);
    static const int prec = tunable("PRECISION", 1);  // as the forward GEMMs: 1 = bf16x3 operands, 0 = exact fp32 MFMA
    const int hi_only = prec && fcl::gemm_mode() == FCL_GEMM_BF16;
    ProfScope ps(hi_only ? "gemm_tn_kernel/bf16" : "gemm_tn_kernel", 2.0 * m * (double)n * k * ntaps, m, (hipStream_t)stream);
    if (prec) hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, c, ldc, m, n, k, shift0, seg_lo, seg_hi, rps,
                                 ntaps, (long long)c_tap_stride, hi_only);
    else hipLaunchKernelGGL(gemm_tn_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, c, ldc, m, n, k, shift0, seg_lo, seg_hi, rps,
                            ntaps, (long long)c_tap_stride, 0);
    return check_hip(hipGetLastError(), "gemm_tn_fwd");
}

int fcl_gemm_tn_fwd(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, int shift,
                    const int32_t* seg_lo, const int32_t* seg_hi, fcl_stream_t stream) {
    return fcl_gemm_tn_taps_fwd(a, lda, b, ldb, c, ldc, m, n, k, shift, 1, 0, seg_lo, seg_hi, stream);
}

int fcl_colsum2_fwd(const float* x, const float* y, const float* g, const float* b, float* out, float* out_x, int m, int c, int mode, fcl_stream_t stream);
int fcl_colsum_fwd(const float* x, const float* y, const float* g, const float* b, float* out, int m, int c, int mode, fcl_stream_t stream) {
    return fcl_colsum2_fwd(x, y, g, b, out, nullptr, m, c, mode, stream);
}

int fcl_colsum2_fwd(const float* x, const float* y, const float* g, const float* b, float* out, float* out_x, int m, int c, int mode, fcl_stream_t stream) {
    FCL_REQUIRE(x && out && m >= 0 && c > 0 && mode >= 0 && mode <= 3, FCL_ERR_INVALID, "colsum_fwd: bad arguments");
    FCL_REQUIRE(mode == 0 || y, FCL_ERR_INVALID, "colsum_fwd: mode needs y");
    FCL_REQUIRE(mode < 2 || (g && b), FCL_ERR_INVALID, "colsum_fwd: modes 2 and 3 need g and b");
    if (m == 0) return 0;
    const int rpb = m >= 8192 ? 128 : 64;
    hipLaunchKernelGGL(colsum_kernel, dim3((c + 63) / 64, (m + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, x, y, g, b, out, m, c, mode, rpb, out_x);
    return check_hip(hipGetLastError(), "colsum_fwd");
}

int fcl_act_bwd(const float* dy, const float* y, const uint8_t* keep, float keep_scale, float* dz, uint16_t* dzp, int cols, size_t n, int act,
                fcl_stream_t stream) {
    FCL_REQUIRE(dy && dz && (y || act == FCL_ACT_NONE) && act >= FCL_ACT_NONE && act <= FCL_ACT_SIGMOID, FCL_ERR_INVALID, "act_bwd: bad arguments");
    FCL_REQUIRE(!dzp || (cols > 0 && (cols & 31) == 0 && n % (size_t)cols == 0 && (reinterpret_cast<uintptr_t>(dzp) & 127u) == 0), FCL_ERR_SHAPE,
                "act_bwd: planes need cols %% 32 == 0, n %% cols == 0 and a 128-byte aligned buffer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1d((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y ? y : dy, keep, keep_scale, dz,
                       (long long)n, act, dzp, cols);
    return check_hip(hipGetLastError(), "act_bwd");
}

int fcl_act_fwd(const float* x, const uint8_t* keep, float keep_scale, float* y, uint16_t* yp, int cols, size_t n, int act, fcl_stream_t stream) {
    FCL_REQUIRE(x && (y || yp) && act >= FCL_ACT_NONE && act <= FCL_ACT_SIGMOID, FCL_ERR_INVALID, "act_fwd: bad arguments");
    FCL_REQUIRE(!yp || (cols > 0 && (cols & 31) == 0 && n % (size_t)cols == 0 && (reinterpret_cast<uintptr_t>(yp) & 127u) == 0), FCL_ERR_SHAPE,
                "act_fwd: planes need cols %% 32 == 0, n %% cols == 0 and a 128-byte aligned buffer");
    if (n == 0) return 0;
    if ((n & 3) == 0 && (!yp || (cols & 3) == 0) && aligned16(x) && (!y || aligned16(y)) && (!keep || (reinterpret_cast<uintptr_t>(keep) & 3u) == 0)) {
        hipLaunchKernelGGL(act_fwd4_kernel, dim3(grid1d((long long)(n >> 2), 256)), dim3(256), 0, (hipStream_t)stream, x, keep, keep_scale, y, (long long)(n >> 2), act,
                           yp, yp ? cols : 0);
        return check_hip(hipGetLastError(), "act_fwd");
    }
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid1d((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, x, keep, keep_scale, y, (long long)n, act, yp, cols);
    return check_hip(hipGetLastError(), "act_fwd");
}

int fcl_unpack_conv1d_grad(const float* dwp, const float* scale, float* dw, int cout, int cin, int k, fcl_stream_t stream) {
    FCL_REQUIRE(dwp && dw && cout > 0 && cin > 0 && k > 0, FCL_ERR_INVALID, "unpack_conv1d_grad: bad arguments");
    hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3(grid1d((long long)k * cout * cin, 256)), dim3(256), 0, (hipStream_t)stream, dwp, scale, dw, cout, cin, k);
    return check_hip(hipGetLastError(), "unpack_conv1d_grad");
}

int fcl_l1_mse_grad(const float* a, const float* b, const uint8_t* row_valid, int m, int c, int b_log, float b_log_offset, float w_l1, float w_mse,
                    double count, float* da, int accumulate, fcl_stream_t stream) {
    FCL_REQUIRE(a && b && da && m >= 0 && c > 0 && count > 0, FCL_ERR_INVALID, "l1_mse_grad: bad arguments");
    if (m == 0) return 0;
    hipLaunchKernelGGL(l1_mse_grad_kernel, dim3(grid1d((long long)m * c, 256)), dim3(256), 0, (hipStream_t)stream, a, b, row_valid, m, c, b_log,
                       b_log_offset, w_l1, w_mse, (float)(1.0 / count), da, accumulate);
    return check_hip(hipGetLastError(), "l1_mse_grad");
}

int fcl_l1_mse_loss_grad(const float* a, const float* b, const uint8_t* row_valid, int m, int c, int b_log, float b_log_offset, float w_l1, float w_mse,
                         double count, float* da, int accumulate, double* sums, uint16_t* da_planes, fcl_stream_t stream) {
    FCL_REQUIRE(!da_planes || ((c & 31) == 0 && (reinterpret_cast<uintptr_t>(da_planes) & 127u) == 0), FCL_ERR_SHAPE,
                "l1_mse_loss_grad: planes need C %% 32 == 0 and a 128-byte aligned buffer");
    FCL_REQUIRE(a && b && da && sums && m >= 0 && c > 0 && count > 0, FCL_ERR_INVALID, "l1_mse_loss_grad: bad arguments");
    FCL_REQUIRE(!(c & 3) && aligned16(a) && aligned16(b) && aligned16(da), FCL_ERR_ALIGN, "l1_mse_loss_grad: C %% 4 == 0 and 16-byte aligned operands required");
    if (m == 0) return 0;
    const int grid = std::min(grid1d((long long)m * (c / 4), 256), 1024);
    hipLaunchKernelGGL(l1_mse_loss_grad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, row_valid, m, c, b_log, b_log_offset, w_l1, w_mse,
                       (float)(1.0 / count), da, accumulate, sums, da_planes);
    return check_hip(hipGetLastError(), "l1_mse_loss_grad");
}

int fcl_layernorm_bwd(const float* x, const float* gamma, const float* beta, float eps, const float* dy, const float* lin_w, const float* ds,
                      const uint8_t* pad_mask, const uint8_t* keep, float keep_scale, float* dx, float* dgamma, float* dbeta, float* dlin_w,
                      float* dlin_b, int m, int c, fcl_stream_t stream) {
    FCL_REQUIRE(x && gamma && beta && dx && dgamma && dbeta && (dy || ds) && m >= 0 && c > 0 && c <= 1024, FCL_ERR_INVALID, "layernorm_bwd: bad arguments");
    FCL_REQUIRE(!ds || lin_w, FCL_ERR_INVALID, "layernorm_bwd: ds needs lin_w");
    if (m == 0) return 0;
    dim3 grid(std::min((m + 3) / 4, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define FCL_LNB(P) hipLaunchKernelGGL((layernorm_bwd_kernel<P>), grid, block, 0, s, x, gamma, beta, eps, dy, lin_w, ds, pad_mask, keep, keep_scale, dx, dgamma, dbeta, dlin_w, dlin_b, m, c)
    if (c <= 64) FCL_LNB(1);
    else if (c <= 256) FCL_LNB(4);
    else if (c <= 512) FCL_LNB(8);
    else FCL_LNB(16);
#undef FCL_LNB
    return check_hip(hipGetLastError(), "layernorm_bwd");
}

int fcl_lstm_cell_bwd(const float* gates, const float* c_old, const float* c_new, const float* dh_out, const float* dh_out2, int ld_dh2,
                      const float* dc_out, float zoneout,
                      const uint8_t* zone_keep_h, const uint8_t* zone_keep_c, const int32_t* row_len, int step, float* dgates, float* dh_old,
                      float* dc_old, uint16_t* dgates_p, int m, int u, fcl_stream_t stream) {
    return fcl::launch_lstm_cell_bwd(gates, c_old, c_new, dh_out, u, dh_out2, ld_dh2, dc_out, zoneout, zone_keep_h, zone_keep_c, row_len, step, dgates, dh_old, u,
                                     dc_old, dgates_p, m, u, (hipStream_t)stream);
}

}  // extern "C"

namespace fcl {
static int cell_bwd_check(const CellBwdArgs& a) {
    FCL_REQUIRE(a.ld_dh >= a.u && a.ld_dho >= a.u, FCL_ERR_SHAPE, "lstm_cell_bwd: row strides smaller than U");
    FCL_REQUIRE(a.gates && a.c_old && a.c_new && a.dh_out && a.dgates && a.dh_old && a.dc_old && a.m >= 0 && a.u > 0, FCL_ERR_INVALID, "lstm_cell_bwd: bad arguments");
    FCL_REQUIRE(!a.dgates_p || (((4 * a.u) & 31) == 0 && (reinterpret_cast<uintptr_t>(a.dgates_p) & 127u) == 0), FCL_ERR_SHAPE,
                "lstm_cell_bwd: planes need 4U %% 32 == 0 and a 128-byte aligned buffer");
    FCL_REQUIRE((a.zk_h == nullptr) == (a.zk_c == nullptr), FCL_ERR_INVALID, "lstm_cell_bwd: zoneout masks come in pairs");
    return 0;
}

int launch_lstm_cell_bwd(const CellBwdArgs& a, hipStream_t stream) {
    const int rc = cell_bwd_check(a);
    if (rc || a.m == 0) return rc;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid1d((long long)a.m * a.u, 256)), dim3(256), 0, stream, a);
    return check_hip(hipGetLastError(), "lstm_cell_bwd");
}

int launch_lstm_cell_bwd_pair(const CellBwdArgs& a0, const CellBwdArgs& a1, hipStream_t stream) {
    int rc = cell_bwd_check(a0);
    if (!rc) rc = cell_bwd_check(a1);
    if (rc) return rc;
    if (a0.m == 0) return launch_lstm_cell_bwd(a1, stream);
    if (a1.m == 0) return launch_lstm_cell_bwd(a0, stream);
    const long long big = (long long)std::max(a0.m, a1.m) * a0.u;
    hipLaunchKernelGGL(lstm_cell_bwd_pair_kernel, dim3(grid1d(big, 256), 2), dim3(256), 0, stream, a0, a1);
    return check_hip(hipGetLastError(), "lstm_cell_bwd pair");
}

int launch_lstm_cell_bwd(const float* gates, const float* c_old, const float* c_new, const float* dh_out, int ld_dh, const float* dh_out2, int ld_dh2,
                         const float* dc_out, float zoneout, const uint8_t* zone_keep_h, const uint8_t* zone_keep_c, const int32_t* row_len, int step,
                         float* dgates, float* dh_old, int ld_dho, float* dc_old, uint16_t* dgates_p, int m, int u, hipStream_t stream) {
    CellBwdArgs a = {gates, c_old, c_new, dh_out, dh_out2, dc_out, zone_keep_h, zone_keep_c, row_len, dgates, dh_old, dc_old, dgates_p,
                     ld_dh, ld_dho, ld_dh2, step, m, u, zoneout};
    return launch_lstm_cell_bwd(a, stream);
}

}  // namespace fcl

extern "C" {

int fcl_scatter_add_rows(const float* src, const int64_t* idx, float* dst, int m, int c, int64_t skip, fcl_stream_t stream) {
    FCL_REQUIRE(src && idx && dst && m >= 0 && c > 0, FCL_ERR_INVALID, "scatter_add_rows: bad arguments");
    if (m == 0) return 0;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid1d((long long)m * c, 256)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, m, c, (long long)skip);
    return check_hip(hipGetLastError(), "scatter_add_rows");
}

int fcl_add2d(float* dst, int ld_dst, const float* src, int ld_src, int rows, int cols, float alpha, const uint8_t* row_valid, fcl_stream_t stream) {
    FCL_REQUIRE(dst && src && rows >= 0 && cols > 0 && ld_dst >= cols && ld_src >= cols, FCL_ERR_INVALID, "add2d: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(add2d_kernel, dim3(grid1d((long long)rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, dst, ld_dst, src, ld_src, rows, cols,
                       alpha, row_valid);
    return check_hip(hipGetLastError(), "add2d");
}

static int bn_stats_launch(const float* z, int m, int c, float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                           double* workspace, bool clear, hipStream_t s) {
    FCL_REQUIRE(z && mean && invstd && workspace && m > 0 && c > 0, FCL_ERR_INVALID, "bn_stats_fwd: bad arguments");
    FCL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), FCL_ERR_INVALID, "bn_stats_fwd: running statistics come in pairs");
    const int groups = (c + 63) / 64;
    if (clear && hipMemsetAsync(workspace, 0, sizeof(double) * (size_t)(2 * c + groups), s) != hipSuccess) return check_hip(hipGetLastError(), "bn_stats_fwd memset");
    const int rpb = m >= 8192 ? 128 : 64;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(groups, (m + rpb - 1) / rpb), dim3(256), 0, s, z, m, c, rpb, workspace,
                       reinterpret_cast<unsigned int*>(workspace + 2 * (size_t)c), eps, momentum, mean, invstd, running_mean, running_var);
    return check_hip(hipGetLastError(), "bn_stats_fwd");
}

int fcl_bn_stats_fwd(const float* z, int m, int c, float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                     double* workspace, fcl_stream_t stream) {
    return bn_stats_launch(z, m, c, eps, momentum, mean, invstd, running_mean, running_var, workspace, true, (hipStream_t)stream);
}

int fcl_bn_stats_ws_fwd(const float* z, int m, int c, float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                        double* zero_workspace, fcl_stream_t stream) {
    return bn_stats_launch(z, m, c, eps, momentum, mean, invstd, running_mean, running_var, zero_workspace, false, (hipStream_t)stream);
}

int fcl_bn_act_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, const uint8_t* keep, float keep_scale,
                   float* y_act, float* y_drop, uint16_t* yp, int m, int c, int act, fcl_stream_t stream) {
    FCL_REQUIRE(z && mean && invstd && gamma && beta && (y_act || y_drop || yp) && m >= 0 && c > 0 && act >= FCL_ACT_NONE && act <= FCL_ACT_TANH, FCL_ERR_INVALID,
                "bn_act_fwd: bad arguments");
    FCL_REQUIRE(!keep || y_drop, FCL_ERR_INVALID, "bn_act_fwd: a keep mask needs y_drop");
    FCL_REQUIRE(!yp || ((c & 31) == 0 && (reinterpret_cast<uintptr_t>(yp) & 127u) == 0), FCL_ERR_SHAPE, "bn_act_fwd: planes need C %% 32 == 0, 128-byte aligned");
    if (m == 0) return 0;
    if ((c & 3) == 0 && aligned16(z) && aligned16(mean) && aligned16(invstd) && aligned16(gamma) && aligned16(beta) && (!y_act || aligned16(y_act)) &&
        (!y_drop || aligned16(y_drop)) && (!keep || (reinterpret_cast<uintptr_t>(keep) & 3u) == 0)) {  // four channels per thread (round 6)
        hipLaunchKernelGGL(bn_act_fwd4_kernel, dim3(grid1d((long long)m * (c >> 2), 256)), dim3(256), 0, (hipStream_t)stream, z, mean, invstd, gamma, beta, keep,
                           keep_scale, y_act, y_drop, (long long)m * (c >> 2), c, act, yp);
        return check_hip(hipGetLastError(), "bn_act_fwd");
    }
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid1d((long long)m * c, 256)), dim3(256), 0, (hipStream_t)stream, z, mean, invstd, gamma, beta, keep, keep_scale,
                       y_act, y_drop, (long long)m * c, c, act, yp);
    return check_hip(hipGetLastError(), "bn_act_fwd");
}

int fcl_bn_bwd(const float* dy, const float* z, const float* mean, const float* invstd, const float* gamma, const float* dbeta, const float* dgamma,
               float* dz, uint16_t* dzp, int m, int c, float* acc_dbeta, float* acc_dgamma, fcl_stream_t stream) {
    FCL_REQUIRE(dy && z && mean && invstd && gamma && dbeta && dgamma && dz && m > 0 && c > 0, FCL_ERR_INVALID, "bn_bwd: bad arguments");
    FCL_REQUIRE((acc_dbeta == nullptr) == (acc_dgamma == nullptr), FCL_ERR_INVALID, "bn_bwd: acc_dbeta / acc_dgamma come in pairs");
    FCL_REQUIRE(!dzp || ((c & 31) == 0 && (reinterpret_cast<uintptr_t>(dzp) & 127u) == 0), FCL_ERR_SHAPE, "bn_bwd: planes need C %% 32 == 0, 128-byte aligned");
    if ((c & 3) == 0 && aligned16(dy) && aligned16(z) && aligned16(mean) && aligned16(invstd) && aligned16(gamma) && aligned16(dbeta) && aligned16(dgamma) && aligned16(dz)) {
        hipLaunchKernelGGL(bn_bwd4_kernel, dim3(grid1d((long long)m * (c >> 2), 256)), dim3(256), 0, (hipStream_t)stream, dy, z, mean, invstd, gamma, dbeta, dgamma, dz,
                           (long long)m * (c >> 2), c, 1.0f / (float)m, dzp, acc_dbeta, acc_dgamma);
        return check_hip(hipGetLastError(), "bn_bwd");
    }
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(grid1d((long long)m * c, 256)), dim3(256), 0, (hipStream_t)stream, dy, z, mean, invstd, gamma, dbeta, dgamma, dz,
                       (long long)m * c, c, 1.0f / (float)m, dzp, acc_dbeta, acc_dgamma);
    return check_hip(hipGetLastError(), "bn_bwd");
}

int fcl_bernoulli_u8(uint8_t* out, size_t n, float p_one, uint32_t seed, const uint32_t* seed_dev, fcl_stream_t stream) {
    FCL_REQUIRE(out && p_one >= 0.f && p_one <= 1.f, FCL_ERR_INVALID, "bernoulli_u8: bad arguments");
    if (n == 0) return 0;
    const unsigned int thresh = (unsigned int)((double)p_one * 16777216.0 + 0.5);  // 24-bit uniform
    hipLaunchKernelGGL(bernoulli_u8_kernel, dim3(grid1d((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, out, (long long)n, thresh, seed, seed_dev);
    return check_hip(hipGetLastError(), "bernoulli_u8");
}

int fcl_bernoulli_batch(const fcl_bernoulli_site_t* sites, int n_sites, fcl_stream_t stream) {
    FCL_REQUIRE(n_sites >= 0 && n_sites <= FCL_BERNOULLI_MAX_SITES && (n_sites == 0 || sites), FCL_ERR_INVALID, "bernoulli_batch: 0 .. %d sites", FCL_BERNOULLI_MAX_SITES);
    BernBatch b = {};
    int blocks = 0;
    for (int k = 0; k < n_sites; ++k) {
        FCL_REQUIRE(sites[k].out && sites[k].n >= 0 && sites[k].p_one >= 0.f && sites[k].p_one <= 1.f, FCL_ERR_INVALID, "bernoulli_batch: bad site %d", k);
        if (sites[k].n == 0) continue;
        FCL_REQUIRE(sites[k].n < (1LL << 42), FCL_ERR_SHAPE, "bernoulli_batch: site too large");
        b.out[b.n] = sites[k].out;
        b.len[b.n] = sites[k].n;
        b.thresh[b.n] = (unsigned int)((double)sites[k].p_one * 16777216.0 + 0.5);
        b.seed[b.n] = sites[k].seed;
        b.first_block[b.n] = blocks;
        blocks += (int)((sites[k].n + 4095) / 4096);
        ++b.n;
    }
    if (b.n == 0) return 0;
    b.first_block[b.n] = blocks;
    hipLaunchKernelGGL(bernoulli_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
    return check_hip(hipGetLastError(), "bernoulli_batch");
}

int fcl_scale(float* x, size_t n, float alpha, fcl_stream_t stream) {
    FCL_REQUIRE(x, FCL_ERR_INVALID, "scale: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_kernel, dim3(grid1d((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, alpha);
    return check_hip(hipGetLastError(), "scale");
}

int fcl_conv1d_in1_dw(const float* dy, int ldy, const float* x, const int32_t* seg_lo, const int32_t* seg_hi, float* dw, float* db, int m, int c, int k,
                      fcl_stream_t stream) {
    FCL_REQUIRE(dy && x && dw && m >= 0 && c > 0 && ldy >= c && k >= 1 && k <= 16 && (k & 1), FCL_ERR_INVALID, "conv1d_in1_dw: bad arguments (odd k <= 16)");
    FCL_REQUIRE((seg_lo == nullptr) == (seg_hi == nullptr), FCL_ERR_INVALID, "conv1d_in1_dw: segment bounds come in pairs");
    if (m == 0) return 0;
    const int rpb = 64;
    hipLaunchKernelGGL(conv1d_in1_dw_kernel<16>, dim3((c + 63) / 64, (m + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, dy, ldy, x, seg_lo, seg_hi, dw, db,
                       m, c, k, rpb);
    return check_hip(hipGetLastError(), "conv1d_in1_dw");
}

int fcl_transpose2d(const float* src, float* dst, int rows, int cols, fcl_stream_t stream) {
    FCL_REQUIRE(src && dst && rows > 0 && cols > 0 && src != dst, FCL_ERR_INVALID, "transpose2d: bad arguments");
    hipLaunchKernelGGL(transpose2d_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, (hipStream_t)stream, src, dst, rows, cols);
    return check_hip(hipGetLastError(), "transpose2d");
}

int fcl_sumsq_accum(const float* x, size_t n, double* out, fcl_stream_t stream) {
    FCL_REQUIRE(x && out, FCL_ERR_INVALID, "sumsq_accum: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sumsq_kernel, dim3(std::min(grid1d((long long)(n + 3) / 4, 256), 2048)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, out);
    return check_hip(hipGetLastError(), "sumsq_accum");
}

int fcl_adam_step_wd(float* p, const float* g, float* m, float* v, size_t n, const double* gradnorm_sq, float max_norm, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int32_t* step_dev, const uint32_t* status, fcl_stream_t stream) {
    FCL_REQUIRE(p && g && m && v && gradnorm_sq && step_dev, FCL_ERR_INVALID, "adam_step: bad arguments");
    FCL_REQUIRE(weight_decay >= 0.f, FCL_ERR_INVALID, "adam_step: weight_decay must be >= 0 (torch.optim.Adam raises on a negative value)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, gradnorm_sq, max_norm, lr,
                       beta1, beta2, eps, weight_decay, step_dev, status);
    hipLaunchKernelGGL(adam_commit_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, gradnorm_sq, step_dev, status);
    return check_hip(hipGetLastError(), "adam_step");
}

int fcl_adam_step(float* p, const float* g, float* m, float* v, size_t n, const double* gradnorm_sq, float max_norm, float lr, float beta1,
                  float beta2, float eps, int32_t* step_dev, const uint32_t* status, fcl_stream_t stream) {
    return fcl_adam_step_wd(p, g, m, v, n, gradnorm_sq, max_norm, lr, beta1, beta2, eps, 0.0f, step_dev, status, stream);
}

}  // extern "C"
