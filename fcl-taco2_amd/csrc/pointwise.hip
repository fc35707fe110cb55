// pointwise.hip — HBM-bound gather / normalisation / integer kernels of the FCL-taco2 path (gfx950).
// Every kernel is coalesced along the channel dimension (channels-last rows), float4 where widths allow.
#include <algorithm>
#include "fcl_common.h"
#include "row_maps.h"

namespace fcl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- plan-time packing ----------------------------------------------------------------------------
__global__ void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                 int cout, int cin, int k) {
    const long long total = (long long)k * cout * cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        const int co = (int)((i / cin) % cout);
        const int j = (int)(i / ((long long)cin * cout));
        float v = w[((size_t)co * cin + ci) * k + j];
        if (scale) v *= scale[co];
        out[i] = v;
    }
}

__global__ void fold_bn_kernel(const float* g, const float* b, const float* mean, const float* var, float eps,
                               float* scale, float* shift, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) {
        const float s = g[i] / sqrtf(var[i] + eps);
        scale[i] = s;
        shift[i] = b[i] - mean[i] * s;
    }
}

__global__ void copy2d_kernel(float* dst, int ld_dst, const float* src, int ld_src, int rows, int cols) {
    const long long total = (long long)rows * cols;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        dst[(size_t)r * ld_dst + c] = src[(size_t)r * ld_src + c];
    }
}

__global__ void pack_frag_bf16_kernel(const float* __restrict__ w, int rows, int cols, int nsteps, unsigned short* __restrict__ hi,
                                      unsigned short* __restrict__ lo, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const size_t blk = i >> 9;
        const int step = (int)(blk % nsteps), tile = (int)(blk / nsteps);
        const int r = tile * 16 + (lane & 15), c = step * 32 + (lane >> 4) * 8 + e;
        const float x = (r < rows && c < cols) ? w[(size_t)r * cols + c] : 0.f;
        const __bf16 h = (__bf16)x;
        hi[i] = __builtin_bit_cast(unsigned short, h);
        lo[i] = __builtin_bit_cast(unsigned short, (__bf16)(x - (float)h));
    }
}

__global__ void u32_add_kernel(unsigned int* p, unsigned int v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p += v;
}

__global__ void add_vec_kernel(const float* a, const float* b, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// ---- H1 embedding / H9 row gather -------------------------------------------------------------------
// one wave per row, float4 lanes
// dst_p (optional): the gathered rows as P32 planes (ceil(c / 32) lines per row, zero past c) for the GEMM that consumes them; dst may then be null
template <typename IdxT>
__global__ void gather_rows_kernel(const float* __restrict__ src, const IdxT* __restrict__ idx, float* __restrict__ dst,
                                   int n, int c, long long src_rows, unsigned short* __restrict__ dst_p) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    const long long r = (long long)idx[wave];
    const bool ok = r >= 0 && (src_rows < 0 || r < src_rows);
    const int ldp = (c + 31) >> 5;
    if ((c & 3) == 0) {
        for (int j = lane * 4; j < ldp * 32; j += 256) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok && j < c) v = *reinterpret_cast<const f32x4*>(src + (size_t)r * c + j);
            if (dst && j < c) *reinterpret_cast<f32x4*>(dst + (size_t)wave * c + j) = v;
            if (dst_p) {
                uint2 hi, lo;
                split4(v, hi, lo);
                unsigned short* line = dst_p + ((size_t)wave * ldp + (j >> 5)) * 64 + (j & 31);
                *reinterpret_cast<uint2*>(line) = hi;
                *reinterpret_cast<uint2*>(line + 32) = lo;
            }
        }
    } else {
        for (int j = lane; j < ldp * 32; j += 64) {
            const float v = (ok && j < c) ? src[(size_t)r * c + j] : 0.f;
            if (dst && j < c) dst[(size_t)wave * c + j] = v;
            if (dst_p) store_p32(dst_p, ldp, wave, j, v);
        }
    }
}

// ---- H4/H5 channel LayerNorm (+ Linear(C->1) + masked_fill) ------------------------------------------
// one wave per row; two-pass (mean, then centred variance) like torch's CPU kernel numerics.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int MAXPER>
__global__ void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 float eps, float* __restrict__ y, unsigned short* __restrict__ yp, const float* __restrict__ lin_w,
                                 const float* __restrict__ lin_b, const uint8_t* __restrict__ pad_mask,
                                 const uint8_t* __restrict__ keep, float keep_scale, float* __restrict__ scalar, int m, int c,
                                 int ldx = 0, long long gstride = 0, int groups = 1) {
    // groups > 1 (fcl_layernorm_group_fwd): G independent LayerNorms of the same shape in one launch -- wave r handles (group g, row mi) = (r / m, r % m),
    // reads x + g * gstride + mi * ldx, its own gamma / beta / head [g][c], and writes GROUP-MAJOR outputs (row g * m + mi)
    const int grow = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (grow >= m * groups) return;
    const int g_ = groups > 1 ? grow / m : 0, mi = grow - g_ * m;
    if (groups > 1) {
        x += (long long)g_ * gstride + (long long)mi * ldx - (long long)grow * c;  // so that x[row * c + j] below addresses this row
        gamma += g_ * c; beta += g_ * c;
        if (lin_w) { lin_w += g_ * c; lin_b += g_; }
        if (pad_mask) pad_mask += mi - grow;  // the pad mask is per row of ONE group
    }
    const int row = grow;
    float v[MAXPER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) {
        const int j = lane + i * 64;
        v[i] = j < c ? x[(size_t)row * c + j] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) {
        const int j = lane + i * 64;
        const float d = j < c ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)c + eps);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) {
        const int j = lane + i * 64;
        if (j < c) {
            float o = (v[i] - mean) * rstd * gamma[j] + beta[j];
            if (keep) o = keep[(size_t)row * c + j] ? o * keep_scale : 0.f;  // training: Dropout after the LayerNorm
            if (y) y[(size_t)row * c + j] = o;
            if (yp) store_p32(yp, (c + 31) >> 5, row, j, o);
            if (lin_w) dot += o * lin_w[j];
        } else if (yp && j < ((c + 31) & ~31)) {
            store_p32(yp, (c + 31) >> 5, row, j, 0.f);  // zero padding of the last 32-column line
        }
    }
    if (lin_w) {
        dot = wave_sum(dot);
        if (lane == 0) {
            float r = dot + lin_b[0];
            if (pad_mask && pad_mask[row]) r = 0.f;
            scalar[row] = r;
        }
    }
}

// ---- H4 duration rounding (INT) -------------------------------------------------------------------
__global__ void duration_round_kernel(const float* __restrict__ x, int64_t* __restrict__ out, int n, int linear_domain,
                                      float offset, const uint8_t* __restrict__ pad_mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    if (!linear_domain) v = expf(v) - offset;
    v = rintf(v);  // round-half-to-even == torch.round
    v = fmaxf(v, 0.f);
    int64_t d = (int64_t)v;
    if (pad_mask && pad_mask[i]) d = 0;
    out[i] = d;
}

// ---- H5 variance embeds + hs + p_embs + e_embs --------------------------------------------------------
__global__ void variance_embed_add_kernel(const float* __restrict__ hs, const float* __restrict__ p,
                                          const float* __restrict__ e, const float* __restrict__ wp,
                                          const float* __restrict__ bp, const float* __restrict__ we,
                                          const float* __restrict__ be, const int* __restrict__ seg_lo,
                                          const int* __restrict__ seg_hi, float* __restrict__ out, float* __restrict__ p_emb,
                                          float* __restrict__ e_emb, int m, int c, int k) {
    const int row = blockIdx.x;
    if (row >= m) return;
    const int lo = seg_lo[row], hi = seg_hi[row], pad = (k - 1) / 2;
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        float ap = bp[ch], ae = be[ch];
        for (int j = 0; j < k; ++j) {
            const int r = row + j - pad;
            if (r >= lo && r < hi) {
                ap = fmaf(wp[ch * k + j], p[r], ap);
                ae = fmaf(we[ch * k + j], e[r], ae);
            }
        }
        const size_t o = (size_t)row * c + ch;
        if (p_emb) p_emb[o] = ap;
        if (e_emb) e_emb[o] = ae;
        if (out) out[o] = (hs[o] + ap) + ae;  // reference order: (h + p_embs) + e_embs
    }
}

// ---- speaker embedding: hs <- cat[hs, F.normalize(spemb)] (..._sa.py:555-557, 636-638) -------------------------------------------------
// out[row, 0:C] = hs[row, 0:C]; out[row, C:C+S] = spk[b] / max(||spk[b]||_2, 1e-12) with b = row / T (the padded [B, T] row layout).  One wave per
// row: the norm of its utterance's vector by a wave reduction (S <= a few hundred floats, L2-resident), then C + S contiguous outputs; outp
// (optional): the row also as P32 planes ((C + S) % 32 == 0).
__global__ __launch_bounds__(256) void concat_spk_kernel(const float* __restrict__ hs, int ldh, const float* __restrict__ spk, int C, int S, int T, int M,
                                                         float* __restrict__ out, unsigned short* __restrict__ outp) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* sv = spk + (size_t)(row / T) * S;
    float q = 0.f;
    for (int i = lane; i < S; i += 64) q = fmaf(sv[i], sv[i], q);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float inv = 1.0f / fmaxf(sqrtf(q), 1e-12f);
    const int W = C + S;
    for (int i = lane; i < W; i += 64) {
        const float v = i < C ? hs[(size_t)row * ldh + i] : sv[i - C] * inv;
        if (out) out[(size_t)row * W + i] = v;
        if (outp) store_p32(outp, W >> 5, row, i, v);
    }
}

// ---- H10 position table ---------------------------------------------------------------------------
__global__ void position_table_kernel(const int* __restrict__ dur, float* __restrict__ pos, int n, int lmax) {
    const long long total = (long long)n * lmax;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / lmax), t = (int)(i % lmax);
        const int d = dur[r];
        pos[i] = t < d ? (float)t / (float)d : 0.f;
    }
}

// ---- H12 masked loss sums ---------------------------------------------------------------------------
__global__ void masked_l1_mse_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                     const uint8_t* __restrict__ valid, int m, int c, int b_log, float off, double* out) {
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    const long long total = (long long)m * c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / c), j = (int)(i - (long long)r * c);
        if (valid && !valid[r]) continue;
        float bv = b[(size_t)r * ldb + j];
        if (b_log) bv = logf(bv + off);
        const float d = a[(size_t)r * lda + j] - bv;
        s1 += fabsf(d);
        s2 += (double)d * d;
        cnt += 1.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
        cnt += __shfl_xor(cnt, o);
    }
    // one atomic triple per workgroup: thousands of same-address fp64 atomics serialise (this was 13 % of a KD step)
    __shared__ double part[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[wave][0] = s1;
        part[wave][1] = s2;
        part[wave][2] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += part[w][threadIdx.x];
        if (v != 0.0) atomicAdd(out + threadIdx.x, v);
    }
}

static inline int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// Batched operand forms of the parameters (fcl_derive_batch): one 32x32 output tile per workgroup, any number of matrices per launch.
// out[(a*B + b), c] = src[a*sa + b*sb + c*sc] (+ src2[same]) ; fp32 and / or P32 planes out.
__global__ void __launch_bounds__(256) derive_batch_kernel(const fcl_derive_t* __restrict__ descs, int n) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n - 1;  // last descriptor whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const fcl_derive_t d = descs[lo];
    const int R = d.a * d.b, tiles_c = (d.c + 31) >> 5;
    const int t = (int)blockIdx.x - d.first_block;
    const int r0 = (t / tiles_c) << 5, c0 = (t % tiles_c) << 5;
    const int j = threadIdx.x & 31, i0 = threadIdx.x >> 5;
    const bool along_c = d.sc == 1 || d.sc == -1 || (d.sb != 1 && d.sb != -1);  // lanes follow the unit-stride direction of the SOURCE
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = i0 + 8 * p;
        const int r = along_c ? r0 + i : r0 + j, c = along_c ? c0 + j : c0 + i;
        float v = 0.f;
        if (r < R && c < d.c) {
            const long long off = (long long)(r / d.b) * d.sa + (long long)(r % d.b) * d.sb + (long long)c * d.sc;
            v = d.src[off];
            if (d.src2) v += d.src2[off];
        }
        if (along_c) tile[i][j] = v; else tile[j][i] = v;
    }
    __syncthreads();
    const int ldp = tiles_c;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = i0 + 8 * p, r = r0 + i, c = c0 + j;
        if (r >= R) continue;
        const float v = tile[i][j];  // zero past column C: plane lines are zero-padded
        if (d.dst && c < d.c) d.dst[(size_t)r * d.c + c] = v;
        if (d.dst_p) store_p32(d.dst_p, ldp, r, c, v);
    }
}

// ---- H10 on the device: row / frame maps of a batch from durations that live in HBM (fcl_row_maps_build) -----------------------------------
// Every output element is a count or a prefix sum over the N compact rows, so each thread walks the durations once (staged through LDS in
// chunks: broadcast reads): O(N) per thread, N/256 + a few workgroups -- ~10 us for a 32-utterance batch, no sort network, no atomics,
// deterministic.  Role by workgroup: rows (stable descending rank + exclusive prefix sum), live-row counts, utterance frame starts / totals.
__global__ __launch_bounds__(256) void row_maps_kernel(const fcl_row_maps_t a, int wg_rows, int wg_live) {
    __shared__ __attribute__((aligned(16))) int d_l[RM_CHUNK];
    const int wg = blockIdx.x;
    const int role = wg < wg_rows ? 0 : (wg < wg_rows + wg_live ? 1 : 2);
    const int idx = (wg - (role == 0 ? 0 : (role == 1 ? wg_rows : wg_rows + wg_live))) * 256 + threadIdx.x;
    const bool on = role == 0 ? idx < a.n : (role == 1 ? idx <= a.lmax_cap : idx <= a.b);
    int my_d = 0, lim = 0;
    if (role == 0 && on) my_d = max(rm_dur(a, idx), 0);
    if (role == 2 && on) lim = a.utt_row0 ? a.utt_row0[idx] : idx * a.t_max;
    int rank = 0, acc = 0, dmax = 0, zeros = 0;
    for (int c0 = 0; c0 < a.n; c0 += RM_CHUNK) {
        const int cn = min(RM_CHUNK, a.n - c0), cn4 = (cn + 3) & ~3;
        __syncthreads();
        for (int j = threadIdx.x; j < cn4; j += 256) d_l[j] = j < cn ? rm_dur(a, c0 + j) : -1;  // (-1: a padding tag, counted nowhere)
        __syncthreads();
        if (!on) continue;
        // four durations per 16-byte LDS read (all lanes read the same address: a broadcast), unrolled so the reads run ahead of the compares --
        // a scalar loop exposes the full LDS latency per element (98 us for 3 200 rows; this form: ~10)
        const int4* d4 = reinterpret_cast<const int4*>(d_l);
        if (role == 0) {
            const int rel = idx - c0;  // elements [0, rel) of this chunk precede the row
#pragma unroll 4
            for (int q = 0; q < cn4 / 4; ++q) {
                const int4 v = d4[q];
                const int e[4] = {max(v.x, 0), max(v.y, 0), max(v.z, 0), max(v.w, 0)};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool before = q * 4 + t < rel;
                    rank += (e[t] > my_d) | ((e[t] == my_d) & before);  // stable: ties keep row order (np.argsort(-dur, kind="stable"))
                    acc += before ? e[t] : 0;                          // exclusive prefix sum = first output frame of the row (H10)
                }
            }
        } else if (role == 1) {
#pragma unroll 4
            for (int q = 0; q < cn4 / 4; ++q) {
                const int4 v = d4[q];
                acc += (v.x > idx) + (v.y > idx) + (v.z > idx) + (v.w > idx);  // rows still alive at step idx (tags are negative)
            }
        } else {
            const int rel = lim - c0;
#pragma unroll 4
            for (int q = 0; q < cn4 / 4; ++q) {
                const int4 v = d4[q];
                const int r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int d = max(r[t], 0);
                    acc += (q * 4 + t < rel) ? d : 0;
                    dmax = max(dmax, d);
                    zeros += r[t] == 0;  // (padding rows and tags are < 0 here)
                }
            }
        }
    }
    if (!on) return;
    if (role == 0) {
        a.src_rows[rank] = a.row_src ? a.row_src[idx] : idx;
        a.dur_sorted[rank] = my_d;
        a.frame_off[rank] = acc;
        if (a.order) a.order[rank] = idx;
    } else if (role == 1) {
        a.live_rows[idx] = acc;
    } else {
        a.utt_frame0[idx] = acc;
        if (idx == a.b) {  // lim = N: acc is the total
            a.totals[0] = acc;
            a.totals[1] = dmax;
            a.totals[2] = zeros;
            a.totals[3] = 0;
            unsigned int bits = 0;
            if (zeros) bits |= FCL_STATUS_ZERO_DURATION;
            if (dmax > a.lmax_cap) bits |= FCL_STATUS_LMAX_CAP;
            if (acc > a.frames_cap) bits |= FCL_STATUS_FRAMES_CAP;
            if (bits) atomicOr(a.status, bits);
        }
    }
}

// The same maps by a COUNTING sort in one workgroup (durations are small integers): per round of 1 024 rows every wave ranks its 64 rows inside
// their duration value with wave ballots (one iteration per distinct value in the wave), the per-wave counts are scanned across the waves and
// carried across the rounds, durations get a block-wide exclusive prefix sum the same way; a second sweep adds the number of rows with a larger
// duration and scatters.  O(N) work, ~5 us for a 32-utterance batch (the form above: 98 us at 3 200 rows, 30 ns per row and thread).
__global__ __launch_bounds__(1024) void row_maps_fast_kernel(const fcl_row_maps_t a) { row_maps_block<1024>(a); }

// second launch (the totals are complete): frame -> utterance bounds, and -- on any violation -- no live rows at all
__global__ __launch_bounds__(256) void row_maps_finish_kernel(const fcl_row_maps_t a) { row_maps_finish_item(a, blockIdx.x * 256 + threadIdx.x); }

}  // namespace fcl

using namespace fcl;

extern "C" {

int fcl_pack_conv1d_weight(const float* w, const float* scale, float* out, int cout, int cin, int k, fcl_stream_t stream) {
    FCL_REQUIRE(w && out && cout > 0 && cin > 0 && k > 0, FCL_ERR_INVALID, "pack_conv1d_weight: bad arguments");
    const long long total = (long long)k * cout * cin;
    hipLaunchKernelGGL(pack_conv_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, w, scale, out, cout, cin, k);
    return check_hip(hipGetLastError(), "pack_conv1d_weight");
}

int fcl_fold_batchnorm(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                       float* scale, float* shift, int c, fcl_stream_t stream) {
    FCL_REQUIRE(gamma && beta && mean && var && scale && shift && c > 0, FCL_ERR_INVALID, "fold_batchnorm: bad arguments");
    hipLaunchKernelGGL(fold_bn_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, scale, shift, c);
    return check_hip(hipGetLastError(), "fold_batchnorm");
}

int fcl_copy2d(float* dst, int ld_dst, const float* src, int ld_src, int rows, int cols, fcl_stream_t stream) {
    FCL_REQUIRE(dst && src && rows > 0 && cols > 0 && ld_dst >= cols && ld_src >= cols, FCL_ERR_INVALID, "copy2d: bad arguments");
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((long long)rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, dst, ld_dst, src, ld_src, rows, cols);
    return check_hip(hipGetLastError(), "copy2d");
}

size_t fcl_frag_bf16_elems(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)((rows + 15) / 16) * ((cols + 31) / 32) * 512;
}

int fcl_pack_frag_bf16(const float* w, int rows, int cols, uint16_t* hi, uint16_t* lo, fcl_stream_t stream) {
    FCL_REQUIRE(w && hi && lo && rows > 0 && cols > 0, FCL_ERR_INVALID, "pack_frag_bf16: bad arguments");
    const size_t n = fcl_frag_bf16_elems(rows, cols);
    hipLaunchKernelGGL(pack_frag_bf16_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, w, rows, cols,
                       (cols + 31) / 32, hi, lo, n);
    return check_hip(hipGetLastError(), "pack_frag_bf16");
}

// Fragment-major fp32 form of W [rows, cols] for v_mfma_f32_16x16x4_f32 (the exact-fp32 feat/prenet kernel): lane (r16, kq) of (16-row tile, 32-k
// step) holds eight floats -- W[row][32 st + 4 kq + 0..3] and W[row][32 st + 16 + 4 kq + 0..3] (the two 16-byte pieces the "lines" form of the
// exact-mode GEMMs reads per row and step: MFMA e of a piece contracts k = 4 kq + e of all four lane groups) -- zero past rows / cols.
__global__ void pack_frag_f32_kernel(const float* __restrict__ w, int rows, int cols, int nsteps, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long long ts = i >> 9;
    const int st = (int)(ts % nsteps), tile = (int)(ts / nsteps);
    const int row = tile * 16 + (lane & 15), k = st * 32 + (j < 4 ? 0 : 16) + 4 * (lane >> 4) + (j & 3);
    out[i] = (row < rows && k < cols) ? w[(size_t)row * cols + k] : 0.f;
}

int fcl_pack_frag_f32(const float* w, int rows, int cols, float* out, fcl_stream_t stream) {
    FCL_REQUIRE(w && out && rows > 0 && cols > 0, FCL_ERR_INVALID, "pack_frag_f32: bad arguments");
    const size_t n = fcl_frag_bf16_elems(rows, cols);  // same element count: (rows / 16 tiles) x (cols / 32 steps) x 64 lanes x 8
    hipLaunchKernelGGL(pack_frag_f32_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, w, rows, cols, (cols + 31) / 32, out,
                       (long long)n);
    return check_hip(hipGetLastError(), "pack_frag_f32");
}

int fcl_u32_add(uint32_t* p, uint32_t v, fcl_stream_t stream) {
    FCL_REQUIRE(p, FCL_ERR_INVALID, "u32_add: null pointer");
    hipLaunchKernelGGL(u32_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, v);
    return check_hip(hipGetLastError(), "u32_add");
}

int fcl_add_vec(const float* a, const float* b, float* out, int n, fcl_stream_t stream) {
    FCL_REQUIRE(a && b && out && n > 0, FCL_ERR_INVALID, "add_vec: bad arguments");
    hipLaunchKernelGGL(add_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return check_hip(hipGetLastError(), "add_vec");
}

int fcl_embedding_fwd(const int64_t* ids, const float* table, float* out, uint16_t* out_p, int m, int v, int e, fcl_stream_t stream) {
    FCL_REQUIRE(ids && table && (out || out_p) && m >= 0 && v > 0 && e > 0, FCL_ERR_INVALID, "embedding_fwd: bad arguments");
    if (m == 0) return 0;
    FCL_REQUIRE((e & 3) != 0 || (aligned16(table) && aligned16(out)), FCL_ERR_ALIGN, "embedding_fwd: 16-byte alignment required");
    FCL_REQUIRE((reinterpret_cast<uintptr_t>(out_p) & 127u) == 0, FCL_ERR_ALIGN, "embedding_fwd: planes must be 128-byte aligned");
    hipLaunchKernelGGL((gather_rows_kernel<int64_t>), dim3((m + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, ids, out, m, e, (long long)v, out_p);
    return check_hip(hipGetLastError(), "embedding_fwd");
}

int fcl_gather_rows_fwd(const float* src, const int32_t* idx, float* dst, uint16_t* dst_p, int n, int c, fcl_stream_t stream) {
    FCL_REQUIRE(src && idx && (dst || dst_p) && n >= 0 && c > 0, FCL_ERR_INVALID, "gather_rows_fwd: bad arguments");
    if (n == 0) return 0;
    FCL_REQUIRE((c & 3) != 0 || (aligned16(src) && aligned16(dst)), FCL_ERR_ALIGN, "gather_rows_fwd: 16-byte alignment required");
    FCL_REQUIRE((reinterpret_cast<uintptr_t>(dst_p) & 127u) == 0, FCL_ERR_ALIGN, "gather_rows_fwd: planes must be 128-byte aligned");
    hipLaunchKernelGGL((gather_rows_kernel<int32_t>), dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n, c, -1LL, dst_p);
    return check_hip(hipGetLastError(), "gather_rows_fwd");
}

int fcl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, uint16_t* yp, const float* lin_w,
                      const float* lin_b, const uint8_t* pad_mask, const uint8_t* keep, float keep_scale, float* scalar, int m, int c,
                      fcl_stream_t stream) {
    FCL_REQUIRE(x && gamma && beta && m >= 0 && c > 0, FCL_ERR_INVALID, "layernorm_fwd: bad arguments");
    FCL_REQUIRE(y || yp || lin_w, FCL_ERR_INVALID, "layernorm_fwd: nothing to compute (y, yp and lin_w all NULL)");
    FCL_REQUIRE((reinterpret_cast<uintptr_t>(yp) & 127u) == 0, FCL_ERR_ALIGN, "layernorm_fwd: planes must be 128-byte aligned");
    FCL_REQUIRE(!lin_w || (lin_b && scalar), FCL_ERR_INVALID, "layernorm_fwd: lin_w needs lin_b and scalar");
    FCL_REQUIRE(c <= 1024, FCL_ERR_SHAPE, "layernorm_fwd: C=%d > 1024 unsupported", c);
    if (m == 0) return 0;
    dim3 grid((m + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (c <= 64) hipLaunchKernelGGL((layernorm_kernel<1>), grid, block, 0, s, x, gamma, beta, eps, y, yp, lin_w, lin_b, pad_mask, keep, keep_scale, scalar, m, c);
    else if (c <= 256) hipLaunchKernelGGL((layernorm_kernel<4>), grid, block, 0, s, x, gamma, beta, eps, y, yp, lin_w, lin_b, pad_mask, keep, keep_scale, scalar, m, c);
    else if (c <= 512) hipLaunchKernelGGL((layernorm_kernel<8>), grid, block, 0, s, x, gamma, beta, eps, y, yp, lin_w, lin_b, pad_mask, keep, keep_scale, scalar, m, c);
    else hipLaunchKernelGGL((layernorm_kernel<16>), grid, block, 0, s, x, gamma, beta, eps, y, yp, lin_w, lin_b, pad_mask, keep, keep_scale, scalar, m, c);
    return check_hip(hipGetLastError(), "layernorm_fwd");
}

int fcl_layernorm_group_fwd(const float* x, int ldx, int64_t x_group_stride, const float* gamma, const float* beta, float eps, float* y, uint16_t* yp,
                            const float* lin_w, const float* lin_b, const uint8_t* pad_mask, float* scalar, int m, int c, int groups,
                            fcl_stream_t stream) {
    FCL_REQUIRE(x && gamma && beta && m >= 0 && c > 0 && groups >= 1 && ldx >= c, FCL_ERR_INVALID, "layernorm_group_fwd: bad arguments");
    FCL_REQUIRE(y || yp || lin_w, FCL_ERR_INVALID, "layernorm_group_fwd: nothing to compute (y, yp and lin_w all NULL)");
    FCL_REQUIRE((reinterpret_cast<uintptr_t>(yp) & 127u) == 0, FCL_ERR_ALIGN, "layernorm_group_fwd: planes must be 128-byte aligned");
    FCL_REQUIRE(!lin_w || (lin_b && scalar), FCL_ERR_INVALID, "layernorm_group_fwd: lin_w needs lin_b and scalar");
    FCL_REQUIRE(c <= 1024, FCL_ERR_SHAPE, "layernorm_group_fwd: C=%d > 1024 unsupported", c);
    if (m == 0) return 0;
    dim3 grid(((long long)m * groups + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    const uint8_t* keep = nullptr;
#define FCL_LNG(P) hipLaunchKernelGGL((layernorm_kernel<P>), grid, block, 0, s, x, gamma, beta, eps, y, yp, lin_w, lin_b, pad_mask, keep, 1.0f, scalar, m, c, ldx, \
                                      (long long)x_group_stride, groups)
    if (c <= 64) FCL_LNG(1);
    else if (c <= 256) FCL_LNG(4);
    else if (c <= 512) FCL_LNG(8);
    else FCL_LNG(16);
#undef FCL_LNG
    return check_hip(hipGetLastError(), "layernorm_group_fwd");
}

int fcl_duration_round_fwd(const float* x, int64_t* out, int n, int linear_domain, float offset, const uint8_t* pad_mask,
                           fcl_stream_t stream) {
    FCL_REQUIRE(x && out && n >= 0, FCL_ERR_INVALID, "duration_round_fwd: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(duration_round_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, n, linear_domain, offset, pad_mask);
    return check_hip(hipGetLastError(), "duration_round_fwd");
}

int fcl_variance_embed_add_fwd(const float* hs, const float* p, const float* e, const float* wp, const float* bp,
                               const float* we, const float* be, const int32_t* seg_lo, const int32_t* seg_hi, float* out,
                               float* p_emb, float* e_emb, int m, int c, int k, fcl_stream_t stream) {
    FCL_REQUIRE(p && e && wp && bp && we && be && seg_lo && seg_hi && m >= 0 && c > 0 && k > 0 && (k & 1), FCL_ERR_INVALID,
                "variance_embed_add_fwd: bad arguments");
    FCL_REQUIRE(out || p_emb || e_emb, FCL_ERR_INVALID, "variance_embed_add_fwd: no output");
    FCL_REQUIRE(!out || hs, FCL_ERR_INVALID, "variance_embed_add_fwd: out needs hs");
    if (m == 0) return 0;
    hipLaunchKernelGGL(variance_embed_add_kernel, dim3(m), dim3(c >= 256 ? 256 : 64), 0, (hipStream_t)stream, hs, p, e, wp, bp, we, be,
                       seg_lo, seg_hi, out, p_emb, e_emb, m, c, k);
    return check_hip(hipGetLastError(), "variance_embed_add_fwd");
}

int fcl_masked_l1_mse_fwd(const float* a, int lda, const float* b, int ldb, const uint8_t* row_valid, int m, int c, int b_log,
                          float b_log_offset, double* out, fcl_stream_t stream) {
    FCL_REQUIRE(a && b && out && m >= 0 && c > 0 && lda >= c && ldb >= c, FCL_ERR_INVALID, "masked_l1_mse_fwd: bad arguments");
    if (m == 0) return 0;
    hipLaunchKernelGGL(masked_l1_mse_kernel, dim3(std::min(grid_for((long long)m * c, 256 * 8), 512)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                       row_valid, m, c, b_log, b_log_offset, out);
    return check_hip(hipGetLastError(), "masked_l1_mse_fwd");
}

int fcl_concat_spk_fwd(const float* hs, int ldh, const float* spk, float* out, uint16_t* out_p, int m, int c, int s, int t, fcl_stream_t stream) {
    FCL_REQUIRE(hs && spk && (out || out_p) && m >= 0 && c > 0 && s > 0 && t > 0 && ldh >= c, FCL_ERR_INVALID, "concat_spk_fwd: bad arguments");
    FCL_REQUIRE(!out_p || (((c + s) & 31) == 0 && (reinterpret_cast<uintptr_t>(out_p) & 127u) == 0), FCL_ERR_SHAPE,
                "concat_spk_fwd: planes need (C + S) %% 32 == 0 and a 128-byte aligned buffer");
    if (m == 0) return 0;
    hipLaunchKernelGGL(concat_spk_kernel, dim3((m + 3) / 4), dim3(256), 0, (hipStream_t)stream, hs, ldh, spk, c, s, t, m, out, out_p);
    return check_hip(hipGetLastError(), "concat_spk_fwd");
}

int fcl_position_table_fwd(const int32_t* dur, float* pos, int n, int lmax, fcl_stream_t stream) {
    FCL_REQUIRE(dur && pos && n >= 0 && lmax > 0, FCL_ERR_INVALID, "position_table_fwd: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(position_table_kernel, dim3(grid_for((long long)n * lmax, 256)), dim3(256), 0, (hipStream_t)stream, dur, pos, n, lmax);
    return check_hip(hipGetLastError(), "position_table_fwd");
}

int fcl_derive_blocks(int a, int b, int c) {
    if (a <= 0 || b <= 0 || c <= 0) return 0;
    const long long n = (((long long)a * b + 31) / 32) * ((c + 31) / 32);
    return n > 0x7fffffffLL ? -1 : (int)n;
}

int fcl_derive_batch(const fcl_derive_t* descs_dev, int n, int total_blocks, fcl_stream_t stream) {
    FCL_REQUIRE(n >= 0 && total_blocks >= 0 && (n == 0 || descs_dev), FCL_ERR_INVALID, "derive_batch: bad arguments");
    if (n == 0 || total_blocks == 0) return 0;
    hipLaunchKernelGGL(derive_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n);
    return check_hip(hipGetLastError(), "derive_batch");
}

}  // extern "C"

namespace fcl {
// argument check of fcl_row_maps_build, shared with the BiLSTM launch that builds the maps in an extra workgroup (bilstm.hip);
// *fusable: the one-workgroup counting-sort form applies (scratch given, lmax_cap below its value buckets)
int row_maps_check(const fcl_row_maps_t* a, bool* fusable) {
    FCL_REQUIRE(a, FCL_ERR_INVALID, "row_maps_build: null argument");
    FCL_REQUIRE(a->b > 0 && a->n > 0 && a->lmax_cap > 0 && a->frames_cap > 0, FCL_ERR_SHAPE, "row_maps_build: bad sizes B=%d N=%d lmax_cap=%d frames_cap=%d",
                a->b, a->n, a->lmax_cap, a->frames_cap);
    FCL_REQUIRE(a->n <= 32768 && a->lmax_cap < RM_DMAX, FCL_ERR_SHAPE, "row_maps_build: at most 32768 rows and %d steps", RM_DMAX - 1);
    FCL_REQUIRE((a->dur_i64 != nullptr) != (a->dur_i32 != nullptr), FCL_ERR_INVALID, "row_maps_build: exactly one of dur_i64 / dur_i32");
    FCL_REQUIRE(a->utt_row0 || (a->t_max > 0 && a->n == a->b * a->t_max), FCL_ERR_SHAPE, "row_maps_build: without utt_row0 the rows are the padded [B, t_max] layout");
    FCL_REQUIRE(a->src_rows && a->dur_sorted && a->frame_off && a->live_rows && a->utt_frame0 && a->frame_lo && a->frame_hi &&
                    a->totals && a->status,
                FCL_ERR_INVALID, "row_maps_build: null pointer");
    static const int fast_on = tunable("ROWMAPS_FAST", 1);
    if (fusable) *fusable = fast_on && a->scratch && a->lmax_cap < RMF_V - 1;
    return 0;
}
}  // namespace fcl

extern "C" {

int fcl_row_maps_build(const fcl_row_maps_t* a, fcl_stream_t stream) {
    bool fast = false;
    {
        const int rc = row_maps_check(a, &fast);
        if (rc) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    if (fast) {
        hipLaunchKernelGGL(row_maps_fast_kernel, dim3(1), dim3(1024), 0, s, *a);
    } else {
        const int wg_rows = (a->n + 255) / 256, wg_live = (a->lmax_cap + 1 + 255) / 256, wg_utt = (a->b + 1 + 255) / 256;
        hipLaunchKernelGGL(row_maps_kernel, dim3(wg_rows + wg_live + wg_utt), dim3(256), 0, s, *a, wg_rows, wg_live);
    }
    FCL_HIP(hipGetLastError());
    const int items = std::max(a->frames_cap, a->lmax_cap + 1);
    hipLaunchKernelGGL(row_maps_finish_kernel, dim3((items + 255) / 256), dim3(256), 0, s, *a);
    return check_hip(hipGetLastError(), "row_maps_build");
}

}  // extern "C"
