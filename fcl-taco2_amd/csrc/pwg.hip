// pwg.hip — Parallel WaveGAN generator (BASELINE configs[4]: the stage the reference shells out to,
// inference_student.sh:20-23).  gfx950 only.  Rows are SAMPLES (time-major, utterances concatenated, zero padding at utterance edges through
// the GEMM's segment bounds); channels are the contiguous dimension, so every convolution of the residual stack is a K-term of the pre-split
// operand GEMM (gemm_planes.hip):
//     z[M, 128]  = x[t-d] W_0^T + x[t] W_1^T + x[t+d] W_2^T + c_up[t] W_aux^T + b          (4 terms: K = 3 x 64 + 80)
//     g[M, 64]   = tanh(z[:, :64]) * sigmoid(z[:, 64:])                                     (pwg_gate_kernel, writes planes)
//     o[M, 128]  = g [W_out ; W_skip]^T + [b_out ; b_skip]                                  (1 term: K = 64)
//     x          = (o[:, :64] + x) * sqrt(0.5) ;  skips += o[:, 64:]                        (pwg_resid_kernel, writes fp32 + planes)
// The upsampling network (nearest-neighbour stretch + 1 x (2s+1) smoothing, four times) and the 1 -> 64 / 64 -> 1 end convolutions are
// HBM-bound row kernels.  Published architecture: kan-bayashi/ParallelWaveGAN, ParallelWaveGANGenerator (v1, LJSpeech); see oracle/pwg_oracle.py.
#include <algorithm>

#include "fcl_common.h"

namespace fcl {

typedef unsigned short u16;

// ---- one stage of UpsampleNetwork: Stretch2d(scale, nearest) + Conv2d(1, 1, (1, 2*scale+1), padding (0, scale), no bias), per channel.
// in: [frames * rate_in rows, C]; out: [frames * rate_in * scale rows, C] (fp32 and / or planes with ldp lines per row, zero past C).
// A row's utterance comes from its frame: frame_utt[frame], bounds utt_off[u] .. utt_off[u+1] (in frames): the zero padding of the smoothing
// convolution applies at UTTERANCE edges, as when each utterance is upsampled on its own.
__global__ __launch_bounds__(256) void pwg_upsample_stage_kernel(const float* __restrict__ in, const int* __restrict__ frame_utt,
                                                                 const int* __restrict__ utt_off, long long rows_out, int rate_in, int scale,
                                                                 const float* __restrict__ w, float* __restrict__ out, u16* __restrict__ out_p, int ldp,
                                                                 int C, int chunk_major) {
    // item = (output row, 4 channels).  The 2*scale+1 taps of a row fall on at most three rows of the stage input (nearest-neighbour stretch), so
    // they collapse to three coefficients per row (sums of the taps that land on each input row and lie inside the utterance: the zero padding
    // of the smoothing convolution); then 3 float4 loads and 12 FMAs per item.  C % 4 == 0.
    const int cq = out_p ? ldp * 8 : C >> 2;  // channel quads per row (planes: the zero padding up to the last 32-column line is written too)
    const long long total = rows_out * cq;
    const int rate_out = rate_in * scale;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cq;
        const int c = (int)(i - r * cq) * 4;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            const int u = frame_utt[r / rate_out];
            const long long lo = (long long)utt_off[u] * rate_out, hi = (long long)utt_off[u + 1] * rate_out;
            const long long i0 = r / scale;  // input row of the centre tap
            float k[3] = {0.f, 0.f, 0.f};    // coefficients of input rows i0 - 1, i0, i0 + 1
            for (int j = -scale; j <= scale; ++j) {
                const long long q = r + j;
                if (q < lo || q >= hi) continue;
                k[(int)(q / scale - i0) + 1] += w[j + scale];
            }
            const long long rows_in = rows_out / scale;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const long long ri = i0 + d - 1;
                if (k[d] != 0.f && ri >= 0 && ri < rows_in) {
                    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(in + ri * C + c);
                    acc += k[d] * v;
                }
            }
            if (out) *reinterpret_cast<f32x4_t*>(out + r * C + c) = acc;
        }
        if (out_p) {
            uint2 h, l;
            split4(acc, h, l);
            u16* line = chunk_major ? out_p + ((size_t)(c >> 5) * rows_out + r) * 64 + (c & 31) : out_p + ((size_t)r * ldp + (c >> 5)) * 64 + (c & 31);
            *reinterpret_cast<uint2*>(line) = h;
            *reinterpret_cast<uint2*>(line + 32) = l;
        }
    }
}

// first_conv: x[m, ch] = w[ch] * z[m] + b[ch]  (Conv1d1x1(1 -> R))
__global__ __launch_bounds__(256) void pwg_first_conv_kernel(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ b,
                                                             float* __restrict__ x, u16* __restrict__ xp, long long M, int R, int chunk_major) {
    const int rq = R >> 2;  // item = (sample, 4 channels): one 8-byte store per plane
    const long long total = M * rq;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / rq;
        const int ch = (int)(i - m * rq) * 4;
        const float zm = z[m];
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(w + ch) * zm + *reinterpret_cast<const f32x4_t*>(b + ch);
        if (x) *reinterpret_cast<f32x4_t*>(x + m * R + ch) = v;
        uint2 h, l;
        split4(v, h, l);
        u16* line = chunk_major ? xp + ((size_t)(ch >> 5) * M + m) * 64 + (ch & 31) : xp + ((size_t)m * (R >> 5) + (ch >> 5)) * 64 + (ch & 31);
        *reinterpret_cast<uint2*>(line) = h;
        *reinterpret_cast<uint2*>(line + 32) = l;
    }
}

// g = tanh(z[:, :H]) * sigmoid(z[:, H:])  -> planes [M, H]
__global__ __launch_bounds__(256) void pwg_gate_kernel(const float* __restrict__ z, u16* __restrict__ gp, long long M, int H) {
    const long long total = M * H;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / H;
        const int ch = (int)(i - m * H);
        const float a = z[m * 2 * H + ch], b = z[m * 2 * H + H + ch];
        const float v = tanhf(a) * (1.0f / (1.0f + __expf(-b)));
        store_p32(gp + (size_t)m * (H >> 5) * 64, H >> 5, 0, ch, v);
    }
}

// x = (o[:, :R] + x) * sqrt(0.5) (fp32 + planes) ; skips += o[:, R:]
__global__ __launch_bounds__(256) void pwg_resid_kernel(const float* __restrict__ o, float* __restrict__ x, u16* __restrict__ xp, float* __restrict__ skips,
                                                        long long M, int R, int first) {
    const long long total = M * R;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / R;
        const int ch = (int)(i - m * R);
        const float v = (o[m * 2 * R + ch] + x[i]) * 0.70710678118654752440f;
        x[i] = v;
        store_p32(xp + (size_t)m * (R >> 5) * 64, R >> 5, 0, ch, v);
        const float s = o[m * 2 * R + R + ch];
        skips[i] = first ? s : skips[i] + s;
    }
}

// y = relu(skips * scale) -> planes [M, S]
__global__ __launch_bounds__(256) void pwg_relu_scale_kernel(const float* __restrict__ skips, float scale, u16* __restrict__ yp, long long M, int S) {
    const long long total = M * S;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / S;
        const int ch = (int)(i - m * S);
        store_p32(yp + (size_t)m * (S >> 5) * 64, S >> 5, 0, ch, fmaxf(skips[i] * scale, 0.f));
    }
}

// wav[m] = sum_ch relu(h[m, ch]) * w[ch] + b   (last ReLU + Conv1d1x1(S -> 1)); one wave per 4 rows of S = 64 channels
__global__ __launch_bounds__(256) void pwg_out_kernel(const float* __restrict__ h, const float* __restrict__ w, float b, float* __restrict__ wav, long long M,
                                                      int S) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long m = wave; m < M; m += nwaves) {
        float acc = 0.f;
        for (int ch = lane; ch < S; ch += 64) acc += fmaxf(h[m * S + ch], 0.f) * w[ch];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) wav[m] = acc + b;
    }
}

// coefficient lines of the frame-rate auxiliary term (include/fcl_hip.h: fcl_pwg_aux_coeff): one 128-byte line per sample
__global__ __launch_bounds__(256) void pwg_aux_coeff_kernel(const float* __restrict__ kc, long long M, int hop, long long frames, u16* __restrict__ kp) {
    for (long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
        const long long f = m / hop;
        const int r = (int)(f & 31);
        const long long w0 = (r >= 2 && r <= 29) ? (f >> 5) * 32 : ((f + 16) >> 5) * 32 - 16;
        const f32x4_t k03 = *reinterpret_cast<const f32x4_t*>(kc + m * 8);
        const float k4 = kc[m * 8 + 4];
        const float kv[5] = {k03[0], k03[1], k03[2], k03[3], k4};
        unsigned hw[16], lw[16];  // 32 columns, two bf16 per word
#pragma unroll
        for (int i = 0; i < 16; ++i) hw[i] = lw[i] = 0u;
        const int p0 = (int)(f - 2 - w0);  // column of frame f - 2 (0 .. 27)
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const long long g = f - 2 + d;
            if (g < 0 || g >= frames) continue;
            float v = 0.f;
            const int c = (int)(g % 5);
#pragma unroll
            for (int e = 0; e < 5; ++e) v = c == e ? kv[e] : v;
            const __bf16 h = (__bf16)v;
            const unsigned hb = __builtin_bit_cast(unsigned short, h), lb = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
            const int col = p0 + d;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i == (col >> 1)) {
                    hw[i] |= hb << ((col & 1) * 16);
                    lw[i] |= lb << ((col & 1) * 16);
                }
        }
        uint4* line = reinterpret_cast<uint4*>(kp + (size_t)m * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            line[i] = make_uint4(hw[4 * i], hw[4 * i + 1], hw[4 * i + 2], hw[4 * i + 3]);
            line[4 + i] = make_uint4(lw[4 * i], lw[4 * i + 1], lw[4 * i + 2], lw[4 * i + 3]);
        }
    }
}

// z ~ N(0, 1): Box-Muller on two hashed 24-bit uniforms per element (counter-based: element i of a given seed is reproducible on any grid)
__global__ __launch_bounds__(256) void pwg_noise_kernel(float* __restrict__ z, long long n, unsigned int seed) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned int lo = (unsigned int)i, hi = (unsigned int)(i >> 32);
        const unsigned int h1 = hash_u32(lo ^ hash_u32(seed ^ (hi * 0x9E3779B9u))), h2 = hash_u32(h1 ^ 0x85EBCA6Bu ^ lo);
        const float u1 = ((h1 >> 8) + 1) * (1.0f / 16777216.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);  // u1 in (0, 1]
        z[i] = sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
    }
}

static unsigned grid_1d(long long n, int per_block) {
    long long b = (n + per_block - 1) / per_block;
    return (unsigned)std::min<long long>(std::max<long long>(b, 1), 1 << 20);
}

}  // namespace fcl

using namespace fcl;

extern "C" {

int fcl_pwg_upsample_stage(const float* in, const int32_t* frame_utt, const int32_t* utt_off, int64_t frames, int rate_in, int scale, const float* w,
                           float* out, uint16_t* out_p, int c, int chunk_major, fcl_stream_t stream) {
    FCL_REQUIRE(in && frame_utt && utt_off && w && (out || out_p) && frames > 0 && rate_in >= 1 && scale >= 1 && c > 0 && (c & 3) == 0, FCL_ERR_INVALID,
                "pwg_upsample_stage: bad arguments (channels must be a multiple of 4)");
    FCL_REQUIRE((reinterpret_cast<uintptr_t>(out_p) & 127u) == 0, FCL_ERR_ALIGN, "pwg_upsample_stage: planes must be 128-byte aligned");
    FCL_REQUIRE(aligned16(in) && aligned16(out), FCL_ERR_ALIGN, "pwg_upsample_stage: in / out must be 16-byte aligned");
    const long long rows_out = (long long)frames * rate_in * scale;
    const int ldp = (c + 31) / 32;
    hipLaunchKernelGGL(pwg_upsample_stage_kernel, dim3(grid_1d(rows_out * (out_p ? ldp * 8 : c / 4), 256)), dim3(256), 0, (hipStream_t)stream, in, frame_utt, utt_off,
                       rows_out, rate_in, scale, w, out, out_p, ldp, c, chunk_major);
    return check_hip(hipGetLastError(), "pwg_upsample_stage");
}

int fcl_pwg_aux_coeff(const float* kc, int64_t m, int hop, int64_t frames, uint16_t* kp, fcl_stream_t stream) {
    FCL_REQUIRE(kc && kp && m > 0 && hop > 0 && frames > 0 && m <= frames * (int64_t)hop, FCL_ERR_INVALID, "pwg_aux_coeff: bad arguments");
    FCL_REQUIRE(aligned16(kc) && (reinterpret_cast<uintptr_t>(kp) & 127u) == 0, FCL_ERR_ALIGN, "pwg_aux_coeff: kc must be 16-byte, kp 128-byte aligned");
    hipLaunchKernelGGL(pwg_aux_coeff_kernel, dim3(grid_1d(m, 256)), dim3(256), 0, (hipStream_t)stream, kc, (long long)m, hop, (long long)frames, kp);
    return check_hip(hipGetLastError(), "pwg_aux_coeff");
}

int fcl_pwg_noise(float* z, int64_t n, uint32_t seed, fcl_stream_t stream) {
    FCL_REQUIRE(z && n > 0, FCL_ERR_INVALID, "pwg_noise: bad arguments");
    hipLaunchKernelGGL(pwg_noise_kernel, dim3(grid_1d(n, 1024)), dim3(256), 0, (hipStream_t)stream, z, (long long)n, seed);
    return check_hip(hipGetLastError(), "pwg_noise");
}

int fcl_pwg_first_conv(const float* z, const float* w, const float* b, float* x, uint16_t* xp, int64_t m, int r, int chunk_major, fcl_stream_t stream) {
    FCL_REQUIRE(z && w && b && xp && m > 0 && r > 0 && (r & 31) == 0, FCL_ERR_INVALID, "pwg_first_conv: bad arguments (R must be a multiple of 32)");
    FCL_REQUIRE(aligned16(w) && aligned16(b) && aligned16(x) && (reinterpret_cast<uintptr_t>(xp) & 127u) == 0, FCL_ERR_ALIGN,
                "pwg_first_conv: w / b / x must be 16-byte aligned, the planes 128-byte aligned");
    hipLaunchKernelGGL(pwg_first_conv_kernel, dim3(grid_1d(m * (r / 4), 256)), dim3(256), 0, (hipStream_t)stream, z, w, b, x, xp, (long long)m, r, chunk_major);
    return check_hip(hipGetLastError(), "pwg_first_conv");
}

int fcl_pwg_layer_fwd(const fcl_pwg_layer_t* a, fcl_stream_t stream) {
    FCL_REQUIRE(a && a->m > 0 && (a->x || a->xp_out) && a->xp && (a->kp || (a->cp && a->w_aux_p)) && a->w_conv_p && a->b_conv && a->w_os_p && a->b_os && a->skips &&
                    a->seg_lo && a->seg_hi,
                FCL_ERR_INVALID, "pwg_layer_fwd: null argument");
    if (a->kp) {
        FCL_REQUIRE(a->xp_out && a->pt_a && a->pt_b && a->ld_pt > 0 && a->hop > 0 && a->hop % 128 == 0, FCL_ERR_INVALID,
                    "pwg_layer_fwd: the frame-rate auxiliary term needs the one-launch form (xp_out), pt_a / pt_b / ld_pt and a hop that is a multiple of 128");
        FCL_REQUIRE(((reinterpret_cast<uintptr_t>(a->kp) | reinterpret_cast<uintptr_t>(a->pt_a) | reinterpret_cast<uintptr_t>(a->pt_b)) & 127u) == 0, FCL_ERR_ALIGN,
                    "pwg_layer_fwd: kp / pt_a / pt_b must be 128-byte aligned");
        FCL_REQUIRE((a->m + a->hop - 1) / a->hop + 16 <= (int64_t)a->ld_pt * 32, FCL_ERR_SHAPE, "pwg_layer_fwd: ld_pt does not cover the frames");
    }
    FCL_REQUIRE(a->r > 0 && (a->r & 31) == 0 && a->aux > 0 && a->dilation >= 1 && a->ksize >= 1 && (a->ksize & 1) && a->ksize + 1 <= FCL_MAX_TERMS, FCL_ERR_SHAPE,
                "pwg_layer_fwd: residual channels must be a multiple of 32, kernel size odd and < %d", FCL_MAX_TERMS);
    FCL_REQUIRE(a->m <= 0x7fffffffLL, FCL_ERR_SHAPE, "pwg_layer_fwd: more than 2^31 samples in one call");
    if (a->xp_out) {
        FCL_REQUIRE(a->r == 64 && a->ksize == 3 && a->aux <= 96 && a->xp_out != a->xp, FCL_ERR_SHAPE,
                    "pwg_layer_fwd: the one-launch block is built for r = 64, ksize = 3, aux <= 96 and needs xp_out != xp");
        return launch_pwg_layer_fused(*a, (hipStream_t)stream);
    }
    FCL_REQUIRE(a->z && a->gp && a->o && a->x, FCL_ERR_WORKSPACE, "pwg_layer_fwd: the unfused path needs x and the z / g / o workspaces");
    hipStream_t s = (hipStream_t)stream;
    const int R = a->r, G = 2 * R, M = (int)a->m;
    const int ldx = R / 32, ldc = (a->aux + 31) / 32;
    GemmArgs g = {};
    for (int j = 0; j < a->ksize; ++j) {
        g.term[j].K = R;
        g.term[j].shift = (j - (a->ksize - 1) / 2) * a->dilation;
        g.term[j].Ap = a->xp; g.term[j].lda_p = ldx;
        g.term[j].Wp = a->w_conv_p + (size_t)j * G * ldx * 64; g.term[j].ldw_p = ldx;
    }
    g.term[a->ksize].K = a->aux;
    g.term[a->ksize].Ap = a->cp; g.term[a->ksize].lda_p = ldc;
    g.term[a->ksize].Wp = a->w_aux_p; g.term[a->ksize].ldw_p = ldc;
    g.nterms = a->ksize + 1;
    g.M = M; g.N = G;
    g.seg_lo = a->seg_lo; g.seg_hi = a->seg_hi;
    g.bias = a->b_conv;
    g.Y = a->z; g.ldy = G;
    int rc = launch_gemm(g, s);
    if (rc) return rc;
    hipLaunchKernelGGL(pwg_gate_kernel, dim3(grid_1d((long long)M * R, 1024)), dim3(256), 0, s, a->z, a->gp, (long long)M, R);
    GemmArgs h = {};
    h.term[0].K = R;
    h.term[0].Ap = a->gp; h.term[0].lda_p = ldx;
    h.term[0].Wp = a->w_os_p; h.term[0].ldw_p = ldx;
    h.nterms = 1;
    h.M = M; h.N = G;
    h.bias = a->b_os;
    h.Y = a->o; h.ldy = G;
    rc = launch_gemm(h, s);
    if (rc) return rc;
    hipLaunchKernelGGL(pwg_resid_kernel, dim3(grid_1d((long long)M * R, 1024)), dim3(256), 0, s, a->o, a->x, a->xp, a->skips, (long long)M, R, a->first_layer);
    return check_hip(hipGetLastError(), "pwg_layer_fwd");
}

int fcl_pwg_last_fwd(const float* skips, float scale, const uint16_t* w1p, const float* b1, const float* w2, float b2, uint16_t* yp, float* h, float* wav,
                     int64_t m, int s_ch, fcl_stream_t stream) {
    FCL_REQUIRE(skips && w1p && b1 && w2 && wav && m > 0 && m <= 0x7fffffffLL && s_ch > 0 && (s_ch & 31) == 0, FCL_ERR_INVALID,
                "pwg_last_fwd: bad arguments (skip channels must be a multiple of 32)");
    hipStream_t s = (hipStream_t)stream;
    static const int fused = tunable("PWG_LAST_FUSED", 1);
    if (s_ch == 64 && fused && aligned16(skips) && (reinterpret_cast<uintptr_t>(w1p) & 15u) == 0)  // one launch, skips read once; no workspaces
        return launch_pwg_last_fused(skips, scale, w1p, b1, w2, b2, wav, m, s);
    FCL_REQUIRE(yp && h, FCL_ERR_WORKSPACE, "pwg_last_fwd: this channel count needs the yp / h workspaces");
    hipLaunchKernelGGL(pwg_relu_scale_kernel, dim3(grid_1d(m * s_ch, 1024)), dim3(256), 0, s, skips, scale, yp, (long long)m, s_ch);
    GemmArgs g = {};
    g.term[0].K = s_ch;
    g.term[0].Ap = yp; g.term[0].lda_p = s_ch / 32;
    g.term[0].Wp = w1p; g.term[0].ldw_p = s_ch / 32;
    g.nterms = 1;
    g.M = (int)m; g.N = s_ch;
    g.bias = b1;
    g.Y = h; g.ldy = s_ch;
    int rc = launch_gemm(g, s);
    if (rc) return rc;
    hipLaunchKernelGGL(pwg_out_kernel, dim3(grid_1d(m, 16)), dim3(256), 0, s, h, w2, b2, wav, (long long)m, s_ch);
    return check_hip(hipGetLastError(), "pwg_last_fwd");
}

}  // extern "C"
