// bilstm.hip — H3: the encoder's 1-layer bidirectional LSTM over packed sequences (gfx950).
//
// Step 1 (both algorithms): input projections for every time step at once, one MFMA GEMM per direction:
//     Gx_d[b*T + t, :] = x[b*T + t, :] . W_ih_d^T + (b_ih_d + b_hh_d)                    [B*T, 4H]
// Step 2, the recurrence, is latency-bound (T dependent steps of a [4H x H] mat-vec per utterance):
//   algo 2 "persistent": one workgroup per (utterance, direction); thread j keeps row j of W_hh in
//          REGISTERS for the whole sequence (H floats/thread; H=128 -> 256 KB per workgroup, the fused
//          mat-vec with persistent weights of the north star), h is broadcast from LDS, the cell update
//          runs on the first H threads; 2 barriers per step, no global traffic but Gx in / h out.
//   algo 1 "steps": one fused LSTM-step GEMM launch (gemm_f32.hip) per time step and direction, rows =
//          utterances, packed-sequence semantics via row_len.  Any H; also the cross-check of algo 2.
#include <map>
#include <mutex>

#include "fcl_common.h"
#include "row_maps.h"
#include "lstm_epilogue.h"

namespace fcl {

typedef float f32x4 __attribute__((ext_vector_type(4)));


// MAPS: the launch carries one more workgroup column (blockIdx.x == B): its first workgroup builds the row / frame maps of the batch
// (row_maps.h: forced durations depend on nothing the encoder computes) beside the 2 B recurrences -- ~25 us of a 100 us launch that leaves
// three quarters of the CUs idle -- instead of two launches on the pass's dependent chain.
template <int H, bool SAVE = false, bool MAPS = false>
__global__ __launch_bounds__(4 * H) void bilstm_persistent_kernel(const float* __restrict__ gx_f, const float* __restrict__ gx_r,
                                                                   const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                                   const int* __restrict__ lens, float* __restrict__ out, int T, BilstmSave sv = BilstmSave(),
                                                                   unsigned short* __restrict__ out_p = nullptr /* optional P32 planes of out */,
                                                                   fcl_row_maps_t mp = fcl_row_maps_t()) {
    if constexpr (MAPS) {
        if (blockIdx.x == gridDim.x - 1) {
            if (blockIdx.y == 0) {
                row_maps_block<4 * H>(mp);
                row_maps_finish_block<4 * H>(mp);
            }
            return;
        }
    }
    __shared__ __attribute__((aligned(16))) float h_s[H];
    __shared__ float g_s[4 * H];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const float* gx = (dir ? gx_r : gx_f) + (size_t)b * T * (4 * H);
    const float* whh = dir ? whh_r : whh_f;
    const int len = lens[b];

    float w[H];
#pragma unroll
    for (int k = 0; k < H; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(whh + (size_t)j * H + k);
        w[k] = v[0]; w[k + 1] = v[1]; w[k + 2] = v[2]; w[k + 3] = v[3];
    }
    if (j < H) h_s[j] = 0.f;
    float c = 0.f;
    // zero the padded tail of this direction's half of the output rows
    for (int t = len + (j / H); t < T; t += 4) {
        out[((size_t)b * T + t) * (2 * H) + dir * H + (j % H)] = 0.f;
        if (out_p) store_p32(out_p, (2 * H + 31) >> 5, b * T + t, dir * H + (j % H), 0.f);
    }
    __syncthreads();

    int t = dir ? len - 1 : 0;
    const int dt = dir ? -1 : 1;
    float gnext = len > 0 ? gx[(size_t)t * (4 * H) + j] : 0.f;
    for (int s = 0; s < len; ++s, t += dt) {
        const float gcur = gnext;
        if (s + 1 < len) gnext = gx[(size_t)(t + dt) * (4 * H) + j];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(&h_s[k]);  // same address in every lane: LDS broadcast
            a0 = fmaf(w[k], hv[0], a0);
            a1 = fmaf(w[k + 1], hv[1], a1);
            a2 = fmaf(w[k + 2], hv[2], a2);
            a3 = fmaf(w[k + 3], hv[3], a3);
        }
        g_s[j] = gcur + ((a0 + a1) + (a2 + a3));
        __syncthreads();
        if (j < H) {
            const float ig = sigmoid_f(g_s[j]), fg = sigmoid_f(g_s[H + j]), gg = tanh_f(g_s[2 * H + j]), og = sigmoid_f(g_s[3 * H + j]);
            if (SAVE) {
                const size_t cell = (size_t)t * sv.B + b;
                float* sg = sv.gates[dir] + cell * (4 * H);
                sg[j] = ig; sg[H + j] = fg; sg[2 * H + j] = gg; sg[3 * H + j] = og;
                sv.c_old[dir][cell * H + j] = c;
                sv.h_old[dir][cell * H + j] = h_s[j];
            }
            c = fg * c + ig * gg;
            if (SAVE) sv.c_new[dir][((size_t)t * sv.B + b) * H + j] = c;
            const float h = og * tanh_f(c);
            h_s[j] = h;
            out[((size_t)b * T + t) * (2 * H) + dir * H + j] = h;
            if (out_p) store_p32(out_p, (2 * H + 31) >> 5, b * T + t, dir * H + j, h);
        }
        __syncthreads();
    }
}

// ---- H = 128, round 5: the same recurrence with the contraction split over the LANES instead of broadcast to every row.
// The row-per-thread kernel above reads all H values of h per thread and step: 32 broadcast `ds_read_b128` per wave = 256 per workgroup, ~2 000 LDS
// cycles of its ~2 500-cycle step (the FMAs are 64 v_pk_fma_f32 per wave).  Here a lane owns an [8 rows x 16 k] block of W_hh (the same 128 weights
// in registers): the rows are the 4 gates of 2 units, the 8 lanes of a group cover the 8 k-slices of those rows, a wave covers 16 units.  Per step a
// lane reads ITS 16 h values (4 ds_read_b128), 64 v_pk_fma_f32 (h splat by op_sel), then a reduce-scatter over the 8 lanes in three DPP exchanges
// (row_half_mirror, quad_perm [2,3,0,1], quad_perm [1,0,3,2]: 4 + 2 + 1 adds) leaves lane q with gate q & 3 of unit 2 R + (q >> 2).  Every lane applies
// ITS gate's activation (2 transcendentals instead of 8 on a quarter of the threads), the quad exchanges the four activated gates by DPP, the cell
// update runs replicated in the quad, lane 0 of the quad publishes h into the other half of a double-buffered LDS vector: ONE barrier per step.
// Gate pre-activations are prefetched four steps ahead (a step is shorter than a global load).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Wrong values beside foreign waves (rounds 5 - 6).  Round 5: beside another stream's kernels these kernels returned 1e-3 .. 1e-2 errors in single units in
// 299 of 300 launches, always computed in lanes 48 - 63 of some wave, never with the device to itself; claiming the SIMD's whole register file (no third wave on the
// SIMD) removed it and the round called it a DPP hazard.  Round 6 found the cause: it is the PACKED-FP32 arithmetic (`v_pk_fma_f32` of ks_matvec and whatever hipcc packs
// from float4 code), whose results come out stale in the last quarter of the wave when ANOTHER wave on the SIMD issues MFMAs (any GEMM of another stream: 100 / 100
// launches wrong beside `gemm_kernel`, 0 / 100 beside element-wise, copy or fill kernels; with `ds_bpermute` in place of DPP nothing changes; with the packed-FP32 feature
// taken away from the compiler -- csrc/Makefile NOPK, the whole library -- 0 / 100 beside everything, `profiles/r6_packed_fp32_hazard_ab.log`).  The library is built
// without packed-FP32 instructions (tests/test_cabi_cpu.py disassembles it), so these kernels share their SIMDs again; -DFCL_KS_EXCLUSIVE restores the round-5 guard
// for builds that re-enable the feature (`make NOPK=`).
__device__ __forceinline__ void ks_exclusive() {
#ifdef FCL_KS_EXCLUSIVE
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
#endif
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
#ifdef FCL_KS_BPERMUTE  // (developer experiment, round 6: the same lane exchange through the LDS crossbar -- ds_bpermute_b32 -- instead of DPP; profiles/r6_dpp_hazard_ab.log)
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int src = CTRL == 0x141 ? ((lane & ~7) | (7 - (lane & 7))) : CTRL == 0x4E ? (lane ^ 2) : CTRL == 0xB1 ? (lane ^ 1) : CTRL == 0x128 ? ((lane & ~15) | ((lane + 8) & 15))
                    : ((lane & ~3) | (CTRL & 3));  // 0x00 / 0x55 / 0xAA / 0xFF: quad broadcast of lane CTRL & 3
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v)));
#else
    asm volatile("s_nop 3" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
#endif
}

// a lane's [8 x 16] weight block against its 16-value slice of the vector in LDS: 4 ds_read_b128 + 64 v_pk_fma_f32 (the vector element splat by op_sel)
__device__ __forceinline__ void ks_matvec(const f32x2 (&w)[4][16], const float* __restrict__ v16, float (&a8)[8]) {
    f32x4 hv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) hv[e] = *reinterpret_cast<const f32x4*>(v16 + 4 * e);
    f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const float hk = hv[kk >> 2][kk & 3];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = __builtin_elementwise_fma(w[p][kk], f32x2{hk, hk}, acc[p]);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        a8[2 * p] = acc[p][0];
        a8[2 * p + 1] = acc[p][1];
    }
}

// reduce-scatter of 8 values over the 8 lanes of a group (q3 = lane & 7): lane q3 returns the group's total of a8[q3].  Three DPP exchanges:
// row_half_mirror (q3 <-> 7 - q3), quad_perm [2,3,0,1], quad_perm [1,0,3,2]; 4 + 2 + 1 adds
__device__ __forceinline__ float ks_reduce_scatter8(const float (&a8)[8], int q3) {
    const bool lo1 = q3 < 4, lo2 = (q3 & 2) == 0, lo3 = (q3 & 1) == 0;
    float k4[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) k4[x] = (lo1 ? a8[x] : a8[4 + x]) + dpp_f<0x141>(lo1 ? a8[4 + x] : a8[x]);
    float k2[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) k2[x] = (lo2 ? k4[x] : k4[2 + x]) + dpp_f<0x4E>(lo2 ? k4[2 + x] : k4[x]);
    return (lo3 ? k2[0] : k2[1]) + dpp_f<0xB1>(lo3 ? k2[1] : k2[0]);
}

// lane (gate g = lane & 3 of its unit) holds the gate's pre-activation z: activation of that gate (tanh(z) = 2 sigmoid(2z) - 1: two transcendentals
// per lane), the quad's four activated gates by DPP, the cell update replicated in the quad
struct KsCell {
    float a, c_new, h;
};
__device__ __forceinline__ KsCell ks_cell(float z, int g, float c) {
    const float sc = g == 2 ? 2.f : 1.f;
    const float r = __builtin_amdgcn_rcpf(1.0f + __expf(-sc * z));
    KsCell o;
    o.a = g == 2 ? 2.0f * r - 1.0f : r;
    const float ig = dpp_f<0x00>(o.a), fg = dpp_f<0x55>(o.a), gg = dpp_f<0xAA>(o.a), og = dpp_f<0xFF>(o.a);
    o.c_new = fg * c + ig * gg;
    o.h = og * tanh_f(o.c_new);
    return o;
}

template <bool SAVE = false, bool MAPS = false>
__global__ __launch_bounds__(512) void bilstm_ksplit_kernel(const float* __restrict__ gx_f, const float* __restrict__ gx_r,
                                                            const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                            const int* __restrict__ lens, float* __restrict__ out, int T, BilstmSave sv = BilstmSave(),
                                                            unsigned short* __restrict__ out_p = nullptr, fcl_row_maps_t mp = fcl_row_maps_t()) {
    constexpr int H = 128;
    if constexpr (MAPS) {
        if (blockIdx.x == gridDim.x - 1) {
            if (blockIdx.y == 0) {
                row_maps_block<4 * H>(mp);
                row_maps_finish_block<4 * H>(mp);
            }
            return;
        }
    }
    __shared__ __attribute__((aligned(16))) float h_s[2][H];
    ks_exclusive();
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const int q = j & 7, R = j >> 3;       // k-slice, row group (2 units x 4 gates)
    const int g = q & 3, U = 2 * R + (q >> 2);  // the gate / unit this lane ends up with after the reduce-scatter
    const float* gx = (dir ? gx_r : gx_f) + (size_t)b * T * (4 * H);
    const float* whh = dir ? whh_r : whh_f;
    const int len = lens[b];

    f32x2 w[4][16];  // pair p = rows i = 2p, 2p + 1 of the block; row i = gate (i & 3) of unit 2R + (i >> 2)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float* r0 = whh + (size_t)(((2 * p) & 3) * H + 2 * R + ((2 * p) >> 2)) * H + 16 * q;
        const float* r1 = whh + (size_t)(((2 * p + 1) & 3) * H + 2 * R + ((2 * p + 1) >> 2)) * H + 16 * q;
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(r0 + kk), v1 = *reinterpret_cast<const f32x4*>(r1 + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[p][kk + e] = f32x2{v0[e], v1[e]};
        }
    }
    if (j < 2 * H) (&h_s[0][0])[j] = 0.f;
    // zero the padded tail of this direction's half of the output rows
    for (int t = len + (j / H); t < T; t += 4) {
        out[((size_t)b * T + t) * (2 * H) + dir * H + (j % H)] = 0.f;
        if (out_p) store_p32(out_p, (2 * H + 31) >> 5, b * T + t, dir * H + (j % H), 0.f);
    }
    __syncthreads();

    const int t0 = dir ? len - 1 : 0, dt = dir ? -1 : 1;
    const float* gp = gx + g * H + U;
    float gq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) gq[d] = d < len ? gp[(size_t)(t0 + d * dt) * (4 * H)] : 0.f;
    float c = 0.f, h_prev = 0.f;

    for (int s0 = 0; s0 < len; s0 += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int s = s0 + d;
            if (s >= len) break;  // uniform
            const int t = t0 + s * dt;
            const float gcur = gq[d];
            if (s + 4 < len) gq[d] = gp[(size_t)(t + 4 * dt) * (4 * H)];
            float a8[8];
            ks_matvec(w, h_s[d & 1] + 16 * q, a8);
            const KsCell cl = ks_cell(gcur + ks_reduce_scatter8(a8, q), g, c);
            const float a = cl.a, c_new = cl.c_new, h = cl.h;
            if (SAVE) {
                const size_t cell = (size_t)t * sv.B + b;
                sv.gates[dir][cell * (4 * H) + g * H + U] = a;
                if (g == 0) {
                    sv.c_old[dir][cell * H + U] = c;
                    sv.h_old[dir][cell * H + U] = h_prev;
                    sv.c_new[dir][cell * H + U] = c_new;
                }
            }
            if (g == 0) {
                h_s[(d & 1) ^ 1][U] = h;
                out[((size_t)b * T + t) * (2 * H) + dir * H + U] = h;
                if (out_p) store_p32(out_p, (2 * H + 31) >> 5, b * T + t, dir * H + U, h);
            }
            c = c_new;
            h_prev = h;
            __syncthreads();
        }
    }
}

// Reverse pass of one (utterance, direction): thread (q, k) = (j / H, j % H) keeps the H weights W_hh[qH .. qH+H, k] (read from the transposed
// matrix, contiguous) in registers; per step the first H threads do the cell backward, all 4H threads a quarter of dh = dgates . W_hh, the quarters
// meet in LDS.  Visits only live cells, in the reverse of the forward's order; the dead tail of dg is zeroed here.
template <int H>
__global__ __launch_bounds__(4 * H) void bilstm_bptt_persistent_kernel(BilstmBwd a, const int* __restrict__ lens, int T) {
    __shared__ __attribute__((aligned(16))) float dg_s[4 * H];
    __shared__ float p_s[4][H];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const int q = j / H, k = j % H;
    const int len = lens[b];
    float w[H];
    const float* wt = a.whh_t[dir] + (size_t)k * (4 * H) + q * H;
#pragma unroll
    for (int r = 0; r < H; r += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wt + r);
        w[r] = v[0]; w[r + 1] = v[1]; w[r + 2] = v[2]; w[r + 3] = v[3];
    }
    float* dg = a.dg[dir];
    for (int t = len; t < T; ++t) dg[((size_t)t * a.B + b) * (4 * H) + j] = 0.f;  // dead cells contribute nothing to the weight gradients
    float dh = 0.f, dc = 0.f;  // carries (threads j < H)
    int t = dir ? 0 : len - 1;  // reverse of the forward visiting order
    const int dt = dir ? 1 : -1;
    for (int s = 0; s < len; ++s, t += dt) {
        const size_t cell = (size_t)t * a.B + b;
        if (j < H) {
            const float* g = a.gates[dir] + cell * (4 * H);
            const float ig = g[j], fg = g[H + j], gg = g[2 * H + j], og = g[3 * H + j];
            const float dho = dh + a.d_out[((size_t)b * T + t) * a.ld + dir * H + j];
            const float tc = tanh_f(a.c_new[dir][cell * H + j]);
            const float dcn = dc + dho * og * (1.0f - tc * tc);
            const float d0 = dcn * gg * ig * (1.0f - ig), d1 = dcn * a.c_old[dir][cell * H + j] * fg * (1.0f - fg);
            const float d2 = dcn * ig * (1.0f - gg * gg), d3 = dho * tc * og * (1.0f - og);
            dg_s[j] = d0; dg_s[H + j] = d1; dg_s[2 * H + j] = d2; dg_s[3 * H + j] = d3;
            float* o = dg + cell * (4 * H);
            o[j] = d0; o[H + j] = d1; o[2 * H + j] = d2; o[3 * H + j] = d3;
            dc = dcn * fg;
        }
        __syncthreads();
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int r = 0; r < H; r += 4) {
            const f32x4 dv = *reinterpret_cast<const f32x4*>(&dg_s[q * H + r]);  // wave-uniform address (H >= 64) or 64/H addresses: broadcast
            a0 = fmaf(w[r], dv[0], a0);
            a1 = fmaf(w[r + 1], dv[1], a1);
            a2 = fmaf(w[r + 2], dv[2], a2);
            a3 = fmaf(w[r + 3], dv[3], a3);
        }
        p_s[q][k] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (j < H) dh = (p_s[0][j] + p_s[1][j]) + (p_s[2][j] + p_s[3][j]);
    }
}

// Reverse pass at H = 128 with the same lane split: a lane owns [16 rows x 8 k] of W_hh (16 consecutive gate rows are contiguous in the transposed
// matrix), 32 r-slices x 16 k-groups; dh[k] = sum over the 32 r-slices: reduce-scatter over 8 lanes, row_ror:8, and a swizzle across the two rows of
// the 32-lane group.  The sum lands in the lane that owns unit k's cell backward (lanes j & 31 < 8: 16 per wave, all eight waves), so dh never leaves
// registers; operands of the coming step are prefetched; the gate-row gradients go to everybody through a double-buffered LDS vector: one barrier per step.
template <int DUMMY = 0>
__global__ __launch_bounds__(512) void bilstm_bptt_ksplit_kernel(BilstmBwd a, const int* __restrict__ lens, int T) {
    constexpr int H = 128;
    __shared__ __attribute__((aligned(16))) float dg_s[2][4 * H];
    ks_exclusive();
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const int rs = j & 31, q3 = rs & 7, kg = j >> 5;
    const bool owner = rs < 8;
    const int u = 8 * kg + q3;  // the unit whose dh this lane ends up with
    const int len = lens[b];
    f32x2 w[4][16];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const float* c0 = a.whh_t[dir] + (size_t)(8 * kg + 2 * pr) * (4 * H) + 16 * rs;
        const float* c1 = c0 + 4 * H;
#pragma unroll
        for (int rr = 0; rr < 16; rr += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(c0 + rr), v1 = *reinterpret_cast<const f32x4*>(c1 + rr);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[pr][rr + e] = f32x2{v0[e], v1[e]};
        }
    }
    float* dg = a.dg[dir];
    for (int t = len; t < T; ++t) dg[((size_t)t * a.B + b) * (4 * H) + j] = 0.f;  // dead cells contribute nothing to the weight gradients
    float dh = 0.f, dc = 0.f;
    const int t0 = dir ? 0 : len - 1, dt = dir ? 1 : -1;  // reverse of the forward's visiting order
    float pre[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int t) {
        const size_t cell = (size_t)t * a.B + b;
        const float* gsv = a.gates[dir] + cell * (4 * H);
        pre[0] = gsv[u]; pre[1] = gsv[H + u]; pre[2] = gsv[2 * H + u]; pre[3] = gsv[3 * H + u];
        pre[4] = a.d_out[((size_t)b * T + t) * a.ld + dir * H + u];
        pre[5] = a.c_new[dir][cell * H + u];
        pre[6] = a.c_old[dir][cell * H + u];
    };
    if (owner && len > 0) fetch(t0);
    for (int s0 = 0; s0 < len; s0 += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int s = s0 + d;
            if (s >= len) break;  // uniform
            const int t = t0 + s * dt;
            if (owner) {
                const float ig = pre[0], fg = pre[1], gg = pre[2], og = pre[3], dout = pre[4], cn = pre[5], co = pre[6];
                if (s + 1 < len) fetch(t + dt);
                const float dho = dh + dout;
                const float tc = tanh_f(cn);
                const float dcn = dc + dho * og * (1.0f - tc * tc);
                const float d0 = dcn * gg * ig * (1.0f - ig), d1 = dcn * co * fg * (1.0f - fg);
                const float d2 = dcn * ig * (1.0f - gg * gg), d3 = dho * tc * og * (1.0f - og);
                float* ds = dg_s[d];
                ds[u] = d0; ds[H + u] = d1; ds[2 * H + u] = d2; ds[3 * H + u] = d3;
                float* o = dg + ((size_t)t * a.B + b) * (4 * H);
                o[u] = d0; o[H + u] = d1; o[2 * H + u] = d2; o[3 * H + u] = d3;
                dc = dcn * fg;
            }
            __syncthreads();
            if (s + 1 < len) {
                float a8[8];
                ks_matvec(w, dg_s[d] + 16 * rs, a8);
                float v = ks_reduce_scatter8(a8, q3);
                v += dpp_f<0x128>(v);                                                                                 // r-slices + 8
                v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // r-slices + 16 (lane ^ 16)
                dh = v;
            }
        }
    }
}

// ---- H = 256 (FCL-taco2-T): one CU's register file (512 KB) cannot hold the 1 MB of W_hh, so an (utterance, direction) recurrence is shared by
// P = H/64 = 4 workgroups, each owning 64 units (all four gates, 256 weight rows, 128 weights per thread in registers).  Per step the workgroups
// exchange their slices of h through global memory and meet at a counter barrier (release fence -> device-scope atomic -> spin -> acquire fence).
// Group members have adjacent block ids and the grid (8 B workgroups of 512 threads) is refused unless it fits the device one workgroup per CU, so
// with ONE such kernel in flight every group is resident.  Two group kernels in flight on different streams can each hold CUs the other's missing
// members need, so the path is SINGLE-STREAM ONLY; the spin is bounded and a timeout is reported, never silent: the kernel ORs
// FCL_STATUS_GROUP_TIMEOUT into the caller's device status word (the launchers refuse the path without one), which fcl_adam_step tests before
// touching the parameters and the host reads back next to the losses.
struct GroupSync {
    unsigned int* flag;  // [groups] zeroed before the launch
    unsigned int* error; // the caller's status word (never reset here)
};

// No fences: an agent-scope release / acquire on gfx942-class parts writes back / invalidates the whole L2 (measured 45 us per step here).  The
// exchanged words are instead written and read with agent-scope RELAXED atomics (write-through / L2-bypassing accesses), the writers drain their
// store counters before the workgroup barrier, and only then is the arrival counted.
__device__ __forceinline__ void xchg_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float xchg_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool group_barrier(const GroupSync& gs, int group, unsigned int target) {
    __builtin_amdgcn_s_waitcnt(0);  // this thread's write-through stores have been acknowledged
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(gs.flag + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 0;
        for (long long spin = 0; spin < (1ll << 24); ++spin) {
            if (__hip_atomic_load(gs.flag + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
        }
        if (!ok) atomicOr(gs.error, (unsigned int)FCL_STATUS_GROUP_TIMEOUT);
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

template <int H, bool SAVE>
__global__ __launch_bounds__(512) void bilstm_group_kernel(const float* __restrict__ gx_f, const float* __restrict__ gx_r, const float* __restrict__ whh_f,
                                                           const float* __restrict__ whh_r, const int* __restrict__ lens, float* __restrict__ out, int T,
                                                           float* __restrict__ hbuf /* [groups][2][H] */, GroupSync gs, BilstmSave sv) {
    static_assert(H == 256, "4 workgroups x 64 units x 2 K-halves");
    constexpr int P = H / 64;
    __shared__ __attribute__((aligned(16))) float h_s[H];
    __shared__ float g_s[2][256];
    const int p = blockIdx.x % P, group = blockIdx.x / P;  // group = b * 2 + dir
    const int b = group >> 1, dir = group & 1;
    const int j = threadIdx.x, row = j & 255, kh = j >> 8;  // row = gate * 64 + unit of this workgroup's slice
    const int gate = row >> 6, unit = p * 64 + (row & 63);
    const float* gx = (dir ? gx_r : gx_f) + (size_t)b * T * (4 * H);
    const float* whh = (dir ? whh_r : whh_f) + (size_t)(gate * H + unit) * H + kh * 128;
    const int len = lens[b];
    float w[128];
#pragma unroll
    for (int k = 0; k < 128; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(whh + k);
        w[k] = v[0]; w[k + 1] = v[1]; w[k + 2] = v[2]; w[k + 3] = v[3];
    }
    if (j < H) h_s[j] = 0.f;
    float c = 0.f;
    if (j < 64)
        for (int t = len; t < T; ++t) out[((size_t)b * T + t) * (2 * H) + dir * H + p * 64 + j] = 0.f;  // padded tail of this slice
    __syncthreads();
    int t = dir ? len - 1 : 0;
    const int dt = dir ? -1 : 1;
    float* hb = hbuf + (size_t)group * 2 * H;
    for (int s = 0; s < len; ++s, t += dt) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < 128; k += 4) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(&h_s[kh * 128 + k]);
            a0 = fmaf(w[k], hv[0], a0);
            a1 = fmaf(w[k + 1], hv[1], a1);
            a2 = fmaf(w[k + 2], hv[2], a2);
            a3 = fmaf(w[k + 3], hv[3], a3);
        }
        g_s[kh][row] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (j < 64) {
            const float* gr = gx + (size_t)t * (4 * H) + p * 64 + j;
            const float ig = sigmoid_f(gr[0] + g_s[0][j] + g_s[1][j]), fg = sigmoid_f(gr[H] + g_s[0][64 + j] + g_s[1][64 + j]);
            const float gg = tanh_f(gr[2 * H] + g_s[0][128 + j] + g_s[1][128 + j]), og = sigmoid_f(gr[3 * H] + g_s[0][192 + j] + g_s[1][192 + j]);
            const int u = p * 64 + j;
            if (SAVE) {
                const size_t cell = (size_t)t * sv.B + b;
                float* sg = sv.gates[dir] + cell * (4 * H);
                sg[u] = ig; sg[H + u] = fg; sg[2 * H + u] = gg; sg[3 * H + u] = og;
                sv.c_old[dir][cell * H + u] = c;
                sv.h_old[dir][cell * H + u] = h_s[u];
            }
            c = fg * c + ig * gg;
            if (SAVE) sv.c_new[dir][((size_t)t * sv.B + b) * H + u] = c;
            const float h = og * tanh_f(c);
            out[((size_t)b * T + t) * (2 * H) + dir * H + u] = h;
            xchg_store(hb + (s & 1) * H + u, h);
        }
        if (!group_barrier(gs, group, (unsigned int)(P * (s + 1)))) return;
        if (j < H) h_s[j] = xchg_load(hb + (s & 1) * H + j);
        __syncthreads();
    }
}

// reverse pass, same grouping: workgroup p does the cell backward of its 64 units, then its 256 gate rows' share of dh = dgates . W_hh for ALL H columns
// (thread (k, rh): 128 weights W_hh[rows of half rh, k] from the transposed matrix); the P partial vectors are exchanged and summed per unit slice.
template <int H>
__global__ __launch_bounds__(512) void bilstm_bptt_group_kernel(BilstmBwd a, const int* __restrict__ lens, int T, float* __restrict__ part /* [groups][2][P][H] */,
                                                                GroupSync gs) {
    static_assert(H == 256, "4 workgroups x 64 units");
    constexpr int P = H / 64;
    __shared__ __attribute__((aligned(16))) float dg_s[256];  // this workgroup's gate-row gradients, (gate, unit-in-slice) order
    __shared__ float p_s[2][H];
    const int p = blockIdx.x % P, group = blockIdx.x / P;
    const int b = group >> 1, dir = group & 1;
    const int j = threadIdx.x, k = j & 255, rh = j >> 8;
    const int len = lens[b];
    float w[128];  // W_hh[(2rh + seg) * H + 64p + r, k] for seg in {0, 1}, r in 0..63
    const float* wt = a.whh_t[dir] + (size_t)k * (4 * H);
#pragma unroll
    for (int seg = 0; seg < 2; ++seg)
#pragma unroll
        for (int r = 0; r < 64; r += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wt + (2 * rh + seg) * H + p * 64 + r);
            w[seg * 64 + r] = v[0]; w[seg * 64 + r + 1] = v[1]; w[seg * 64 + r + 2] = v[2]; w[seg * 64 + r + 3] = v[3];
        }
    float* dg = a.dg[dir];
    if (j < 256)
        for (int t = len; t < T; ++t) dg[((size_t)t * a.B + b) * (4 * H) + (j >> 6) * H + p * 64 + (j & 63)] = 0.f;  // dead cells of this slice
    float dh = 0.f, dc = 0.f;
    int t = dir ? 0 : len - 1;
    const int dt = dir ? 1 : -1;
    float* pb = part + (size_t)group * 2 * P * H;
    for (int s = 0; s < len; ++s, t += dt) {
        const size_t cell = (size_t)t * a.B + b;
        if (j < 64) {
            const int u = p * 64 + j;
            const float* g = a.gates[dir] + cell * (4 * H);
            const float ig = g[u], fg = g[H + u], gg = g[2 * H + u], og = g[3 * H + u];
            const float dho = dh + a.d_out[((size_t)b * T + t) * a.ld + dir * H + u];
            const float tc = tanh_f(a.c_new[dir][cell * H + u]);
            const float dcn = dc + dho * og * (1.0f - tc * tc);
            const float d0 = dcn * gg * ig * (1.0f - ig), d1 = dcn * a.c_old[dir][cell * H + u] * fg * (1.0f - fg);
            const float d2 = dcn * ig * (1.0f - gg * gg), d3 = dho * tc * og * (1.0f - og);
            dg_s[j] = d0; dg_s[64 + j] = d1; dg_s[128 + j] = d2; dg_s[192 + j] = d3;
            float* o = dg + cell * (4 * H);
            o[u] = d0; o[H + u] = d1; o[2 * H + u] = d2; o[3 * H + u] = d3;
            dc = dcn * fg;
        }
        __syncthreads();
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int r = 0; r < 128; r += 4) {
            const f32x4 dv = *reinterpret_cast<const f32x4*>(&dg_s[rh * 128 + r]);
            a0 = fmaf(w[r], dv[0], a0);
            a1 = fmaf(w[r + 1], dv[1], a1);
            a2 = fmaf(w[r + 2], dv[2], a2);
            a3 = fmaf(w[r + 3], dv[3], a3);
        }
        p_s[rh][k] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (j < H) xchg_store(pb + ((s & 1) * P + p) * H + j, p_s[0][j] + p_s[1][j]);
        if (!group_barrier(gs, group, (unsigned int)(P * (s + 1)))) return;
        if (j < 64) {
            const float* q = pb + (s & 1) * P * H + p * 64 + j;
            float v = 0.f;
#pragma unroll
            for (int pp = 0; pp < P; ++pp) v += xchg_load(q + pp * H);
            dh = v;
        }
    }
}

// ---- H = 256, round 5: the group kernels with (1) the lane-split contraction of bilstm_ksplit_kernel (a workgroup's [256 rows x 256 k] block: 16
// k-slices per row group, so the reduce-scatter is followed by one row_ror:8 add), (2) every per-step global operand prefetched, and (3) the exchange
// as ONE memory round trip: a published word carries its value AND the step number (8-byte relaxed agent-scope store), consumers poll the words
// themselves until the tag is the step's (tags are 16 bits: sequences shorter than 65 535 steps, checked by the launchers).  The counter protocol above costs four dependent round trips per step (drain stores, atomic add, spin, load).
// Double buffering by step parity is safe: nobody can publish step s + 2 before it has consumed every member's step s + 1, which a member publishes only
// after it has consumed step s.  The exchange area is zeroed before each launch (tags start at 1).
// The two dwords of a published word each carry 16 bits of the value and the step's 16-bit tag: a reader accepts the word only when BOTH halves carry
// the tag, so nothing depends on the 8-byte store reaching a polling reader as one piece.  (Measured: with {value, tag} dwords and readers polling while the
// word is written, `tools/stress_bilstm_concurrent.py` -- the kernel beside another stream's work, members not co-resident -- returned a step's tag with the
// value of two steps before in 299 of 300 launches; alone on the device it never showed.)
__device__ __forceinline__ void ll_store(unsigned long long* p, float v, unsigned int tag) {
    const unsigned int u = __float_as_uint(v), t16 = tag & 0xffffu;
    const unsigned long long w = (unsigned long long)((u & 0xffff0000u) | t16) | ((unsigned long long)((u << 16) | t16) << 32);
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool ll_poll(const unsigned long long* p, unsigned int tag, float& v) {
    const unsigned int t16 = tag & 0xffffu;
    for (int spin = 0; spin < (1 << 22); ++spin) {
        const unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int w0 = (unsigned int)w, w1 = (unsigned int)(w >> 32);
        if ((w0 & 0xffffu) == t16 && (w1 & 0xffffu) == t16) {
            v = __uint_as_float((w0 & 0xffff0000u) | (w1 >> 16));
            return true;
        }
    }
    return false;
}

template <bool SAVE>
__global__ __launch_bounds__(512) void bilstm_group_ks_kernel(const float* __restrict__ gx_f, const float* __restrict__ gx_r, const float* __restrict__ whh_f,
                                                              const float* __restrict__ whh_r, const int* __restrict__ lens, float* __restrict__ out, int T,
                                                              unsigned long long* __restrict__ hbuf /* [groups][2][H] */, GroupSync gs, BilstmSave sv) {
    constexpr int H = 256, P = 4;
    __shared__ __attribute__((aligned(16))) float h_s[2][H];
    __shared__ int ok_s;
    ks_exclusive();
    const int p = blockIdx.x % P, group = blockIdx.x / P;  // group = b * 2 + dir
    const int b = group >> 1, dir = group & 1;
    const int j = threadIdx.x, q = j & 15, q3 = q & 7, R = j >> 4;  // k-slice (16 of 16 k), row group (2 units x 4 gates of this workgroup's 64 units)
    const int g = q3 & 3, U = p * 64 + 2 * R + (q3 >> 2);           // after the reduction (both halves of the 16 lanes hold the same sums)
    const bool first_half = (q & 8) == 0, publisher = first_half && g == 0;
    const float* gx = (dir ? gx_r : gx_f) + (size_t)b * T * (4 * H);
    const float* whh = dir ? whh_r : whh_f;
    const int len = lens[b];
    f32x2 w[4][16];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const float* r0 = whh + (size_t)(((2 * pr) & 3) * H + p * 64 + 2 * R + ((2 * pr) >> 2)) * H + 16 * q;
        const float* r1 = whh + (size_t)(((2 * pr + 1) & 3) * H + p * 64 + 2 * R + ((2 * pr + 1) >> 2)) * H + 16 * q;
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(r0 + kk), v1 = *reinterpret_cast<const f32x4*>(r1 + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[pr][kk + e] = f32x2{v0[e], v1[e]};
        }
    }
    (&h_s[0][0])[j] = 0.f;
    if (j == 0) ok_s = 1;
    if (j < 64)
        for (int t = len; t < T; ++t) out[((size_t)b * T + t) * (2 * H) + dir * H + p * 64 + j] = 0.f;  // padded tail of this slice
    __syncthreads();
    const int t0 = dir ? len - 1 : 0, dt = dir ? -1 : 1;
    const float* gp = gx + g * H + U;
    float gq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) gq[d] = d < len ? gp[(size_t)(t0 + d * dt) * (4 * H)] : 0.f;
    float c = 0.f, h_prev = 0.f;
    unsigned long long* hb = hbuf + (size_t)group * 2 * H;
    for (int s0 = 0; s0 < len; s0 += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int s = s0 + d;
            if (s >= len) break;  // uniform
            const int t = t0 + s * dt;
            const float gcur = gq[d];
            if (s + 4 < len) gq[d] = gp[(size_t)(t + 4 * dt) * (4 * H)];
            float a8[8];
            ks_matvec(w, h_s[d & 1] + 16 * q, a8);
            float z = ks_reduce_scatter8(a8, q3);
            z = gcur + (z + dpp_f<0x128>(z));  // row_ror:8: the other eight k-slices
            const KsCell cl = ks_cell(z, g, c);
            if (SAVE && first_half) {
                const size_t cell = (size_t)t * sv.B + b;
                sv.gates[dir][cell * (4 * H) + g * H + U] = cl.a;
                if (g == 0) {
                    sv.c_old[dir][cell * H + U] = c;
                    sv.h_old[dir][cell * H + U] = h_prev;
                    sv.c_new[dir][cell * H + U] = cl.c_new;
                }
            }
            if (publisher) {
                ll_store(hb + (d & 1) * H + U, cl.h, (unsigned int)(s + 1));
                out[((size_t)b * T + t) * (2 * H) + dir * H + U] = cl.h;
            }
            c = cl.c_new;
            h_prev = cl.h;
            if (j < H && s + 1 < len) {  // the next step's h: all four slices, this workgroup's own included
                float v = 0.f;
                if (!ll_poll(hb + (d & 1) * H + j, (unsigned int)(s + 1), v)) {
                    atomicOr(gs.error, (unsigned int)FCL_STATUS_GROUP_TIMEOUT);
                    ok_s = 0;
                }
                h_s[(d & 1) ^ 1][j] = v;
            }
            __syncthreads();
            if (!ok_s) return;
        }
    }
}

// reverse pass: a lane owns [16 rows x 8 k] of this workgroup's [256 gate rows x 256 k] block of W_hh (from the transposed matrix: 16 consecutive rows of
// one gate are contiguous); 16 r-slices x 32 k-groups.  The 64 owner lanes (wave 0) poll the four partial sums of their unit from the previous step, do the
// cell backward with operands prefetched a step ahead and hand this workgroup's gate-row gradients to everybody through LDS: one barrier per step.
__global__ __launch_bounds__(512) void bilstm_bptt_group_ks_kernel(BilstmBwd a, const int* __restrict__ lens, int T,
                                                                   unsigned long long* __restrict__ part /* [groups][2][P][H] */, GroupSync gs) {
    constexpr int H = 256, P = 4;
    __shared__ __attribute__((aligned(16))) float dg_s[2][256];  // (gate, unit-in-slice) order
    __shared__ int ok_s;
    ks_exclusive();
    const int p = blockIdx.x % P, group = blockIdx.x / P;
    const int b = group >> 1, dir = group & 1;
    const int j = threadIdx.x, rs = j & 15, q3 = rs & 7, kg = j >> 4;
    const int len = lens[b];
    f32x2 w[4][16];  // pair pr: k = 8 kg + 2 pr, + 1; rr: local row 16 rs + rr = (gate, unit-in-slice)
    {
        const int lr = 16 * rs, grow = (lr >> 6) * H + p * 64 + (lr & 63);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const float* c0 = a.whh_t[dir] + (size_t)(8 * kg + 2 * pr) * (4 * H) + grow;
            const float* c1 = c0 + 4 * H;
#pragma unroll
            for (int rr = 0; rr < 16; rr += 4) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(c0 + rr), v1 = *reinterpret_cast<const f32x4*>(c1 + rr);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[pr][rr + e] = f32x2{v0[e], v1[e]};
            }
        }
    }
    float* dg = a.dg[dir];
    if (j < 256)
        for (int t = len; t < T; ++t) dg[((size_t)t * a.B + b) * (4 * H) + (j >> 6) * H + p * 64 + (j & 63)] = 0.f;  // dead cells of this slice
    if (j == 0) ok_s = 1;
    __syncthreads();
    float dh = 0.f, dc = 0.f;
    const int t0 = dir ? 0 : len - 1, dt = dir ? 1 : -1;  // reverse of the forward's visiting order
    unsigned long long* pb = part + (size_t)group * 2 * P * H;
    const int u = p * 64 + (j & 63);
    float pre[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // ig, fg, gg, og, d_out, c_new, c_old of the owner's unit at the coming step
    auto fetch = [&](int t) {
        const size_t cell = (size_t)t * a.B + b;
        const float* gsv = a.gates[dir] + cell * (4 * H);
        pre[0] = gsv[u]; pre[1] = gsv[H + u]; pre[2] = gsv[2 * H + u]; pre[3] = gsv[3 * H + u];
        pre[4] = a.d_out[((size_t)b * T + t) * a.ld + dir * H + u];
        pre[5] = a.c_new[dir][cell * H + u];
        pre[6] = a.c_old[dir][cell * H + u];
    };
    if (j < 64 && len > 0) fetch(t0);
    for (int s0 = 0; s0 < len; s0 += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int s = s0 + d;
            if (s >= len) break;  // uniform
            const int t = t0 + s * dt;
            if (j < 64) {
                const float ig = pre[0], fg = pre[1], gg = pre[2], og = pre[3], dout = pre[4], cn = pre[5], co = pre[6];
                if (s + 1 < len) fetch(t + dt);
                if (s > 0) {  // dh = the four workgroups' partial sums of the previous step
                    float v[4];
                    bool ok = true;
#pragma unroll
                    for (int pp = 0; pp < P; ++pp) ok = ll_poll(pb + ((d ^ 1) * P + pp) * H + u, (unsigned int)s, v[pp]) && ok;
                    if (!ok) {
                        atomicOr(gs.error, (unsigned int)FCL_STATUS_GROUP_TIMEOUT);
                        ok_s = 0;
                    }
                    dh = (v[0] + v[1]) + (v[2] + v[3]);
                }
                const float dho = dh + dout;
                const float tc = tanh_f(cn);
                const float dcn = dc + dho * og * (1.0f - tc * tc);
                const float d0 = dcn * gg * ig * (1.0f - ig), d1 = dcn * co * fg * (1.0f - fg);
                const float d2 = dcn * ig * (1.0f - gg * gg), d3 = dho * tc * og * (1.0f - og);
                float* ds = dg_s[d];
                ds[j] = d0; ds[64 + j] = d1; ds[128 + j] = d2; ds[192 + j] = d3;
                float* o = dg + ((size_t)t * a.B + b) * (4 * H);
                o[u] = d0; o[H + u] = d1; o[2 * H + u] = d2; o[3 * H + u] = d3;
                dc = dcn * fg;
            }
            __syncthreads();
            if (!ok_s) return;
            if (s + 1 < len) {
                float a8[8];
                ks_matvec(w, dg_s[d] + 16 * rs, a8);
                float v = ks_reduce_scatter8(a8, q3);
                v += dpp_f<0x128>(v);
                if ((rs & 8) == 0) ll_store(pb + (d * P + p) * H + 8 * kg + q3, v, (unsigned int)(s + 1));
            }
        }
    }
}

__global__ void fill_kernel(float* p, long long n, float v) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace fcl

using namespace fcl;

namespace fcl {

// both directions of the training forward / reverse pass in one launch each; false = H unsupported (the caller falls back to per-step launches)
// H = 256: groups of 4 workgroups.  ws: [2B] flags + 1 error word (zeroed here) followed by the exchange buffer.
static bool bilstm_ksplit_enabled() {
    static const int on = tunable("BILSTM_KSPLIT", 1);
    return on != 0;
}

// Builds WITH packed-FP32 instructions (-DFCL_KS_EXCLUSIVE, `make NOPK=`) keep the round-5 guard, and verify it: before the first launch of a
// lane-split kernel the runtime is asked what the loaded code object really requests (hipFuncGetAttributes: numRegs, in the hardware's 8-register granules 249 .. 256 all
// allocate 256); a kernel that does not claim the whole file is NOT launched -- its callers fall back to the row-per-thread kernels -- and the reason is left in
// fcl_last_error().  The default build has no packed-FP32 instruction to guard against (tests/test_cabi_cpu.py) and skips the check.
template <typename K>
static bool ks_claims_simd(K kernel, const char* name) {
#ifndef FCL_KS_EXCLUSIVE
    return true;  // (no packed-FP32 instructions in this build: nothing to guard -- see ks_exclusive())
#endif
    static const int guard = tunable("KS_GUARD", 1);
    if (!guard) return true;
    static std::mutex mu;
    static std::map<const void*, bool> seen;
    std::lock_guard<std::mutex> lock(mu);
    const void* key = reinterpret_cast<const void*>(kernel);
    auto it = seen.find(key);
    if (it != seen.end()) return it->second;
    hipFuncAttributes at;
    const bool got = hipFuncGetAttributes(&at, key) == hipSuccess;
    if (!got) (void)hipGetLastError();
    const bool ok = got && at.numRegs >= 249;
    if (!ok) set_error("bilstm: %s does not claim the SIMD's register file (numRegs %d, 256 expected): the lane-split kernels are off, the row-per-thread kernels run", name,
                       got ? at.numRegs : -1);
    seen[key] = ok;
    return ok;
}

// 1 KB (error word, flags from +1 KB) | flags [2B] | exchange area: [groups][2][4][H] 8-byte words (value + step tag; the counter-protocol kernels use it as floats)
size_t bilstm_group_workspace_bytes(int B, int H) { return 1024 + sizeof(unsigned int) * 2 * (size_t)B + 32 + sizeof(unsigned long long) * 2 * (size_t)B * 2 * 4 * H; }
static char* group_exchange_base(void* ws, int B) {  // 16-byte aligned, behind the error word and the flags
    return (char*)(((uintptr_t)ws + 1024 + sizeof(unsigned int) * 2 * (size_t)B + 15) & ~(uintptr_t)15);
}

static bool group_ok(int B, int H, void* ws, size_t ws_bytes, const unsigned int* status) {
    static const int enabled = tunable("BILSTM_GROUP", 1);
    if (!enabled || H != 256 || !ws || ws_bytes < bilstm_group_workspace_bytes(B, H) || !status) return false;  // no status word, no spinning kernel
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    // The grid must FIT the device (one 512-thread workgroup per CU: two waves per SIMD at 168 - 194 VGPRs leave no room for a second one of these;
    // other streams' smaller waves may share the SIMDs since round 6).  That does not make the group's members co-resident beside other streams' kernels: a member may wait for a CU a foreign workgroup holds, and
    // the members already resident spin meanwhile.  Progress then rests on the foreign kernels finishing (they never wait for these), on in-order dispatch of
    // this grid, and on the bounded spin: after 2^22 polls a workgroup gives up and raises FCL_STATUS_GROUP_TIMEOUT -- the optimizer skips that update ON THE
    // DEVICE and the step's LossReport.resolve() raises FclError with the status text (training.py), so starvation is loud, never a silently wrong gradient.
    // Two spinning group kernels on two streams (or two processes) of one device can starve each other; the engines issue one group kernel at a time per
    // device (the KD update's other recurrence is the student's single-workgroup H = 128 kernel, which does not spin).
    return 2 * B * 4 <= cus;
}

bool launch_bilstm_group(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r, const int* lens, float* out, int B, int T, int H,
                         const BilstmSave* sv, void* ws, size_t ws_bytes, unsigned int* status, hipStream_t s) {
    if (!group_ok(B, H, ws, ws_bytes, status)) return false;
    unsigned int* flags = (unsigned int*)ws;
    static const int ll_on = tunable("BILSTM_GROUP_LL", 1);
    const bool ll = ll_on && T < 65535 &&  // 16-bit step tags
                    (sv ? ks_claims_simd(bilstm_group_ks_kernel<true>, "bilstm_group_ks_kernel<true>") : ks_claims_simd(bilstm_group_ks_kernel<false>, "bilstm_group_ks_kernel<false>"));
    char* xb = group_exchange_base(ws, B);
    if (hipMemsetAsync(flags, 0, (size_t)(xb - (char*)ws) + (ll ? sizeof(unsigned long long) * 2 * (size_t)B * 2 * H : 0), s) != hipSuccess) return false;
    GroupSync gs{flags + 256, status};
    float* hbuf = (float*)xb;
    dim3 grid(2 * B * 4);
    if (ll) {
        ProfScope ps(sv ? "bilstm_group_ks_kernel/train" : "bilstm_group_ks_kernel", 2.0 * 2 * B * (double)T * 4 * H * H, (double)B * T, s);
        if (sv) hipLaunchKernelGGL((bilstm_group_ks_kernel<true>), grid, dim3(512), 0, s, gx_f, gx_r, whh_f, whh_r, lens, out, T, (unsigned long long*)hbuf, gs, *sv);
        else hipLaunchKernelGGL((bilstm_group_ks_kernel<false>), grid, dim3(512), 0, s, gx_f, gx_r, whh_f, whh_r, lens, out, T, (unsigned long long*)hbuf, gs, BilstmSave());
        return true;
    }
    ProfScope ps(sv ? "bilstm_group_kernel<256>/train" : "bilstm_group_kernel<256>", 2.0 * 2 * B * (double)T * 4 * H * H, (double)B * T, s);
    if (sv) hipLaunchKernelGGL((bilstm_group_kernel<256, true>), grid, dim3(512), 0, s, gx_f, gx_r, whh_f, whh_r, lens, out, T, hbuf, gs, *sv);
    else hipLaunchKernelGGL((bilstm_group_kernel<256, false>), grid, dim3(512), 0, s, gx_f, gx_r, whh_f, whh_r, lens, out, T, hbuf, gs, BilstmSave());
    return true;
}

bool launch_bilstm_bptt_group(const BilstmBwd& a, const int* lens, int B, int T, int H, void* ws, size_t ws_bytes, unsigned int* status, hipStream_t s) {
    if (!group_ok(B, H, ws, ws_bytes, status)) return false;
    unsigned int* flags = (unsigned int*)ws;
    static const int ll_on = tunable("BILSTM_GROUP_LL", 1);
    const bool ll = ll_on && T < 65535 && ks_claims_simd(bilstm_bptt_group_ks_kernel, "bilstm_bptt_group_ks_kernel");  // 16-bit step tags
    char* xb = group_exchange_base(ws, B);
    if (hipMemsetAsync(flags, 0, (size_t)(xb - (char*)ws) + (ll ? sizeof(unsigned long long) * 2 * (size_t)B * 2 * 4 * H : 0), s) != hipSuccess) return false;
    GroupSync gs{flags + 256, status};
    float* part = (float*)xb;
    if (ll) {
        ProfScope ps("bilstm_bptt_group_ks_kernel", 2.0 * 2 * B * (double)T * 4 * H * H, (double)B * T, s);
        hipLaunchKernelGGL(bilstm_bptt_group_ks_kernel, dim3(2 * B * 4), dim3(512), 0, s, a, lens, T, (unsigned long long*)part, gs);
        return true;
    }
    ProfScope ps("bilstm_bptt_group_kernel<256>", 2.0 * 2 * B * (double)T * 4 * H * H, (double)B * T, s);
    hipLaunchKernelGGL((bilstm_bptt_group_kernel<256>), dim3(2 * B * 4), dim3(512), 0, s, a, lens, T, part, gs);
    return true;
}

bool launch_bilstm_train_persistent(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r, const int* lens, float* out, int B, int T,
                                    int H, const BilstmSave& sv, hipStream_t s) {
    dim3 grid(B, 2);
    if (H != 8 && H != 16 && H != 32 && H != 64 && H != 128) return false;
    if (H == 128 && bilstm_ksplit_enabled() && ks_claims_simd(bilstm_ksplit_kernel<true, false>, "bilstm_ksplit_kernel<true, false>")) {
        ProfScope ps("bilstm_ksplit_kernel/train", 2.0 * 2 * B * (double)T * 4 * H * H, (double)B * T, s);
        hipLaunchKernelGGL((bilstm_ksplit_kernel<true, false>), grid, dim3(512), 0, s, gx_f, gx_r, whh_f, whh_r, lens, out, T, sv, (unsigned short*)nullptr,
                           fcl_row_maps_t());
        return true;
    }
    ProfScope ps("bilstm_persistent_kernel/train", 2.0 * 2 * B * (double)T * 4 * H * H, (double)B * T, s);
#define FCL_BILSTM_CASE(HH) \
    case HH: hipLaunchKernelGGL((bilstm_persistent_kernel<HH, true>), grid, dim3(4 * HH), 0, s, gx_f, gx_r, whh_f, whh_r, lens, out, T, sv); return true;
    switch (H) {
        FCL_BILSTM_CASE(8)
        FCL_BILSTM_CASE(16)
        FCL_BILSTM_CASE(32)
        FCL_BILSTM_CASE(64)
        FCL_BILSTM_CASE(128)
    }
#undef FCL_BILSTM_CASE
    return false;
}

bool launch_bilstm_bptt_persistent(const BilstmBwd& a, const int* lens, int B, int T, int H, hipStream_t s) {
    dim3 grid(B, 2);
    if (H != 8 && H != 16 && H != 32 && H != 64 && H != 128) return false;
    if (H == 128 && bilstm_ksplit_enabled() && ks_claims_simd(bilstm_bptt_ksplit_kernel<0>, "bilstm_bptt_ksplit_kernel<0>")) {
        ProfScope ps("bilstm_bptt_ksplit_kernel", 2.0 * 2 * B * (double)T * 4 * H * H, (double)B * T, s);
        hipLaunchKernelGGL((bilstm_bptt_ksplit_kernel<0>), grid, dim3(512), 0, s, a, lens, T);
        return true;
    }
    ProfScope ps("bilstm_bptt_persistent_kernel", 2.0 * 2 * B * (double)T * 4 * H * H, (double)B * T, s);
#define FCL_BILSTM_CASE(HH) \
    case HH: hipLaunchKernelGGL((bilstm_bptt_persistent_kernel<HH>), grid, dim3(4 * HH), 0, s, a, lens, T); return true;
    switch (H) {
        FCL_BILSTM_CASE(8)
        FCL_BILSTM_CASE(16)
        FCL_BILSTM_CASE(32)
        FCL_BILSTM_CASE(64)
        FCL_BILSTM_CASE(128)
    }
#undef FCL_BILSTM_CASE
    return false;
}

}  // namespace fcl

extern "C" {

// workspace: Gx_f [B*T,4H] | Gx_r [B*T,4H] | h[2][2 dirs][B,H] | c[2 dirs][B,H]
size_t fcl_bilstm_workspace_bytes(int b, int t, int h) {
    if (b <= 0 || t <= 0 || h <= 0) return 0;
    return sizeof(float) * ((size_t)2 * b * t * 4 * h + (size_t)6 * b * h) + 256 + (h == 256 ? bilstm_group_workspace_bytes(b, h) : 0);
}

int fcl_bilstm_fwd(const float* x, const int32_t* lens, const float* w_ih_f, const float* w_hh_f, const float* b_f,
                   const float* w_ih_r, const float* w_hh_r, const float* b_r, float* out, uint16_t* out_p, const uint16_t* x_p,
                   const uint16_t* w_ih_f_p, const uint16_t* w_ih_r_p, int b, int t, int c, int h,
                   int algo, void* workspace, size_t workspace_bytes, uint32_t* status, const fcl_row_maps_t* row_maps, fcl_stream_t stream) {
    FCL_REQUIRE((x || x_p) && lens && w_ih_f && w_hh_f && b_f && w_ih_r && w_hh_r && b_r && out, FCL_ERR_INVALID, "bilstm_fwd: null argument");
    FCL_REQUIRE((x_p == nullptr) == (w_ih_f_p == nullptr) && (x_p == nullptr) == (w_ih_r_p == nullptr), FCL_ERR_INVALID,
                "bilstm_fwd: x_p / w_ih_f_p / w_ih_r_p come together");
    FCL_REQUIRE(x || x_p, FCL_ERR_INVALID, "bilstm_fwd: no input");
    FCL_REQUIRE(b > 0 && t > 0 && c > 0 && h > 0 && (c & 3) == 0 && (h & 3) == 0, FCL_ERR_SHAPE, "bilstm_fwd: bad sizes B=%d T=%d C=%d H=%d", b, t, c, h);
    FCL_REQUIRE(workspace && workspace_bytes >= fcl_bilstm_workspace_bytes(b, t, h), FCL_ERR_WORKSPACE, "bilstm_fwd: workspace too small");
    FCL_REQUIRE(aligned16(workspace) && aligned16(out), FCL_ERR_ALIGN, "bilstm_fwd: 16-byte alignment required");
    // row_maps: also build the batch's row / frame maps (fcl_row_maps_build).  With the persistent recurrence at H = 128 they ride in the same
    // launch (one extra workgroup); every other path builds them with their own two launches after the recurrence
    bool maps_fusable = false;
    if (row_maps) {
        const int rc = row_maps_check(row_maps, &maps_fusable);
        if (rc) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    float* gx_f = reinterpret_cast<float*>(workspace);
    float* gx_r = gx_f + (size_t)b * t * 4 * h;
    float* hbuf = gx_r + (size_t)b * t * 4 * h;  // [2 ping-pong][2 dirs][B,H]
    float* cbuf = hbuf + (size_t)4 * b * h;      // [2 dirs][B,H]

    for (int d = 0; d < 2; ++d) {
        GemmArgs g = {};
        g.term[0] = GemmTerm{x, d ? w_ih_r : w_ih_f, c, c, c, 0};
        if (x_p && w_ih_f_p && w_ih_r_p) {  // pre-split operands of the input projection (written by the last encoder convolution / at plan time)
            g.term[0].Ap = x_p;
            g.term[0].Wp = d ? w_ih_r_p : w_ih_f_p;
            g.term[0].lda_p = g.term[0].ldw_p = (c + 31) / 32;
        }
        g.nterms = 1;
        g.M = b * t;
        g.N = 4 * h;
        g.bias = d ? b_r : b_f;
        g.Y = d ? gx_r : gx_f;
        g.ldy = 4 * h;
        int rc = launch_gemm(g, s);
        if (rc) return rc;
    }
    const bool can_persist = (h == 8 || h == 16 || h == 32 || h == 64 || h == 128);
    // algo 3 (H = 256, FCL-taco2-T): 4 cooperating workgroups per (utterance, direction).  Opt-in: it is 2.6x faster than 2T per-step launches on an
    // otherwise idle GPU (the training step), but its 8B spinning workgroups own every CU for the whole recurrence, which costs 9 % of throughput when
    // several synthesis passes are in flight on other streams (3.52 vs 3.85 M frames/s) — so algo 0 keeps the per-step launches there.
    static const int group_infer = tunable("BILSTM_GROUP_INFER", 0);
    FCL_REQUIRE(algo != 3 || status, FCL_ERR_INVALID, "bilstm_fwd: algo 3 (cooperating workgroups) needs a device status word");
    static const int fuse_on = tunable("ROWMAPS_FUSE", 1);
    struct MapsAfter {  // the non-fused paths below return from several places
        const fcl_row_maps_t* m;
        fcl_stream_t st;
        int finish(int rc) const { return (rc || !m) ? rc : fcl_row_maps_build(m, st); }
    } after{row_maps, stream};
    if (h == 256 && status && (algo == 3 || (algo == 0 && group_infer))) {
        void* gws = hbuf;  // the Gx buffers are final, so the tail of the workspace (per-step state of algo 1) is free for the flags and the exchange buffer
        const size_t gbytes = workspace_bytes - sizeof(float) * ((size_t)2 * b * t * 4 * h);
        if (launch_bilstm_group(gx_f, gx_r, w_hh_f, w_hh_r, lens, out, b, t, h, nullptr, gws, gbytes, status, s)) {
            FCL_HIP(hipGetLastError());
            return after.finish(out_p ? fcl_pack_planes(out, 2 * h, b * t, 2 * h, out_p, stream) : 0);
        }
    }
    if (algo == 0 || algo == 3) algo = can_persist ? 2 : 1;
    FCL_REQUIRE(algo == 1 || (algo == 2 && can_persist), FCL_ERR_INVALID, "bilstm_fwd: algo %d unavailable for H=%d", algo, h);
    const bool fuse_wanted = row_maps && maps_fusable && fuse_on;
    if (algo == 2 && h == 128 && bilstm_ksplit_enabled() &&
        (fuse_wanted ? ks_claims_simd(bilstm_ksplit_kernel<false, true>, "bilstm_ksplit_kernel<false, true>")
                     : ks_claims_simd(bilstm_ksplit_kernel<false, false>, "bilstm_ksplit_kernel<false, false>"))) {
        const bool fuse = fuse_wanted;
        ProfScope ps(fuse ? "bilstm_ksplit_kernel+maps" : "bilstm_ksplit_kernel", 2.0 * 2 * b * (double)t * 4 * h * h, (double)b * t, s);
        if (fuse)
            hipLaunchKernelGGL((bilstm_ksplit_kernel<false, true>), dim3(b + 1, 2), dim3(512), 0, s, gx_f, gx_r, w_hh_f, w_hh_r, lens, out, t, BilstmSave(), out_p,
                               *row_maps);
        else
            hipLaunchKernelGGL((bilstm_ksplit_kernel<false, false>), dim3(b, 2), dim3(512), 0, s, gx_f, gx_r, w_hh_f, w_hh_r, lens, out, t, BilstmSave(), out_p,
                               fcl_row_maps_t());
        const int rc = check_hip(hipGetLastError(), "bilstm k-split launch");
        return fuse ? rc : after.finish(rc);
    }
    if (algo == 2 && h == 128 && row_maps && maps_fusable && fuse_on) {
        ProfScope ps("bilstm_persistent_kernel+maps", 2.0 * 2 * b * (double)t * 4 * h * h, (double)b * t, s);
        hipLaunchKernelGGL((bilstm_persistent_kernel<128, false, true>), dim3(b + 1, 2), dim3(512), 0, s, gx_f, gx_r, w_hh_f, w_hh_r, lens, out, t, BilstmSave(),
                           out_p, *row_maps);
        return check_hip(hipGetLastError(), "bilstm persistent (+ row maps) launch");
    }
    if (algo == 2) {
        ProfScope ps("bilstm_persistent_kernel", 2.0 * 2 * b * (double)t * 4 * h * h, (double)b * t, s);
        dim3 grid(b, 2);
#define FCL_BILSTM_CASE(HH) \
    case HH: hipLaunchKernelGGL((bilstm_persistent_kernel<HH>), grid, dim3(4 * HH), 0, s, gx_f, gx_r, w_hh_f, w_hh_r, lens, out, t, BilstmSave(), out_p); break;
        switch (h) {
            FCL_BILSTM_CASE(8)
            FCL_BILSTM_CASE(16)
            FCL_BILSTM_CASE(32)
            FCL_BILSTM_CASE(64)
            FCL_BILSTM_CASE(128)
        }
#undef FCL_BILSTM_CASE
        return after.finish(check_hip(hipGetLastError(), "bilstm persistent launch"));
    }
    // algo 1: per-step launches
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, s, hbuf, (long long)6 * b * h, 0.f);
    auto step_args = [&](int d, int st, int cur) {
        float* hp[2] = {hbuf + (size_t)d * b * h, hbuf + (size_t)(2 + d) * b * h};
        const int tt = d ? t - 1 - st : st;
        LstmStepArgs a = {};
        a.term[0] = GemmTerm{hp[cur], d ? w_hh_r : w_hh_f, h, h, h, 0};
        a.nterms = 1;
        a.M = b;
        a.U = h;
        a.G = d ? gx_r : gx_f;
        a.g_row_mul = t;
        a.g_row_add = tt;
        a.step = tt;
        a.h_in = hp[cur];
        a.h_out = hp[cur ^ 1];
        a.c = cbuf + (size_t)d * b * h;
        a.zoneout = 0.f;
        a.row_len = lens;
        a.out2 = out;
        a.out2_row_mul = t;
        a.out2_row_add = tt;
        a.ld2 = 2 * h;
        a.out2_col_off = d * h;
        return a;
    };
    // the two directions are independent recurrences: while a step is small enough for the wave-per-gate kernel both go out in ONE launch per time
    // step (T dependent launches instead of 2T)
    static const int pair = tunable("BILSTM_PAIR", 1);
    if (pair && lstm_step_is_small(b, h)) {
        int cur = 0;
        for (int st = 0; st < t; ++st) {
            int rc = launch_lstm_small_pair(step_args(0, st, cur), step_args(1, st, cur), s);
            if (rc) return rc;
            cur ^= 1;
        }
    } else {
        for (int d = 0; d < 2; ++d) {
            int cur = 0;
            for (int st = 0; st < t; ++st) {
                int rc = launch_lstm_step(step_args(d, st, cur), s);
                if (rc) return rc;
                cur ^= 1;
            }
        }
    }
    return after.finish(out_p ? fcl_pack_planes(out, 2 * h, b * t, 2 * h, out_p, stream) : 0);  // the per-step path writes fp32 only: split once at the end
}

}  // extern "C"
