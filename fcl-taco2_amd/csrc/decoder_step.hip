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
#include "lstm_epilogue.h"

namespace fcl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// acc[t] = A_l[16 x K] . W_t[16 rows x K]^T for NT column tiles that share the A fragments.
// A_l: LDS, row stride lda_l floats.  wrow[t] = &W[(n_t + r16) * ldw] — ALWAYS a valid row (callers clamp the row
// index and discard the surplus columns), so every load is unconditional and stays in flight (guide §5 trap (c)).
// W fragments are fetched one 64-k chunk ahead of the MFMAs that consume them.
// rot: workgroups start the K walk at different 64-k chunks (chunk index rotated by rot) so that the ~150 workgroups
// of a launch do not all request the same weight lines from the same L2 channels at the same moment.
template <int NT>
__device__ __forceinline__ void rowtile_mma(const float* A_l, int lda_l, const float* const (&wrow)[NT], int K, int r16, int kq,
                                            f32x4 (&out)[NT], int rot = 0) {
    f32x4 acc0[NT], acc1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc0[t] = acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ap = A_l + r16 * lda_l + kq * 4;
    const float* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = wrow[t] + kq * 4;
    const int nfull = K >> 6;  // whole 64-k chunks
    f32x4 bn[NT][4];
    int cc = nfull > 0 ? rot % nfull : 0;
    if (nfull > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) bn[t][i] = *reinterpret_cast<const f32x4*>(wp[t] + (cc << 6) + i * 16);
    }
    for (int c = 0; c < nfull; ++c) {
        const int k = cc << 6;
        const int cn = cc + 1 == nfull ? 0 : cc + 1;
        f32x4 b[NT][4], a[4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) b[t][i] = bn[t][i];
        if (c + 1 < nfull) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) bn[t][i] = *reinterpret_cast<const f32x4*>(wp[t] + (cn << 6) + i * 16);
        }
        cc = cn;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(ap + k + i * 16);
        __builtin_amdgcn_sched_barrier(0);  // keep the next chunk's W loads ABOVE this chunk's MFMAs (hipcc sinks them otherwise)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][0], b[t][i][0], acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][1], b[t][i][1], acc1[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][2], b[t][i][2], acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][3], b[t][i][3], acc1[t], 0, 0, 0);
            }
        }
    }
    for (int k = nfull << 6; k < K; k += 16) {  // tail: K % 4 == 0, lanes past K contribute zeros
        const bool in = k + kq * 4 < K;
        // lanes past K: keep the address valid (the first float4 of the row -- `wp + 0` would still carry the lane's kq * 4 offset, i.e. up to 12 floats
        // past the end of a row shorter than 16, and past the end of the MATRIX in its last row: a fault when the weight ends a mapped segment) and
        // zero the operand instead of skipping the load
        f32x4 a = *reinterpret_cast<const f32x4*>(in ? ap + k : A_l + r16 * lda_l);
        if (!in) a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(in ? wp[t] + k : wrow[t]);
            acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], bb[0], acc0[t], 0, 0, 0);
            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], bb[1], acc1[t], 0, 0, 0);
            acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], bb[2], acc0[t], 0, 0, 0);
            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], bb[3], acc1[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = acc0[t] + acc1[t];
}

// cooperative load of a [16 x K] row tile (rows m0.., zero beyond M) into LDS with row stride ld_l
__device__ __forceinline__ void load_rowtile(float* dst, int ld_l, const float* src, int ld, int K, int m0, int M) {
    const int per_row = K >> 2;
    for (int i = threadIdx.x; i < 16 * per_row; i += blockDim.x) {
        const int r = i / per_row, c = (i - r * per_row) * 4;
        const int m = min(m0 + r, M - 1);  // clamp (M >= 1): rows past M are computed on a copy and never stored
        *reinterpret_cast<f32x4*>(dst + r * ld_l + c) = *reinterpret_cast<const f32x4*>(src + (size_t)m * ld + c);
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
    // rows of this launch: the host's counts, or -- device-driven loop -- the smaller of the host's bounds and the device's live counts
    // device-driven loops: LOADS and the MFMA chains use the host's row bounds (rows between the device count and the bound hold stale but finite state);
    // only the STORES are limited to the device's live-row counts -- their scalar loads are then off the kernel's critical path (first use: an epilogue)
    const int M_feat = a.M_feat, M_pre = a.M_pre;
    const int Ms_feat = (a.live && a.t_prev >= 0) ? min(a.M_feat, a.live[a.t_prev]) : a.M_feat, Ms_pre = a.live ? min(a.M_pre, a.live[a.t_cur]) : a.M_pre;
    if (a.live && a.w0 && blockIdx.x == 0 && threadIdx.x == 0 && a.live[a.t_cur] > a.M_pre) atomicOr(a.status, (unsigned int)FCL_STATUS_ROWS_CAP);
    if ((int)blockIdx.x * (16) >= (a.h1 ? Ms_feat : Ms_pre)) return;  // tile beyond the device's live rows (uniform per workgroup, before any barrier)
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
        load_rowtile(A1, ldU, a.h1, a.U, a.U, m0, M_feat);
        __syncthreads();
        for (int tile = wave; tile * 16 < a.O; tile += nwaves) {
            const float* const wr[1] = {a.wf_h + (size_t)min(tile * 16 + r16, a.O - 1) * a.U};
            const int nc = tile * 16 + col, ncc = min(nc, a.O - 1);
            float f0v[4];
            int fo[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // epilogue operands fetched before the MFMA chain (clamped, unconditional)
                const int mc = min(m0 + rq * 4 + r, M_feat - 1);
                f0v[r] = a.F0[(size_t)mc * a.O + ncc];
                fo[r] = a.frame_off[mc];
            }
            f32x4 accv[1];
            rowtile_mma<1>(A1, ldU, wr, a.U, r16, kq, accv, blockIdx.x);
            const f32x4 acc = accv[0];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rq * 4 + r, m = m0 + row;
                float v = 0.f;
                if (m < Ms_feat && nc < a.O) {
                    v = acc[r] + f0v[r];
                    a.before[(size_t)(fo[r] + a.t_prev) * a.O + nc] = v;
                }
                if (nc < a.O) A2[row * ldO + nc] = (a.out_act && m < Ms_feat) ? act_apply(v, a.out_act) : v;
            }
        }
    } else {
        for (int i = threadIdx.x; i < 16 * ldO; i += blockDim.x) A2[i] = 0.f;  // prev_out = 0 at t = 0
    }
    if (!a.w0 || m0 >= M_pre) return;  // last step: feat only / rows that just finished
    __syncthreads();
    if (a.teacher_in) {  // teacher forcing: prenet input is y_{t-1}, not the decoder's own output
        load_rowtile(A2, ldO, a.teacher_in, a.teacher_ld, a.O, m0, M_pre);
        __syncthreads();
    }
    // ---- H6 prenet layer 0 ---------------------------------------------------------------------------
    for (int tile = wave; tile * 16 < a.P; tile += 2 * nwaves) {
        const int tile2 = tile + nwaves;
        const float* const wr[2] = {a.w0 + (size_t)min(tile * 16 + r16, a.P - 1) * a.O, a.w0 + (size_t)min(tile2 * 16 + r16, a.P - 1) * a.O};
        f32x4 accv[2];
        rowtile_mma<2>(A2, ldO, wr, a.O, r16, kq, accv, blockIdx.x);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? tile2 : tile) * 16 + col;
            if (nc < a.P) {
                const float bn = a.b0[nc];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rq * 4 + r, m = m0 + row;
                    float v = fmaxf(accv[tt][r] + bn, 0.f);
                    if (m < M_pre) v = drop_apply(v, a.drop_mode, a.keep0, a.P, m, nc, a.P, a.keep_scale, a.drop_p, seed0);
                    A3[row * ldP + nc] = v;
                }
            }
        }
    }
    __syncthreads();
    // ---- H6 prenet layer 1 -> global (+ KD tap) --------------------------------------------------------
    for (int tile = wave; tile * 16 < a.P; tile += 2 * nwaves) {
        const int tile2 = tile + nwaves;
        const float* const wr[2] = {a.w1 + (size_t)min(tile * 16 + r16, a.P - 1) * a.P, a.w1 + (size_t)min(tile2 * 16 + r16, a.P - 1) * a.P};
        f32x4 accv[2];
        rowtile_mma<2>(A3, ldP, wr, a.P, r16, kq, accv, blockIdx.x);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? tile2 : tile) * 16 + col;
            if (nc < a.P) {
                const float bn = a.b1[nc];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + rq * 4 + r;
                    if (m >= Ms_pre) continue;
                    float v = fmaxf(accv[tt][r] + bn, 0.f);
                    v = drop_apply(v, a.drop_mode, a.keep1, a.P, m, nc, a.P, a.keep_scale, a.drop_p, seed1);
                    a.pre_out[(size_t)m * a.P + nc] = v;
                    if (a.tap_prenet) a.tap_prenet[(size_t)(a.frame_off[m] + a.t_cur) * a.P + nc] = v;
                }
            }
        }
    }
}

// =====================================================================================================
// bf16x3 variants of the two small-tile kernels: activations are split (hi/lo bf16 planes) when they enter LDS,
// weights arrive pre-split (fcl_split_bf16 at plan time) and stream from L2 straight into 16x16x32 bf16 MFMA fragments:
// 3 MFMAs of 16 cycles per 32-k step instead of 8 fp32 MFMAs of 32 cycles, same fp32 accumulators and epilogues.
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ void split1(float x, u16& hi, u16& lo) {
    const __bf16 h = (__bf16)x;
    hi = __builtin_bit_cast(u16, h);
    lo = __builtin_bit_cast(u16, (__bf16)(x - (float)h));
}

// A planes: Ah/Al [16][ldk] bf16 in LDS.  fhi/flo[t]: this lane's slot in the first fragment block of column tile t
// (fragment-major planes, fcl_pack_frag_bf16): step st lives 512 elements further, zero-padded past K.
template <int NT>
__device__ __forceinline__ void rowtile_mma_x3(const u16* Ah, const u16* Al, int ldk, const u16* const (&fhi)[NT], const u16* const (&flo)[NT],
                                               int K, int r16, int kq, f32x4 (&out)[NT]) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u16* ah = Ah + r16 * ldk + kq * 8;
    const u16* al = Al + r16 * ldk + kq * 8;
    const int nsteps = (K + 31) >> 5;
    s16x8 bhn[NT], bln[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bhn[t] = *reinterpret_cast<const s16x8*>(fhi[t]);
        bln[t] = *reinterpret_cast<const s16x8*>(flo[t]);
    }
    for (int st = 0; st < nsteps; ++st) {
        const int k = st << 5;
        s16x8 bh[NT], bl[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { bh[t] = bhn[t]; bl[t] = bln[t]; }
        if (st + 1 < nsteps) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bhn[t] = *reinterpret_cast<const s16x8*>(fhi[t] + (size_t)(st + 1) * 512);
                bln[t] = *reinterpret_cast<const s16x8*>(flo[t] + (size_t)(st + 1) * 512);
            }
        }
        s16x8 a_hi = {0, 0, 0, 0, 0, 0, 0, 0}, a_lo = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k + kq * 8 < K) {  // K % 8 == 0; the LDS row holds exactly K valid elements
            a_hi = *reinterpret_cast<const s16x8*>(ah + k);
            a_lo = *reinterpret_cast<const s16x8*>(al + k);
        }
        __builtin_amdgcn_sched_barrier(0);  // next step's W loads stay above this step's MFMAs
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, bh[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, bl[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, bh[t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = acc[t];
}

// Register-resident W fragments: the weights do not depend on the activation chain, so a wave can request ALL the
// fragments it will need (NS 32-k steps x NT column tiles x 2 planes) up front and pay the L2/Infinity-Cache latency
// once per kernel instead of once per k-step; the MFMAs then issue back to back from registers.
template <int NT, int NS>
struct WFrag {
    s16x8 hi[NT][NS], lo[NT][NS];
    __device__ __forceinline__ void load(const u16* const (&fhi)[NT], const u16* const (&flo)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int st = 0; st < NS; ++st) {  // every wave-wide load is one contiguous 1 KB fragment block
                hi[t][st] = *reinterpret_cast<const s16x8*>(fhi[t] + st * 512);
                lo[t][st] = *reinterpret_cast<const s16x8*>(flo[t] + st * 512);
            }
    }
    // CHECK = false: the LDS rows hold NS * 32 valid (or zero-padded) elements, so no lane needs masking and the loop is branch-free
    template <bool CHECK = true>
    __device__ __forceinline__ void mma(const u16* Ah, const u16* Al, int ldk, int K, int r16, int kq, f32x4 (&out)[NT]) const {
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const u16* ah = Ah + r16 * ldk + kq * 8;
        const u16* al = Al + r16 * ldk + kq * 8;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            s16x8 a_hi = {0, 0, 0, 0, 0, 0, 0, 0}, a_lo = {0, 0, 0, 0, 0, 0, 0, 0};
            if (!CHECK || st * 32 + kq * 8 < K) {
                a_hi = *reinterpret_cast<const s16x8*>(ah + st * 32);
                a_lo = *reinterpret_cast<const s16x8*>(al + st * 32);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, hi[t][st], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, lo[t][st], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, hi[t][st], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) out[t] = acc[t];
    }
};

// cooperative load of a [16 x K] fp32 row tile, split into the two LDS planes (rows clamped to M-1)
__device__ __forceinline__ void load_rowtile_split(u16* Ah, u16* Al, int ldk, const float* src, int ld, int K, int m0, int M) {
    const int per_row = K >> 2;
    for (int i = threadIdx.x; i < 16 * per_row; i += blockDim.x) {
        const int r = i / per_row, c = (i - r * per_row) * 4;
        const int m = min(m0 + r, M - 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)m * ld + c);
        u16 h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split1(v[e], h[e], l[e]);
        *reinterpret_cast<uint2*>(Ah + r * ldk + c) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        *reinterpret_cast<uint2*>(Al + r * ldk + c) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
    }
}

// Workgroup barrier for data exchanged through LDS only: LDS accesses retired (lgkmcnt), then s_barrier.  Unlike __syncthreads() it leaves
// global loads in flight (vmcnt untouched) — the register-resident weight fragments requested ahead of their phase.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// SU/SO/SP = compile-time 32-k step counts of U/O/P (weights preloaded into registers); 0 = generic streaming loops.
template <int SU, int SO, int SP>
__global__ __launch_bounds__(512) void feat_prenet_x3_kernel(const FeatPrenetArgs a) {
    // rows of this launch: the host's counts, or -- device-driven loop -- the smaller of the host's bounds and the device's live counts
    // device-driven loops: LOADS and the MFMA chains use the host's row bounds (rows between the device count and the bound hold stale but finite state);
    // only the STORES are limited to the device's live-row counts -- their scalar loads are then off the kernel's critical path (first use: an epilogue)
    const int M_feat = a.M_feat, M_pre = a.M_pre;
    const int Ms_feat = (a.live && a.t_prev >= 0) ? min(a.M_feat, a.live[a.t_prev]) : a.M_feat, Ms_pre = a.live ? min(a.M_pre, a.live[a.t_cur]) : a.M_pre;
    if (a.live && a.w0 && blockIdx.x == 0 && threadIdx.x == 0 && a.live[a.t_cur] > a.M_pre) atomicOr(a.status, (unsigned int)FCL_STATUS_ROWS_CAP);
    if ((int)blockIdx.x * (16) >= (a.h1 ? Ms_feat : Ms_pre)) return;  // tile beyond the device's live rows (uniform per workgroup, before any barrier)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ldU = a.U + 8, ldO = a.O + 8, ldP = a.P + 8;  // bf16 elements per plane row
    u16* A1h = reinterpret_cast<u16*>(smem);
    u16* A1l = A1h + 16 * ldU;
    u16* A2h = A1l + 16 * ldU;
    u16* A2l = A2h + 16 * ldO;
    u16* A3h = A2l + 16 * ldO;
    u16* A3l = A3h + 16 * ldP;
    const int m0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int col = lane & 15, rq = lane >> 4;
    const unsigned int sbump = a.seed_dev ? *a.seed_dev * 0x9E3779B9u : 0u;
    const unsigned int seed0 = hash_u32(a.seed0 + sbump), seed1 = hash_u32(a.seed1 + sbump);

    constexpr bool PRE = SU > 0;
    // weight fragments for the prenet layers are requested first thing: they are consumed two barriers later
    WFrag<2, PRE ? SO : 1> f0;
    WFrag<2, PRE ? SP : 1> f1;
    const bool has_pre = a.w0 && m0 < M_pre;
    const int ptile = wave, ptile2 = wave + nwaves;  // P/16 <= 2*nwaves is checked by the launcher for the PRE path
    const int nsU = (a.U + 31) >> 5, nsO = (a.O + 31) >> 5, nsP = (a.P + 31) >> 5, ptmax = ((a.P + 15) >> 4) - 1;
    const size_t lane8 = (size_t)lane * 8;
    auto frag = [&](const u16* base, int tile, int ns) { return base + (size_t)tile * ns * 512 + lane8; };
    // epilogue biases of this wave's two prenet column tiles: requested now, consumed two / three barriers later (they used to be dependent
    // global loads behind each layer's MFMA chain: ~1 us of exposed latency each on the step's critical path)
    float pb0[2] = {0.f, 0.f}, pb1[2] = {0.f, 0.f};
    if (has_pre) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = min((tt ? ptile2 : ptile) * 16 + col, a.P - 1);
            pb0[tt] = a.b0[nc];
            pb1[tt] = a.b1[nc];
        }
    }
    if (PRE && has_pre) {
        const u16* const wh0[2] = {frag(a.w0_hi, ptile, nsO), frag(a.w0_hi, min(ptile2, ptmax), nsO)};
        const u16* const wl0[2] = {frag(a.w0_lo, ptile, nsO), frag(a.w0_lo, min(ptile2, ptmax), nsO)};
        f0.load(wh0, wl0);
    }
    if (a.h1) {
        WFrag<1, PRE ? SU : 1> ff;
        if (PRE && wave * 16 < a.O) {
            const u16* const wh[1] = {frag(a.wf_hi, wave, nsU)};
            const u16* const wl[1] = {frag(a.wf_lo, wave, nsU)};
            ff.load(wh, wl);
        }
        load_rowtile_split(A1h, A1l, ldU, a.h1, a.U, a.U, m0, M_feat);
        __syncthreads();
        if (a.dbg_phase == 1) return;
        for (int tile = wave; tile * 16 < a.O; tile += nwaves) {
            const u16* const wh[1] = {frag(a.wf_hi, tile, nsU)};
            const u16* const wl[1] = {frag(a.wf_lo, tile, nsU)};
            const int nc = tile * 16 + col, ncc = min(nc, a.O - 1);
            float f0v[4];
            int fo[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mc = min(m0 + rq * 4 + r, M_feat - 1);
                f0v[r] = a.F0[(size_t)mc * a.O + ncc];
                fo[r] = a.frame_off[mc];
            }
            f32x4 accv[1];
            if (PRE) ff.mma(A1h, A1l, ldU, a.U, r16, kq, accv);
            else rowtile_mma_x3<1>(A1h, A1l, ldU, wh, wl, a.U, r16, kq, accv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rq * 4 + r, m = m0 + row;
                float v = 0.f;
                if (m < Ms_feat && nc < a.O) {
                    v = accv[0][r] + f0v[r];
                    a.before[(size_t)(fo[r] + a.t_prev) * a.O + nc] = v;
                    if (a.before_p) store_p32(a.before_p, (a.O + 31) >> 5, fo[r] + a.t_prev, nc, v);
                }
                if (nc < a.O) split1((a.out_act && m < Ms_feat) ? act_apply(v, a.out_act) : v, A2h[row * ldO + nc], A2l[row * ldO + nc]);
            }
        }
        if (a.before_p && (a.O & 31)) {  // zero padding of the last 32-column line of this tile's frames
            const int padc = 32 - (a.O & 31);
            for (int i = threadIdx.x; i < 16 * padc; i += blockDim.x) {
                const int m = m0 + i / padc;
                if (m < Ms_feat) store_p32(a.before_p, (a.O + 31) >> 5, a.frame_off[m] + a.t_prev, a.O + i % padc, 0.f);
            }
        }
    } else {
        for (int i = threadIdx.x; i < 16 * ldO; i += blockDim.x) { A2h[i] = 0; A2l[i] = 0; }  // prev_out = 0 at t = 0
    }
    if (!has_pre || a.dbg_phase == 2) return;
    if (PRE) {  // layer-1 fragments: in flight while layer 0 computes
        const u16* const wh1[2] = {frag(a.w1_hi, ptile, nsP), frag(a.w1_hi, min(ptile2, ptmax), nsP)};
        const u16* const wl1[2] = {frag(a.w1_lo, ptile, nsP), frag(a.w1_lo, min(ptile2, ptmax), nsP)};
        f1.load(wh1, wl1);
    }
    lds_barrier();  // NOT __syncthreads(): its vmcnt(0) would park every wave until the 32 fragment loads just issued have landed
    if (a.teacher_in) {
        load_rowtile_split(A2h, A2l, ldO, a.teacher_in, a.teacher_ld, a.O, m0, M_pre);
        __syncthreads();
    }
    for (int tile = wave; tile * 16 < a.P; tile += 2 * nwaves) {
        const int tile2 = tile + nwaves;
        const u16* const wh[2] = {frag(a.w0_hi, tile, nsO), frag(a.w0_hi, min(tile2, ptmax), nsO)};
        const u16* const wl[2] = {frag(a.w0_lo, tile, nsO), frag(a.w0_lo, min(tile2, ptmax), nsO)};
        f32x4 accv[2];
        if (PRE) f0.mma(A2h, A2l, ldO, a.O, r16, kq, accv);
        else rowtile_mma_x3<2>(A2h, A2l, ldO, wh, wl, a.O, r16, kq, accv);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? tile2 : tile) * 16 + col;
            if (nc < a.P) {
                const float bn = tile == ptile ? pb0[tt] : a.b0[nc];  // first pass of the tile loop: the prefetched value
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rq * 4 + r, m = m0 + row;
                    float v = fmaxf(accv[tt][r] + bn, 0.f);
                    if (m < M_pre) v = drop_apply(v, a.drop_mode, a.keep0, a.P, m, nc, a.P, a.keep_scale, a.drop_p, seed0);
                    split1(v, A3h[row * ldP + nc], A3l[row * ldP + nc]);
                }
            }
        }
    }
    lds_barrier();
    if (a.dbg_phase == 3) return;
    for (int tile = wave; tile * 16 < a.P; tile += 2 * nwaves) {
        const int tile2 = tile + nwaves;
        const u16* const wh[2] = {frag(a.w1_hi, tile, nsP), frag(a.w1_hi, min(tile2, ptmax), nsP)};
        const u16* const wl[2] = {frag(a.w1_lo, tile, nsP), frag(a.w1_lo, min(tile2, ptmax), nsP)};
        f32x4 accv[2];
        if (PRE) f1.mma(A3h, A3l, ldP, a.P, r16, kq, accv);
        else rowtile_mma_x3<2>(A3h, A3l, ldP, wh, wl, a.P, r16, kq, accv);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? tile2 : tile) * 16 + col;
            if (nc < a.P) {
                const float bn = tile == ptile ? pb1[tt] : a.b1[nc];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + rq * 4 + r;
                    if (m >= Ms_pre) continue;
                    float v = fmaxf(accv[tt][r] + bn, 0.f);
                    v = drop_apply(v, a.drop_mode, a.keep1, a.P, m, nc, a.P, a.keep_scale, a.drop_p, seed1);
                    if (a.pre_out) a.pre_out[(size_t)m * a.P + nc] = v;
                    if (a.pre_out_p) store_p32(a.pre_out_p, (a.P + 31) >> 5, m, nc, v);
                    if (a.tap_prenet) a.tap_prenet[(size_t)(a.frame_off[m] + a.t_cur) * a.P + nc] = v;
                }
            }
        }
    }
}

// ---- round 4: the same chain with TRANSPOSED accumulators and (optionally) prenet layer 1's columns split over NS workgroups per row tile.
// Where the round-3 kernel's (feat_prenet_fast_kernel, deleted in round 5) 14 us went (r4 phase stamps, 2 400 rows: launch floor 3.6, feat_out +2.5..3.4, layer 0 +3.0..3.5, layer 1 +3.6..3.9):
// not into the weight stream alone -- a first column-split form that only cut the fragments per workgroup from 442 to 245 KB ran exactly as long --
// but into the per-ELEMENT epilogues.  With the activations as the MFMA's A operand a lane ends up with ONE column of FOUR rows: every finished
// element then costs its own bias / mask load, its own counter hash, its own fp32 -> (hi, lo) split, two 2-byte LDS stores and up to five scattered
// 2- / 4-byte global stores with 64-bit address arithmetic: ~30 VALU + ~5 memory instructions per element, 16 - 32 elements per lane and phase, two
// waves per SIMD.  Swapping the operands (weights as A, activations as B: the same fragments, the same products in the same order, D^T instead of D)
// gives a lane FOUR CONSECUTIVE COLUMNS of ONE row: one float4 bias / F0 load, one 4-byte mask load, two hashes (16 bits per decision), one packed
// split (v_cvt_pk_bf16_f32), two 8-byte LDS stores = the next layer's operand, and 8- / 16-byte global stores (the P32 line's hi and lo halves,
// the fp32 frame) -- about a quarter of the instructions.
// Column split NS (gridDim.y): a workgroup owns 16 RT rows x 256 / NS columns of layer 1 and recomputes feat_out and layer 0 for its rows (61 k of
// the 127 k weights): 82 + 98 + 262 / NS KB of fragments per workgroup, NS x the workgroups; split 0 alone stores feat_out.  NS = 1: no split.
// The layer-1 operand tile re-uses the h1 tile's LDS (96 KB at RT = 4).  All fragments are requested at entry (layer 1's once the h1 staging
// registers are free); no __syncthreads (its vmcnt(0) would wait for every fragment before the first phase).
// RNG mode draws a different (equally distributed) stream than the other feat/prenet kernels: 16 bits of a counter hash per decision instead of 24.
__device__ __forceinline__ void p32_store4(unsigned short* __restrict__ p, int ld, long long m, int n, uint2 hi, uint2 lo) {
    unsigned short* line = p + ((size_t)m * ld + (n >> 5)) * 64 + (n & 31);
    *reinterpret_cast<uint2*>(line) = hi;
    *reinterpret_cast<uint2*>(line + 32) = lo;
}

template <int NT, int NS_>
__device__ __forceinline__ void mma_t(const WFrag<NT, NS_>& w, const u16* Ah, const u16* Al, int ldk, int r16, int kq, f32x4 (&out)[NT]) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u16* ah = Ah + r16 * ldk + kq * 8;
    const u16* al = Al + r16 * ldk + kq * 8;
#pragma unroll
    for (int st = 0; st < NS_; ++st) {
        const s16x8 a_hi = *reinterpret_cast<const s16x8*>(ah + st * 32);
        const s16x8 a_lo = *reinterpret_cast<const s16x8*>(al + st * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi[t][st], a_lo, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.lo[t][st], a_hi, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi[t][st], a_hi, acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = acc[t];
}

template <int DROP, int RT, int NS>
__global__ __launch_bounds__(512) void feat_prenet_split_kernel(const FeatPrenetArgs a) {
    constexpr int SU = 8, SO = 3, SP = 8, U = 256, OP = 96, P = 256, CT = 16 / NS, NW1 = CT < 8 ? CT : 8, T1 = CT / NW1;
    static_assert(NS == 1 || NS == 2 || NS == 4 || NS == 8, "layer-1 column split");
    const int M_feat = a.M_feat, M_pre = a.M_pre;
    const int Ms_feat = (a.live && a.t_prev >= 0) ? min(a.M_feat, a.live[a.t_prev]) : a.M_feat, Ms_pre = a.live ? min(a.M_pre, a.live[a.t_cur]) : a.M_pre;
    const int split = blockIdx.y;
    if (a.live && a.w0 && blockIdx.x == 0 && split == 0 && threadIdx.x == 0 && a.live[a.t_cur] > a.M_pre) atomicOr(a.status, (unsigned int)FCL_STATUS_ROWS_CAP);
    if ((int)blockIdx.x * (16 * RT) >= (a.h1 ? Ms_feat : Ms_pre)) return;  // tile beyond the device's live rows (uniform per workgroup, before any barrier)
    constexpr int ldU = U + 16, ldO = OP + 16, ldP = P + 16, ROWS = 16 * RT;
    static_assert(ldU == ldP, "the layer-1 operand tile re-uses the h1 tile's LDS");
    extern __shared__ __attribute__((aligned(16))) u16 fp_lds[];
    u16* A1h = fp_lds;
    u16* A1l = A1h + ROWS * ldU;
    u16* A2h = A1l + ROWS * ldU;
    u16* A2l = A2h + ROWS * ldO;
    u16* A3h = A1h;  // written in phase 2, when every wave is past its last read of the h1 tile (the barrier that ends phase 1)
    u16* A3l = A1l;
    const int O = a.O;
    const int m0 = blockIdx.x * ROWS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;   // operand fragments: row / column r16, k-group kq
    const int arow = lane & 15, cq = lane >> 4;  // transposed accumulators: activation row arow, columns cq * 4 .. cq * 4 + 3 of the column tile
    const bool has_feat = a.h1 != nullptr, has_pre = a.w0 != nullptr && m0 < M_pre;
    if (!has_pre && split != 0) return;  // feat-only launch (after the last step): one workgroup per row tile does it
    const bool feat_wave = has_feat && wave * 16 < O;
    const bool l1_wave = has_pre && wave >= 8 - NW1;
    const size_t lane8 = (size_t)lane * 8;
    auto frag = [&](const u16* base, int tile, int ns) { return base + (size_t)tile * ns * 512 + lane8; };
    const int t0 = wave, t1 = wave + 8;                   // this wave's layer-0 column tiles
    const int c1 = split * CT + (wave - (8 - NW1)) * T1;  // first of this wave's T1 layer-1 column tiles (l1_wave only)

    // ---- phase 0 operands first (loads return in order: the first wait then covers the h1 rows alone) ---------------------------------------
    constexpr int HV = RT * 2;  // float4 loads of h1 per thread: ROWS x 64 float4 over 512 threads
    f32x4 hreg[HV];
    if (has_feat) {
#pragma unroll
        for (int j = 0; j < HV; ++j) {
            const int i = threadIdx.x + j * 512, r = i >> 6, c = (i & 63) * 4;
            hreg[j] = *reinterpret_cast<const f32x4*>(a.h1 + (size_t)min(m0 + r, M_feat - 1) * U + c);
        }
    }
    // The CU's address path serves its 8 waves' requests in ISSUE order (one 1 KB wave-load per 16 clocks): a wave that runs ahead and queues its 30 - 44
    // fragment loads puts them in front of the other waves' h1 loads, and phase 0 ends at a barrier -- so the phase used to wait for ~2 us of weight
    // traffic it does not need (r4).  An s_barrier (no data wait: ~100 clocks) between the request groups makes the queue's order the order of use:
    // every wave's h1 rows, then feat_out's fragments, then layer 0's, then layer 1's.
    asm volatile("s_barrier" ::: "memory");
    // ---- weight fragments and epilogue operands of phases 1 and 2, requested now ------------------------------------------------------------------
    WFrag<1, SU> ff;
    WFrag<2, SO> f0;
    WFrag<T1, SP> f1;
    const int fnc = wave * 16 + cq * 4;  // feat_out columns fnc .. fnc + 3 of this lane (whole groups of four: O % 4 == 0)
    f32x4 f0v[RT];
    int fo[RT];
    if (feat_wave) {
        const u16* const wh[1] = {frag(a.wf_hi, wave, SU)};
        const u16* const wl[1] = {frag(a.wf_lo, wave, SU)};
        ff.load(wh, wl);
    }
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        f0v[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        fo[q] = 0;
        if (feat_wave && fnc < O) {
            const int mc = min(m0 + q * 16 + arow, M_feat - 1);
            f0v[q] = *reinterpret_cast<const f32x4*>(a.F0 + (size_t)mc * O + fnc);
            if (split == 0) fo[q] = a.frame_off[mc];
        }
    }
    unsigned int seed0 = 0, seed1 = 0;
    const unsigned int thr16 = (unsigned int)(a.drop_p * 65536.0f);
    if (DROP == 2) {
        const unsigned int sbump = a.seed_dev ? *a.seed_dev * 0x9E3779B9u : 0u;
        seed0 = hash_u32(a.seed0 + sbump);
        seed1 = hash_u32(a.seed1 + sbump);
    }
    f32x4 pb0[2], pb1[T1];
    unsigned int k0[RT][2], k1[RT][T1];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) pb0[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (has_feat) asm volatile("s_barrier" ::: "memory");  // (feat_out's fragments are queued before anybody's layer-0 fragments)
    if (has_pre) {
        const u16* const wh0[2] = {frag(a.w0_hi, t0, SO), frag(a.w0_hi, t1, SO)};
        const u16* const wl0[2] = {frag(a.w0_lo, t0, SO), frag(a.w0_lo, t1, SO)};
        f0.load(wh0, wl0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? t1 : t0) * 16 + cq * 4;
            pb0[tt] = *reinterpret_cast<const f32x4*>(a.b0 + nc);
            if (DROP == 1) {
#pragma unroll
                for (int q = 0; q < RT; ++q) k0[q][tt] = *reinterpret_cast<const unsigned int*>(a.keep0 + (size_t)min(m0 + q * 16 + arow, M_pre - 1) * P + nc);
            }
        }
    }
    // ---- phase 0: h1 tiles -> LDS planes; prenet-input tiles zeroed (prev_out = 0 at t = 0; zero padding past O otherwise) ------------------
    for (int i = threadIdx.x; i < ROWS * ldO / 8; i += 512) {
        reinterpret_cast<uint4*>(A2h)[i] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(A2l)[i] = make_uint4(0, 0, 0, 0);
    }
    if (has_feat) {
#pragma unroll
        for (int j = 0; j < HV; ++j) {
            const int i = threadIdx.x + j * 512, r = i >> 6, c = (i & 63) * 4;
            uint2 h, l;
            split4(hreg[j], h, l);
            *reinterpret_cast<uint2*>(A1h + r * ldU + c) = h;
            *reinterpret_cast<uint2*>(A1l + r * ldU + c) = l;
        }
    }
    // layer-1 fragments: requested once the h1 staging registers are free (they are consumed last; the stream behind feat_out / layer 0 is
    // unbroken) -- NS = 1 (all 16 column tiles: 128 VGPRs of fragments): once feat_out's fragments are dead too, i.e. after phase 1
#pragma unroll
    for (int tt = 0; tt < T1; ++tt) pb1[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto request_l1 = [&]() {
        if (l1_wave) {
            const u16* wh1[T1];
            const u16* wl1[T1];
#pragma unroll
            for (int tt = 0; tt < T1; ++tt) {
                wh1[tt] = frag(a.w1_hi, c1 + tt, SP);
                wl1[tt] = frag(a.w1_lo, c1 + tt, SP);
            }
            f1.load(wh1, wl1);
#pragma unroll
            for (int tt = 0; tt < T1; ++tt) {
                const int nc = (c1 + tt) * 16 + cq * 4;
                pb1[tt] = *reinterpret_cast<const f32x4*>(a.b1 + nc);
                if (DROP == 1) {
#pragma unroll
                    for (int q = 0; q < RT; ++q) k1[q][tt] = *reinterpret_cast<const unsigned int*>(a.keep1 + (size_t)min(m0 + q * 16 + arow, M_pre - 1) * P + nc);
                }
            }
        }
    };
    if (NS > 1) {
        asm volatile("s_barrier" ::: "memory");  // (layer 0's fragments are queued before anybody's layer-1 fragments)
        request_l1();
    }
    lds_barrier();
    if (a.dbg_phase == 1) return;
    // ---- phase 1: H8 feat_out of the previous step (+ H10 scatter by split 0) -----------------------------------------------------------------
    if (feat_wave) {
#pragma unroll
        for (int q = 0; q < RT; ++q) {
            f32x4 accv[1];
            mma_t<1, SU>(ff, A1h + q * 16 * ldU, A1l + q * 16 * ldU, ldU, r16, kq, accv);
            const int row = q * 16 + arow, m = m0 + row;
            if (fnc < O) {
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                uint2 h, l;
                if (m < Ms_feat) {
                    v = accv[0] + f0v[q];
                    if (split == 0) {
                        const long long fr = (long long)fo[q] + a.t_prev;
                        *reinterpret_cast<f32x4*>(a.before + (size_t)fr * O + fnc) = v;
                        if (a.before_p) {
                            split4(v, h, l);
                            p32_store4(a.before_p, (O + 31) >> 5, fr, fnc, h, l);
                        }
                    }
                    if (a.out_act) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], a.out_act);
                    }
                }
                split4(v, h, l);
                *reinterpret_cast<uint2*>(A2h + row * ldO + fnc) = h;
                *reinterpret_cast<uint2*>(A2l + row * ldO + fnc) = l;
            }
        }
    }
    if (has_feat && split == 0 && a.before_p && (O & 31)) {  // zero padding of the last 32-column line of this tile's frames
        const int padc = 32 - (O & 31);
        for (int i = threadIdx.x; i < ROWS * padc; i += 512) {
            const int m = m0 + i / padc;
            if (m < Ms_feat) store_p32(a.before_p, (O + 31) >> 5, a.frame_off[m] + a.t_prev, O + i % padc, 0.f);
        }
    }
    if (!has_pre || a.dbg_phase == 2) return;
    if (NS == 1) request_l1();
    lds_barrier();
    if (a.teacher_in) {  // teacher forcing: prenet input is y_{t-1}, not the decoder's own output
#pragma unroll
        for (int q = 0; q < RT; ++q) load_rowtile_split(A2h + q * 16 * ldO, A2l + q * 16 * ldO, ldO, a.teacher_in, a.teacher_ld, O, m0 + q * 16, M_pre);
        lds_barrier();
    }
    // ---- phase 2: H6 prenet layer 0 (all 256 columns; recomputed by every column split) ----------------------------------------------------------
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        f32x4 accv[2];
        mma_t<2, SO>(f0, A2h + q * 16 * ldO, A2l + q * 16 * ldO, ldO, r16, kq, accv);
        const int row = q * 16 + arow, m = m0 + row;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? t1 : t0) * 16 + cq * 4;
            f32x4 v = accv[tt] + pb0[tt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            v = drop4<DROP>(v, DROP == 1 ? k0[q][tt] : 0u, (unsigned int)m * (unsigned int)P + (unsigned int)nc, seed0, thr16, a.keep_scale);
            uint2 h, l;
            split4(v, h, l);
            *reinterpret_cast<uint2*>(A3h + row * ldP + nc) = h;
            *reinterpret_cast<uint2*>(A3l + row * ldP + nc) = l;
        }
    }
    lds_barrier();
    if (a.dbg_phase == 3 || !l1_wave) return;
    // ---- phase 3: H6 prenet layer 1, this split's columns -> global (+ KD tap, + P32 planes) --------------------------------------------------
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        f32x4 accv[T1];
        mma_t<T1, SP>(f1, A3h + q * 16 * ldP, A3l + q * 16 * ldP, ldP, r16, kq, accv);
        const int m = m0 + q * 16 + arow;
        if (m >= Ms_pre) continue;
#pragma unroll
        for (int tt = 0; tt < T1; ++tt) {
            const int nc = (c1 + tt) * 16 + cq * 4;
            f32x4 v = accv[tt] + pb1[tt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            v = drop4<DROP>(v, DROP == 1 ? k1[q][tt] : 0u, (unsigned int)m * (unsigned int)P + (unsigned int)nc, seed1, thr16, a.keep_scale);
            if (a.pre_out) *reinterpret_cast<f32x4*>(a.pre_out + (size_t)m * P + nc) = v;
            if (a.pre_out_p) {
                uint2 h, l;
                split4(v, h, l);
                p32_store4(a.pre_out_p, SP, m, nc, h, l);
            }
            if (a.tap_prenet) *reinterpret_cast<f32x4*>(a.tap_prenet + (size_t)(a.frame_off[m] + a.t_cur) * P + nc) = v;
        }
    }
}


// The same launch in EXACT fp32 (FCL_PRECISION=0; round 4): v_mfma_f32_16x16x4_f32 on fragment-major fp32 weights (fcl_pack_frag_f32), one row tile per
// workgroup, no column split.  Same structure as feat_prenet_split_kernel -- every weight fragment register-resident (a lane's two float4 per tile and
// 32-k step are exactly the bytes of its hi + lo pair), transposed accumulators (the weight fragment is the MFMA's A operand), vector epilogues,
// three phases chained through LDS -- with fp32 tiles in LDS instead of split planes.  Replaces feat_prenet_kernel (weights re-streamed from L2 at
// every k-step: 19.2 us per launch at 2 400 rows, the largest item of the exact-mode pass) where the shape is covered.
template <int NT, int NS_>
struct WFragF {
    f32x4 w0[NT][NS_], w1[NT][NS_];
    __device__ __forceinline__ void load(const float* const (&f)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int st = 0; st < NS_; ++st) {
                w0[t][st] = *reinterpret_cast<const f32x4*>(f[t] + st * 512);
                w1[t][st] = *reinterpret_cast<const f32x4*>(f[t] + st * 512 + 4);
            }
    }
};

template <int NT, int NS_>
__device__ __forceinline__ void mma_tf(const WFragF<NT, NS_>& w, const float* A, int ldk, int r16, int kq, f32x4 (&out)[NT]) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* a = A + r16 * ldk + kq * 4;
#pragma unroll
    for (int st = 0; st < NS_; ++st) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a + st * 32), a1 = *reinterpret_cast<const f32x4*>(a + st * 32 + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w0[t][st][e], a0[e], acc[t], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w1[t][st][e], a1[e], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = acc[t];
}

template <int DROP>
__global__ __launch_bounds__(512) void feat_prenet_split_f32_kernel(const FeatPrenetArgs a) {
    constexpr int SU = 8, SO = 3, SP = 8, U = 256, OP = 96, P = 256;
    const int M_feat = a.M_feat, M_pre = a.M_pre;
    const int Ms_feat = (a.live && a.t_prev >= 0) ? min(a.M_feat, a.live[a.t_prev]) : a.M_feat, Ms_pre = a.live ? min(a.M_pre, a.live[a.t_cur]) : a.M_pre;
    if (a.live && a.w0 && blockIdx.x == 0 && threadIdx.x == 0 && a.live[a.t_cur] > a.M_pre) atomicOr(a.status, (unsigned int)FCL_STATUS_ROWS_CAP);
    if ((int)blockIdx.x * 16 >= (a.h1 ? Ms_feat : Ms_pre)) return;
    constexpr int ldU = U + 8, ldO = OP + 8, ldP = P + 8;  // floats: rows 32 bytes apart modulo the 256-byte bank period, as the plane tiles are
    static_assert(ldU == ldP, "the layer-1 operand tile re-uses the h1 tile's LDS");
    extern __shared__ __attribute__((aligned(16))) float fpf_lds[];
    float* A1 = fpf_lds;            // [16][ldU]  h1 tile, then (phase 2 on) layer 0's output
    float* A2 = A1 + 16 * ldU;      // [16][ldO]  feat_out / prenet input
    const int O = a.O;
    const int m0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int arow = lane & 15, cq = lane >> 4;
    const bool has_feat = a.h1 != nullptr, has_pre = a.w0 != nullptr && m0 < M_pre;
    const bool feat_wave = has_feat && wave * 16 < O;
    const size_t lane8 = (size_t)lane * 8;
    auto frag = [&](const float* base, int tile, int ns) { return base + (size_t)tile * ns * 512 + lane8; };
    const int t0 = wave, t1 = wave + 8;  // this wave's column tiles of both prenet layers
    // ---- phase 0 operands first
    f32x4 hreg[2];
    if (has_feat) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = threadIdx.x + j * 512, r = i >> 6, c = (i & 63) * 4;
            hreg[j] = *reinterpret_cast<const f32x4*>(a.h1 + (size_t)min(m0 + r, M_feat - 1) * U + c);
        }
    }
    asm volatile("s_barrier" ::: "memory");  // (queue order = order of use: see feat_prenet_split_kernel)
    WFragF<1, SU> ff;
    WFragF<2, SO> f0;
    WFragF<2, SP> f1;
    const int fnc = wave * 16 + cq * 4;
    f32x4 f0v = (f32x4){0.f, 0.f, 0.f, 0.f};
    int fo = 0;
    if (feat_wave) {
        const float* const wf[1] = {frag(a.wf_ff, wave, SU)};
        ff.load(wf);
        if (fnc < O) {
            const int mc = min(m0 + arow, M_feat - 1);
            f0v = *reinterpret_cast<const f32x4*>(a.F0 + (size_t)mc * O + fnc);
            fo = a.frame_off[mc];
        }
    }
    unsigned int seed0 = 0, seed1 = 0;
    const unsigned int thr16 = (unsigned int)(a.drop_p * 65536.0f);
    if (DROP == 2) {
        const unsigned int sbump = a.seed_dev ? *a.seed_dev * 0x9E3779B9u : 0u;
        seed0 = hash_u32(a.seed0 + sbump);
        seed1 = hash_u32(a.seed1 + sbump);
    }
    f32x4 pb0[2], pb1[2];
    unsigned int k0[2] = {0u, 0u}, k1[2] = {0u, 0u};
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) pb0[tt] = pb1[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (has_feat) asm volatile("s_barrier" ::: "memory");
    if (has_pre) {
        const float* const w0p[2] = {frag(a.w0_ff, t0, SO), frag(a.w0_ff, t1, SO)};
        f0.load(w0p);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? t1 : t0) * 16 + cq * 4;
            pb0[tt] = *reinterpret_cast<const f32x4*>(a.b0 + nc);
            if (DROP == 1) k0[tt] = *reinterpret_cast<const unsigned int*>(a.keep0 + (size_t)min(m0 + arow, M_pre - 1) * P + nc);
        }
    }
    // ---- phase 0: h1 tile -> LDS; the prenet-input tile zeroed (prev_out = 0 at t = 0; zero padding past O otherwise)
    for (int i = threadIdx.x; i < 16 * ldO / 4; i += 512) reinterpret_cast<f32x4*>(A2)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (has_feat) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = threadIdx.x + j * 512, r = i >> 6, c = (i & 63) * 4;
            *reinterpret_cast<f32x4*>(A1 + r * ldU + c) = hreg[j];
        }
    }
    lds_barrier();
    // ---- phase 1: feat_out of the previous step (+ frame scatter)
    if (feat_wave) {
        f32x4 accv[1];
        mma_tf<1, SU>(ff, A1, ldU, r16, kq, accv);
        const int m = m0 + arow;
        if (fnc < O) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (m < Ms_feat) {
                v = accv[0] + f0v;
                *reinterpret_cast<f32x4*>(a.before + (size_t)((long long)fo + a.t_prev) * O + fnc) = v;
                if (a.out_act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], a.out_act);
                }
            }
            *reinterpret_cast<f32x4*>(A2 + arow * ldO + fnc) = v;
        }
    }
    if (!has_pre) return;
    {   // layer 1's fragments: requested once feat_out's are dead (128 VGPRs)
        const float* const w1p[2] = {frag(a.w1_ff, t0, SP), frag(a.w1_ff, t1, SP)};
        f1.load(w1p);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? t1 : t0) * 16 + cq * 4;
            pb1[tt] = *reinterpret_cast<const f32x4*>(a.b1 + nc);
            if (DROP == 1) k1[tt] = *reinterpret_cast<const unsigned int*>(a.keep1 + (size_t)min(m0 + arow, M_pre - 1) * P + nc);
        }
    }
    lds_barrier();
    // ---- phase 2: prenet layer 0 -> LDS (over the h1 tile: every wave is past its last read of it)
    {
        f32x4 accv[2];
        mma_tf<2, SO>(f0, A2, ldO, r16, kq, accv);
        const int m = m0 + arow;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? t1 : t0) * 16 + cq * 4;
            f32x4 v = accv[tt] + pb0[tt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            v = drop4<DROP>(v, k0[tt], (unsigned int)m * (unsigned int)P + (unsigned int)nc, seed0, thr16, a.keep_scale);
            *reinterpret_cast<f32x4*>(A1 + arow * ldP + nc) = v;
        }
    }
    lds_barrier();
    // ---- phase 3: prenet layer 1 -> global (+ KD tap)
    {
        f32x4 accv[2];
        mma_tf<2, SP>(f1, A1, ldP, r16, kq, accv);
        const int m = m0 + arow;
        if (m >= Ms_pre) return;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int nc = (tt ? t1 : t0) * 16 + cq * 4;
            f32x4 v = accv[tt] + pb1[tt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            v = drop4<DROP>(v, k1[tt], (unsigned int)m * (unsigned int)P + (unsigned int)nc, seed1, thr16, a.keep_scale);
            if (a.pre_out) *reinterpret_cast<f32x4*>(a.pre_out + (size_t)m * P + nc) = v;
            if (a.tap_prenet) *reinterpret_cast<f32x4*>(a.tap_prenet + (size_t)(a.frame_off[m] + a.t_cur) * P + nc) = v;
        }
    }
}

// PRE: both terms have K = 256 (the student's decoder LSTMs) -> all 2 x 8 weight fragments are requested at kernel entry.
template <bool PRE>
__global__ __launch_bounds__(256) void lstm_small_x3_kernel(const LstmStepArgs a) {
    // device-driven loops: loads and MFMAs run on the host's row bound, only the final store is limited to the device's live-row count (its scalar
    // load is then off the critical path: first use after the K walk)
    const int M = a.M, Ms = live_rows_of(a.M, a.m_dev);
    if ((int)blockIdx.y * 16 >= Ms) return;  // tile beyond the device's live rows (uniform per workgroup, before any barrier): a step of a loose bound costs a launch, not a pass
    constexpr int KC = 512;
    __shared__ __attribute__((aligned(16))) u16 A_h[16 * (KC + 16)];  // row stride = 2 (mod 4) sixteen-byte slots: conflict-free fragment reads
    __shared__ __attribute__((aligned(16))) u16 A_lo[16 * (KC + 16)];
    __shared__ float g_l[4][16][17];
    const int m0 = blockIdx.y * 16, u0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    // epilogue operands first (latency hides under the K walk)
    const int erow = threadIdx.x >> 4, euc = threadIdx.x & 15;
    const int em = m0 + erow, eu = u0 + euc;
    const bool evalid = em < M && eu < a.U;
    CellIn ci;
    if (evalid) ci = cell_prefetch(a, em, eu);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    WFrag<1, 8> wf[2];
    if (PRE) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {  // W rows g*U + u0.. form fragment tile (g*U + u0)/16 (U % 16 == 0); 8 steps per row tile
            const size_t fo = (size_t)((g * a.U + u0) >> 4) * 8 * 512 + (size_t)lane * 8;
            const u16* const wh[1] = {a.term[t].Whi + fo};
            const u16* const wl[1] = {a.term[t].Wlo + fo};
            wf[t].load(wh, wl);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t) __syncthreads();
            load_rowtile_split(A_h, A_lo, 256 + 16, a.term[t].A, a.term[t].lda, 256, m0, M);
            __syncthreads();
            f32x4 part[1];
            wf[t].mma(A_h, A_lo, 256 + 16, 256, r16, kq, part);
            acc += part[0];
        }
    }
    for (int t = 0; t < (PRE ? 0 : a.nterms); ++t) {
        const GemmTerm T = a.term[t];
        for (int k0 = 0; k0 < T.K; k0 += KC) {
            const int kc = min(KC, T.K - k0);
            __syncthreads();
            load_rowtile_split(A_h, A_lo, kc + 16, T.A + k0, T.lda, kc, m0, M);
            __syncthreads();
            const size_t fo = ((size_t)((g * a.U + u0) >> 4) * ((T.K + 31) >> 5) + (k0 >> 5)) * 512 + (size_t)lane * 8;
            const u16* const wh[1] = {T.Whi + fo};
            const u16* const wl[1] = {T.Wlo + fo};
            f32x4 part[1];
            rowtile_mma_x3<1>(A_h, A_lo, kc + 16, wh, wl, kc, r16, kq, part);
            acc += part[0];
        }
    }
    {
        const int col = lane & 15, rq = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) g_l[g][rq * 4 + r][col] = acc[r];
    }
    __syncthreads();
    if (!evalid || em >= Ms) return;
    const float pre[4] = {g_l[0][erow][euc], g_l[1][erow][euc], g_l[2][erow][euc], g_l[3][erow][euc]};
    cell_finish(a, em, eu, pre, ci);
}


// The small step in EXACT fp32 with register-resident weights (FCL_PRECISION=0; round 4): the structure of lstm_small_x3_kernel<true> on fragment-major
// fp32 weights (fcl_gemm_term_t.Wff) and v_mfma_f32_16x16x4_f32, the weight fragment as the A operand (a lane's accumulators: four consecutive units
// of one row).  Both terms K = 256.
__global__ __launch_bounds__(256) void lstm_small_ff_kernel(const LstmStepArgs a) {
    const int M = a.M, Ms = live_rows_of(a.M, a.m_dev);
    if ((int)blockIdx.y * 16 >= Ms) return;
    constexpr int LD = 256 + 8;
    __shared__ __attribute__((aligned(16))) float A_f[16 * LD];
    __shared__ float g_l[4][16][17];
    const int m0 = blockIdx.y * 16, u0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int erow = threadIdx.x >> 4, euc = threadIdx.x & 15;
    const int em = m0 + erow, eu = u0 + euc;
    const bool evalid = em < M && eu < a.U;
    CellIn ci;
    if (evalid) ci = cell_prefetch(a, em, eu);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    WFragF<1, 8> wf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float* const w[1] = {a.term[t].Wff + (size_t)((g * a.U + u0) >> 4) * 8 * 512 + (size_t)lane * 8};
        wf[t].load(w);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t) __syncthreads();
        load_rowtile(A_f, LD, a.term[t].A, a.term[t].lda, 256, m0, M);
        __syncthreads();
        f32x4 part[1];
        mma_tf<1, 8>(wf[t], A_f, LD, r16, kq, part);
        acc += part[0];
    }
    {
        const int arow = lane & 15, cq = lane >> 4;  // accumulators: row arow, units cq * 4 .. + 3 of the tile
#pragma unroll
        for (int i = 0; i < 4; ++i) g_l[g][arow][cq * 4 + i] = acc[i];
    }
    __syncthreads();
    if (!evalid || em >= Ms) return;
    const float pre[4] = {g_l[0][erow][euc], g_l[1][erow][euc], g_l[2][erow][euc], g_l[3][erow][euc]};
    cell_finish(a, em, eu, pre, ci);
}

// ---- small-M LSTM step: wave g owns gate g of 16 units x 16 rows ------------------------------------
constexpr int SMALL_KC = 512;  // K chunk resident in LDS

__device__ __forceinline__ void lstm_small_body(const LstmStepArgs& a) {
    // device-driven loops: loads and MFMAs run on the host's row bound, only the final store is limited to the device's live-row count (its scalar
    // load is then off the critical path: first use after the K walk)
    const int M = a.M, Ms = live_rows_of(a.M, a.m_dev);
    if ((int)blockIdx.y * 16 >= Ms) return;  // tile beyond the device's live rows (uniform per workgroup, before any barrier): a step of a loose bound costs a launch, not a pass
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
            load_rowtile(A_l, kc + 4, T.A + k0, T.lda, kc, m0, M);
            __syncthreads();
            const float* const wr[1] = {T.W + (size_t)(g * a.U + min(u, a.U - 1)) * T.ldw + k0};
            f32x4 part[1];
            rowtile_mma<1>(A_l, kc + 4, wr, kc, r16, kq, part, blockIdx.y);
            acc += part[0];
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
    if (m >= Ms || uu >= a.U) return;
    const CellIn ci = cell_prefetch(a, m, uu);
    const float pre[4] = {g_l[0][row][uc], g_l[1][row][uc], g_l[2][row][uc], g_l[3][row][uc]};
    cell_finish(a, m, uu, pre, ci);
}

__global__ __launch_bounds__(256) void lstm_small_kernel(const LstmStepArgs a) { lstm_small_body(a); }

// Two independent steps of the same shape in one launch (blockIdx.z picks the problem): the forward and the backward direction of a BiLSTM time
// step (bilstm.hip algo 1) -- half the dependent launches of the per-step path.
__global__ __launch_bounds__(256) void lstm_small_pair_kernel(const LstmStepArgs a0, const LstmStepArgs a1) {
    if (blockIdx.z == 0) lstm_small_body(a0);
    else lstm_small_body(a1);
}

static bool fp_exact_split() {
    static const int v = tunable("FP_EXACT_SPLIT", 1);  // 0: the exact-fp32 mode keeps its fp32-operand feat/prenet and small-step kernels
    return v != 0;
}

int launch_lstm_small(const LstmStepArgs& a, hipStream_t s) {
    double ksum = 0;
    for (int i = 0; i < a.nterms; ++i) ksum += a.term[i].K;
    dim3 grid((a.U + 15) / 16, (a.M + 15) / 16);
    bool planes = true;
    for (int i = 0; i < a.nterms; ++i) planes = planes && a.term[i].Whi && a.term[i].Wlo && (a.term[i].K & 7) == 0 && a.term[i].ldw == a.term[i].K;
    planes = planes && (a.U & 15) == 0;
    if (planes) {
        ProfScope ps("lstm_small_kernel/bf16x3", 2.0 * a.M * 4.0 * a.U * ksum, a.M, s);
        if (a.nterms == 2 && a.term[0].K == 256 && a.term[1].K == 256) hipLaunchKernelGGL(lstm_small_x3_kernel<true>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(lstm_small_x3_kernel<false>, grid, dim3(256), 0, s, a);
    } else if (a.nterms == 2 && a.term[0].Wff && a.term[1].Wff && a.term[0].K == 256 && a.term[1].K == 256 && (a.U & 15) == 0 && fp_exact_split()) {
        ProfScope ps("lstm_small_ff_kernel/f32", 2.0 * a.M * 4.0 * a.U * ksum, a.M, s);
        hipLaunchKernelGGL(lstm_small_ff_kernel, grid, dim3(256), 0, s, a);
    } else {
        ProfScope ps("lstm_small_kernel", 2.0 * a.M * 4.0 * a.U * ksum, a.M, s);
        hipLaunchKernelGGL(lstm_small_kernel, grid, dim3(256), 0, s, a);
    }
    return check_hip(hipGetLastError(), "lstm_small launch");
}

int launch_lstm_small_pair(const LstmStepArgs& a0, const LstmStepArgs& a1, hipStream_t s) {
    double fl = 0;
    for (const LstmStepArgs* a : {&a0, &a1}) {
        double ksum = 0;
        for (int i = 0; i < a->nterms; ++i) ksum += a->term[i].K;
        fl += 2.0 * a->M * 4.0 * a->U * ksum;
    }
    const int mmax = a0.M > a1.M ? a0.M : a1.M;  // (tiles beyond a problem's own rows exit at once)
    dim3 grid((a0.U + 15) / 16, (mmax + 15) / 16, 2);
    ProfScope ps("lstm_small_pair_kernel", fl, a0.M + a1.M, s);
    hipLaunchKernelGGL(lstm_small_pair_kernel, grid, dim3(256), 0, s, a0, a1);
    return check_hip(hipGetLastError(), "lstm_small_pair launch");
}

// two independent small steps that launch_lstm_step would both send to lstm_small_kernel (fp32 operands, no fragment-major weights): one launch;
// *handled = false when either step has another form (the caller launches them one after the other)
int launch_lstm_small_pair_any(const LstmStepArgs& a0, const LstmStepArgs& a1, hipStream_t s, bool* handled) {
    *handled = false;
    static const int on = tunable("LSTM_SMALL_PAIR", 1);
    if (!on || a0.U != a1.U || a0.m_dev || a1.m_dev) return 0;
    for (const LstmStepArgs* a : {&a0, &a1})
        for (int i = 0; i < a->nterms; ++i)
            if (!a->term[i].A || !a->term[i].W || a->term[i].Whi || a->term[i].Wff) return 0;
    *handled = true;
    return launch_lstm_small_pair(a0, a1, s);
}

int launch_feat_prenet(const FeatPrenetArgs& a, hipStream_t s) {
    const int rows = a.h1 ? a.M_feat : a.M_pre;
    if (rows <= 0) return 0;
    FCL_REQUIRE(!(a.U & 3) && !(a.O & 3) && !(a.P & 3), FCL_ERR_SHAPE, "feat_prenet: U/O/P must be multiples of 4");
    const size_t lds = sizeof(float) * 16 * ((size_t)(a.U + 4) + (a.O + 4) + (a.P + 4));
    FCL_REQUIRE(lds <= 160 * 1024, FCL_ERR_SHAPE, "feat_prenet: tile does not fit LDS (%zu B)", lds);
    {
        const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(feat_prenet_kernel), 160 * 1024);
        if (rc) return rc;
    }
    double fl = 0;
    if (a.h1) fl += 2.0 * a.M_feat * a.O * a.U;
    if (a.w0) fl += 2.0 * a.M_pre * ((double)a.P * a.O + (double)a.P * a.P);
    static const int dbg = tunable("FP_DBG", 0);
    const_cast<FeatPrenetArgs&>(a).dbg_phase = dbg;
    const bool planes = a.wf_hi && a.wf_lo && a.w0_hi && a.w0_lo && a.w1_hi && a.w1_lo && !(a.U & 7) && !(a.O & 7) && !(a.P & 7);
    FCL_REQUIRE(planes || (!a.pre_out_p && !a.before_p), FCL_ERR_INVALID, "feat_prenet: P32 outputs need the bf16x3 weight planes");
    FCL_REQUIRE(!a.w0 || a.pre_out || a.pre_out_p, FCL_ERR_INVALID, "feat_prenet: no prenet output buffer");
    FCL_REQUIRE(!a.pre_out_p || !(a.P & 31), FCL_ERR_SHAPE, "feat_prenet: P32 prenet output needs P %% 32 == 0");
    if (planes) {
        const size_t lds3 = 2 * sizeof(unsigned short) * 16 * ((size_t)(a.U + 8) + (a.O + 8) + (a.P + 8));
        {
            const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(feat_prenet_x3_kernel<0, 0, 0>), 160 * 1024);
            if (rc) return rc;
        }
        ProfScope ps("feat_prenet_kernel/bf16x3", fl, rows, s);
        static const int fast = tunable("FEAT_PRENET_FAST", 1);
        if (fast && a.U == 256 && a.O > 64 && a.O <= 96 && a.P == 256) {
            const dim3 b(512);
            // round 4: feat_prenet_split_kernel (transposed accumulators: vectorised epilogues; optional column split of layer 1).  Measured on one box,
            // 2 400 live rows, rocprofv3 durations: the round-3 kernel (RT = 2) 14.1 us; this one at (NS, RT) = (1, 1) 9.5, (1, 2) 11.9, (2, 2) 10.2,
            // (2, 1) 9.6, (4, 1) 17 (600 workgroups x 245 KB: the L2 -> CU traffic of a launch, not the per-workgroup stream, is what a split costs).
            // The 4-stream bench line does not move with any of them (44.5 - 45.7 M frames/s for all, same box): with four passes in flight the
            // pass is bound by the sum of workgroup-time, which RT = 1 doubles while it halves the latency.  Default: no split, one row tile per
            // workgroup (lowest single-pass latency: 339 -> 269 us of a 1.38 ms eager pass) below 4 096 rows, two above.
            // FCL_FP_SPLIT = 1 / 2 / 4 / 8; FCL_FP_SPLIT_RT = row tiles per workgroup (0: by row count)
            static const int ns_t = tunable("FP_SPLIT", 1), rt_t = tunable("FP_SPLIT_RT", 0);
            {
                const int rt_s = rt_t > 0 ? rt_t : (rows >= 4096 ? 2 : 1);
#define FCL_FPS_CASE(RT_, NS_)                                                                                                           \
    do {                                                                                                                                 \
        constexpr size_t lds_s = 2 * sizeof(unsigned short) * 16 * RT_ * ((256 + 16) + (96 + 16));                                        \
        const dim3 g((rows + 16 * RT_ - 1) / (16 * RT_), NS_);                                                                           \
        const void* fn = a.drop_mode == 1   ? reinterpret_cast<const void*>(feat_prenet_split_kernel<1, RT_, NS_>)                        \
                         : a.drop_mode == 2 ? reinterpret_cast<const void*>(feat_prenet_split_kernel<2, RT_, NS_>)                        \
                                            : reinterpret_cast<const void*>(feat_prenet_split_kernel<0, RT_, NS_>);                       \
        const int rc = ensure_dyn_lds(fn, (int)lds_s);                                                                                   \
        if (rc) return rc;                                                                                                               \
        if (a.drop_mode == 1) hipLaunchKernelGGL((feat_prenet_split_kernel<1, RT_, NS_>), g, b, lds_s, s, a);                            \
        else if (a.drop_mode == 2) hipLaunchKernelGGL((feat_prenet_split_kernel<2, RT_, NS_>), g, b, lds_s, s, a);                       \
        else hipLaunchKernelGGL((feat_prenet_split_kernel<0, RT_, NS_>), g, b, lds_s, s, a);                                             \
    } while (0)
#define FCL_FPS_RT(NS_)                                                                                                                  \
    do {                                                                                                                                 \
        if (rt_s >= 4) FCL_FPS_CASE(4, NS_);                                                                                             \
        else if (rt_s >= 2) FCL_FPS_CASE(2, NS_);                                                                                        \
        else FCL_FPS_CASE(1, NS_);                                                                                                       \
    } while (0)
                if (ns_t >= 8) FCL_FPS_RT(8);
                else if (ns_t >= 4) FCL_FPS_RT(4);
                else if (ns_t >= 2) FCL_FPS_RT(2);
                else FCL_FPS_RT(1);
#undef FCL_FPS_RT
#undef FCL_FPS_CASE
            }
        } else if (a.U == 256 && a.O == 80 && a.P == 256) {
            hipLaunchKernelGGL((feat_prenet_x3_kernel<8, 3, 8>), dim3((rows + 15) / 16), dim3(512), lds3, s, a);
        } else {
            hipLaunchKernelGGL((feat_prenet_x3_kernel<0, 0, 0>), dim3((rows + 15) / 16), dim3(512), lds3, s, a);
        }
    } else if (a.wf_ff && a.w0_ff && a.w1_ff && a.U == 256 && a.P == 256 && a.O > 64 && a.O <= 96 && !a.teacher_in && !a.pre_out_p && !a.before_p &&
               (!a.w0 || a.pre_out) && fp_exact_split()) {
        // exact-fp32 mode with fragment-major fp32 weights: the register-resident form (19.2 -> ~9 us per launch at 2 400 rows)
        ProfScope ps("feat_prenet_split_kernel/f32", fl, rows, s);
        constexpr size_t lds_f = sizeof(float) * 16 * ((256 + 8) + (96 + 8));
        const dim3 g((rows + 15) / 16), b(512);
        const void* fn = a.drop_mode == 1   ? reinterpret_cast<const void*>(feat_prenet_split_f32_kernel<1>)
                         : a.drop_mode == 2 ? reinterpret_cast<const void*>(feat_prenet_split_f32_kernel<2>)
                                            : reinterpret_cast<const void*>(feat_prenet_split_f32_kernel<0>);
        const int rc = ensure_dyn_lds(fn, (int)lds_f);
        if (rc) return rc;
        if (a.drop_mode == 1) hipLaunchKernelGGL(feat_prenet_split_f32_kernel<1>, g, b, lds_f, s, a);
        else if (a.drop_mode == 2) hipLaunchKernelGGL(feat_prenet_split_f32_kernel<2>, g, b, lds_f, s, a);
        else hipLaunchKernelGGL(feat_prenet_split_f32_kernel<0>, g, b, lds_f, s, a);
    } else {
        ProfScope ps("feat_prenet_kernel", fl, rows, s);
        hipLaunchKernelGGL(feat_prenet_kernel, dim3((rows + 15) / 16), dim3(512), lds, s, a);
    }
    return check_hip(hipGetLastError(), "feat_prenet launch");
}

}  // namespace fcl
