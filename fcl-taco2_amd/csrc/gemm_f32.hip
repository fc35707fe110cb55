// gemm_f32.hip — fp32 MFMA (v_mfma_f32_16x16x4_f32) multi-term GEMM and fused LSTM-step kernels, gfx950.
//
// One tiling serves every dense contraction on the FCL-taco2 path (SURVEY.md §8a H2,H4-H8,H11):
//   * Linear layers          : 1 term, W is the PyTorch [out, in] matrix used in place (K-contiguous).
//   * Conv1d (channels-last) : k terms, one per tap, A rows shifted by (j - pad) and zero-filled outside
//                              the row's segment [seg_lo, seg_hi) — an LDS-tiled stencil-as-GEMM.
//   * LSTM / LSTMCell step   : 2-3 terms (input part, recurrent part), the four gates of a hidden unit
//                              accumulate in the same lane so the cell + zoneout update is the epilogue.
// Geometry: workgroup = WM x WN waves (64 lanes each); every wave owns a 16-row x 64-column strip as
// four 16x16 accumulator tiles; K is consumed in 32-wide chunks staged global -> registers -> LDS
// (double-buffered, one barrier per chunk).  A and W fragments are both "row x 4 consecutive k" float4
// LDS reads: lane (r = lane&15, q = lane>>4) feeds element e of its float4 as MFMA k-slot q, so the four
// MFMAs of a float4 cover k = {4q+e}; A and W use the same assignment, hence the sum over k is complete.
// f32-in MFMA is an exact fmaf chain (guide §3), so results differ from the CPU reference only by
// summation order.
#include <string.h>

#include "fcl_common.h"
#include "lstm_epilogue.h"

namespace fcl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef FCL_BK
#define FCL_BK 32
#endif
constexpr int BK = FCL_BK;      // k-chunk (floats): 32 = one 128-B line per row
constexpr int F4 = BK / 4;      // float4 per row per chunk
constexpr int F4_SHIFT = F4 == 8 ? 3 : 2;
static_assert(BK == 32 || BK == 16, "BK must be 16 or 32");
#ifndef FCL_NBUF
#define FCL_NBUF 2
#endif
constexpr int NBUF = FCL_NBUF;  // LDS buffers per workgroup: 2 = one barrier per chunk; 1 = half the LDS (more workgroups per CU), two barriers
constexpr int LDS_LD = BK + 4;  // padded LDS row stride (floats); keeps float4 alignment and spreads rows over banks

// TM = 16-row MFMA tiles per wave along M.  TM = 1: a wave reads 10 operand fragments from LDS per 12 MFMAs (bf16x3) and the kernel is LDS-bound;
// TM = 2 (32 x 64 per wave, 64 x 128 per workgroup with WM = WN = 2): 12 fragments per 24 MFMAs, LDS and MFMA time balance.
template <int WM, int WN, bool LSTM, int TM = 1>
struct Geo {
    static constexpr int BM = 16 * WM * TM;
    static constexpr int BN = 64 * WN;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int NA = (BM * F4 + THREADS - 1) / THREADS;  // float4 loads of A per thread per chunk
    static constexpr int NB = (BN * F4 + THREADS - 1) / THREADS;
    // per row: fp32 path 36 floats (32 + pad); bf16x3 path two bf16 planes of 40 elements (32 + 8 pad) = 40 floats
    static constexpr int LDS_FLOATS = NBUF * (BM + BN) * 40;
};

// ---- bf16x3 operand split (PREC 1): x = hi + lo, hi = bf16_rn(x), lo = bf16_rn(x - hi);  a.b ~= ah.bh + ah.bl + al.bh
// with fp32 accumulation on the bf16 MFMA pipe (16x the fp32-MFMA rate, 3 MFMAs per product => 5.3x), error ~2^-16
// relative per product: 8.6e-6 max-abs on the mel end to end in emulation, 100x inside the 1e-3 parity bar.
constexpr int LDK = BK + 8;  // bf16 elements per LDS plane row: 80 B stride -> conflict-free b128 fragment reads

// Shared main loop.  n0: first output column (generic) or first hidden unit (LSTM).  NU: N (generic) or U.
template <int WM, int WN, bool LSTM, int PREC = 0, int TM = 1>
__device__ __forceinline__ void mainloop(const GemmTerm* __restrict__ terms, int nterms, int M, int m0, int n0,
                                         int NU, const int* __restrict__ seg_lo, const int* __restrict__ seg_hi,
                                         float* lds, f32x4 (&acc)[TM][4], bool hi_only = false) {
    using G = Geo<WM, WN, LSTM, TM>;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kq = lane >> 4;

    // per-thread loader coordinates
    int a_row[G::NA], a_lo[G::NA], a_hi[G::NA];
    bool a_ok[G::NA];
#pragma unroll
    for (int i = 0; i < G::NA; ++i) {
        const int idx = tid + i * G::THREADS;
        const int row = idx >> F4_SHIFT;
        const int m = m0 + row;
        a_ok[i] = (idx < G::BM * F4) && (m < M);
        a_row[i] = m;
        a_lo[i] = 0;
        a_hi[i] = 0x7fffffff;
        if (seg_lo != nullptr && a_ok[i]) {
            a_lo[i] = seg_lo[m];
            a_hi[i] = seg_hi[m];
        }
    }
    long long b_row[G::NB];  // W row index, -1 = zero row
#pragma unroll
    for (int i = 0; i < G::NB; ++i) {
        const int idx = tid + i * G::THREADS;
        const int row = idx >> F4_SHIFT;  // tile column 0..BN-1
        long long wr = -1;
        if (idx < G::BN * F4) {
            if (LSTM) {
                const int u = n0 + (row >> 6) * 16 + (row & 15);
                const int g = (row >> 4) & 3;
                if (u < NU) wr = (long long)g * NU + u;
            } else {
                const int n = n0 + row;
                if (n < NU) wr = n;
            }
        }
        b_row[i] = wr;
    }
    const int c4 = (tid & (F4 - 1)) * 4;  // idx & (F4-1) is the same for every i because THREADS % F4 == 0

    f32x4 ra[G::NA], rb[G::NB];
    auto fetch = [&](int t, int k0) {
        const GemmTerm T = terms[t];
        const int k = k0 + c4;
        const bool kin = k < T.K;
#pragma unroll
        for (int i = 0; i < G::NA; ++i) {
            const int src = a_row[i] + T.shift;
            const bool ok = a_ok[i] && kin && src >= a_lo[i] && src < a_hi[i];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(T.A + (size_t)src * T.lda + k);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < G::NB; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_row[i] >= 0 && kin) v = *reinterpret_cast<const f32x4*>(T.W + (size_t)b_row[i] * T.ldw + k);
            rb[i] = v;
        }
    };
    static_assert(PREC == 0 || BK == 32, "the bf16x3 path consumes one 32-k MFMA step per chunk");
    constexpr int BUF_FLOATS = (G::BM + G::BN) * 40;
    auto stash = [&](int buf) {
        if constexpr (PREC == 0) {
            float* A_l = lds + buf * BUF_FLOATS;
            float* B_l = A_l + G::BM * LDS_LD;
#pragma unroll
            for (int i = 0; i < G::NA; ++i) {
                const int idx = tid + i * G::THREADS;
                if (idx < G::BM * F4) *reinterpret_cast<f32x4*>(A_l + (idx >> F4_SHIFT) * LDS_LD + c4) = ra[i];
            }
#pragma unroll
            for (int i = 0; i < G::NB; ++i) {
                const int idx = tid + i * G::THREADS;
                if (idx < G::BN * F4) *reinterpret_cast<f32x4*>(B_l + (idx >> F4_SHIFT) * LDS_LD + c4) = rb[i];
            }
        } else {  // planes: [A hi][A lo][B hi][B lo], rows of LDK bf16
            unsigned short* P = reinterpret_cast<unsigned short*>(lds + buf * BUF_FLOATS);
            unsigned short* Ah = P, *Al = P + G::BM * LDK, *Bh = P + 2 * G::BM * LDK, *Bl = Bh + G::BN * LDK;
#pragma unroll
            for (int i = 0; i < G::NA; ++i) {
                const int idx = tid + i * G::THREADS;
                if (idx < G::BM * F4) {
                    uint2 hi, lo;
                    split4(ra[i], hi, lo);
                    *reinterpret_cast<uint2*>(Ah + (idx >> F4_SHIFT) * LDK + c4) = hi;
                    *reinterpret_cast<uint2*>(Al + (idx >> F4_SHIFT) * LDK + c4) = lo;
                }
            }
#pragma unroll
            for (int i = 0; i < G::NB; ++i) {
                const int idx = tid + i * G::THREADS;
                if (idx < G::BN * F4) {
                    uint2 hi, lo;
                    split4(rb[i], hi, lo);
                    *reinterpret_cast<uint2*>(Bh + (idx >> F4_SHIFT) * LDK + c4) = hi;
                    *reinterpret_cast<uint2*>(Bl + (idx >> F4_SHIFT) * LDK + c4) = lo;
                }
            }
        }
    };

    int t = 0, k0 = 0;
    fetch(0, 0);
    stash(0);
    __syncthreads();
    int buf = 0;
    while (true) {
        // advance to the next (term, k0) chunk
        int tn = t, kn = k0 + BK;
        if (kn >= terms[t].K) { tn = t + 1; kn = 0; }
        const bool has_next = tn < nterms;
        if (has_next) fetch(tn, kn);

        if constexpr (PREC == 0) {
            const float* A_l = lds + buf * BUF_FLOATS;
            const float* B_l = A_l + G::BM * LDS_LD;
#pragma unroll
            for (int s = 0; s < BK / 16; ++s) {
                f32x4 af[TM];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    af[tm] = *reinterpret_cast<const f32x4*>(A_l + ((wm * TM + tm) * 16 + r16) * LDS_LD + s * 16 + kq * 4);
                f32x4 bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bf[j] = *reinterpret_cast<const f32x4*>(B_l + (wn * 64 + j * 16 + r16) * LDS_LD + s * 16 + kq * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[tm][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tm][e], bf[j][e], acc[tm][j], 0, 0, 0);
                }
            }
        } else {  // one v_mfma_f32_16x16x32_bf16 k-step per chunk: lane (r16, kq) feeds 8 consecutive k of row r16
            const unsigned short* P = reinterpret_cast<const unsigned short*>(lds + buf * BUF_FLOATS);
            const unsigned short* Ah = P, *Al = P + G::BM * LDK, *Bh = P + 2 * G::BM * LDK, *Bl = Bh + G::BN * LDK;
            s16x8 ah[TM], al[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int ao = ((wm * TM + tm) * 16 + r16) * LDK + kq * 8;
                ah[tm] = *reinterpret_cast<const s16x8*>(Ah + ao);
                al[tm] = *reinterpret_cast<const s16x8*>(Al + ao);
            }
            s16x8 bh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bo = (wn * 64 + j * 16 + r16) * LDK + kq * 8;
                bh[j] = *reinterpret_cast<const s16x8*>(Bh + bo);
                bl[j] = *reinterpret_cast<const s16x8*>(Bl + bo);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                if (!hi_only) {  // FCL_GEMM_BF16: bf16-rounded operands, the hi.hi product alone
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[tm][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tm], bh[j], acc[tm][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[tm][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bl[j], acc[tm][j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[tm][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bh[j], acc[tm][j], 0, 0, 0);
            }
        }
        if (!has_next) break;
        if constexpr (NBUF == 1) {
            __syncthreads();  // every wave is done reading the only buffer
            stash(0);
        } else {
            stash(buf ^ 1);
            buf ^= 1;
        }
        __syncthreads();
        t = tn;
        k0 = kn;
    }
}

// XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs by linear id (x fastest), each XCD with its own L2; the N tiles of
// one M-tile share the A rows, so with the natural order every XCD fetches every A tile.  Remap: the workgroups one XCD receives (ids = xcd mod 8)
// walk a CONTIGUOUS range of tiles (bijective for any grid size), so the N tiles of an M-tile meet in one L2; W is read by every XCD either way.
__device__ __forceinline__ void xcd_tile(int& bx, int& by) {
    const int nx = gridDim.x, nwg = gridDim.x * gridDim.y;
    const int orig = blockIdx.y * nx + blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    by = t / nx;
    bx = t - by * nx;
}

// --------------------------------------------------------------------------------------------------
template <int WM, int WN, int PREC, int TM = 1>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmArgs a) {
    using G = Geo<WM, WN, false, TM>;
    __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS];
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = by * G::BM, n0 = bx * G::BN;
    f32x4 acc[TM][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tm][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mainloop<WM, WN, false, PREC, TM>(a.term, a.nterms, a.M, m0, n0, a.N, a.seg_lo, a.seg_hi, lds, acc, PREC != 0 && a.hi_only != 0);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, rq = lane >> 4;
    const unsigned int seed = hash_u32(a.rng_seed + (a.seed_dev ? *a.seed_dev * 0x9E3779B9u : 0u));
    // plain launches (bias / activation / residual only: most of them) skip the per-element option tests -- the element code of these epilogues is
    // instruction-bound (gemm_planes.hip pgemm_epilogue: ~5 us per workgroup with every option tested per element)
    if (!a.rank1_a && !a.C0 && a.drop_mode == 0 && !a.Y2) {
        const int a_M = a.M, a_N = a.N, a_act = a.act, a_ldy = a.ldy, a_ldr = a.ldr;
        const float* const a_bias = a.bias;
        const float* const a_R = a.R;
        float* const a_Y = a.Y;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + col;
                if (n >= a_N) continue;
                const float bn = a_bias ? a_bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + (wm * TM + tm) * 16 + rq * 4 + r;
                    if (m >= a_M) continue;
                    float v = acc[tm][j][r] + bn;
                    if (a_act == FCL_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (a_act == FCL_ACT_TANH) v = tanh_f(v);
                    if (a_R) v += a_R[(size_t)m * a_ldr + n];
                    a_Y[(size_t)m * a_ldy + n] = v;
                }
            }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + col;
        if (n >= a.N) continue;
        const float bn = a.bias ? a.bias[n] : 0.f;
        const float r1w = a.rank1_w ? a.rank1_w[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + (wm * TM + tm) * 16 + rq * 4 + r;
            if (m >= a.M) continue;
            float v = acc[tm][j][r] + bn;
            if (a.rank1_a) v += a.rank1_a[(size_t)m * a.rank1_lda] * r1w;
            if (a.C0) v += a.C0[(size_t)m * a.ldc0 + n];
            if (a.act == FCL_ACT_RELU) v = fmaxf(v, 0.f);
            else if (a.act == FCL_ACT_TANH) v = tanh_f(v);
            if (a.drop_mode == 1) {
                v = a.keep[(size_t)m * a.ldkeep + n] ? v * a.keep_scale : 0.f;
            } else if (a.drop_mode == 2) {
                const unsigned int h = hash_u32(((unsigned int)m * (unsigned int)a.N + (unsigned int)n) ^ seed);
                v = ((h >> 8) * (1.0f / 16777216.0f) >= a.drop_p) ? v * a.keep_scale : 0.f;
            }
            if (a.R) v += a.R[(size_t)m * a.ldr + n];
            a.Y[(size_t)m * a.ldy + n] = v;
            if (a.Y2) a.Y2[(size_t)(a.y2_row_base[m] + a.y2_row_add) * a.ldy2 + n] = v;
        }
    }
}

template <int WM, int WN, int MODE, int PREC, int TM = 1>
__global__ __launch_bounds__(64 * WM * WN) void lstm_step_kernel(const LstmStepArgs a, const int hi_only) {
    using G = Geo<WM, WN, true, TM>;
    __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS];
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = by * G::BM, u0 = bx * (16 * WN);
    const int M = a.M, Ms = live_rows_of(a.M, a.m_dev);  // device-driven loops: loads on the host's bound, stores on the device's count
    if (m0 >= Ms) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int u = u0 + wn * 16 + (lane & 15);
    const int rq = lane >> 4;
    // epilogue operands (G0 / bias / position / old state) are requested BEFORE the K loop so their latency hides
    // under the MFMAs; rows / units past the edge read a clamped, valid address and are never stored.
    CellIn ci[TM][4];
    const int uc = min(u, a.U - 1);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) ci[tm][r] = cell_prefetch<MODE>(a, min(m0 + (wm * TM + tm) * 16 + rq * 4 + r, M - 1), uc);
    f32x4 acc[TM][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tm][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mainloop<WM, WN, true, PREC, TM>(a.term, a.nterms, M, m0, u0, a.U, nullptr, nullptr, lds, acc, PREC != 0 && hi_only != 0);
    if (u >= a.U) return;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + (wm * TM + tm) * 16 + rq * 4 + r;
            if (m >= Ms) continue;
            const float pre[4] = {acc[tm][0][r], acc[tm][1][r], acc[tm][2][r], acc[tm][3][r]};
            cell_finish(a, m, u, pre, ci[tm][r]);
        }
}

// --------------------------------------------------------------------------------------------------
// FCL_PRECISION: 1 (default) = bf16x3 split operands on the bf16 MFMA pipe, fp32 accumulate: ~1e-5 relative error,
// 1.4x the end-to-end throughput of 0 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32), which stays available for audits.
static int tm2_min_wg() {
    static const int v = tunable("GEMM_TM2_MIN_WG", 128);
    return v;
}

static int half_tile_wg() {
    static const int v = tunable("GEMM_HALF_TILE_WG", 0);  // measured: 32 x 64 tiles lose (39 vs 47 TFLOP/s) even when 64 x 64 leaves CUs idle; off
    return v;
}

static int precision() {
    static const int p = tunable("PRECISION", 1);
    return p;
}

// ---- split-K GEMM for a handful of rows (the per-step GEMMs of the BPTT recurrences: M = live rows of a tail step or the B utterances of a
// BiLSTM step).  One workgroup = a 16 x 16 output tile, its 4 waves each contract a quarter of K with exact fp32 MFMAs straight from global
// memory (each lane: one float4 of A and one of W per 16 k's; the 4 MFMAs of a chunk use k = kb + 4*lanegroup + e on both operands), the four
// partial tiles meet in LDS.  N/16 x ceil(M/16) workgroups instead of the single 16 x 256 tile a big-tile kernel would give such a GEMM.
// Round 5: TR x TC MFMA tiles per workgroup (16 TR x 16 TC outputs): the 16 x 16 form re-reads W once per 16 rows and A once per 16 columns -- at
// FCL-taco2-T size (M = 256, N = 2 048, K = 4 096) ~1 GB of L2 traffic per launch, 55 us; 32 x 32 halves both.
template <int TR, int TC>
__global__ __launch_bounds__(256) void gemm_smallm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, int K,
                                                          const float* __restrict__ bias, const float* __restrict__ R, int ldr, float* __restrict__ Y,
                                                          int ldy, int M, int N) {
    __shared__ f32x4 part[4][TR * TC][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16 * TC, m0 = blockIdx.y * 16 * TR;
    const int kchunk = ((K + 63) / 64) * 16;  // K range of one wave, a multiple of 16
    const int k_lo = wave * kchunk, k_hi = min(K, k_lo + kchunk);
    const float* ap[TR];
    const float* wp[TC];
#pragma unroll
    for (int i = 0; i < TR; ++i) ap[i] = A + (size_t)min(m0 + 16 * i + r16, M - 1) * lda + q * 4;  // clamped loads, masked stores
#pragma unroll
    for (int c = 0; c < TC; ++c) wp[c] = W + (size_t)min(n0 + 16 * c + r16, N - 1) * ldw + q * 4;
    f32x4 acc[TR][TC];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int c = 0; c < TC; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // UN chunks of 16 k per trip with all their loads issued before the first MFMA: at a handful of rows the kernel is a stream of W (33 MB at FCL-taco2-T
    // size) and two loads in flight per lane left it at 1.5 TB/s
    constexpr int UN = TR * TC == 1 ? 4 : (TR * TC == 2 ? 2 : 1);  // (2 x 2 tiles: 16 MFMAs per chunk already cover the loads; deeper was 10 % slower)
    for (int kb = k_lo; kb < k_hi; kb += 16 * UN) {
        f32x4 av[UN][TR], wv[UN][TC];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool in = kb + 16 * u + q * 4 < k_hi;  // K % 4 == 0: a float4 is all in or all out
#pragma unroll
            for (int i = 0; i < TR; ++i) av[u][i] = in ? *reinterpret_cast<const f32x4*>(ap[i] + kb + 16 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < TC; ++c) wv[u][c] = in ? *reinterpret_cast<const f32x4*>(wp[c] + kb + 16 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int c = 0; c < TC; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i][e], wv[u][c][e], acc[i][c], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int c = 0; c < TC; ++c) part[wave][i * TC + c][lane] = acc[i][c];
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < (TR * TC + 3) / 4; ++tt) {  // tile tt * 4 + wave is summed and stored by this wave
        const int tile = tt * 4 + wave;
        if (tile >= TR * TC) continue;
        const f32x4 p0 = part[0][tile][lane], p1 = part[1][tile][lane], p2 = part[2][tile][lane], p3 = part[3][tile][lane];
        const int n = n0 + 16 * (tile % TC) + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + 16 * (tile / TC) + q * 4 + e;  // C layout of the 16x16 MFMA: lane (r16, q) holds rows 4q..4q+3 of column r16
            if (m < M && n < N) {
                float v = (p0[e] + p1[e]) + (p2[e] + p3[e]);
                if (bias) v += bias[n];
                if (R) v += R[(size_t)m * ldr + n];
                Y[(size_t)m * ldy + n] = v;
            }
        }
    }
}

template <int WM, int WN, int TM = 1>
static void launch_gemm_cfg(const GemmArgs& a, hipStream_t s, const char* name, double flops) {
    using G = Geo<WM, WN, false, TM>;
    dim3 grid((a.N + G::BN - 1) / G::BN, (a.M + G::BM - 1) / G::BM);
    char full[64];  // the rocprofv3 instantiation name <WM, WN, PREC, TM>, so HIP-event and rocprof / PMC rows can be joined by name
    GemmArgs b = a;
    b.hi_only = precision() && gemm_mode() == FCL_GEMM_BF16;
    snprintf(full, sizeof(full), "gemm_kernel<%d,%d,%d,%d>%s", WM, WN, precision() ? 1 : 0, TM, b.hi_only ? "/bf16" : precision() ? "/bf16x3" : "");
    static const int shapes = tunable("PROF_SHAPES", 0);  // developer aid: the profile records split by shape (M x N x sum K)
    if (shapes && g_prof_on) {
        long long ks = 0;
        for (int i = 0; i < a.nterms; ++i) ks += a.term[i].K;
        const size_t l = strlen(full);
        snprintf(full + l, sizeof(full) - l, " %dx%dx%lld", a.M, a.N, ks);
    }
    (void)name;
    ProfScope ps(full, flops, a.M, s);
    if (precision()) hipLaunchKernelGGL((gemm_kernel<WM, WN, 1, TM>), grid, dim3(G::THREADS), 0, s, b);
    else hipLaunchKernelGGL((gemm_kernel<WM, WN, 0, TM>), grid, dim3(G::THREADS), 0, s, b);
}

template <int WM, int WN, int TM = 1>
static void launch_lstm_cfg(const LstmStepArgs& a, hipStream_t s, const char* name, double flops) {
    using G = Geo<WM, WN, true, TM>;
    dim3 grid((a.U + 16 * WN - 1) / (16 * WN), (a.M + G::BM - 1) / G::BM);
    const bool plain = !a.zone_keep_h && !a.row_len;
    const int mode = (plain && a.G && a.rank1_w && !a.bias) ? 0 : (plain && a.bias && !a.G && !a.rank1_w) ? 1 : -1;
    char full[64];  // <WM, WN, MODE, PREC, TM> as rocprofv3 prints the instantiation
    const int hi_only = precision() && gemm_mode() == FCL_GEMM_BF16;
    snprintf(full, sizeof(full), "lstm_step_kernel<%d,%d,%d,%d,%d>%s", WM, WN, mode, precision() ? 1 : 0, TM, hi_only ? "/bf16" : precision() ? "/bf16x3" : "");
    (void)name;
    ProfScope ps(full, flops, a.M, s);
    if (precision()) {
        if (mode == 0) hipLaunchKernelGGL((lstm_step_kernel<WM, WN, 0, 1, TM>), grid, dim3(G::THREADS), 0, s, a, hi_only);
        else if (mode == 1) hipLaunchKernelGGL((lstm_step_kernel<WM, WN, 1, 1, TM>), grid, dim3(G::THREADS), 0, s, a, hi_only);
        else hipLaunchKernelGGL((lstm_step_kernel<WM, WN, -1, 1, TM>), grid, dim3(G::THREADS), 0, s, a, hi_only);
    } else {
        if (mode == 0) hipLaunchKernelGGL((lstm_step_kernel<WM, WN, 0, 0, TM>), grid, dim3(G::THREADS), 0, s, a, hi_only);
        else if (mode == 1) hipLaunchKernelGGL((lstm_step_kernel<WM, WN, 1, 0, TM>), grid, dim3(G::THREADS), 0, s, a, hi_only);
        else hipLaunchKernelGGL((lstm_step_kernel<WM, WN, -1, 0, TM>), grid, dim3(G::THREADS), 0, s, a, hi_only);
    }
}

// ---- exact fp32 on the LDS-DMA machinery of gemm_planes.hip (round 4) ------------------------------------------------------------------------
// Under FCL_PRECISION=0 a contraction whose operands are plain row-major fp32 matrices with K, lda, ldw multiples of 32 floats and 128-byte aligned
// bases IS a set of 128-byte lines in the geometry the pre-split kernels stream (a P32 line holds 32 hi | 32 lo bf16 halves of 32 columns, an fp32
// line the 32 columns themselves): the same loaders, ring, swizzle and epilogues run it with v_mfma_f32_16x16x4_f32 (pchunk_mma, HI = 2).
extern thread_local bool t_exact_lines;  // gemm_planes.hip
static bool exact_lines_on() {
    static const int v = tunable("EXACT_LINES", 1);
    return v != 0;
}
static bool exact_lines_ok(const GemmTerm* t, int n) {
    if (precision() || !exact_lines_on()) return false;
    for (int i = 0; i < n; ++i) {
        if (!t[i].A || !t[i].W || (t[i].K & 31) || (t[i].lda & 31) || (t[i].ldw & 31) || t[i].lda < t[i].K || t[i].ldw < t[i].K || t[i].a_chunk_stride) return false;
        if ((reinterpret_cast<uintptr_t>(t[i].A) | reinterpret_cast<uintptr_t>(t[i].W)) & 127u) return false;
    }
    return true;
}
template <typename Args>
static void exact_lines_terms(Args& b, int n) {  // the fp32 rows as "planes": lines per row = floats per row / 32
    for (int i = 0; i < n; ++i) {
        b.term[i].Ap = reinterpret_cast<const uint16_t*>(b.term[i].A);
        b.term[i].Wp = reinterpret_cast<const uint16_t*>(b.term[i].W);
        b.term[i].lda_p = b.term[i].lda / 32;
        b.term[i].ldw_p = b.term[i].ldw / 32;
    }
}
struct ExactScope {
    ExactScope() { t_exact_lines = true; }
    ~ExactScope() { t_exact_lines = false; }
};

// --------------------------------------------------------------------------------------------------
static int check_terms(const GemmTerm* t, int n, int maxn, bool need_seg, const int* lo) {
    FCL_REQUIRE(n >= 1 && n <= maxn, FCL_ERR_INVALID, "gemm: nterms %d out of range [1,%d]", n, maxn);
    for (int i = 0; i < n; ++i) {
        FCL_REQUIRE(t[i].K > 0 && (t[i].K & 3) == 0, FCL_ERR_SHAPE, "gemm: term %d K=%d must be a positive multiple of 4", i, t[i].K);
        const bool planes = t[i].Ap && t[i].Wp;  // pre-split P32 operands: the fp32 pointers are optional then
        if (planes) {
            FCL_REQUIRE(t[i].lda_p * 32 >= t[i].K && t[i].ldw_p * 32 >= t[i].K, FCL_ERR_SHAPE, "gemm: term %d plane strides %d/%d lines are narrower than K=%d", i,
                        t[i].lda_p, t[i].ldw_p, t[i].K);
            FCL_REQUIRE(((reinterpret_cast<uintptr_t>(t[i].Ap) | reinterpret_cast<uintptr_t>(t[i].Wp)) & 127u) == 0, FCL_ERR_ALIGN,
                        "gemm: term %d planes must be 128-byte aligned", i);
        }
        if (!planes || (t[i].A && t[i].W)) {  // the fp32 pair is optional next to planes (either pointer alone is ignored then)
            FCL_REQUIRE(t[i].A && t[i].W, FCL_ERR_INVALID, "gemm: term %d has a null operand", i);
            FCL_REQUIRE((t[i].lda & 3) == 0 && (t[i].ldw & 3) == 0, FCL_ERR_SHAPE, "gemm: term %d lda=%d/ldw=%d must be multiples of 4", i, t[i].lda, t[i].ldw);
            FCL_REQUIRE(aligned16(t[i].A) && aligned16(t[i].W), FCL_ERR_ALIGN, "gemm: term %d operands must be 16-byte aligned", i);
        }
        if (t[i].shift != 0) FCL_REQUIRE(need_seg && lo, FCL_ERR_INVALID, "gemm: shifted term %d needs seg_lo/seg_hi", i);
    }
    return 0;
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    FCL_REQUIRE(a.M >= 0 && a.N > 0, FCL_ERR_SHAPE, "gemm: bad M=%d N=%d", a.M, a.N);
    if (a.M == 0) return 0;
    int rc = check_terms(a.term, a.nterms, FCL_MAX_TERMS, true, a.seg_lo);
    if (rc) return rc;
    FCL_REQUIRE(a.Y || a.Yp, FCL_ERR_INVALID, "gemm: null output");
    FCL_REQUIRE(a.drop_mode != 1 || a.keep, FCL_ERR_INVALID, "gemm: drop_mode 1 needs a keep mask");
    FCL_REQUIRE(!a.Y2 || a.y2_row_base, FCL_ERR_INVALID, "gemm: Y2 needs y2_row_base");
    if (precision() && planes_ok(a.term, a.nterms)) return launch_gemm_planes(a, s);  // pre-split operands: the LDS-DMA kernels
    FCL_REQUIRE(a.Y && !a.Yp, FCL_ERR_INVALID, "gemm: the fp32-operand kernels write fp32 outputs only (planes output needs planes inputs)");
    static const int exact_min_m = tunable("EXACT_LINES_MIN_M", 257);  // (fewer rows: the split-K / 64-row fp32-operand kernels below)
    if (a.M >= exact_min_m && exact_lines_ok(a.term, a.nterms)) {
        GemmArgs b = a;
        exact_lines_terms(b, a.nterms);
        ExactScope sc;
        return launch_gemm_planes(b, s);
    }
    for (int i = 0; i < a.nterms; ++i) FCL_REQUIRE(a.term[i].A && a.term[i].W, FCL_ERR_INVALID, "gemm: term %d has no fp32 operands for the fp32-operand kernels", i);
    double ksum = 0;
    for (int i = 0; i < a.nterms; ++i) ksum += a.term[i].K;
    const double flops = 2.0 * a.M * (double)a.N * ksum;
    const GemmTerm& t0 = a.term[0];
    static const int smallm = tunable("GEMM_SMALLM", 256);  // (r3: 64 -> 256; the BPTT steps with 65 - 256 live rows ran 34 us each on one lonely 64-row tile per 16 columns)
    if (a.M <= smallm && a.nterms == 1 && t0.shift == 0 && !a.seg_lo && !a.rank1_a && !a.C0 && a.act == FCL_ACT_NONE && a.drop_mode == 0 && !a.keep &&
        !a.Y2 && t0.K >= 256) {  // plain Y = A.W^T (+bias) (+R) on a few rows: split K over the waves instead of one lonely big tile
        ProfScope ps("gemm_smallm_kernel", flops, a.M, s);
        static const int big_min = tunable("GEMM_SMALLM_32_MIN_WG", 256);  // 32 x 32 (16 x 32) tiles while they still give a workgroup per CU
        const long long wg32 = (long long)((a.N + 31) / 32) * ((a.M + 31) / 32), wg16x32 = (long long)((a.N + 31) / 32) * ((a.M + 15) / 16);
        if (wg32 >= big_min)
            hipLaunchKernelGGL((gemm_smallm_kernel<2, 2>), dim3((a.N + 31) / 32, (a.M + 31) / 32), dim3(256), 0, s, t0.A, t0.lda, t0.W, t0.ldw, t0.K, a.bias, a.R,
                               a.ldr, a.Y, a.ldy, a.M, a.N);
        else if (wg16x32 >= big_min)
            hipLaunchKernelGGL((gemm_smallm_kernel<1, 2>), dim3((a.N + 31) / 32, (a.M + 15) / 16), dim3(256), 0, s, t0.A, t0.lda, t0.W, t0.ldw, t0.K, a.bias, a.R,
                               a.ldr, a.Y, a.ldy, a.M, a.N);
        else
            hipLaunchKernelGGL((gemm_smallm_kernel<1, 1>), dim3((a.N + 15) / 16, (a.M + 15) / 16), dim3(256), 0, s, t0.A, t0.lda, t0.W, t0.ldw, t0.K, a.bias, a.R,
                               a.ldr, a.Y, a.ldy, a.M, a.N);
        return check_hip(hipGetLastError(), "gemm_smallm launch");
    }
    // 64x64 tiles measured best for every GEMM of the path (the 32x128 / 16x256 variants only win for a single 16/32-row tile)
    static const int force = tunable("GEMM_CFG", 0);  // experiments only: 1 -> <4,1>, 2 -> <2,2>, 3 -> <1,4>
    static const int tm2 = tunable("GEMM_TM2", 1);    // 64 x 128 workgroup tiles with 32 x 64 per wave where the grid still fills the chip
    if (force == 0 && tm2 && a.N >= 128 && (long long)((a.M + 63) / 64) * ((a.N + 127) / 128) >= tm2_min_wg()) {
        launch_gemm_cfg<2, 2, 2>(a, s, "gemm_kernel<2,2,tm2>", flops);
    } else if (force == 0 && a.M > 32 && (long long)((a.M + 63) / 64) * ((a.N + 63) / 64) < half_tile_wg()) {
        launch_gemm_cfg<2, 1>(a, s, "gemm_kernel<2,1>", flops);  // too few 64 x 64 tiles to fill 256 CUs: 32 x 64 tiles, twice the workgroups
    } else if (force == 1 || (force == 0 && a.M > 32)) {
        launch_gemm_cfg<4, 1>(a, s, "gemm_kernel<4,1>", flops);
    } else if (force == 2 || (force == 0 && a.M > 16)) {
        launch_gemm_cfg<2, 2>(a, s, "gemm_kernel<2,2>", flops);
    } else {
        launch_gemm_cfg<1, 4>(a, s, "gemm_kernel<1,4>", flops);
    }
    return check_hip(hipGetLastError(), "gemm launch");
}

bool lstm_step_is_small(int M, int U) {
    static const int small_m = tunable("LSTM_SMALL_M", 0);  // 0 = by width: the wave-per-gate small-tile kernel re-streams W per 16-row tile,
    // which stops paying earlier at U = 1024 (FCL-taco2-T); with the pre-split operand kernels available (32-row tiles) at ~500 rows for U = 256
    static const bool planes_on = (tunable("PRECISION", 1) != 0 && tunable("PLANES", 1) != 0) || (tunable("PRECISION", 1) == 0 && tunable("EXACT_LINES", 1) != 0);
    // (U >= 512: 256 -> 64 rows in round 3 -- FCL-taco2-T synthesis 6.88 -> 7.18 M frames/s, teacher update 12.93 -> 12.78 ms: at 4 096 gate columns the
    // 32-row pre-split tiles beat the wave-per-gate kernel's W re-streaming from 65 rows on)
    return M <= (small_m ? small_m : (U >= 512 ? (planes_on ? 64 : 256) : (planes_on ? 512 : 1024)));
}

// the argument checks of one LSTM step, shared by launch_lstm_step and by the callers that hand two steps to a pair launch (train_loops.hip: ADVICE r5 --
// the wavefront's pair path used to skip them)
int validate_lstm_step(const LstmStepArgs& a) {
    FCL_REQUIRE(a.M >= 0 && a.U > 0, FCL_ERR_SHAPE, "lstm_step: bad M=%d U=%d", a.M, a.U);
    if (a.M == 0) return 0;
    int rc = check_terms(a.term, a.nterms, 3, false, nullptr);
    if (rc) return rc;
    FCL_REQUIRE(a.h_in && a.h_out && a.c && a.h_in != a.h_out, FCL_ERR_INVALID, "lstm_step: h_in/h_out/c must be set and h_out must not alias h_in");
    FCL_REQUIRE(!a.rank1_w || a.dur, FCL_ERR_INVALID, "lstm_step: rank1_w needs dur");
    FCL_REQUIRE((a.zone_keep_h == nullptr) == (a.zone_keep_c == nullptr), FCL_ERR_INVALID, "lstm_step: zoneout masks come in pairs");
    return 0;
}
// true when launch_lstm_step would run this step on the pre-split operands (bf16x3 / bf16 arithmetic): FCL_PRECISION / FCL_PLANES on, every term with planes
bool lstm_step_on_planes(const LstmStepArgs& a) { return precision() && planes_ok(a.term, a.nterms); }

int launch_lstm_step(const LstmStepArgs& a, hipStream_t s) {
    int rc = validate_lstm_step(a);
    if (rc) return rc;
    if (a.M == 0) return 0;
    const bool small = lstm_step_is_small(a.M, a.U);
    bool has_f32 = true;
    for (int i = 0; i < a.nterms; ++i) has_f32 = has_f32 && a.term[i].A && a.term[i].W;
    if (precision() && planes_ok(a.term, a.nterms) && (!small || !has_f32)) return launch_lstm_planes(a, s);  // pre-split operands, LDS-DMA ring
    FCL_REQUIRE(has_f32, FCL_ERR_INVALID, "lstm_step: the fp32-operand kernels need A and W of every term");
    if (small) return launch_lstm_small(a, s);
    if (!a.h_out_p && exact_lines_ok(a.term, a.nterms)) {  // FCL_PRECISION=0: the step on the LDS-DMA kernels with exact fp32 MFMAs
        LstmStepArgs b = a;
        exact_lines_terms(b, a.nterms);
        ExactScope sc;
        return launch_lstm_planes(b, s);
    }
    double ksum = 0;
    for (int i = 0; i < a.nterms; ++i) ksum += a.term[i].K;
    const double flops = 2.0 * a.M * 4.0 * a.U * ksum;
    const long long wg64 = (long long)((a.M + 63) / 64) * ((a.U + 15) / 16);
    static const int tm2 = tunable("GEMM_TM2", 1);
    if (tm2 && a.U >= 32 && (long long)((a.M + 63) / 64) * ((a.U + 31) / 32) >= tm2_min_wg()) {
        launch_lstm_cfg<2, 2, 2>(a, s, "lstm_step_kernel<2,2,tm2>", flops);
    } else if (wg64 >= 256 || a.U <= 16) {
        launch_lstm_cfg<4, 1>(a, s, "lstm_step_kernel<4,1>", flops);
    } else if (a.U <= 32 || (long long)((a.M + 31) / 32) * ((a.U + 31) / 32) >= 192) {
        launch_lstm_cfg<2, 2>(a, s, "lstm_step_kernel<2,2>", flops);
    } else {
        launch_lstm_cfg<1, 4>(a, s, "lstm_step_kernel<1,4>", flops);
    }
    return check_hip(hipGetLastError(), "lstm_step launch");
}

}  // namespace fcl
