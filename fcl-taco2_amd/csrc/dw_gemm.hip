// dw_gemm.hip — the weight-gradient GEMM of the training step (tts.py:137-179 `loss.backward()`), gfx950, round 5.
//
//   dW[n, k] += sum_m dY[m, n] * X[row(m) + shift, k]          (contraction over the ROWS of two row-major fp32 activations)
//
// Both operands are stored the "wrong" way round for an MFMA (a lane needs 8 consecutive CONTRACTION indices of one column), which is why the
// round-1 kernel (backward.hip gemm_tn_kernel) loads one column per lane with scalar dword loads (256 bytes per wave instruction: a quarter of the
// texture addresser's rate) and why the transposed-planes route pays two packing passes.  Here the transposition costs nothing:
//   * 16-byte coalesced global loads (a wave = 2 rows x 128 columns), one bf16 hi / lo split per element in registers, 8-byte LDS stores into
//     ROW-major bf16 planes [m][col];
//   * `ds_read_b64_tr_b16`: a 16-lane group reads a [4 m][16 col] block and every lane receives the 4 m-values of ITS column — two such reads
//     are one MFMA fragment (probe: tools/probe/tr_probe.hip; lane i of the group addresses row i / 4, columns 4 (i % 4) .. + 3);
//   * 128 x 128 output tile, four waves of 64 x 64 (48 bf16x3 MFMAs per wave and 32-row chunk against 32 transposing reads), two LDS stages,
//     one barrier per chunk, the next chunk's global loads in flight under the MFMAs;
//   * 32-byte units of a plane row are XOR-swizzled with (m & 3) | ((m >> 3) & 1) << 2 so that the eight rows a half-wave's two lane groups
//     read land in eight distinct bank octets (a plane row is 256 bytes = all 64 banks).
// Split over the contraction across workgroups (gridDim.z slices), fp32 atomics into the gradient buffer, as the old kernel.
#include <algorithm>
#include "fcl_common.h"

namespace fcl {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DW_BM = 32;  // contraction rows per chunk (one 16x16x32 MFMA step)

struct DwArgs {
    const float* A;
    int lda;
    const float* B;
    int ldb;
    float* C;
    int ldc;
    int M, N, K, shift0;
    const int* seg_lo;
    const int* seg_hi;
    int rows_per_slice, ntaps;
    long long c_tap_stride;
};

// One operand's share of a stage: bf16 planes [32 rows][32 T columns] (hi, then lo), T = 16-column MFMA tiles per wave (two waves side by side).
// 32-byte units (16 columns) of a plane row are XOR-swizzled so that the eight rows a half-wave's two lane groups read -- rows r0 .. r0 + 3 and
// r0 + 8 .. r0 + 11 -- land in eight distinct bank octets: a 256-byte row (T = 4) covers all 64 banks, so the unit index takes (r & 3) and bit 3
// of r; two 128-byte rows (T = 2) share the 64 banks, row parity picks the half, so the unit index takes bit 1 and bit 3 of r.
template <int T>
struct DwOp {
    static constexpr int ROWB = 64 * T;             // bytes per plane row
    static constexpr int PLANE = DW_BM * ROWB;      // bytes per plane
    static constexpr int F4_ROW = 8 * T;            // float4 per source row
    static constexpr int ROWS_PASS = 256 / F4_ROW;  // rows one pass of the 256 loader threads covers
    static constexpr int PASSES = DW_BM / ROWS_PASS;  // = T
    __device__ static __forceinline__ int swz(int r) { return T == 4 ? ((r & 3) | (((r >> 3) & 1) << 2)) : (((r >> 1) & 1) | (((r >> 3) & 1) << 1)); }
    __device__ static __forceinline__ int byte_of(int r, int col) { return r * ROWB + ((((col >> 4) ^ swz(r))) << 5) + ((col & 15) << 1); }
};

template <int ROWB>
__device__ __forceinline__ s16x8 dw_frag(const unsigned char* plane, int off) {
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(plane + off));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(plane + off + 4 * ROWB));
    return (s16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
}

// TA / TB: MFMA tiles per wave along n / k (output tile 32 TA x 32 TB, four waves 2 x 2).  (4, 4): 128 x 128, 64 KB of LDS, the large outputs;
// (2, 2): 64 x 64, 32 KB: outputs with few tiles, where the slices' atomics (tile bytes x slices) are what a launch costs.
template <int TA, int TB, bool SEG, bool HI_ONLY>
__global__ __launch_bounds__(256, 2) void dw_mfma_kernel(const DwArgs g) {
    typedef DwOp<TA> OA;
    typedef DwOp<TB> OB;
    constexpr int STAGE = 2 * OA::PLANE + 2 * OB::PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    // XCD-aware order (as gemm_tn_kernel): the ids one XCD receives walk a contiguous range of (slice, tap, k tile, n tile)
    const int nx = gridDim.x, nxy = gridDim.x * gridDim.y, nwg = nxy * gridDim.z;
    const int orig = (blockIdx.z * gridDim.y + blockIdx.y) * nx + blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tl = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    const int bz = tl / nxy, by = (tl - bz * nxy) / nx, bx = tl - bz * nxy - by * nx;
    const int n0 = bx * 32 * TA, k0 = by * 32 * TB;
    const int tap = bz % g.ntaps, slice = bz / g.ntaps;
    const int shift = g.shift0 + tap;
    float* __restrict__ C = g.C + (size_t)tap * g.c_tap_stride;
    const int M = g.M, N = g.N, K = g.K;
    const int m_lo = slice * g.rows_per_slice, m_hi = min(M, m_lo + g.rows_per_slice);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;

    // ---- loader role: per operand, float4 #(row r0 + ROWS_PASS j, columns c4 .. c4 + 3), j < PASSES
    const int a_c4 = (tid % OA::F4_ROW) * 4, a_r0 = tid / OA::F4_ROW;
    const int b_c4 = (tid % OB::F4_ROW) * 4, b_r0 = tid / OB::F4_ROW;
    const bool a_col = n0 + a_c4 < N, b_col = k0 + b_c4 < K;  // N, K are multiples of 4: a float4 is inside or outside as a whole
    const float* __restrict__ ap = g.A + n0 + (a_col ? a_c4 : 0);
    const float* __restrict__ bp = g.B + k0 + (b_col ? b_c4 : 0);
    f32x4 ra[OA::PASSES], rb[OB::PASSES];
    int blo[SEG ? OB::PASSES : 1], bhi[SEG ? OB::PASSES : 1];  // segment bounds of the B rows in flight
    unsigned okm = 0;  // bit j: A row valid, bit 4 + j: B row valid (applied at the split: no load result is consumed early)
    // No address depends on a loaded value: B is read at the clamped shifted row whatever its segment says, the bounds travel beside it and
    // decide at the split (a dependent seg -> address -> load chain stalled every chunk of the Conv1d form for a memory round trip)
    auto fetch = [&](int mc) {
        okm = 0;
#pragma unroll
        for (int j = 0; j < OA::PASSES; ++j) {
            const int m = mc + a_r0 + OA::ROWS_PASS * j;
            ra[j] = *reinterpret_cast<const f32x4*>(ap + (size_t)min(m, M - 1) * g.lda);
            okm |= (m < m_hi && a_col) ? (1u << j) : 0u;
        }
#pragma unroll
        for (int j = 0; j < OB::PASSES; ++j) {
            const int m = mc + b_r0 + OB::ROWS_PASS * j;
            const int src = m + shift;
            rb[j] = *reinterpret_cast<const f32x4*>(bp + (size_t)min(max(src, 0), M - 1) * g.ldb);
            if (SEG) {
                const int mcl = min(m, M - 1);
                blo[j] = g.seg_lo[mcl];
                bhi[j] = g.seg_hi[mcl];
            }
            okm |= (m < m_hi && src >= 0 && src < M && b_col) ? (16u << j) : 0u;
        }
    };
    auto stash = [&](unsigned char* st, int mc) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        uint2 h, l;
#pragma unroll
        for (int j = 0; j < OA::PASSES; ++j) {
            const int off = OA::byte_of(a_r0 + OA::ROWS_PASS * j, a_c4);
            split4(((okm >> j) & 1u) ? ra[j] : z, h, l);
            *reinterpret_cast<uint2*>(st + off) = h;
            *reinterpret_cast<uint2*>(st + OA::PLANE + off) = l;
        }
#pragma unroll
        for (int j = 0; j < OB::PASSES; ++j) {
            const int off = OB::byte_of(b_r0 + OB::ROWS_PASS * j, b_c4);
            bool ok = (okm >> (4 + j)) & 1u;
            if (SEG) {
                const int src = mc + b_r0 + OB::ROWS_PASS * j + shift;
                ok = ok && src >= blo[j] && src < bhi[j];
            }
            split4(ok ? rb[j] : z, h, l);
            *reinterpret_cast<uint2*>(st + 2 * OA::PLANE + off) = h;
            *reinterpret_cast<uint2*>(st + 2 * OA::PLANE + OB::PLANE + off) = l;
        }
    };
    // ---- fragment role: lane i of lane group kq reads row kq * 8 (+ 4) + i / 4, 8 bytes at column quad i % 4 of the tile's 16-column unit
    //      (the swizzle term of row r and of row r + 4 is the same for both row widths: the second half of a fragment is + 4 rows, unswizzled)
    const int fi = lane & 15, kq = lane >> 4;
    const int f_row = kq * 8 + (fi >> 2);
    int offA[TA], offB[TB];
#pragma unroll
    for (int t = 0; t < TA; ++t) offA[t] = OA::byte_of(f_row, (wn * TA + t) * 16) + (fi & 3) * 8;
#pragma unroll
    for (int t = 0; t < TB; ++t) offB[t] = OB::byte_of(f_row, (wk * TB + t) * 16) + (fi & 3) * 8;
    f32x4 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (m_lo < m_hi) {
        fetch(m_lo);
        stash(lds, m_lo);
        if (m_lo + DW_BM < m_hi) fetch(m_lo + DW_BM);
        __syncthreads();
        int cur = 0;
        for (int mc = m_lo; mc < m_hi; mc += DW_BM) {
            const unsigned char* st = lds + cur * STAGE;
            if (mc + DW_BM < m_hi) {
                stash(lds + (cur ^ 1) * STAGE, mc + DW_BM);  // chunk mc + 32 (its loads were issued one iteration ago); nobody reads that stage now
                if (mc + 2 * DW_BM < m_hi) fetch(mc + 2 * DW_BM);  // in flight under the MFMAs below
            }
            s16x8 bh[TB], bl[TB];
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                bh[b] = dw_frag<OB::ROWB>(st + 2 * OA::PLANE, offB[b]);
                bl[b] = dw_frag<OB::ROWB>(st + 2 * OA::PLANE + OB::PLANE, offB[b]);
            }
#pragma unroll
            for (int a = 0; a < TA; ++a) {
                const s16x8 ah = dw_frag<OA::ROWB>(st, offA[a]), al = dw_frag<OA::ROWB>(st + OA::PLANE, offA[a]);
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    if (!HI_ONLY) {  // FCL_GEMM_BF16: bf16-rounded operands, the hi.hi product alone
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b], acc[a][b], 0, 0, 0);
                    }
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], acc[a][b], 0, 0, 0);
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    // ---- accumulate: C/D map of the 16x16 MFMA: column = lane & 15 (k), row = (lane >> 4) * 4 + reg (n)
    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int k = k0 + (wk * TB + b) * 16 + col;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + (wn * TA + a) * 16 + rq * 4 + r;
                if (n < N) atomicAdd(C + (size_t)n * g.ldc + k, acc[a][b][r]);
            }
        }
}

template <int TA, int TB>
static void dw_launch(const DwArgs& g, dim3 grid, bool seg, bool hi_only, hipStream_t stream) {
    if (seg) {
        if (hi_only) hipLaunchKernelGGL((dw_mfma_kernel<TA, TB, true, true>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((dw_mfma_kernel<TA, TB, true, false>), grid, dim3(256), 0, stream, g);
    } else {
        if (hi_only) hipLaunchKernelGGL((dw_mfma_kernel<TA, TB, false, true>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((dw_mfma_kernel<TA, TB, false, false>), grid, dim3(256), 0, stream, g);
    }
}

// the launcher behind fcl_gemm_tn_taps_fwd (backward.hip) for the bf16x3 / bf16 modes; returns false when the shape stays on the old kernel
bool launch_dw_mfma(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, int shift0, int ntaps, size_t c_tap_stride,
                    const int32_t* seg_lo, const int32_t* seg_hi, int hi_only, hipStream_t stream) {
    static const int on = tunable("DW_MFMA", 1);
    static const int min_rows = tunable("DW_MFMA_MIN_ROWS", 256);
    if (!on || m < min_rows || n < 32 || k < 32) return false;  // (k = 4: the position column's gradient)
    // every slice ends in tile-bytes of atomics, so (output bytes x slices) is what a small output pays: few 128 x 128 tiles -> 64 x 64 tiles,
    // which reach the same number of workgroups with a quarter of the slices
    static const int big_min = tunable("DW_BIG_TILES_MIN", 16), wgs_big = tunable("DW_WORKGROUPS", 512), wgs_small = tunable("DW_WORKGROUPS_SMALL", 768);
    static const int min_chunks = tunable("DW_MIN_CHUNKS", 8);
    const int tiles_big = ((n + 127) / 128) * ((k + 127) / 128) * ntaps;
    static const int big_rows = tunable("DW_BIG_TILES_MIN_ROWS", 8192);  // (short contractions: the 64 x 64 form wins whatever the output, r5 sweep)
    const bool big = tiles_big >= big_min && m >= big_rows;
    const int te = big ? 128 : 64;
    const int tiles = ((n + te - 1) / te) * ((k + te - 1) / te) * ntaps;
    int slices = ((big ? wgs_big : wgs_small) + tiles - 1) / tiles;
    int rps = ((m + slices - 1) / slices + DW_BM - 1) / DW_BM * DW_BM;
    if (rps < min_chunks * DW_BM) rps = min_chunks * DW_BM;
    slices = (m + rps - 1) / rps;
    DwArgs g = {a, lda, b, ldb, c, ldc, m, n, k, shift0, seg_lo, seg_hi, rps, ntaps, (long long)c_tap_stride};
    dim3 grid((n + te - 1) / te, (k + te - 1) / te, slices * ntaps);
    ProfScope ps(hi_only ? "dw_mfma_kernel/bf16" : "dw_mfma_kernel", 2.0 * m * (double)n * k * ntaps, m, stream);
    if (big) dw_launch<4, 4>(g, grid, seg_lo != nullptr, hi_only != 0, stream);
    else dw_launch<2, 2>(g, grid, seg_lo != nullptr, hi_only != 0, stream);
    return true;
}

}  // namespace fcl
