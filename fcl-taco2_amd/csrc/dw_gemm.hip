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

constexpr int DW_BM = 32;                   // contraction rows per chunk (one 16x16x32 MFMA step)
constexpr int DW_ROWB = 256;                // bytes per plane row: 128 bf16 columns
constexpr int DW_PLANE = DW_BM * DW_ROWB;   // 8 KB
constexpr int DW_STAGE = 4 * DW_PLANE;      // A hi | A lo | B hi | B lo

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
    int hi_only;
};

__device__ __forceinline__ s16x8 dw_frag(const unsigned char* plane, int off) {
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(plane + off));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(plane + off + 4 * DW_ROWB));
    return (s16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
}

template <bool SEG, bool HI_ONLY>
__global__ __launch_bounds__(256, 2) void dw_mfma_kernel(const DwArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * DW_STAGE];
    // XCD-aware order (as gemm_tn_kernel): the ids one XCD receives walk a contiguous range of (slice, tap, k tile, n tile)
    const int nx = gridDim.x, nxy = gridDim.x * gridDim.y, nwg = nxy * gridDim.z;
    const int orig = (blockIdx.z * gridDim.y + blockIdx.y) * nx + blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tl = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    const int bz = tl / nxy, by = (tl - bz * nxy) / nx, bx = tl - bz * nxy - by * nx;
    const int n0 = bx * 128, k0 = by * 128;
    const int tap = bz % g.ntaps, slice = bz / g.ntaps;
    const int shift = g.shift0 + tap;
    float* __restrict__ C = g.C + (size_t)tap * g.c_tap_stride;
    const int M = g.M, N = g.N, K = g.K;
    const int m_lo = slice * g.rows_per_slice, m_hi = min(M, m_lo + g.rows_per_slice);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;

    // ---- loader role: float4 #(row lr0 + 8 j, columns lc4 .. lc4 + 3), j = 0..3, of both operands
    const int lc4 = (tid & 31) * 4, lr0 = tid >> 5;
    const bool a_col = n0 + lc4 < N, b_col = k0 + lc4 < K;  // N, K are multiples of 4: a float4 is inside or outside as a whole
    const float* __restrict__ ap = g.A + n0 + (a_col ? lc4 : 0);
    const float* __restrict__ bp = g.B + k0 + (b_col ? lc4 : 0);
    f32x4 ra[4], rb[4];
    unsigned okm = 0;  // bit j: A row valid, bit 4 + j: B row valid (applied at the split: no load result is consumed early)
    auto fetch = [&](int mc) {
        okm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mc + lr0 + 8 * j;
            const int mcl = min(m, M - 1);
            ra[j] = *reinterpret_cast<const f32x4*>(ap + (size_t)mcl * g.lda);
            const int src = m + shift;
            bool ok = m < m_hi;
            okm |= (ok && a_col) ? (1u << j) : 0u;
            if (SEG) ok = ok && src >= g.seg_lo[mcl] && src < g.seg_hi[mcl];
            else ok = ok && src >= 0 && src < M;
            rb[j] = *reinterpret_cast<const f32x4*>(bp + (size_t)(ok ? src : mcl) * g.ldb);
            okm |= (ok && b_col) ? (16u << j) : 0u;
        }
    };
    const int st_off = lr0 * DW_ROWB + ((lc4 & 15) << 1);
    const int st_u = lc4 >> 4, st_s = lr0 & 3;
    auto stash = [&](unsigned char* st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 va = ((okm >> j) & 1u) ? ra[j] : z, vb = ((okm >> (4 + j)) & 1u) ? rb[j] : z;
            const int off = st_off + j * 8 * DW_ROWB + ((st_u ^ (st_s | ((j & 1) << 2))) << 5);
            uint2 h, l;
            split4(va, h, l);
            *reinterpret_cast<uint2*>(st + off) = h;
            *reinterpret_cast<uint2*>(st + DW_PLANE + off) = l;
            split4(vb, h, l);
            *reinterpret_cast<uint2*>(st + 2 * DW_PLANE + off) = h;
            *reinterpret_cast<uint2*>(st + 3 * DW_PLANE + off) = l;
        }
    };
    // ---- fragment role: lane i of lane group kq reads row kq * 8 (+ 4) + i / 4, 8 bytes at column quad i % 4 of the tile's 16-column unit
    const int fi = lane & 15, kq = lane >> 4;
    const int f_sw = (fi >> 2) | ((kq & 1) << 2);
    const int f_base = (kq * 8 + (fi >> 2)) * DW_ROWB + (fi & 3) * 8;
    int offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        offA[t] = f_base + (((wn * 4 + t) ^ f_sw) << 5);
        offB[t] = f_base + (((wk * 4 + t) ^ f_sw) << 5);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (m_lo < m_hi) {
        fetch(m_lo);
        stash(lds);
        if (m_lo + DW_BM < m_hi) fetch(m_lo + DW_BM);
        __syncthreads();
        int cur = 0;
        for (int mc = m_lo; mc < m_hi; mc += DW_BM) {
            unsigned char* st = lds + cur * DW_STAGE;
            if (mc + DW_BM < m_hi) {
                stash(lds + (cur ^ 1) * DW_STAGE);  // chunk mc + 32 (its loads were issued one iteration ago); nobody reads that stage now
                if (mc + 2 * DW_BM < m_hi) fetch(mc + 2 * DW_BM);  // in flight under the MFMAs below
            }
            s16x8 bh[4], bl[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                bh[b] = dw_frag(st + 2 * DW_PLANE, offB[b]);
                bl[b] = dw_frag(st + 3 * DW_PLANE, offB[b]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const s16x8 ah = dw_frag(st, offA[a]), al = dw_frag(st + DW_PLANE, offA[a]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
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
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = k0 + wk * 64 + b * 16 + col;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + a * 16 + rq * 4 + r;
                if (n < N) atomicAdd(C + (size_t)n * g.ldc + k, acc[a][b][r]);
            }
        }
}

// the launcher behind fcl_gemm_tn_taps_fwd (backward.hip) for the bf16x3 / bf16 modes; returns false when the shape stays on the old kernel
bool launch_dw_mfma(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, int shift0, int ntaps, size_t c_tap_stride,
                    const int32_t* seg_lo, const int32_t* seg_hi, int hi_only, hipStream_t stream) {
    static const int on = tunable("DW_MFMA", 1);
    static const int min_rows = tunable("DW_MFMA_MIN_ROWS", 256);
    if (!on || m < min_rows || n < 32 || k < 32) return false;  // (k = 4: the position column's gradient)
    const int tiles = ((n + 127) / 128) * ((k + 127) / 128) * ntaps;
    static const int wgs = tunable("DW_WORKGROUPS", 512);  // two 64 KB workgroups per CU
    int slices = (wgs + tiles - 1) / tiles;
    int rps = ((m + slices - 1) / slices + DW_BM - 1) / DW_BM * DW_BM;
    if (rps < 4 * DW_BM) rps = 4 * DW_BM;  // every slice ends in 16 K atomics per tile
    slices = (m + rps - 1) / rps;
    DwArgs g = {a, lda, b, ldb, c, ldc, m, n, k, shift0, seg_lo, seg_hi, rps, ntaps, (long long)c_tap_stride, hi_only};
    dim3 grid((n + 127) / 128, (k + 127) / 128, slices * ntaps);
    ProfScope ps(hi_only ? "dw_mfma_kernel/bf16" : "dw_mfma_kernel", 2.0 * m * (double)n * k * ntaps, m, stream);
    if (seg_lo) {
        if (hi_only) hipLaunchKernelGGL((dw_mfma_kernel<true, true>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((dw_mfma_kernel<true, false>), grid, dim3(256), 0, stream, g);
    } else {
        if (hi_only) hipLaunchKernelGGL((dw_mfma_kernel<false, true>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((dw_mfma_kernel<false, false>), grid, dim3(256), 0, stream, g);
    }
    return true;
}

}  // namespace fcl
