// gemm_planes.hip — the bf16x3 GEMM / LSTM-step core on PRE-SPLIT operand planes, gfx950.
//
// Same contractions as gemm_f32.hip, different data path.  gemm_f32.hip loads fp32 operands, splits them into
// bf16 hi / lo in registers and writes them to LDS every k-chunk: a serial chain (global load -> wait -> ~14 VALU per float4 -> ds_write ->
// barrier) that left the MFMA pipe idle 7/8 of the time at the path's GEMM sizes.  Here every operand arrives already split, in the "P32"
// layout written once by its producer (weights at plan time, activations by the epilogue / pointwise kernel that creates them):
//     P32 of X [R, K]: uint16 [R][ld lines][2][32]: per row and 32-k block ONE 128-byte line = 32 bf16 hi | 32 bf16 lo, zero past K.
// A k-chunk of a tile is then a set of whole 128-byte lines that go global -> LDS by direct LDS-DMA (global_load_lds_dwordx4: no VGPR staging,
// no VALU), through an NST-deep ring with COUNTED vmcnt waits and one raw s_barrier per chunk (guide §5 "Pipelining across barriers").
// LDS image of a chunk: rows of 128 B; the 16-byte piece c = plane*4 + k/8 of row r sits at position c ^ ((r >> 1) & 7), which makes the
// MFMA fragment reads (ds_read_b128: lane (r16, kq) -> row r16, pieces kq and 4+kq) bank-conflict free for the documented lane groups.
// LDS-DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE address (guide §5.4 rule 21: linear destination,
// permuted source, the same permutation on the read).  Rows outside the matrix / outside a conv segment read a static all-zero line.
// Numerics: a.b ~= al.bh + ah.bl + ah.bh on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, exactly the products of the in-kernel split.
#include <string.h>

#include <type_traits>

#include "fcl_common.h"
#include "lstm_epilogue.h"

namespace fcl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned char u8;

__device__ __attribute__((aligned(128))) unsigned int g_zero_line[32];  // 128 bytes of zeros (static device memory is zero-initialised)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// XCD-aware tile order (see gemm_f32.hip): the workgroups one XCD receives walk a contiguous range of tiles.  Round 3: the linear tile order
// itself is GROUPED (groups of g_tile_group row tiles, rows innermost, as in the usual matmul swizzle) instead of row-major: the ~32 workgroups
// an XCD runs at a time then form 8 row tiles x 4 column tiles, not 1 x 32, so per K-chunk step they pull 8 A + 4 W chunk-lines through the
// XCD's L2 instead of 1 A + 32 W -- with W larger than the 4 MB L2 (every frame-sized GEMM of the training step, every FCL-taco2-T LSTM step)
// the row-major order streamed the whole W from the Infinity Cache for every row of tiles.
__constant__ int g_tile_group = 8;
__constant__ int g_plstm_dbg = 0;  // developer timing aid (FCL_PLSTM_DBG=1): the LSTM step returns after its main loop (states are then garbage)
__device__ __forceinline__ void xcd_tile_p(int& bx, int& by) {
    const int nx = gridDim.x, ny = gridDim.y, nwg = nx * ny;
    const int orig = blockIdx.y * nx + blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int G = g_tile_group;
    if (G <= 1) {
        by = t / nx;
        bx = t - by * nx;
        return;
    }
    const int per = G * nx, grp = t / per, first = grp * G, gsz = min(ny - first, G), rem = t - grp * per;
    bx = rem / gsz;
    by = first + (rem - bx * gsz);
}

// one 32-k chunk of a wave's TM x TN tiles: fragment reads (bank-conflict free by the piece permutation) and 3 bf16 MFMAs per tile pair
// HI = 0: bf16x3 split (hi | lo planes, three MFMAs per product); 1: the hi planes alone (autocast); 2 (round 4): EXACT fp32 -- the "planes" are
// plain row-major fp32 matrices (a 128-byte line = 32 consecutive floats of a row instead of 32 hi | 32 lo halves: same line geometry, same LDS-DMA
// ring, same swizzle), consumed by v_mfma_f32_16x16x4_f32: lane (r16, kq) reads floats 4 kq .. 4 kq + 3 and 16 + 4 kq .. + 3 of its row's line (the two
// 16-byte pieces the bf16 form reads as hi and lo) and MFMA e of a piece contracts k = 4 kq + e of all four lane groups -- a permutation of the 32
// k's that is the same for both operands.  8 MFMAs of 32 cycles per (row tile, column tile) and chunk: the loop is MFMA-bound, as an exact-fp32
// GEMM should be (FCL_PRECISION=0; until round 4 that mode ran on gemm_f32.hip's register-staged kernels at 0.32 of the fp32 matrix peak).
template <int TM, int TN, int HI>
__device__ __forceinline__ void pchunk_mma(const u8* sb, int a_hi, int a_lo, int b_hi, int b_lo, f32x4 (&acc)[TM][TN]) {
    if constexpr (HI == 2) {
        f32x4 a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            a0[tm] = *reinterpret_cast<const f32x4*>(sb + a_hi + tm * 16 * 128);
            a1[tm] = *reinterpret_cast<const f32x4*>(sb + a_lo + tm * 16 * 128);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            b0[tn] = *reinterpret_cast<const f32x4*>(sb + b_hi + tn * 16 * 128);
            b1[tn] = *reinterpret_cast<const f32x4*>(sb + b_lo + tn * 16 * 128);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[tm][e], b0[tn][e], acc[tm][tn], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[tm][e], b1[tn][e], acc[tm][tn], 0, 0, 0);
        return;
    }
    s16x8 ah[TM], al[TM], bh[TN], bl[TN];
    if constexpr (HI == 1) {  // FCL_GEMM_BF16 (autocast): operands rounded to bf16 = the hi plane alone, one MFMA per product, fp32 accumulation
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) ah[tm] = *reinterpret_cast<const s16x8*>(sb + a_hi + tm * 16 * 128);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bh[tn] = *reinterpret_cast<const s16x8*>(sb + b_hi + tn * 16 * 128);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
        return;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        ah[tm] = *reinterpret_cast<const s16x8*>(sb + a_hi + tm * 16 * 128);
        al[tm] = *reinterpret_cast<const s16x8*>(sb + a_lo + tm * 16 * 128);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        bh[tn] = *reinterpret_cast<const s16x8*>(sb + b_hi + tn * 16 * 128);
        bl[tn] = *reinterpret_cast<const s16x8*>(sb + b_lo + tn * 16 * 128);
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
    }
}

// LW = 0: every wave loads its share of a chunk and computes.  LW = 2: two extra LOADER waves per workgroup do nothing but the LDS-DMA of all row
// groups (and the counted waits), the WM x WN compute waves nothing but fragment reads and MFMAs; one barrier per chunk joins them.  A wave that
// issues its own global_load_lds cannot issue MFMAs meanwhile (in-order issue), so with ~1 workgroup per CU the two phases used to alternate.
template <int WM, int WN, int TM, int TN, int NST, int LW = 0>
struct PGeo {
    static constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN, NW = WM * WN, THREADS = 64 * (NW + LW), CTHREADS = 64 * NW;
    static constexpr int NL = LW > 0 ? LW : NW;                              // waves that issue LDS-DMA
    static constexpr int GA = BM / 8 / NL, GB = BN / 8 / NL, GPW = GA + GB;  // LDS-DMA row groups (8 rows = 1 KB) per loading wave per chunk
    static constexpr int STAGE = (BM + BN) * 128;
    static constexpr int LDS_BYTES = NST * STAGE;
    static_assert((BM / 8) % NL == 0 && (BN / 8) % NL == 0 && (NL % 2) == 0, "tile rows must split evenly over the loading waves");
    static_assert(NST >= 2 && NST <= 4, "ring depth (2: 64 KB for a 128 x 128 tile -> TWO workgroups per CU, one's epilogue beside the other's main loop)");
    static_assert(2 * GPW <= 60, "s_waitcnt vmcnt is a 6-bit field");
};

// The shared main loop.  LSTM: tile column c of the wave strip wn is gate (c >> 4) & 3 of unit u0 + wn*16 + (c & 15), i.e. W row g*NU + u
// (TN must be 4); generic: W row n0 + c.  NU = N (generic) or U.  Returns false in a loader wave (LW > 0), which is done and must return.
template <int WM, int WN, int TM, int TN, int NST, bool LSTM, int LW, int HI>
__device__ __forceinline__ bool pmainloop(const GemmTerm* __restrict__ terms, int nterms, int M, int m0, int n0, int NU, const int* __restrict__ seg_lo,
                                          const int* __restrict__ seg_hi, u8* smem, f32x4 (&acc)[TM][TN], int ksplit_chunks = 0) {
    using G = PGeo<WM, WN, TM, TN, NST, LW>;
    static_assert(!LSTM || TN == 4, "LSTM tiles keep the four gates of a unit in one lane");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LW > 0 && wave >= G::NW;
    const int lw = LW > 0 ? wave - G::NW : wave;  // index among the loading waves
    const int wm = wave / WN, wn = wave % WN;
    int nchunks = 0;
    for (int t = 0; t < nterms; ++t) nchunks += (terms[t].K + 31) >> 5;
    int kskip = 0;  // split contraction (single term): slice blockIdx.z owns chunks [kskip, kskip + nchunks)
    if (ksplit_chunks > 0) {
        kskip = (int)blockIdx.z * ksplit_chunks;
        nchunks = max(0, min(ksplit_chunks, nchunks - kskip));
    }

    if (LW == 0 || loader) {
        // ---- loader coordinates: group g = j*NL + lw covers rows g*8 .. g*8+7 of its region; lane -> (row = lane >> 3, LDS piece = lane & 7)
        const unsigned coff = (unsigned)(((lane & 7) ^ (((lw & 1) << 2) | (lane >> 4))) * 16);  // SOURCE piece for this lane's LDS slot
        const u8* zline = reinterpret_cast<const u8*>(g_zero_line) + coff;
        int am[G::GA], alo[G::GA];
        unsigned alen[G::GA];
#pragma unroll
        for (int j = 0; j < G::GA; ++j) {
            const int m = m0 + (j * G::NL + lw) * 8 + (lane >> 3);
            am[j] = m;
            alo[j] = 0;
            alen[j] = m < M ? (unsigned)M : 0u;  // rows past M: empty segment -> zero line
            if (seg_lo != nullptr && m < M) {
                alo[j] = seg_lo[m];
                alen[j] = (unsigned)(seg_hi[m] - alo[j]);
            }
        }
        long long brow[G::GB];  // W row index, -1 = zero line
#pragma unroll
        for (int j = 0; j < G::GB; ++j) {
            const int c = (j * G::NL + lw) * 8 + (lane >> 3);  // tile column
            long long wr = -1;
            if (LSTM) {
                const int u = n0 + (c >> 6) * 16 + (c & 15);
                const int g = (c >> 4) & 3;
                if (u < NU) wr = (long long)g * NU + u;
            } else {
                if (n0 + c < NU) wr = n0 + c;
            }
            brow[j] = wr;
        }
        // issue-side state: per-lane source pointers, advanced one 128-byte line per chunk (zero-line pointers do not move)
        const u8* pa[G::GA];
        const u8* pb[G::GB];
        unsigned ia[G::GA], ib[G::GB];
        int rem = 0, it = 0;
        auto setup_term = [&](int t) {
            const GemmTerm T = terms[t];
            const u8* Ab = reinterpret_cast<const u8*>(T.Ap);
            const u8* Wb = reinterpret_cast<const u8*>(T.Wp);
#pragma unroll
            for (int j = 0; j < G::GA; ++j) {
                const int src = am[j] + T.shift;
                const bool ok = (unsigned)(src - alo[j]) < alen[j];
                if (T.a_chunk_stride) {  // chunk-major A planes: consecutive rows of one chunk are consecutive 128-byte lines
                    pa[j] = ok ? Ab + (size_t)src * 128 + (size_t)kskip * (size_t)T.a_chunk_stride + coff : zline;
                    ia[j] = ok ? (unsigned)T.a_chunk_stride : 0u;
                } else {
                    pa[j] = ok ? Ab + ((size_t)src * (size_t)T.lda_p + kskip) * 128 + coff : zline;
                    ia[j] = ok ? 128u : 0u;
                }
            }
#pragma unroll
            for (int j = 0; j < G::GB; ++j) {
                const bool ok = brow[j] >= 0;
                pb[j] = ok ? Wb + ((size_t)brow[j] * (size_t)T.ldw_p + kskip) * 128 + coff : zline;
                ib[j] = ok ? 128u : 0u;
            }
            rem = ((T.K + 31) >> 5) - kskip;
        };
        auto issue = [&](int stage) {
            u8* sbase = smem + stage * G::STAGE + lw * 1024;
#pragma unroll
            for (int j = 0; j < G::GA; ++j) {
                glds16(pa[j], sbase + j * G::NL * 1024);
                pa[j] += ia[j];
            }
#pragma unroll
            for (int j = 0; j < G::GB; ++j) {
                glds16(pb[j], sbase + G::BM * 128 + j * G::NL * 1024);
                pb[j] += ib[j];
            }
            if (--rem == 0 && ++it < nterms) setup_term(it);  // rare, wave-uniform
        };
        setup_term(0);
        int issued = 0;
#pragma unroll
        for (int p = 0; p < NST - 1; ++p) {
            if (issued < nchunks) { issue(p); ++issued; }
        }
        if (LW > 0) {  // loader wave: waits + barriers + refills only
            int is = NST - 1;
            for (int i = 0; i < nchunks; ++i) {
                const int left = nchunks - 1 - i;
                if (NST == 2) wait_vm<0>();  // two stages: only the chunk consumed next is in flight
                else if (NST >= 4 && left >= 2) wait_vm<2 * G::GPW>();
                else if (left >= 1) wait_vm<G::GPW>();
                else wait_vm<0>();
                asm volatile("s_barrier" ::: "memory");
                if (left >= NST - 1) {
                    issue(is);
                    is = is + 1 == NST ? 0 : is + 1;
                }
            }
            return false;
        }
        // ---- LW == 0: the same waves also compute
        const int r16 = lane & 15, kq = lane >> 4;
        const int sw = r16 >> 1;
        const int a_hi = (wm * TM * 16 + r16) * 128 + ((kq ^ sw) << 4);
        const int a_lo = (wm * TM * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
        const int b_hi = G::BM * 128 + (wn * TN * 16 + r16) * 128 + ((kq ^ sw) << 4);
        const int b_lo = G::BM * 128 + (wn * TN * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
        int cs = 0, is = NST - 1;
        for (int i = 0; i < nchunks; ++i) {  // chunk i is consumed while chunks i+1 .. i+NST-2 stay in flight and chunk i+NST-1 is issued
            const int left = nchunks - 1 - i;
            if (NST == 2) wait_vm<0>();
            else if (NST >= 4 && left >= 2) wait_vm<2 * G::GPW>();
            else if (left >= 1) wait_vm<G::GPW>();
            else wait_vm<0>();
            asm volatile("s_barrier" ::: "memory");  // chunk i has landed for every wave; every wave is done reading the buffer refilled now
            if (left >= NST - 1) {
                issue(is);
                is = is + 1 == NST ? 0 : is + 1;
            }
            pchunk_mma<TM, TN, HI>(smem + cs * G::STAGE, a_hi, a_lo, b_hi, b_lo, acc);
            cs = cs + 1 == NST ? 0 : cs + 1;
        }
        return true;
    }
    // ---- compute wave of a loader-specialised workgroup
    {
        const int r16 = lane & 15, kq = lane >> 4;
        const int sw = r16 >> 1;
        const int a_hi = (wm * TM * 16 + r16) * 128 + ((kq ^ sw) << 4);
        const int a_lo = (wm * TM * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
        const int b_hi = G::BM * 128 + (wn * TN * 16 + r16) * 128 + ((kq ^ sw) << 4);
        const int b_lo = G::BM * 128 + (wn * TN * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
        int cs = 0;
        for (int i = 0; i < nchunks; ++i) {
            asm volatile("s_barrier" ::: "memory");  // the loader waves have seen chunk i land before they arrive here
            pchunk_mma<TM, TN, HI>(smem + cs * G::STAGE, a_hi, a_lo, b_hi, b_lo, acc);
            cs = cs + 1 == NST ? 0 : cs + 1;
        }
    }
    return true;
}

// --------------------------------------------------------------------------------------------------------------------------------------
// The GEMM epilogue shared by pgemm_kernel and pconv_kernel (compute waves only; BM x BN tile at (m0, n0), accumulators in the MFMA layout).
// Row permutation of a 16-row MFMA tile used by the Conv1d stencil (pconv_kernel): MFMA row i of the fragment holds tile row rho16(i).
// ds_read_b128 is serviced in four fixed 16-lane groups, each = eight rows read at k-piece a plus eight rows read at piece a ^ 1 ({0-3, 12-15} and
// {4-11} of the 16 rows).  With the line swizzle piece ^ ((row >> 1) & 7) two such rows collide iff they have the same parity and sit in the same
// aligned group of four rows -- never for a tile that starts at a multiple of 4 rows (every GEMM), but the stencil reads the SAME tile at row offsets
// -pad .. +pad: 2 of 16 lanes collide for odd offsets, 4 for offsets = 2 (mod 4) (r2 profile: 27 % / 19 % of the LDS cycles of the two
// instantiations).  rho16 hands the even rows to one piece-set and the odd rows to the other, so two rows of different sets always differ in
// parity (address bit 7 = other half of the 64 banks) at EVERY offset: conflict-free by construction (all 16 alignments enumerated offline).
__device__ __forceinline__ int rho16(int i) { return i < 4 ? 2 * i : (i < 12 ? 2 * (i - 4) + 1 : 2 * (i - 8)); }

template <int WN, int TM, int TN, int BM, int BN, int CTHREADS, int LDS_BYTES, bool RP = false>
__device__ __forceinline__ void pgemm_epilogue(const GemmArgs& a_, f32x4 (&acc)[TM][TN], u8* smem, int m0, int n0, long long ob = 0, long long oy = 0,
                                               long long oyp = 0) {
    // (ob / oy / oyp: this group's offsets into bias / Y / Yp for the grouped Conv1d; 0 otherwise)
    const GemmArgs& a = a_;
    const float* const e_bias = a.bias ? a.bias + ob : nullptr;
    float* const e_Y = a.Y ? a.Y + oy : nullptr;
    unsigned short* const e_Yp = a.Yp ? a.Yp + oyp : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, rq = lane >> 4;
    const unsigned int seed = hash_u32(a.rng_seed + (a.seed_dev ? *a.seed_dev * 0x9E3779B9u : 0u));
    // Every argument the element loop reads, ONCE, into locals: the loop contains stores through pointers that came from the argument block (Y2),
    // so the compiler re-read each a.<field> from the kernel-argument segment for every one of the 32 elements (156 s_load + 94 s_waitcnt lgkmcnt(0)
    // in the ISA of the 128 x 128 instantiation: ~5 us per workgroup, the fixed cost every kernel of this family paid per tile -- r3 probe)
    const int a_M = a.M, a_N = a.N, a_act = a.act, a_drop = a.drop_mode, a_ldkeep = a.ldkeep, a_ldr = a.ldr, a_ldc0 = a.ldc0, a_r1lda = a.rank1_lda;
    const int a_ldy2 = a.ldy2, a_y2add = a.y2_row_add;
    const float a_kscale = a.keep_scale, a_dropp = a.drop_p;
    const float* const a_r1a = a.rank1_a;
    const float* const a_r1w = a.rank1_w;
    const float* const a_C0 = a.C0;
    const float* const a_R = a.R;
    const uint8_t* const a_keep = a.keep;
    float* const a_Y2 = a.Y2;
    const int* const a_y2base = a.y2_row_base;
    // Epilogue through LDS: the MFMA accumulator layout gives every lane ONE column of four rows, i.e. 4-byte (fp32) or 2-byte (planes) stores,
    // 32 - 64 store instructions per lane.  The finished tile is staged as fp32 in the (now idle) ring and written out row-wise: 16 bytes per
    // lane, whole 128-byte lines for the planes (hi | lo of 32 columns), a quarter / an eighth of the store instructions.
    // staging row stride (floats): + 4 keeps float4 alignment and spreads rows over banks; a two-stage ring (64 KB) holds the 128 x 128 tile only
    // unpadded (rows 4 apart then share a bank pair: 2-way on ds_write_b32, which costs nothing extra; the row-wise reads stay conflict-free)
    constexpr int LDT = BM * (BN + 4) * 4 <= LDS_BYTES ? BN + 4 : BN;
    static_assert(BM * LDT * 4 <= LDS_BYTES, "the staging tile must fit the ring");
    float* tile = reinterpret_cast<float*>(smem);
    __syncthreads();  // every wave is done with the last chunk
    if (a.dbg_phase == 3) {
        if (acc[0][0][0] + acc[TM - 1][TN - 1][3] == 12345.678f) a.Y[1] = 1.f;
        return;
    }
    if (a.dbg_phase == 4) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[((wm * TM + tm) * 16 + rq * 4 + r) * LDT + (wn * TN + tn) * 16 + col] = acc[tm][tn][r];
        __syncthreads();
        if (tile[threadIdx.x] == 12345.678f) a.Y[1] = 1.f;
        return;
    }
    // The element loop is INSTRUCTION-bound: 32 elements per lane, and with every option tested per element (rank-1 term, C0, two dropout modes,
    // residual, second output) ~40 instructions and ~10 branches each = 1 300 instructions per wave, two waves per SIMD: ~5 us per workgroup
    // (r3 probe: 31 of the 80 us of a 24 320 x 1 024 x 256 GEMM).  The options are wave-uniform, so they are tested ONCE: the plain forms
    // (bias + activation, what most launches are) run a 3-instruction element, everything else the general loop.
    const bool plain = !a_r1a && !a_C0 && a_drop == 0 && !a_Y2 && a.dbg_phase != 5;  // (FCL_PGEMM_DBG=5: the general loop, for A/B timing)
    auto plain_loop = [&](auto act_c, auto res_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool RES = decltype(res_c)::value;  // + R[m, n] behind the activation (the postnet's `before + postnet(before)`, encoder skips)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int cn = (wn * TN + tn) * 16 + col, n = n0 + cn;
                const bool nin = n < a_N;
                const float bn = (e_bias && nin) ? e_bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rm = (wm * TM + tm) * 16 + rq * 4 + r, m = m0 + (RP ? (wm * TM + tm) * 16 + rho16(rq * 4 + r) : rm);
                    float v = acc[tm][tn][r] + bn;
                    if (ACT == FCL_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (ACT == FCL_ACT_TANH) v = tanh_f(v);
                    const bool in = nin && m < a_M;
                    if (RES) v += a_R[(size_t)(in ? m : 0) * a_ldr + (in ? n : 0)];
                    tile[rm * LDT + cn] = in ? v : 0.f;  // zeros past N / M: the planes' padding
                }
            }
    };
    using std::integral_constant;
    if (plain && !a_R && a_act == FCL_ACT_NONE) plain_loop(integral_constant<int, FCL_ACT_NONE>(), integral_constant<bool, false>());
    else if (plain && !a_R && a_act == FCL_ACT_RELU) plain_loop(integral_constant<int, FCL_ACT_RELU>(), integral_constant<bool, false>());
    else if (plain && !a_R && a_act == FCL_ACT_TANH) plain_loop(integral_constant<int, FCL_ACT_TANH>(), integral_constant<bool, false>());
    else if (plain && a_act == FCL_ACT_NONE) plain_loop(integral_constant<int, FCL_ACT_NONE>(), integral_constant<bool, true>());
    else if (plain && a_act == FCL_ACT_RELU) plain_loop(integral_constant<int, FCL_ACT_RELU>(), integral_constant<bool, true>());
    else {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int cn = (wn * TN + tn) * 16 + col, n = n0 + cn;
            const bool nin = n < a_N;
            const float bn = (e_bias && nin) ? e_bias[n] : 0.f;
            const float r1w = (a_r1w && nin) ? a_r1w[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // staging row = MFMA row order (conflict-free ds_write pattern); RP: the tile row it holds is rho16 of it
                const int rm = (wm * TM + tm) * 16 + rq * 4 + r, m = m0 + (RP ? (wm * TM + tm) * 16 + rho16(rq * 4 + r) : rm);
                float v = 0.f;
                if (nin && m < a_M) {
                    v = acc[tm][tn][r] + bn;
                    if (a_r1a) v += a_r1a[(size_t)m * a_r1lda] * r1w;
                    if (a_C0) v += a_C0[(size_t)m * a_ldc0 + n];
                    if (a_act == FCL_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (a_act == FCL_ACT_TANH) v = tanh_f(v);
                    if (a_drop == 1) {
                        v = a_keep[(size_t)m * a_ldkeep + n] ? v * a_kscale : 0.f;
                    } else if (a_drop == 2) {
                        const unsigned int h = hash_u32(((unsigned int)m * (unsigned int)a_N + (unsigned int)n) ^ seed);
                        v = ((h >> 8) * (1.0f / 16777216.0f) >= a_dropp) ? v * a_kscale : 0.f;
                    }
                    if (a_R) v += a_R[(size_t)m * a_ldr + n];
                    if (a_Y2) a_Y2[(size_t)(a_y2base[m] + a_y2add) * a_ldy2 + n] = v;
                }
                tile[rm * LDT + cn] = v;  // zeros past N / M: the planes' padding
            }
        }
    }
    __syncthreads();
    if (a.dbg_phase == 2) return;  // developer timing aid: no write-out
    const int rows = RP ? BM : min(BM, a_M - m0);  // staging rows to write out; RP: staging row rm holds tile row (rm & ~15) + rho16(rm & 15)
    auto trow = [&](int rm) { return RP ? (rm & ~15) + rho16(rm & 15) : rm; };
    if (a.bn_ws) {  // BatchNorm statistics of the staged outputs (GemmArgs::bn_ws): rows past M and columns past N are staged as zeros
        __shared__ int bn_last;
        constexpr int RG = CTHREADS / BN;  // row groups: RG threads per column
        static_assert(CTHREADS % BN == 0 && RG >= 1, "one or more whole threads per tile column");
        const int cn = threadIdx.x % BN, rg = threadIdx.x / BN, n = n0 + cn;
        double sm = 0.0, sq = 0.0;
        for (int rm = rg; rm < BM; rm += RG) {
            const double d = (double)tile[rm * LDT + cn];
            sm += d;
            sq += d * d;
        }
        if (n < a_N) {
            // RETURNING atomics: the wave waits for the old values, i.e. both adds are performed at the coherence point before the workgroup draws its ticket
            const double o1 = atomicAdd(a.bn_ws + n, sm);
            const double o2 = atomicAdd(a.bn_ws + a_N + n, sq);
            asm volatile("" ::"v"(o1), "v"(o2));
        }
        __syncthreads();
        const int col_tile = n0 / BN, row_tiles = (a_M + BM - 1) / BM;
        if (threadIdx.x == 0) bn_last = atomicAdd(a.bn_tickets + col_tile, 1u) == (unsigned)(row_tiles - 1);
        __syncthreads();
        if (bn_last && rg == 0 && n < a_N) {
            const double S = atomicAdd(a.bn_ws + n, 0.0), Q = atomicAdd(a.bn_ws + a_N + n, 0.0);  // (atomic reads: served where the other workgroups' atomics landed)
            const double mu = S / a_M;
            double var = Q / a_M - mu * mu;
            if (var < 0.0) var = 0.0;
            a.bn_mean[n] = (float)mu;
            a.bn_invstd[n] = (float)(1.0 / sqrt(var + (double)a.bn_eps));
            if (a.bn_rmean) {
                const double unbiased = a_M > 1 ? var * (double)a_M / (double)(a_M - 1) : var;
                a.bn_rmean[n] = (float)((1.0 - a.bn_momentum) * a.bn_rmean[n] + a.bn_momentum * mu);
                a.bn_rvar[n] = (float)((1.0 - a.bn_momentum) * a.bn_rvar[n] + a.bn_momentum * unbiased);
            }
            a.bn_ws[n] = 0.0;
            a.bn_ws[a_N + n] = 0.0;
            if (cn == 0) a.bn_tickets[col_tile] = 0u;
        }
    }
    if (a.loss_t) {  // MSE epilogue (GemmArgs::loss_t): the staged y becomes the gradient in place, row-wise (16 bytes of the target per lane), sums per workgroup
        __shared__ double loss_part[16][3];
        const float* const lt = a.loss_t;
        const uint8_t* const lv = a.loss_valid;
        const int ldlt = a.ld_lt;
        const float gsc = a.loss_gscale;
        double s1 = 0.0, s2 = 0.0, cnt = 0.0;
        for (int i = threadIdx.x; i < rows * (BN / 4); i += CTHREADS) {
            const int rm = i / (BN / 4), c4 = (i - rm * (BN / 4)) * 4, n = n0 + c4, gm = m0 + trow(rm);
            if (n >= a_N || gm >= a_M) continue;  // (the staged padding is zero already)
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (!lv || lv[gm]) {
                const f32x4 y = *reinterpret_cast<const f32x4*>(tile + rm * LDT + c4);
                const f32x4 t = *reinterpret_cast<const f32x4*>(lt + (size_t)gm * ldlt + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = y[e] - t[e];
                    s1 += fabsf(d);
                    s2 += (double)d * d;
                    g[e] = loss_grad1(d, 0.f, 1.f, gsc);
                }
                cnt += 4.0;
            }
            *reinterpret_cast<f32x4*>(tile + rm * LDT + c4) = g;
        }
        double v[3] = {s1, s2, cnt};
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o);
        if (lane == 0)
#pragma unroll
            for (int q = 0; q < 3; ++q) loss_part[wave][q] = v[q];
        __syncthreads();  // the gradients are staged, the waves' partial sums are in LDS
        if (threadIdx.x < 3) {
            double s = 0.0;
            for (int w = 0; w < CTHREADS / 64; ++w) s += loss_part[w][threadIdx.x];
            if (s != 0.0) atomicAdd(a.loss_sums + threadIdx.x, s);
        }
    }
    const int a_ldy = a.ldy, a_ldyp = a.ldyp, a_nblk = a.nblk;  // (locals: the loops below store through argument pointers, see above)
    const long long a_blk = a.blk_stride;
    if (a.accumulate) {  // weight gradients (split contraction, accumulation over micro-batches): one float per lane, consecutive lanes on
                         // consecutive addresses, so a wave's atomic instruction touches two cache lines
        float* const Yacc = a.Y;
        for (int i = threadIdx.x; i < rows * BN; i += CTHREADS) {
            const int rm = i / BN, cn = i - rm * BN, n = n0 + cn, gm = m0 + trow(rm);
            if (n >= a_N || gm >= a_M) continue;
            float* dst = a_nblk > 0 ? Yacc + (size_t)(n / a_nblk) * a_blk + (size_t)gm * a_ldy + (n % a_nblk) : Yacc + (size_t)gm * a_ldy + n;
            atomicAdd(dst, tile[rm * LDT + cn]);
        }
        return;
    }
    if (e_Y) {
        const bool vec = (a_ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(e_Y) & 15u) == 0;
        for (int i = threadIdx.x; i < rows * (BN / 4); i += CTHREADS) {
            const int rm = i / (BN / 4), c4 = (i - rm * (BN / 4)) * 4, n = n0 + c4, gm = m0 + trow(rm);
            if (n >= a_N || gm >= a_M) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tile + rm * LDT + c4);
            float* dst = e_Y + (size_t)gm * a_ldy + n;
            if (vec && n + 3 < a_N) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < a_N) dst[e] = v[e];
            }
        }
    }
    if (e_Yp) {  // item = (row, 32-column line, quarter q): 8 values -> 16 bytes of hi at piece q and 16 bytes of lo at piece 4 + q
        const int np = a_ldyp * 32;
        for (int i = threadIdx.x; i < rows * (BN / 8); i += CTHREADS) {
            const int rm = i / (BN / 8), c8 = (i - rm * (BN / 8)) * 8, n = n0 + c8, gm = m0 + trow(rm);
            if (n >= np || gm >= a_M) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(tile + rm * LDT + c8), v1 = *reinterpret_cast<const f32x4*>(tile + rm * LDT + c8 + 4);
            uint2 h0, l0, h1, l1;
            split4(v0, h0, l0);
            split4(v1, h1, l1);
            u16* line = e_Yp + ((size_t)gm * a_ldyp + (n >> 5)) * 64 + (n & 31);
            *reinterpret_cast<uint4*>(line) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4*>(line + 32) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
    }
}

// (HIP's second __launch_bounds__ argument is waves per SIMD: two stages = 64 KB of LDS = two 12-wave workgroups per CU = 6 waves per SIMD, i.e. at
// most 80 VGPRs -- the three-stage instantiation needs 78, so asking for it costs nothing)
template <int WM, int WN, int TM, int TN, int NST, int LW, int HI>
__global__ __launch_bounds__(64 * (WM * WN + LW), (NST == 2 && WM * WN + LW == 12 && HI != 2) ? 6 : 1) void pgemm_kernel(const GemmArgs a) {
    using G = PGeo<WM, WN, TM, TN, NST, LW>;
    extern __shared__ __attribute__((aligned(1024))) u8 smem[];
    int bx, by;
    xcd_tile_p(bx, by);
    const int m0 = by * G::BM, n0 = bx * G::BN;
    if (a.m_dev && m0 >= *a.m_dev) return;  // tile beyond the device's row count (uniform per workgroup, before any barrier / LDS-DMA)
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!pmainloop<WM, WN, TM, TN, NST, false, LW, HI>(a.term, a.nterms, a.M, m0, n0, a.N, a.seg_lo, a.seg_hi, smem, acc, a.ksplit_chunks)) return;  // loader wave
    if (a.dbg_phase == 1) {  // developer timing aid (FCL_PGEMM_DBG=1): main loop only, results are garbage
        float sdbg = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) sdbg += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        if (sdbg == 12345.678f) a.Y[0] = sdbg;
        return;
    }

    pgemm_epilogue<WN, TM, TN, G::BM, G::BN, G::CTHREADS, G::LDS_BYTES>(a, acc, smem, m0, n0);
}

// NST = 2 (two ring stages, 64 KB: TWO workgroups per CU, 6 waves per SIMD, <= 80 VGPRs): the epilogue operands are then NOT requested before
// the K loop (48 VGPRs held across it) but where they are used -- the other workgroup's main loop covers their latency.
template <int WM, int WN, int TM, int NST, int MODE, int LW, int HI>
__device__ __forceinline__ void plstm_body(const LstmStepArgs& a, u8* smem) {
    using G = PGeo<WM, WN, TM, 4, NST, LW>;
    int bx, by;
    xcd_tile_p(bx, by);
    const int m0 = by * G::BM, u0 = bx * (16 * WN);
    // device-driven loops: the LDS-DMA stream and the MFMAs run on the host's row bound; only the write-out is limited to the device's live-row
    // count, whose scalar load is then hidden behind the whole K loop
    const int M = a.M, Ms = live_rows_of(a.M, a.m_dev);
    if (m0 >= Ms) return;  // tile beyond the device's live rows (uniform per workgroup, before any barrier / LDS-DMA)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int u = u0 + wn * 16 + (lane & 15);
    const int rq = lane >> 4;
    // epilogue operands (G0 / bias / position / old state) are requested BEFORE the K loop (plain loads: they are older than every LDS-DMA
    // piece, so the counted vmcnt waits of the loop cover them and their latency hides under it)
    constexpr bool PRE = NST != 2;
    CellIn ci[PRE ? TM : 1][4];
    const int uc = min(u, a.U - 1);
    if (PRE && (LW == 0 || wave < G::NW)) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r) ci[tm][r] = cell_prefetch<MODE>(a, min(m0 + (wm * TM + tm) * 16 + rq * 4 + r, M - 1), uc);
    }
    f32x4 acc[TM][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tm][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!pmainloop<WM, WN, TM, 4, NST, true, LW, HI>(a.term, a.nterms, M, m0, u0, a.U, nullptr, nullptr, smem, acc)) return;  // loader wave
    if (g_plstm_dbg == 1) {
        float sdbg = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sdbg += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        if (sdbg == 12345.678f) a.h_out[0] = sdbg;
        return;
    }
    // Epilogue through LDS (see pgemm_kernel): the new h and c of the tile are staged as [row][unit] fp32 and written out row-wise, 16 bytes
    // per lane; the tile's 16 WN units are (part of) ONE 128-byte P32 line per row, so the planes of h go out as whole 16-byte pieces too.
    constexpr int UW = 16 * WN, LDT = UW + 4;
    static_assert(2 * G::BM * LDT * 4 <= G::LDS_BYTES, "the staging tiles must fit the ring");
    float* th = reinterpret_cast<float*>(smem);
    float* tc = th + G::BM * LDT;
    __syncthreads();  // every wave is done with the last chunk
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rm = (wm * TM + tm) * 16 + rq * 4 + r, m = m0 + rm;
            float h_w = 0.f, c_w = 0.f;
            if (m < Ms && u < a.U) {  // (cell_math also stores the optional taps / saved gates: live rows only)
                const float pre[4] = {acc[tm][0][r], acc[tm][1][r], acc[tm][2][r], acc[tm][3][r]};
                if (PRE) {
                    cell_math<MODE>(a, m, u, pre, ci[PRE ? tm : 0][r], h_w, c_w);
                } else {
                    const CellIn cl = cell_prefetch<MODE>(a, m, u);
                    cell_math<MODE>(a, m, u, pre, cl, h_w, c_w);
                }
            }
            th[rm * LDT + wn * 16 + (lane & 15)] = h_w;
            tc[rm * LDT + wn * 16 + (lane & 15)] = c_w;
        }
    __syncthreads();
    const int rows = min(G::BM, Ms - m0);  // (<= 0: nothing to write)
    const bool vec = (a.U & 3) == 0;
    for (int i = threadIdx.x; i < rows * (UW / 4); i += G::CTHREADS) {
        const int rm = i / (UW / 4), c4 = (i - rm * (UW / 4)) * 4, uu = u0 + c4;
        if (uu >= a.U) continue;
        const f32x4 hv = *reinterpret_cast<const f32x4*>(th + rm * LDT + c4), cv = *reinterpret_cast<const f32x4*>(tc + rm * LDT + c4);
        const size_t off = (size_t)(m0 + rm) * a.U + uu;
        if (vec && uu + 3 < a.U) {
            *reinterpret_cast<f32x4*>(a.h_out + off) = hv;
            *reinterpret_cast<f32x4*>(a.c + off) = cv;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (uu + e < a.U) { a.h_out[off + e] = hv[e]; a.c[off + e] = cv[e]; }
        }
    }
    if (a.h_out_p) {  // U % 32 == 0 (whole lines): item = (row, 8 units) -> 16 bytes of hi and 16 bytes of lo
        for (int i = threadIdx.x; i < rows * (UW / 8); i += G::CTHREADS) {
            const int rm = i / (UW / 8), c8 = (i - rm * (UW / 8)) * 8, uu = u0 + c8;
            if (uu >= a.U) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(th + rm * LDT + c8), v1 = *reinterpret_cast<const f32x4*>(th + rm * LDT + c8 + 4);
            uint2 h0, l0, h1, l1;
            split4(v0, h0, l0);
            split4(v1, h1, l1);
            u16* line = a.h_out_p + ((size_t)(m0 + rm) * a.ld_hp + (uu >> 5)) * 64 + (uu & 31);
            *reinterpret_cast<uint4*>(line) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4*>(line + 32) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
    }
}

template <int WM, int WN, int TM, int NST, int MODE, int LW, int HI>
__global__ __launch_bounds__(64 * (WM * WN + LW), (NST == 2 && WM * WN + LW == 12 && HI != 2) ? 6 : 1) void plstm_kernel(const LstmStepArgs a) {
    extern __shared__ __attribute__((aligned(1024))) u8 smem[];
    plstm_body<WM, WN, TM, NST, MODE, LW, HI>(a, smem);
}

// Two INDEPENDENT steps in one launch (blockIdx.z picks the problem; the grid covers the larger row count, tiles beyond a problem's rows exit at
// once).  Under teacher forcing layer 0 of step t + 1 needs the ground-truth frame and h0(t), not h1(t): it runs beside layer 1 of step t
// (fcl_decoder_train_fwd) -- half the dependent launches of the training forward and twice the workgroups per launch (fewer part-filled rounds).
template <int WM, int WN, int TM, int NST, int LW, int HI>
__global__ __launch_bounds__(64 * (WM * WN + LW), (NST == 2 && WM * WN + LW == 12 && HI != 2) ? 6 : 1) void plstm_pair_kernel(const LstmStepArgs a0, const LstmStepArgs a1) {
    extern __shared__ __attribute__((aligned(1024))) u8 smem[];
    if (blockIdx.z == 0) plstm_body<WM, WN, TM, NST, -1, LW, HI>(a0, smem);
    else plstm_body<WM, WN, TM, NST, -1, LW, HI>(a1, smem);
}

// experiments only: FCL_TILE_GROUP=<g> overrides the grouped tile order's group size (1 = row-major, rounds 1-2) on the calling device
static void tile_group_override() {
    static const int done = [] {
        const int v = tunable("TILE_GROUP", -1);
        if (v >= 1) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_group), &v, sizeof(int));
        const int d = tunable("PLSTM_DBG", 0);
        if (d) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_plstm_dbg), &d, sizeof(int));
        return 1;
    }();
    (void)done;
}

// --------------------------------------------------------------------------------------------------------------------------------------
bool planes_ok(const GemmTerm* t, int n) {
    tile_group_override();
    static const int on = tunable("PLANES", 1);
    if (!on) return false;
    for (int i = 0; i < n; ++i)
        if (!t[i].Ap || !t[i].Wp || t[i].lda_p * 32 < t[i].K || t[i].ldw_p * 32 < t[i].K) return false;
    return true;
}

static int loader_waves() {
    // dedicated LDS-DMA waves per workgroup (0: every wave loads and computes).  Measured: 2 loaders -9 % per isolated launch over 0; 4 loaders another
    // +1.5 % (S, batch 32: 41.8 -> 42.4 M frames/s over 3 runs each) to +3.6 % (batch 64) on the whole pass -- the issue rate of global_load_lds per
    // wave is part of the per-chunk time
    static const int t = tunable("PLANES_LOADERS", 4);
    static const int v = t >= 4 ? 4 : (t ? 2 : 0);
    return v;
}

template <int WM, int WN, int TM, int TN, int NST, int LW, int HI>
static int launch_pgemm_lw(const GemmArgs& a, hipStream_t s, double flops) {
    using G = PGeo<WM, WN, TM, TN, NST, LW>;
    auto k = pgemm_kernel<WM, WN, TM, TN, NST, LW, HI>;
    const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(k), G::LDS_BYTES);
    if (rc) return rc;
    const int ncols = a.Yp ? max(a.N, a.ldyp * 32) : a.N;  // the tiles also cover the zero padding of the output planes
    dim3 grid((ncols + G::BN - 1) / G::BN, (a.M + G::BM - 1) / G::BM);
    char full[64];
    snprintf(full, sizeof(full), "pgemm_kernel<%d,%d,%d,%d,%d,%d>%s%s", WM, WN, TM, TN, NST, LW, HI == 2 ? "/f32" : HI ? "/bf16" : "", a.accumulate ? "/dW" : "");
    static const int shapes = tunable("PROF_SHAPES", 0);  // developer aid: the profile records split by shape (M x N x sum K)
    if (shapes && g_prof_on) {
        long long ks = 0;
        for (int i = 0; i < a.nterms; ++i) ks += a.term[i].K;
        const size_t l = strlen(full);
        snprintf(full + l, sizeof(full) - l, " %dx%dx%lld", a.M, a.N, ks);
    }
    double fill = 0.0;  // LDS-DMA bytes of the launch: every tile streams (BM + BN) 128-byte lines per 32-k chunk of every term
    {
        long long nch = 0;
        for (int i = 0; i < a.nterms; ++i) nch += (a.term[i].K + 31) >> 5;
        fill = (double)grid.x * grid.y * (double)nch * (G::BM + G::BN) * 128.0;
    }
    ProfScope ps(full, flops, a.M, s, fill);
    if (a.accumulate && a.nterms == 1) {  // weight gradient: few output tiles, long contraction -> slices of the contraction over gridDim.z, ~512 workgroups
        GemmArgs b = a;
        const int total = (a.term[0].K + 31) >> 5, tiles = (int)(grid.x * grid.y);
        static const int wg_target = tunable("DW_WORKGROUPS", 256), min_chunks = tunable("DW_MIN_CHUNKS", 16);
        int splits = std::max(1, std::min(wg_target / std::max(tiles, 1), total / std::max(min_chunks, 1)));
        b.ksplit_chunks = (total + splits - 1) / splits;
        grid.z = (unsigned)((total + b.ksplit_chunks - 1) / b.ksplit_chunks);
        hipLaunchKernelGGL(k, grid, dim3(G::THREADS), G::LDS_BYTES, s, b);
        return check_hip(hipGetLastError(), "pgemm launch");
    }
    static const int dbg = tunable("PGEMM_DBG", 0);
    if (dbg) {
        GemmArgs b = a;
        b.dbg_phase = dbg;
        hipLaunchKernelGGL(k, grid, dim3(G::THREADS), G::LDS_BYTES, s, b);
        return check_hip(hipGetLastError(), "pgemm launch");
    }
    hipLaunchKernelGGL(k, grid, dim3(G::THREADS), G::LDS_BYTES, s, a);
    return check_hip(hipGetLastError(), "pgemm launch");
}

// set by launch_gemm / launch_lstm_step (gemm_f32.hip) around a dispatch whose "planes" are plain fp32 rows: the exact-fp32 instantiations
thread_local bool t_exact_lines = false;

template <int WM, int WN, int TM, int TN, int NST>
static int launch_pgemm_cfg(const GemmArgs& a, hipStream_t s, double flops) {
    if (t_exact_lines) return launch_pgemm_lw<WM, WN, TM, TN, NST, 4, 2>(a, s, flops);
    if (gemm_mode() == FCL_GEMM_BF16)  // autocast: bf16-rounded operands (the hi planes alone), one MFMA per product
        return loader_waves() ? launch_pgemm_lw<WM, WN, TM, TN, NST, 2, true>(a, s, flops) : launch_pgemm_lw<WM, WN, TM, TN, NST, 0, true>(a, s, flops);
    if (loader_waves() == 4) return launch_pgemm_lw<WM, WN, TM, TN, NST, 4, false>(a, s, flops);
    return loader_waves() ? launch_pgemm_lw<WM, WN, TM, TN, NST, 2, false>(a, s, flops) : launch_pgemm_lw<WM, WN, TM, TN, NST, 0, false>(a, s, flops);
}

// --------------------------------------------------------------------------------------------------------------------------------------
// Conv1d as an LDS-tiled stencil on the same machinery.  pgemm_kernel treats the k taps as k independent K-terms: every tap re-fetches "its" A rows
// (the same rows shifted by one) and a workgroup's main loop is bound by the bytes it pulls through LDS-DMA ((BM + BN) * Cin * k * 4 at ~47 GB/s
// per CU).  Here a 32-column chunk of the input tile is loaded ONCE with a halo of 8 rows on either side (BM + 16 rows, double-buffered) and the
// k taps read it at row offsets -pad .. +pad; only the W chunk changes per step: (BM + 16 + k * BN) * 128 B per channel chunk instead of
// k * (BM + BN) * 128 B (-39 % for k = 5, 64 x 64 tiles).  Utterance edges: a row outside the segment of the OUTPUT row it contributes to is
// zeroed per lane at fragment-read time (one shared tile serves rows of two utterances when a tile straddles a boundary).
template <int WM, int WN, int TM, int TN, int NL_ = 2, int NSTW_ = 3>
struct CGeo {
    static constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN, NW = WM * WN, NL = NL_, THREADS = 64 * (NW + NL), CTHREADS = 64 * NW;
    static constexpr int HALO = NL_ == 4 ? 16 : 8, AROWS = BM + 2 * HALO, A_BYTES = AROWS * 128, W_BYTES = BN * 128, NSTW = NSTW_;  // W ring stages (2: 68 KB for the 128 x 128 tile -> two workgroups per CU)
    static constexpr int LDS_BYTES = 2 * A_BYTES + NSTW * W_BYTES;
    static constexpr int GAH = AROWS / 8 / NL, GB = BN / 8 / NL;
    static_assert((AROWS / 8) % NL == 0 && (BN / 8) % NL == 0, "tile rows must split evenly over the loader waves");
    static_assert(GAH + GB <= 60, "s_waitcnt vmcnt is a 6-bit field");
};

// HI: 0 = bf16x3 planes, 1 = the hi planes alone (autocast), 2 (round 5) = EXACT fp32 lines (FCL_PRECISION=0: a row-major fp32 activation / the
// tap-major fp32 weights ARE 128-byte lines of 32 floats, see pchunk_mma<HI = 2>): the exact mode's convolutions used to run as k-term GEMMs that
// re-fetch the input tile for every tap
template <int WM, int WN, int TM, int TN, int HI, int NL, int NSTW = 3>
__global__ __launch_bounds__(64 * (WM * WN + NL)) void pconv_kernel(const GemmArgs a) {
    using G = CGeo<WM, WN, TM, TN, NL, NSTW>;
    extern __shared__ __attribute__((aligned(1024))) u8 smem[];
    int bx, by;
    xcd_tile_p(bx, by);
    const int m0 = by * G::BM, n0 = bx * G::BN;
    if (a.m_dev && m0 >= *a.m_dev) return;  // tile beyond the device's row count (uniform per workgroup, before any barrier / LDS-DMA)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = a.conv_k, pad = (k - 1) >> 1;
    GemmTerm T0 = a.term[0];
    const long long gz = blockIdx.z;  // grouped Conv1d: group gz of gridDim.z independent problems
    T0.Ap += gz * a.g_a;
    T0.Wp += gz * a.g_w;
    const int cc = (T0.K + 31) >> 5, S = cc * k;  // channel chunks, steps (chunk-major, tap-minor)
    u8* const wring = smem + 2 * G::A_BYTES;

    if (wave >= G::NW) {  // ------------------------------------------------------------------------------------------------ loader waves
        const int lw = wave - G::NW;
        const unsigned coff = (unsigned)(((lane & 7) ^ (((lw & 1) << 2) | (lane >> 4))) * 16);
        const u8* zline = reinterpret_cast<const u8*>(g_zero_line) + coff;
        const u8* pa[G::GAH];   // chunk 0 of this lane's halo rows (zero line outside the matrix)
        unsigned ia[G::GAH];
#pragma unroll
        for (int j = 0; j < G::GAH; ++j) {
            const int row = m0 - G::HALO + (j * G::NL + lw) * 8 + (lane >> 3);
            const bool ok = row >= 0 && row < a.M;
            pa[j] = ok ? reinterpret_cast<const u8*>(T0.Ap) + (size_t)row * ((size_t)T0.lda_p * 128) + coff : zline;
            ia[j] = ok ? 128u : 0u;
        }
        const u8* pw[G::GB];    // tap 0, chunk 0 of this lane's W rows
        bool wok[G::GB];
        const size_t tap_stride = (size_t)a.N * ((size_t)T0.ldw_p * 128);  // tap-major [k * Cout, Cin] planes
#pragma unroll
        for (int j = 0; j < G::GB; ++j) {
            const int n = n0 + (j * G::NL + lw) * 8 + (lane >> 3);
            wok[j] = n < a.N;
            pw[j] = wok[j] ? reinterpret_cast<const u8*>(T0.Wp) + (size_t)n * ((size_t)T0.ldw_p * 128) + coff : zline;
        }
        int is = 0, ic = 0, ij = 0;  // next bundle to issue: step is = (chunk ic, tap ij)
        auto issue = [&]() {  // bundle(is) = W chunk of the step, plus the A halo tile of the chunk on its first tap
            if (ij == 0) {
                u8* abase = smem + (ic & 1) * G::A_BYTES + lw * 1024;
#pragma unroll
                for (int j = 0; j < G::GAH; ++j) {
                    glds16(pa[j], abase + j * G::NL * 1024);
                    pa[j] += ia[j];
                }
            }
            u8* wbase = wring + (is % G::NSTW) * G::W_BYTES + lw * 1024;
            const size_t woff = (size_t)ij * tap_stride + (size_t)ic * 128;
#pragma unroll
            for (int j = 0; j < G::GB; ++j) glds16(wok[j] ? pw[j] + woff : pw[j], wbase + j * G::NL * 1024);
            ++is;
            if (++ij == k) { ij = 0; ++ic; }
        };
        issue();
        if (G::NSTW == 2) {  // two W stages: one bundle ahead (the other workgroup on the CU covers the exposed DMA latency)
            for (int s = 0; s < S; ++s) {
                wait_vm<0>();
                asm volatile("s_barrier" ::: "memory");
                if (s + 1 < S) issue();
            }
            return;
        }
        if (S > 1) issue();
        for (int s = 0; s < S; ++s) {
            // everything but the newest bundle (step s + 1) has landed; that bundle carries an A tile when s + 1 opens a chunk
            if (s + 1 < S) {
                if ((s + 1) % k == 0) wait_vm<G::GB + G::GAH>();
                else wait_vm<G::GB>();
            } else {
                wait_vm<0>();
            }
            asm volatile("s_barrier" ::: "memory");
            if (s + 2 < S) issue();
        }
        return;
    }
    // -------------------------------------------------------------------------------------------------------------------- compute waves
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kq = lane >> 4, sw = r16 >> 1;
    const int ar16 = rho16(r16);  // the A-tile row this lane's MFMA row holds (see rho16: conflict-free fragment reads at every tap offset)
    int lo_off[TM], hi_off[TM];  // segment of this lane's output rows, relative to the row: tap shift sh contributes iff lo_off <= sh < hi_off
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + (wm * TM + tm) * 16 + ar16;
        lo_off[tm] = hi_off[tm] = 0;
        if (m < a.M) {
            lo_off[tm] = a.seg_lo[m] - m;
            hi_off[tm] = a.seg_hi[m] - m;
        }
    }
    const int b_hi = (wn * TN * 16 + r16) * 128 + ((kq ^ sw) << 4), b_lo = (wn * TN * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int c = 0, j = 0;
    const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < S; ++s) {
        asm volatile("s_barrier" ::: "memory");
        const int sh = j - pad;
        const u8* abase = smem + (c & 1) * G::A_BYTES;
        const u8* wb = wring + (s % G::NSTW) * G::W_BYTES;
        if constexpr (HI == 2) {
            f32x4 a0[TM], a1[TM], b0[TN], b1[TN];
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int lr = (wm * TM + tm) * 16 + ar16 + G::HALO + sh;
                const int swz = (lr >> 1) & 7;
                const bool ok = sh >= lo_off[tm] && sh < hi_off[tm];
                a0[tm] = *reinterpret_cast<const f32x4*>(abase + lr * 128 + ((kq ^ swz) << 4));
                a1[tm] = *reinterpret_cast<const f32x4*>(abase + lr * 128 + (((4 + kq) ^ swz) << 4));
                if (!ok) {
                    a0[tm] = z4;
                    a1[tm] = z4;
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                b0[tn] = *reinterpret_cast<const f32x4*>(wb + b_hi + tn * 16 * 128);
                b1[tn] = *reinterpret_cast<const f32x4*>(wb + b_lo + tn * 16 * 128);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[tm][e], b0[tn][e], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[tm][e], b1[tn][e], acc[tm][tn], 0, 0, 0);
        } else {
        s16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int lr = (wm * TM + tm) * 16 + ar16 + G::HALO + sh;
                const int swz = (lr >> 1) & 7;
                const bool ok = sh >= lo_off[tm] && sh < hi_off[tm];
                ah[tm] = *reinterpret_cast<const s16x8*>(abase + lr * 128 + ((kq ^ swz) << 4));
                if (!HI) al[tm] = *reinterpret_cast<const s16x8*>(abase + lr * 128 + (((4 + kq) ^ swz) << 4));
                if (!ok) {
                    ah[tm] = zero8;
                    al[tm] = zero8;
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                bh[tn] = *reinterpret_cast<const s16x8*>(wb + b_hi + tn * 16 * 128);
                if (!HI) bl[tn] = *reinterpret_cast<const s16x8*>(wb + b_lo + tn * 16 * 128);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                if (!HI) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            }
        }
        if (++j == k) { j = 0; ++c; }
    }
    pgemm_epilogue<WN, TM, TN, G::BM, G::BN, G::CTHREADS, G::LDS_BYTES, true>(a, acc, smem, m0, n0, gz * a.g_bias, gz * a.g_y, gz * a.g_yp);
}

template <int WM, int WN, int TM, int TN, int NL, int NSTW = 3>
static int launch_pconv_nl(const GemmArgs& a, hipStream_t s, double flops) {
    using G = CGeo<WM, WN, TM, TN, NL, NSTW>;
    const bool hi = gemm_mode() == FCL_GEMM_BF16 && !t_exact_lines;
    const void* fn = t_exact_lines ? reinterpret_cast<const void*>(pconv_kernel<WM, WN, TM, TN, 2, NL, NSTW>)
                     : hi          ? reinterpret_cast<const void*>(pconv_kernel<WM, WN, TM, TN, 1, NL, NSTW>)
                                   : reinterpret_cast<const void*>(pconv_kernel<WM, WN, TM, TN, 0, NL, NSTW>);
    const int rc = ensure_dyn_lds(fn, G::LDS_BYTES);
    if (rc) return rc;
    const int ncols = a.Yp ? max(a.N, a.ldyp * 32) : a.N;
    const int groups = (a.g_a || a.g_w || a.g_y || a.g_yp) ? max(1, a.nblk) : 1;  // (nblk carries the group count of a grouped Conv1d)
    dim3 grid((ncols + G::BN - 1) / G::BN, (a.M + G::BM - 1) / G::BM, groups);
    char full[48];
    snprintf(full, sizeof(full), "pconv_kernel<%d,%d,%d,%d,%d>%s%s", WM, WN, TM, TN, NL, NSTW == 2 ? "/2st" : "", t_exact_lines ? "/f32" : hi ? "/bf16" : "");
    // LDS-DMA bytes: per 32-channel chunk of the input a tile streams its row window once (BM + halo rows) and one BN-row weight slab per tap
    const double fill = (double)grid.x * grid.y * grid.z * (double)((a.term[0].K + 31) >> 5) * ((double)G::AROWS + (double)a.conv_k * G::BN) * 128.0;
    ProfScope ps(full, flops, a.M, s, fill);
    static const int dbg = tunable("PGEMM_DBG", 0);
    GemmArgs b = a;
    b.dbg_phase = dbg;
    if (t_exact_lines) hipLaunchKernelGGL((pconv_kernel<WM, WN, TM, TN, 2, NL, NSTW>), grid, dim3(G::THREADS), G::LDS_BYTES, s, b);
    else if (hi) hipLaunchKernelGGL((pconv_kernel<WM, WN, TM, TN, 1, NL, NSTW>), grid, dim3(G::THREADS), G::LDS_BYTES, s, b);
    else hipLaunchKernelGGL((pconv_kernel<WM, WN, TM, TN, 0, NL, NSTW>), grid, dim3(G::THREADS), G::LDS_BYTES, s, b);
    return check_hip(hipGetLastError(), "pconv launch");
}

template <int WM, int WN, int TM, int TN>
static int launch_pconv_cfg(const GemmArgs& a, hipStream_t s, double flops) {
    static const int nl = tunable("PCONV_LOADERS", 2);
    // 128 x 128 tiles (the postnet): two W stages = 68 KB, 10 waves, 96 VGPRs -> two workgroups per CU.  Measured r3: the fresh feed LOSES 1 %
    // with it (39.2 vs 39.6 M frames/s over three runs each: one bundle ahead exposes the W latency of a 5-tap step) -> off
    static const int two = tunable("PCONV_2STAGE", 0);
    if (two && WM * TM == 8 && WN * TN == 8 && nl < 4) return launch_pconv_nl<WM, WN, TM, TN, 2, 2>(a, s, flops);
    return nl >= 4 ? launch_pconv_nl<WM, WN, TM, TN, 4>(a, s, flops) : launch_pconv_nl<WM, WN, TM, TN, 2>(a, s, flops);
}

int launch_gemm_planes(const GemmArgs& a, hipStream_t s) {
    double ksum = 0;
    for (int i = 0; i < a.nterms; ++i) ksum += a.term[i].K;
    const double flops = 2.0 * a.M * (double)a.N * ksum;
    static const int force = tunable("PGEMM_CFG", 0);
    const long long t64x128 = (long long)((a.M + 63) / 64) * ((a.N + 127) / 128);
    const long long t128x128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    // measured on MI355X (tools/probe/planes_gemm_probe): 8-wave 128 x 128 tiles where they still give >= ~150 workgroups, 64 x 128 with three
    // stages (two workgroups per CU) down to ~250, 64 x 64 below that (the encoder-side GEMMs: M = 3 200, N = 256-384)
    static const int pconv = tunable("PCONV", 1);
    // Conv1d: the stencil kernel (shared A halo tile).  In isolation it is within +-10 % of the K-term form (3 200 x 256 x 5 x 256: 20.7 vs 18.8 us;
    // 25 026 x 128 x 5 x 128: 24.7 vs 25.6); its smaller LDS footprint (44 vs 64 KB, 86 vs 96 KB) and 10 instead of 12 waves are what raise the
    // pass rate with several passes in flight (+1.7 ... +4 %, batch 64 +3.6 %).  At Cin >= 512 (FCL-taco2-T: 33.8 vs 28.0, 124 vs 115 us) the
    // single-stream training step loses 2 % with it, so those keep the K-term form (PCONV=2 forces the stencil everywhere).
    // exact-fp32 lines (round 5, pconv_kernel<HI = 2>): built, parity-green under FCL_PRECISION=0 and SLOWER -- 22.6 vs 23.2 M frames/s on the
    // four-stream line: that mode's loops are bound by their 32-cycle fp32 MFMAs, not by the input-tile re-fetch the stencil saves -> opt-in
    static const int pconv_exact = tunable("PCONV_EXACT", 0);
    if (pconv && (!t_exact_lines || pconv_exact) && a.conv_k >= 3 && a.conv_k <= 17 && !a.accumulate && (a.term[0].K <= 384 || pconv >= 2)) {
        // (round 6: 150 -> 60 -- the 128-row stencil wherever it has 60 tiles: +0.4 ... +1.0 % on the four-pass line in three same-box scans, everything else flat:
        // profiles/r6_tile_sweep.log, r6_tunable_scan_synth_pconv.log)
        static const int cbig_min = tunable("PCONV_BIG_MIN", 60);
        if (force == 1 || (force == 0 && t128x128 >= cbig_min && a.N >= 128)) return launch_pconv_cfg<4, 2, 2, 4>(a, s, flops);
        if (force == 2 || (force == 0 && t64x128 >= 250 && a.N >= 96)) return launch_pconv_cfg<2, 2, 2, 4>(a, s, flops);
        return launch_pconv_cfg<2, 2, 2, 2>(a, s, flops);
    }
    if (a.accumulate && force == 0)  // split contraction fills the device whatever the tile count: the largest tiles that fit the output
        return (a.M >= 128 && a.N >= 128) ? launch_pgemm_cfg<4, 2, 2, 4, 3>(a, s, flops) : launch_pgemm_cfg<2, 2, 2, 2, 4>(a, s, flops);
    // 256 x 128 tiles (64 x 64 per wave): 48 KB instead of 64 KB of LDS-DMA per 256 x 128 x 32 MACs -- the main loop is bound by the global -> LDS
    // fill rate, so -25 % bytes per FLOP.  Measured r3 (tools/time_pgemm.py): +6 ... +12 % on frame-sized GEMMs with K >= 512 (12 400 x 4 096 x 512:
    // 203 -> 226 TFLOP/s fp32-equivalent), -10 ... -50 % at K <= 256 (the epilogue of a 256-row tile is not overlapped by anything at one workgroup
    // per CU), and no change of the KD / teacher update (12.58 / 12.93 vs 12.60 / 12.95 ms: few of their GEMMs qualify) -> off unless asked for
    // After the epilogue pass (plain element loop) the 256-row tile no longer pays for its epilogue: 24 300 x 1 024 x 256 64 -> 58 us,
    // 12 400 x 4 096 x 512 222 -> 188, 31 000 x 512 x 512 65 -> 54.5; still slower at N = 128 and below two rounds of workgroups -> ON for N >= 256,
    // K >= 256, >= 512 tiles (KD / teacher update: 11.84 / 12.42 vs 11.88 / 12.40 ms)
    static const int big_min = tunable("PGEMM_BIG_MIN_WG", 1 << 30);  // (off again: the two-stage 128 x 128 configuration below beats it on every shape)
    const long long t256x128 = (long long)((a.M + 255) / 256) * ((a.N + 127) / 128);
    if (force == 3 || (force == 0 && t256x128 >= big_min && a.N >= 256 && ksum >= 256)) return launch_pgemm_cfg<4, 2, 4, 4, 3>(a, s, flops);
    if (force == 6) return launch_pgemm_cfg<4, 2, 2, 4, 4>(a, s, flops);  // 128 x 128 tiles, FOUR ring stages (three chunks = 96 KB in flight per CU)
    // 128 x 128 tiles with TWO ring stages: 64 KB of LDS and 80 VGPRs = two workgroups per CU, so one's epilogue (staging + stores, 2 - 3 us that
    // nothing overlapped at one workgroup per CU) runs beside the other's main loop.  Only where the launch has more than one round of workgroups
    // (a single round runs one per CU whatever it could share): 24 300 x 1 024 x 256 63 -> 52 us, 12 400 x 4 096 x 512 199 -> 152 (343 TFLOP/s
    // fp32-equivalent), 31 000 x 512 x 512 64 -> 49, 2 480 x 4 096 x 1 024 72 -> 66; N = 128 is 13 % slower with it (tools/time_pgemm.py)
    static const int two_stage_min = tunable("PGEMM_2STAGE_MIN_WG", 300);
    if (force == 7 || (force == 0 && t128x128 >= two_stage_min && a.N >= 256)) return launch_pgemm_cfg<4, 2, 2, 4, 2>(a, s, flops);
    static const int gbig_min = tunable("PGEMM_BIG_MIN", 150);
    if (force == 1 || (force == 0 && t128x128 >= gbig_min && a.N >= 128)) return launch_pgemm_cfg<4, 2, 2, 4, 3>(a, s, flops);
    if (force == 2 || (force == 0 && t64x128 >= 250 && a.N >= 96)) return launch_pgemm_cfg<2, 2, 2, 4, 3>(a, s, flops);
    return launch_pgemm_cfg<2, 2, 2, 2, 4>(a, s, flops);
}

template <int WM, int WN, int TM, int NST, int LW, int HI>
static int launch_plstm_lw(const LstmStepArgs& a, hipStream_t s, double flops) {
    using G = PGeo<WM, WN, TM, 4, NST, LW>;
    const bool plain = !a.zone_keep_h && !a.row_len && !a.save_gates && !a.out2;  // (MODE >= 0 compiles these options out of the cell code)
    const int mode = (plain && a.G && a.rank1_w && !a.bias) ? 0 : (plain && a.bias && !a.G && !a.rank1_w) ? 1 : -1;
    dim3 grid((a.U + 16 * WN - 1) / (16 * WN), (a.M + G::BM - 1) / G::BM);
    char full[64];
    snprintf(full, sizeof(full), "plstm_kernel<%d,%d,%d,%d,%d,%d>%s", WM, WN, TM, NST, mode, LW, HI == 2 ? "/f32" : HI ? "/bf16" : "");
    const void* fn = mode == 0 ? reinterpret_cast<const void*>(plstm_kernel<WM, WN, TM, NST, 0, LW, HI>)
                   : mode == 1 ? reinterpret_cast<const void*>(plstm_kernel<WM, WN, TM, NST, 1, LW, HI>)
                               : reinterpret_cast<const void*>(plstm_kernel<WM, WN, TM, NST, -1, LW, HI>);
    const int rc = ensure_dyn_lds(fn, G::LDS_BYTES);
    if (rc) return rc;
    double fill = 0.0;
    {
        long long nch = 0;
        for (int i = 0; i < a.nterms; ++i) nch += (a.term[i].K + 31) >> 5;
        fill = (double)grid.x * grid.y * (double)nch * (G::BM + G::BN) * 128.0;
    }
    ProfScope ps(full, flops, a.M, s, fill);
    if (mode == 0) hipLaunchKernelGGL((plstm_kernel<WM, WN, TM, NST, 0, LW, HI>), grid, dim3(G::THREADS), G::LDS_BYTES, s, a);
    else if (mode == 1) hipLaunchKernelGGL((plstm_kernel<WM, WN, TM, NST, 1, LW, HI>), grid, dim3(G::THREADS), G::LDS_BYTES, s, a);
    else hipLaunchKernelGGL((plstm_kernel<WM, WN, TM, NST, -1, LW, HI>), grid, dim3(G::THREADS), G::LDS_BYTES, s, a);
    return check_hip(hipGetLastError(), "plstm launch");
}

template <int WM, int WN, int TM, int NST>
static int launch_plstm_cfg(const LstmStepArgs& a, hipStream_t s, double flops) {
    if (t_exact_lines) return launch_plstm_lw<WM, WN, TM, NST, 4, 2>(a, s, flops);
    if (gemm_mode() == FCL_GEMM_BF16)
        return loader_waves() ? launch_plstm_lw<WM, WN, TM, NST, 2, true>(a, s, flops) : launch_plstm_lw<WM, WN, TM, NST, 0, true>(a, s, flops);
    if (loader_waves() == 4) return launch_plstm_lw<WM, WN, TM, NST, 4, false>(a, s, flops);
    return loader_waves() ? launch_plstm_lw<WM, WN, TM, NST, 2, false>(a, s, flops) : launch_plstm_lw<WM, WN, TM, NST, 0, false>(a, s, flops);
}

int launch_lstm_planes(const LstmStepArgs& a, hipStream_t s) {
    FCL_REQUIRE(!a.h_out_p || ((a.U & 31) == 0 && a.ld_hp * 32 >= a.U && (reinterpret_cast<uintptr_t>(a.h_out_p) & 127u) == 0), FCL_ERR_SHAPE,
                "lstm_step: h_out_p needs U %% 32 == 0, ld_hp >= U / 32 and a 128-byte aligned buffer");
    double ksum = 0;
    for (int i = 0; i < a.nterms; ++i) ksum += a.term[i].K;
    const double flops = 2.0 * a.M * 4.0 * a.U * ksum;
    static const int force = tunable("PLSTM_CFG", 0);
    const long long t128 = (long long)((a.M + 127) / 128) * ((a.U + 31) / 32);
    const long long t64 = (long long)((a.M + 63) / 64) * ((a.U + 31) / 32);
    const long long t96 = (long long)((a.M + 95) / 96) * ((a.U + 31) / 32);
    static const int use96 = tunable("PLSTM_TILE96", 0);
    // 96-row tiles where they still fit one wave of workgroups and 128-row tiles would leave CUs idle: a workgroup's main loop takes
    // (BM + BN) * K * 4 bytes / ~47 GB/s, so 96 + 128 instead of 128 + 128 rows is -12.5 % per workgroup
    if ((force == 5 || (force == 0 && use96 && t128 >= 150 && t128 < 230 && t96 <= 256)) && loader_waves() > 0 && gemm_mode() != FCL_GEMM_BF16)
        return loader_waves() == 4 ? launch_plstm_lw<3, 2, 2, 3, 4, false>(a, s, flops) : launch_plstm_lw<3, 2, 2, 3, 2, false>(a, s, flops);  // (6 compute waves: loader-specialised only)
    // thresholds as tunables (r3, 4 passes in flight, B = 32: 64-row tiles everywhere -- two workgroups per CU, also from different streams -- are +1.5 ...
    // +4.5 % on the replayed pass and within noise on the fresh feed, but -22 % on FCL-taco2-T synthesis and +6 % on the KD update: the 128-row tiles
    // stay where they are; PLSTM_BIG_MIN=300 PLSTM_MID_MIN=80 is the S-only variant)
    if (force == 6) return launch_plstm_cfg<4, 2, 2, 4>(a, s, flops);  // 128-row tiles, four ring stages
    // 128-row tiles, TWO ring stages = two workgroups per CU (see pgemm): where the launch has more than one round of workgroups (FCL-taco2-T,
    // batch >= 64 at FCL-taco2-S): the step kernel itself +10 % (FCL-taco2-T: frac 0.33 -> 0.36), batch 64 +2.7 %, T synthesis +1 %.  Synthesis
    // steps only: in the KD update the frozen teacher's forward runs BESIDE the student's critical path, and a teacher step that holds two
    // workgroups' worth of every CU slows that path more than it gains (12.20 vs 11.98 ms)
    // Round 6 re-scan on the final build (tools/tunable_scan_synth*.sh, profiles/r6_tunable_scan_synth*.log): 300 -> 125, i.e. the first decoder steps of an FCL-taco2-S batch
    // (>= 1 921 live rows) take the two-stage 128-row tile instead of 64-row tiles: four passes in flight 43.7 -> 45.2 M frames/s (+3.4 %), calibrated capacities +5 %,
    // replayed pass +3 %, batch 64 +1 %; ONE pass alone 21.26 -> 20.91 M (-1.6 %: the step alone is slower without its pre-loop operand requests).  100: the same at four
    // passes, -2.7 % alone.
    static const int two_stage_min = tunable("PLSTM_2STAGE_MIN_WG", 125);
    static const int two_stage_min_exact = tunable("PLSTM_2STAGE_MIN_WG_EXACT", 300);  // (exact-fp32 lines are MFMA-bound: 125 costs that mode 5 %, 23.2 -> 22.0 M frames/s)
    if (force == 7 || (force == 0 && t128 >= (t_exact_lines ? two_stage_min_exact : two_stage_min) && !a.zone_keep_h && !a.save_gates)) return launch_plstm_cfg<4, 2, 2, 2>(a, s, flops);
    // narrow synthesis steps (U <= 256, no training-side outputs: FCL-taco2-S inference) have their own pair of thresholds: 64-row tiles (two
    // workgroups per CU, also of different passes) up to 300 128-row tiles -- same-box A/B, three rounds, (150, 200) vs (300, 80): replayed pass
    // 47.5 -> 48.6 M frames/s, fresh feed 39.4 -> 39.8 M; the wide / training steps keep the 128-row tiles (-22 % on FCL-taco2-T synthesis otherwise)
    static const int big_min_t = tunable("PLSTM_BIG_MIN", 150), big_min_s = tunable("PLSTM_BIG_MIN_S", 300);
    static const int mid_min_t = tunable("PLSTM_MID_MIN", 200), mid_min_s = tunable("PLSTM_MID_MIN_S", 80);
    const bool narrow = a.U <= 256 && !a.zone_keep_h && !a.save_gates;
    const int big_min = narrow ? big_min_s : big_min_t, mid_min = narrow ? mid_min_s : mid_min_t;
    if (force == 1 || (force == 0 && t128 >= big_min)) return launch_plstm_cfg<4, 2, 2, 3>(a, s, flops);
    if (force == 2 || (force == 0 && t64 >= mid_min)) return launch_plstm_cfg<2, 2, 2, 3>(a, s, flops);
    static const int row32_m = tunable("PLSTM_ROW32_M", 1100);
    if (force == 4 || (force == 0 && a.M <= row32_m)) return launch_plstm_cfg<2, 2, 1, 4>(a, s, flops);  // 32 x 128 tiles: M = 500 .. 1100 (measured)
    return launch_plstm_cfg<2, 1, 2, 4>(a, s, flops);  // 64 x 64 gate-column tiles (16 units): M <~ 1600 at U = 256
}

template <int WM, int WN, int TM, int NST, int HI>
static int launch_plstm_pair_cfg(const LstmStepArgs& a0, const LstmStepArgs& a1, hipStream_t s, double flops) {
    constexpr int LW = 4;
    using G = PGeo<WM, WN, TM, 4, NST, LW>;
    auto k = plstm_pair_kernel<WM, WN, TM, NST, LW, HI>;
    const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(k), G::LDS_BYTES);
    if (rc) return rc;
    const int mmax = a0.M > a1.M ? a0.M : a1.M;
    dim3 grid((a0.U + 16 * WN - 1) / (16 * WN), (mmax + G::BM - 1) / G::BM, 2);
    char full[64];
    snprintf(full, sizeof(full), "plstm_pair_kernel<%d,%d,%d,%d,%d>%s", WM, WN, TM, NST, LW, HI ? "/bf16" : "");
    double fill = 0.0;
    for (const LstmStepArgs* a : {&a0, &a1}) {
        long long nch = 0;
        for (int i = 0; i < a->nterms; ++i) nch += (a->term[i].K + 31) >> 5;
        fill += (double)grid.x * (double)((a->M + G::BM - 1) / G::BM) * (double)nch * (G::BM + G::BN) * 128.0;
    }
    ProfScope ps(full, flops, a0.M + a1.M, s, fill);
    hipLaunchKernelGGL(k, grid, dim3(G::THREADS), G::LDS_BYTES, s, a0, a1);
    return check_hip(hipGetLastError(), "plstm pair launch");
}

// two independent pre-split LSTM steps of the same width in one launch; *handled = false when the pair form does not cover the configuration
// (the caller then launches the two steps one after the other)
int launch_lstm_planes_pair(const LstmStepArgs& a0, const LstmStepArgs& a1, hipStream_t s, bool* handled) {
    *handled = false;
    static const int on = tunable("PLSTM_PAIR", 1);
    if (!on || a0.U != a1.U || t_exact_lines || loader_waves() != 4 || a0.m_dev || a1.m_dev) return 0;
    for (const LstmStepArgs* a : {&a0, &a1})
        if (a->h_out_p && !((a->U & 31) == 0 && a->ld_hp * 32 >= a->U && (reinterpret_cast<uintptr_t>(a->h_out_p) & 127u) == 0)) return 0;
    *handled = true;
    double flops = 0;
    for (const LstmStepArgs* a : {&a0, &a1}) {
        double ksum = 0;
        for (int i = 0; i < a->nterms; ++i) ksum += a->term[i].K;
        flops += 2.0 * a->M * 4.0 * a->U * ksum;
    }
    const int M = a0.M > a1.M ? a0.M : a1.M, U = a0.U;
    const bool hi = gemm_mode() == FCL_GEMM_BF16;
    // the training-side thresholds of launch_lstm_planes on the larger of the two row counts (both problems share one tile shape)
    static const int big_min = tunable("PLSTM_PAIR_BIG_MIN", tunable("PLSTM_BIG_MIN", 150)), mid_min = tunable("PLSTM_PAIR_MID_MIN", tunable("PLSTM_MID_MIN", 200)),
                     row32_m = tunable("PLSTM_PAIR_ROW32_M", tunable("PLSTM_ROW32_M", 1100));
    const long long t128 = (long long)((M + 127) / 128) * ((U + 31) / 32), t64 = (long long)((M + 63) / 64) * ((U + 31) / 32);
#define FCL_PAIR(WM_, WN_, TM_, NST_) (hi ? launch_plstm_pair_cfg<WM_, WN_, TM_, NST_, 1>(a0, a1, s, flops) : launch_plstm_pair_cfg<WM_, WN_, TM_, NST_, 0>(a0, a1, s, flops))
    // two ring stages = 64 KB of LDS = TWO workgroups per CU for the pair's 128-row tiles, from ..MIN_WG to below ..MAX_WG tiles per problem.  Round 6 (tools/tunable_scan.sh,
    // profiles/r6_tunable_scan.log): on for 256 <= tiles < 512 -- the teacher's own update (16 utterances: 10 x 32 tiles per problem, 640 per pair on 256 CUs = 2.5 rounds
    // of one-per-CU workgroups): 9.42 -> 9.36 ms; everywhere (FCL_PLSTM_PAIR_NST2=1) the KD update, whose pairs have 160 and 640 tiles, loses 1 %
    static const int two_stage = tunable("PLSTM_PAIR_2STAGE_MIN_WG", 256), two_stage_max = tunable("PLSTM_PAIR_2STAGE_MAX_WG", 512);
    static const int all2 = tunable("PLSTM_PAIR_NST2", 0);  // (r5 A/B: two ring stages everywhere = <= 64 KB of LDS and <= 80 VGPRs per workgroup, so that
                                                           // the frozen teacher's steps and the student's can share a CU in the KD update)
    if (t128 >= two_stage && t128 < two_stage_max) return FCL_PAIR(4, 2, 2, 2);
    if (t128 >= big_min) return all2 ? FCL_PAIR(4, 2, 2, 2) : FCL_PAIR(4, 2, 2, 3);
    if (t64 >= mid_min) return all2 ? FCL_PAIR(2, 2, 2, 2) : FCL_PAIR(2, 2, 2, 3);
    if (M <= row32_m) return FCL_PAIR(2, 2, 1, 4);
    return FCL_PAIR(2, 1, 2, 4);
#undef FCL_PAIR
}

// --------------------------------------------------------------------------------------------------------------------------------------
// One Parallel WaveGAN residual block in ONE launch (pwg.hip holds the unfused form and the algebra): 128 samples x 128 gate columns per workgroup.
//   main loop  z = sum of ksize dilated taps of x + the auxiliary term (the shared LDS-DMA ring, 3 x 2 + 3 chunks for the v1 generator)
//   epilogue 1 z + b -> LDS (fp32); g = tanh(z[:, :64]) * sigmoid(z[:, 64:]) -> LDS as pre-split A planes in the ring's swizzled chunk layout
//   phase 2    o = g [W_out ; W_skip]^T  (2 chunks; W_os was staged in LDS by the compute waves while the loaders filled the ring)
//   epilogue 2 x_out = (o[:, :64] + b_out + x) * sqrt(0.5) as planes (a SECOND buffer: neighbouring workgroups still read x for their taps),
//              skips (+)= o[:, 64:] + b_skip.  HBM traffic per sample and layer: x 256 B in (+ halo) + 256 B out, aux 384 B, skips 512 B.
// tanh(a) * sigmoid(b) = (1 - e^-2a) / ((1 + e^-2a) (1 + e^-b)): two exponentials and one reciprocal (v_rcp_f32, 1 ulp: `/` and __fdividef both
// expand to the 12-instruction IEEE division sequence; the clamp keeps e^-2a finite; tanh is +-1 to fp32 precision beyond |a| = 10)
__device__ __forceinline__ float pwg_gate(float a, float b) {
    const float ea = __expf(-2.0f * fminf(fmaxf(a, -10.f), 10.f)), eb = __expf(-fmaxf(b, -80.f));
    return (1.0f - ea) * __builtin_amdgcn_rcpf((1.0f + ea) * (1.0f + eb));
}

void* g_debug_ptr = nullptr;

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the epilogue operands prefetched
// from global memory right after the main loop (their latency is meant to hide behind the gate and phase 2)
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct PwgFusedArgs {
    GemmTerm term[4];
    int nterms, M;
    const int *seg_lo, *seg_hi;
    const float *b_conv, *b_os;
    const u16 *w_os_p, *xp_in;
    u16* xp_out;
    float* skips;
    long long xcs;  // chunk stride of the x planes in uint16 elements (= M * 64): x is stored chunk-major
    long long* ts;  // developer aid (FCL_PWG_TS): 8 wall-clock stamps (10 ns units) per workgroup
    int first;
    const u16 *pt_a, *pt_b;  // frame-rate auxiliary term (fcl_pwg_layer_t.kp): the last term's W tile is a frame window of pt_a / pt_b, chosen per tile
    int hop;
    int dbg;  // developer timing aid (FCL_PWG_DBG): 1 / 2 / 3 = return after the main loop / the gate / phase 2 (results are then garbage)
};

// WM = 4: 128-row tiles, W_os resident in LDS (160 KB, one workgroup per CU); WM = 2: 64-row tiles, W_os fragments straight from L2 into
// registers, everything else inside the 72 KB ring (two workgroups per CU: one's epilogue overlaps the other's main loop).
template <int WM, int NST, bool HI>
__global__ __launch_bounds__(64 * (WM * 2 + 2)) void pwg_layer_kernel(const PwgFusedArgs a) {
    using G = PGeo<WM, 2, 2, 4, NST, 2>;
    constexpr int TM = 2, TN = 4, WN = 2, BM = G::BM;
    constexpr bool WLDS = WM == 4 && NST == 3;
    constexpr int LDT = 132, ZT_BYTES = BM * LDT * 4;
    constexpr int WOS = G::LDS_BYTES;                                          // WLDS only: 32 KB after the ring
    constexpr int GA = WLDS ? WOS + 32768 : (ZT_BYTES + 1023) / 1024 * 1024;    // gate planes: after W_os, or inside the ring behind the staging tile
    static_assert(G::BN == 128 && ZT_BYTES <= G::LDS_BYTES && (WLDS || GA + BM * 256 <= G::LDS_BYTES), "tile geometry");
    extern __shared__ __attribute__((aligned(1024))) u8 smem[];
    int bx, by;
    xcd_tile_p(bx, by);
    const int m0 = by * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define PWG_STAMP(k) do { if (a.ts && tid == 0) a.ts[(size_t)by * 8 + (k)] = (long long)wall_clock64(); } while (0)
    PWG_STAMP(0);
    if (WLDS && wave < G::NW) {  // W_os [128, 64] planes -> LDS, two 16 KB chunks in the ring's layout (piece p of row n at slot p ^ ((n >> 1) & 7))
        for (int i = tid; i < 2048; i += G::CTHREADS) {
            const int n = i >> 4, c = (i >> 3) & 1, p = i & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(a.w_os_p + ((size_t)(n * 2 + c) * 64 + p * 8));
            *reinterpret_cast<uint4*>(smem + WOS + c * 16384 + n * 128 + ((p ^ ((n >> 1) & 7)) << 4)) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    PWG_STAMP(1);
    float bz[TN], bo[TN];  // this lane's bias columns of both GEMMs, requested before the main loop
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int cn = (((tid >> 6) % WN) * TN + tn) * 16 + (tid & 15);
        bz[tn] = a.b_conv[cn];
        bo[tn] = a.b_os[cn];
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!pmainloop<WM, 2, TM, TN, NST, false, 2, HI>(a.term, a.nterms, a.M, m0, 0, 128, a.seg_lo, a.seg_hi, smem, acc)) return;  // loader wave

    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, rq = lane >> 4;
    const int r16 = lane & 15, kq = lane >> 4, sw = r16 >> 1;
    s16x8 wh[2][TN], wl[2][TN];
    if (!WLDS) {  // phase-2 B fragments (W_os rows wn*64 + tn*16 + r16, chunk c, pieces kq / 4 + kq): requested now, consumed after the gate
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const u16* w = a.w_os_p + ((size_t)((wn * TN + tn) * 16 + r16) * 2 + c) * 64 + kq * 8;
                wh[c][tn] = *reinterpret_cast<const s16x8*>(w);
                if (!HI) wl[c][tn] = *reinterpret_cast<const s16x8*>(w + 32);
            }
    }
    // final-epilogue operands (the old x planes and the skip accumulator of this thread's items) are requested NOW: their latency hides behind the
    // gate and phase 2
    constexpr int ITEMS = BM * 8 / G::CTHREADS;
    const int rows = min(BM, a.M - m0);
    uint4 pxh[ITEMS], pxl[ITEMS];
    f32x4 ps0[ITEMS], ps1[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int i = tid + it * G::CTHREADS, r = i >> 3, c0 = (i & 7) * 8;
        pxh[it] = pxl[it] = make_uint4(0, 0, 0, 0);
        ps0[it] = ps1[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            const size_t m = (size_t)(m0 + r);
            const size_t lo_off = (size_t)(c0 >> 5) * a.xcs + m * 64 + (c0 & 31);
            pxh[it] = *reinterpret_cast<const uint4*>(a.xp_in + lo_off);
            pxl[it] = *reinterpret_cast<const uint4*>(a.xp_in + lo_off + 32);
            if (!a.first) {
                ps0[it] = *reinterpret_cast<const f32x4*>(a.skips + m * 64 + c0);
                ps1[it] = *reinterpret_cast<const f32x4*>(a.skips + m * 64 + c0 + 4);
            }
        }
    }
    PWG_STAMP(2);
    float* zt = reinterpret_cast<float*>(smem);
    if (a.dbg == 1) {
        if (acc[0][0][0] == 12345.f) a.skips[0] = 1.f;
        return;
    }
    lds_sync();  // every compute wave is done with the ring
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int cn = (wn * TN + tn) * 16 + col;
            const float b = bz[tn];
#pragma unroll
            for (int r = 0; r < 4; ++r) zt[((wm * TM + tm) * 16 + rq * 4 + r) * LDT + cn] = acc[tm][tn][r] + b;
        }
    lds_sync();
    PWG_STAMP(6);
    for (int i = tid; i < BM * 8; i += G::CTHREADS) {  // gate: 8 columns per item -> one hi and one lo piece of the A planes
        const int r = i >> 3, p = i & 7, c0 = p * 8;
        f32x4 v0, v1;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(zt + r * LDT + c0), a1 = *reinterpret_cast<const f32x4*>(zt + r * LDT + c0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(zt + r * LDT + 64 + c0), b1 = *reinterpret_cast<const f32x4*>(zt + r * LDT + 64 + c0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = pwg_gate(a0[e], b0[e]);
            v1[e] = pwg_gate(a1[e], b1[e]);
        }
        uint2 h0, l0, h1, l1;
        split4(v0, h0, l0);
        split4(v1, h1, l1);
        u8* base = smem + GA + (p >> 2) * (BM * 128) + r * 128;
        const int sws = (r >> 1) & 7, piece = p & 3;
        *reinterpret_cast<uint4*>(base + ((piece ^ sws) << 4)) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(base + (((4 + piece) ^ sws) << 4)) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    lds_sync();
    PWG_STAMP(3);
    if (a.dbg == 2) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const int ar = (wm * TM * 16 + r16) * 128, br = (wn * TN * 16 + r16) * 128;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if constexpr (WLDS) {
                pchunk_mma<TM, TN, HI>(smem, GA + c * (BM * 128) + ar + ((kq ^ sw) << 4), GA + c * (BM * 128) + ar + (((4 + kq) ^ sw) << 4),
                                       WOS + c * 16384 + br + ((kq ^ sw) << 4), WOS + c * 16384 + br + (((4 + kq) ^ sw) << 4), acc);
            } else {
                s16x8 ah[TM], al[TM];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    ah[tm] = *reinterpret_cast<const s16x8*>(smem + GA + c * (BM * 128) + ar + tm * 2048 + ((kq ^ sw) << 4));
                    if (!HI) al[tm] = *reinterpret_cast<const s16x8*>(smem + GA + c * (BM * 128) + ar + tm * 2048 + (((4 + kq) ^ sw) << 4));
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        if (!HI) {
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tm], wh[c][tn], acc[tm][tn], 0, 0, 0);
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], wl[c][tn], acc[tm][tn], 0, 0, 0);
                        }
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], wh[c][tn], acc[tm][tn], 0, 0, 0);
                    }
            }
        }
    }
    PWG_STAMP(4);
    if (a.dbg == 3) {
        if (acc[0][0][0] == 12345.f) a.skips[0] = 1.f;
        return;
    }
    if (!WLDS) lds_sync();  // GA sits in the ring next to the staging tile's rows: everyone has read its fragments before o is staged
    // (WLDS: zt is free since the gate pass, phase 2 reads GA / WOS only)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int cn = (wn * TN + tn) * 16 + col;
            const float b = bo[tn];
#pragma unroll
            for (int r = 0; r < 4; ++r) zt[((wm * TM + tm) * 16 + rq * 4 + r) * LDT + cn] = acc[tm][tn][r] + b;
        }
    lds_sync();
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int i = tid + it * G::CTHREADS, r = i >> 3, c0 = (i & 7) * 8;
        if (r >= rows) continue;
        const size_t m = (size_t)(m0 + r);
        const size_t lo_off = (size_t)(c0 >> 5) * a.xcs + m * 64 + (c0 & 31);
        const unsigned hw[4] = {pxh[it].x, pxh[it].y, pxh[it].z, pxh[it].w}, lw_[4] = {pxl[it].x, pxl[it].y, pxl[it].z, pxl[it].w};
        f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // element 2e (low half-word) and 2e + 1 (high half-word) of the piece
            const float x0 = __builtin_bit_cast(float, hw[e] << 16) + __builtin_bit_cast(float, lw_[e] << 16);
            const float x1 = __builtin_bit_cast(float, hw[e] & 0xFFFF0000u) + __builtin_bit_cast(float, lw_[e] & 0xFFFF0000u);
            const float n0 = (zt[r * LDT + c0 + 2 * e] + x0) * 0.70710678118654752440f, n1 = (zt[r * LDT + c0 + 2 * e + 1] + x1) * 0.70710678118654752440f;
            if (e < 2) { v0[2 * e] = n0; v0[2 * e + 1] = n1; } else { v1[2 * (e - 2)] = n0; v1[2 * (e - 2) + 1] = n1; }
        }
        uint2 h0, l0, h1, l1;
        split4(v0, h0, l0);
        split4(v1, h1, l1);
        *reinterpret_cast<uint4*>(a.xp_out + lo_off) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(a.xp_out + lo_off + 32) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        float* sk = a.skips + m * 64 + c0;
        *reinterpret_cast<f32x4*>(sk) = ps0[it] + *reinterpret_cast<const f32x4*>(zt + r * LDT + 64 + c0);
        *reinterpret_cast<f32x4*>(sk + 4) = ps1[it] + *reinterpret_cast<const f32x4*>(zt + r * LDT + 64 + c0 + 4);
    }
    PWG_STAMP(5);
#undef PWG_STAMP
}

// ---- persistent form of the block: one workgroup per CU walks a contiguous run of 128-sample tiles of its XCD.  What it buys over one workgroup per
// tile: (1) the loader waves request the NEXT tile's segment bounds and first two chunks while the compute waves are still in the epilogue (the
// 3.7 us of fixed latency in front of a tile's first MFMA disappear), (2) W_os and the biases are staged once per workgroup, not once per tile,
// (3) no launch gap between tiles.  For (1) the epilogue must stay out of the ring: the gate runs in REGISTERS -- the W tile is loaded with its
// rows permuted so that a lane holds tanh column c and sigmoid column c of the same sample (tn tiles alternate tanh / sigmoid) -- its result goes
// straight into the 32 KB gate-plane buffer, and the second GEMM's result is staged in two 64-row halves through that same buffer once phase 2 has
// read it.  LDS: ring 96 KB | W_os 32 KB | gate planes / o staging 32 KB.
// AUXF: the auxiliary term at frame rate -- 3 x 2 + 1 chunks per tile instead of 3 x 2 + 3, the last one (coefficient lines x a 32-frame window of
// the projected features) with a W tile that depends on the tile's frame.
template <bool HI, int LW, bool AUXF>
__global__ __launch_bounds__(64 * (8 + LW)) void pwg_layer_pkernel(const PwgFusedArgs a, const int ntiles) {
    using G = PGeo<4, 2, 2, 4, 3, LW>;
    constexpr int TM = 2, TN = 4, WN = 2, BM = 128, NST = 3, NCH = AUXF ? 7 : 9;
    constexpr int WOS = G::LDS_BYTES, GA = WOS + 32768;
    static_assert(G::BM == BM && G::BN == 128 && G::STAGE == 32768, "tile geometry");
    extern __shared__ __attribute__((aligned(1024))) u8 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile schedule: XCD x (= workgroup id mod 8) owns a contiguous range of tiles, its workgroups stride through it -> a tile and its +-d
    // neighbours meet in one L2
    const int wg = blockIdx.x, nwg = gridDim.x, xcd = wg & 7;
    const int q = ntiles >> 3, rr = ntiles & 7;
    const int t_end = xcd * q + min(xcd, rr) + q + (xcd < rr ? 1 : 0);
    const int per_x = (nwg - xcd + 7) >> 3;
    int t = xcd * q + min(xcd, rr) + (wg >> 3);
    if (t >= t_end) return;  // every wave of the workgroup takes the same exit

    if (wave >= G::NW) {  // ------------------------------------------------------------------------------------------------ loader waves
        const int lw = wave - G::NW;
        const unsigned coff = (unsigned)(((lane & 7) ^ (((lw & 1) << 2) | (lane >> 4))) * 16);
        const u8* zline = reinterpret_cast<const u8*>(g_zero_line) + coff;
        long long brow[G::GB];
#pragma unroll
        for (int j = 0; j < G::GB; ++j) {  // tile column c = 64 wn + 16 tn + jj  <-  gate row (tn odd: sigmoid half) 32 wn + 16 (tn / 2) + jj
            const int c = (j * G::NL + lw) * 8 + (lane >> 3);
            brow[j] = ((c >> 4) & 1 ? 64 : 0) + 32 * (c >> 6) + 16 * ((c >> 5) & 1) + (c & 15);
        }
        int am[G::GA], alo[G::GA];
        unsigned alen[G::GA];
        const u8* pa[G::GA];
        const u8* pb[G::GB];
        unsigned ia[G::GA];
        int rem = 0, it = 0, slot = 0;
        const u8* wtile = nullptr;  // AUXF: this tile's frame window
        // AUXF: a tile lies inside one frame, hence inside one utterance: its bounds are two scalars.  They are fetched one tile ahead through the
        // scalar cache (lgkmcnt -- the vector-memory counter orders these loads with the LDS-DMA stream, and per-row bounds requested after the last
        // chunk held the loaders, and with them the epilogue's first barrier, for an HBM round trip: 3.2 -> 1.8 us for the gate phase)
        int blo = 0, bhi = 0;
        typedef const int __attribute__((address_space(4))) cint_k;  // constant address space: uniform loads become s_load, waits are the compiler's
        auto fetch_bounds = [&](int m0) {
            if (AUXF) {
                blo = *reinterpret_cast<cint_k*>(reinterpret_cast<uintptr_t>(a.seg_lo + m0));
                bhi = *reinterpret_cast<cint_k*>(reinterpret_cast<uintptr_t>(a.seg_hi + m0));
            }
        };
        auto setup_term = [&](int ti) {
            const GemmTerm T = a.term[ti];
            const u8* Ab = reinterpret_cast<const u8*>(T.Ap);
            const u8* Wb = AUXF && ti == a.nterms - 1 ? wtile : reinterpret_cast<const u8*>(T.Wp);
#pragma unroll
            for (int j = 0; j < G::GA; ++j) {
                const int src = am[j] + T.shift;
                const bool ok = (unsigned)(src - alo[j]) < alen[j];
                pa[j] = ok ? Ab + (size_t)src * 128 + coff : zline;  // chunk-major A planes
                ia[j] = ok ? (unsigned)T.a_chunk_stride : 0u;
            }
#pragma unroll
            for (int j = 0; j < G::GB; ++j) pb[j] = Wb + (size_t)brow[j] * ((size_t)T.ldw_p * 128) + coff;
            rem = (T.K + 31) >> 5;
        };
        auto begin_tile = [&](int m0) {
#pragma unroll
            for (int j = 0; j < G::GA; ++j) {
                const int m = m0 + (j * G::NL + lw) * 8 + (lane >> 3);
                am[j] = m;
                alo[j] = 0;
                alen[j] = 0u;
                if (AUXF) {
                    alo[j] = blo;
                    alen[j] = m < a.M ? (unsigned)(bhi - blo) : 0u;
                } else if (m < a.M) {
                    alo[j] = a.seg_lo[m];
                    alen[j] = (unsigned)(a.seg_hi[m] - alo[j]);
                }
            }
            if (AUXF) {  // window rule of fcl_pwg_aux_coeff
                const int f = m0 / a.hop, r = f & 31;
                const bool sel_a = r >= 2 && r <= 29;
                wtile = reinterpret_cast<const u8*>(sel_a ? a.pt_a : a.pt_b) + (size_t)(sel_a ? f >> 5 : (f + 16) >> 5) * 128;
            }
            it = 0;
            setup_term(0);
        };
        auto issue = [&]() {
            u8* sbase = smem + slot * G::STAGE + lw * 1024;
#pragma unroll
            for (int j = 0; j < G::GA; ++j) {
                glds16(pa[j], sbase + j * G::NL * 1024);
                pa[j] += ia[j];
            }
#pragma unroll
            for (int j = 0; j < G::GB; ++j) {
                glds16(pb[j], sbase + BM * 128 + j * G::NL * 1024);
                pb[j] += 128;
            }
            if (--rem == 0 && ++it < a.nterms) setup_term(it);
            slot = slot + 1 == NST ? 0 : slot + 1;
        };
        fetch_bounds(t * BM);
        begin_tile(t * BM);
        issue();
        issue();
        if (t + per_x < t_end) fetch_bounds((t + per_x) * BM);
        while (true) {
#pragma unroll 1
            for (int i = 0; i < NCH; ++i) {
                const int left = NCH - 1 - i;
                if (left >= 1) wait_vm<G::GPW>();
                else wait_vm<0>();
                asm volatile("s_barrier" ::: "memory");
                if (left >= NST - 1) issue();
            }
            const int tn_ = t + per_x;
            const bool more = tn_ < t_end;
            if (more) {  // the next tile's first two chunks travel while the compute waves run this tile's epilogue
                begin_tile(tn_ * BM);
                issue();
                issue();
                // ... and the bounds of the tile after it: requested right before the loaders idle through the epilogue's barriers, so that no
                // lgkmcnt wait of the next main loop (the term descriptors are scalar loads too) ever stands behind them
                if (tn_ + per_x < t_end) fetch_bounds((tn_ + per_x) * BM);
            }
#pragma unroll
            for (int e = 0; e < 5; ++e) asm volatile("s_barrier" ::: "memory");  // the epilogue's five barriers
            if (!more) return;
            t = tn_;
        }
    }
    // -------------------------------------------------------------------------------------------------------------------- compute waves
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, rq = lane >> 4;
    const int r16 = lane & 15, kq = lane >> 4, sw = r16 >> 1;
    for (int i = tid; i < 2048; i += G::CTHREADS) {  // W_os [128, 64] planes -> LDS once per workgroup
        const int n = i >> 4, c = (i >> 3) & 1, p = i & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(a.w_os_p + ((size_t)(n * 2 + c) * 64 + p * 8));
        *reinterpret_cast<uint4*>(smem + WOS + c * 16384 + n * 128 + ((p ^ ((n >> 1) & 7)) << 4)) = v;
    }
    float bz[TN], bo[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        bz[tn] = a.b_conv[((tn & 1) ? 64 : 0) + 32 * wn + 16 * (tn >> 1) + col];  // the permuted gate row of this lane's column
        bo[tn] = a.b_os[(wn * TN + tn) * 16 + col];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int a_hi = (wm * TM * 16 + r16) * 128 + ((kq ^ sw) << 4), a_lo = (wm * TM * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
    const int b_hi = BM * 128 + (wn * TN * 16 + r16) * 128 + ((kq ^ sw) << 4), b_lo = BM * 128 + (wn * TN * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
    float* ot = reinterpret_cast<float*>(smem + GA);  // o staging: 64 rows x 128 columns, column index XOR-ed with ((row >> 2) & 3) << 4
    int cs = 0;
    while (true) {
        const int m0 = t * BM;
        // this thread's two final-epilogue items (one per 64-row half): old x planes and skip accumulator, requested now
        uint4 pxh[2], pxl[2];
        f32x4 ps0[2], ps1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + h * 64 + (tid >> 3), c0 = (tid & 7) * 8;
            pxh[h] = pxl[h] = make_uint4(0, 0, 0, 0);
            ps0[h] = ps1[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (m < a.M) {
                const size_t off = (size_t)(c0 >> 5) * a.xcs + (size_t)m * 64 + (c0 & 31);
                pxh[h] = *reinterpret_cast<const uint4*>(a.xp_in + off);
                pxl[h] = *reinterpret_cast<const uint4*>(a.xp_in + off + 32);
                if (!a.first) {
                    ps0[h] = *reinterpret_cast<const f32x4*>(a.skips + (size_t)m * 64 + c0);
                    ps1[h] = *reinterpret_cast<const f32x4*>(a.skips + (size_t)m * 64 + c0 + 4);
                }
            }
        }
#define PWG_PSTAMP(k) do { if (a.ts && tid == 0) a.ts[(size_t)t * 8 + (k)] = (long long)wall_clock64(); } while (0)
        PWG_PSTAMP(0);
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int i = 0; i < NCH; ++i) {
            asm volatile("s_barrier" ::: "memory");
            pchunk_mma<TM, TN, HI>(smem + cs * G::STAGE, a_hi, a_lo, b_hi, b_lo, acc);
            cs = cs + 1 == NST ? 0 : cs + 1;
        }
        PWG_PSTAMP(1);
        // ---- gate in registers -> pre-split A planes of g (chunk wn, columns 16 u + col)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = pwg_gate(acc[tm][2 * u][r] + bz[2 * u], acc[tm][2 * u + 1][r] + bz[2 * u + 1]);
                    const int R = (wm * TM + tm) * 16 + rq * 4 + r, k = 16 * u + col;
                    const __bf16 gh = (__bf16)g;
                    u8* line = smem + GA + wn * 16384 + R * 128 + (k & 7) * 2;
                    const int sws = (R >> 1) & 7;
                    *reinterpret_cast<u16*>(line + (((k >> 3) ^ sws) << 4)) = __builtin_bit_cast(u16, gh);
                    if (!HI) *reinterpret_cast<u16*>(line + (((4 + (k >> 3)) ^ sws) << 4)) = __builtin_bit_cast(u16, (__bf16)(g - (float)gh));
                }
        lds_sync();  // B1: g complete
        PWG_PSTAMP(2);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const int ar = (wm * TM * 16 + r16) * 128, br = (wn * TN * 16 + r16) * 128;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                pchunk_mma<TM, TN, HI>(smem, GA + c * 16384 + ar + ((kq ^ sw) << 4), GA + c * 16384 + ar + (((4 + kq) ^ sw) << 4),
                                       WOS + c * 16384 + br + ((kq ^ sw) << 4), WOS + c * 16384 + br + (((4 + kq) ^ sw) << 4), acc);
        }
        lds_sync();  // B2: every wave has read its g fragments; the buffer becomes the staging tile of o
        PWG_PSTAMP(3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((wm >> 1) == h) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            ot[((wm & 1) * 32 + tm * 16 + rq * 4 + r) * 128 + (((wn * TN + tn) * 16 + col) ^ (rq << 4))] = acc[tm][tn][r] + bo[tn];
            }
            lds_sync();  // B3 / B5
            {
                const int rl = tid >> 3, c0 = (tid & 7) * 8, m = m0 + h * 64 + rl;
                if (m < a.M) {
                    const int cx = c0 ^ (((rl >> 2) & 3) << 4);
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(ot + rl * 128 + cx), o1 = *reinterpret_cast<const f32x4*>(ot + rl * 128 + cx + 4);
                    const f32x4 k0 = *reinterpret_cast<const f32x4*>(ot + rl * 128 + 64 + cx), k1 = *reinterpret_cast<const f32x4*>(ot + rl * 128 + 64 + cx + 4);
                    const unsigned hw[4] = {pxh[h].x, pxh[h].y, pxh[h].z, pxh[h].w}, lw_[4] = {pxl[h].x, pxl[h].y, pxl[h].z, pxl[h].w};
                    const float ov[8] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
                    f32x4 v0, v1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = __builtin_bit_cast(float, hw[e] << 16) + __builtin_bit_cast(float, lw_[e] << 16);
                        const float x1 = __builtin_bit_cast(float, hw[e] & 0xFFFF0000u) + __builtin_bit_cast(float, lw_[e] & 0xFFFF0000u);
                        const float n0 = (ov[2 * e] + x0) * 0.70710678118654752440f, n1 = (ov[2 * e + 1] + x1) * 0.70710678118654752440f;
                        if (e < 2) { v0[2 * e] = n0; v0[2 * e + 1] = n1; } else { v1[2 * (e - 2)] = n0; v1[2 * (e - 2) + 1] = n1; }
                    }
                    uint2 h0, l0, h1, l1;
                    split4(v0, h0, l0);
                    split4(v1, h1, l1);
                    const size_t off = (size_t)(c0 >> 5) * a.xcs + (size_t)m * 64 + (c0 & 31);
                    *reinterpret_cast<uint4*>(a.xp_out + off) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    *reinterpret_cast<uint4*>(a.xp_out + off + 32) = make_uint4(l0.x, l0.y, l1.x, l1.y);
                    float* sk = a.skips + (size_t)m * 64 + c0;
                    *reinterpret_cast<f32x4*>(sk) = ps0[h] + k0;
                    *reinterpret_cast<f32x4*>(sk + 4) = ps1[h] + k1;
                }
            }
            if (h == 0) lds_sync();  // B4: half 0 has been read out
        }
        PWG_PSTAMP(4);
#undef PWG_PSTAMP
        t += per_x;
        if (t >= t_end) return;
    }
}

template <int WM, int NST, bool HI>
static int launch_pwg_cfg(const PwgFusedArgs& a, long long m, double flops, hipStream_t s) {
    using G = PGeo<WM, 2, 2, 4, NST, 2>;
    constexpr int LDS = G::LDS_BYTES + ((WM == 4 && NST == 3) ? 2 * 32768 : 0);
    auto k = pwg_layer_kernel<WM, NST, HI>;
    const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(k), LDS);
    if (rc) return rc;
    dim3 grid(1, (unsigned)((m + G::BM - 1) / G::BM));
    char full[48];
    snprintf(full, sizeof(full), "pwg_layer_kernel<%d,%d>%s", WM, NST, HI ? "/bf16" : "");
    ProfScope ps(full, flops, (int)m, s);
    hipLaunchKernelGGL(k, grid, dim3(G::THREADS), LDS, s, a);
    return check_hip(hipGetLastError(), "pwg_layer launch");
}

int launch_pwg_layer_fused(const fcl_pwg_layer_t& L, hipStream_t s) {
    PwgFusedArgs a = {};
    const int R = L.r, ldx = R / 32, ldc = (L.aux + 31) / 32;
    for (int j = 0; j < L.ksize; ++j) {
        a.term[j].K = R;
        a.term[j].shift = (j - (L.ksize - 1) / 2) * L.dilation;
        a.term[j].Ap = L.xp; a.term[j].lda_p = ldx; a.term[j].a_chunk_stride = (int64_t)L.m * 128;  // the one-launch block keeps x and aux chunk-major
        a.term[j].Wp = L.w_conv_p + (size_t)j * 2 * R * ldx * 64; a.term[j].ldw_p = ldx;
    }
    a.term[L.ksize].K = L.aux;
    a.term[L.ksize].Ap = L.cp; a.term[L.ksize].lda_p = ldc; a.term[L.ksize].a_chunk_stride = (int64_t)L.m * 128;
    a.term[L.ksize].Wp = L.w_aux_p; a.term[L.ksize].ldw_p = ldc;
    a.nterms = L.ksize + 1;
    a.M = (int)L.m;
    a.seg_lo = L.seg_lo; a.seg_hi = L.seg_hi;
    a.b_conv = L.b_conv; a.b_os = L.b_os; a.w_os_p = L.w_os_p;
    a.xp_in = L.xp; a.xp_out = L.xp_out; a.skips = L.skips; a.first = L.first_layer;
    a.xcs = (long long)L.m * 64;
    static const int dbg = tunable("PWG_DBG", 0), exp_terms = tunable("PWG_EXP_TERMS", 0), want_ts = tunable("PWG_TS", 0);
    a.dbg = dbg;
    static long long* ts_buf = nullptr;
    if (want_ts) {  // developer aid: one stamp buffer for the process, dumped by tools/pwg_stamps.py through fcl_debug_ptr
        if (!ts_buf && hipMalloc(&ts_buf, sizeof(long long) * 8 * ((L.m + 127) / 128 + 1)) != hipSuccess) ts_buf = nullptr;
        a.ts = ts_buf;
        g_debug_ptr = ts_buf;
    }
    if (exp_terms > 0) a.nterms = exp_terms;  // developer timing aid: fewer K-terms (results are then garbage)
    const bool hi = gemm_mode() == FCL_GEMM_BF16;
    const double flops = 2.0 * (double)L.m * 2.0 * R * ((double)L.ksize * R + L.aux + R);
    static const int persist = tunable("PWG_PERSIST", 1);
    const bool auxf = L.kp != nullptr;
    if (auxf) {  // one chunk: coefficient lines (one line per sample) x the tile's frame window
        GemmTerm& T = a.term[L.ksize];
        T.K = 32;
        T.Ap = L.kp; T.lda_p = 1; T.a_chunk_stride = 0;
        T.Wp = L.pt_a; T.ldw_p = L.ld_pt;
        a.pt_a = L.pt_a; a.pt_b = L.pt_b; a.hop = L.hop;
        FCL_REQUIRE(persist && !dbg && exp_terms <= 0, FCL_ERR_INVALID, "pwg_layer_fwd: the frame-rate auxiliary term runs on the persistent kernel only");
    }
    if (persist && !dbg && exp_terms <= 0 && (auxf || L.aux > 64)) {  // (the persistent kernel is written for 3 x 2 + 3 chunks: r = 64, ksize = 3, 64 < aux <= 96)
        constexpr int LDS = PGeo<4, 2, 2, 4, 3, 2>::LDS_BYTES + 2 * 32768;
        static const int plw = tunable("PWG_LOADERS", 4);  // 4 loader waves: 151.9 -> 145.1 ms per 30 blocks
        const int lwv = plw >= 4 ? 4 : 2;
        typedef void (*kern_t)(const PwgFusedArgs, const int);
        static const kern_t table[2][2][2] = {{{pwg_layer_pkernel<false, 2, false>, pwg_layer_pkernel<false, 2, true>},
                                               {pwg_layer_pkernel<false, 4, false>, pwg_layer_pkernel<false, 4, true>}},
                                              {{pwg_layer_pkernel<true, 2, false>, pwg_layer_pkernel<true, 2, true>},
                                               {pwg_layer_pkernel<true, 4, false>, pwg_layer_pkernel<true, 4, true>}}};
        const kern_t fn = table[hi ? 1 : 0][lwv == 4 ? 1 : 0][auxf ? 1 : 0];
        const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(fn), LDS);
        if (rc) return rc;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int ntiles = (int)((L.m + 127) / 128);
        static const int wg_per_cu = tunable("PWG_PERSIST_WGS", 0);  // 0: one workgroup per CU (160 KB of LDS each)
        int nwg = std::min(ntiles, wg_per_cu > 0 ? wg_per_cu : cus);
        ProfScope ps(hi ? "pwg_layer_pkernel/bf16" : "pwg_layer_pkernel", flops, (int)L.m, s);
        hipLaunchKernelGGL(fn, dim3((unsigned)nwg), dim3(64 * (8 + lwv)), LDS, s, a, ntiles);
        return check_hip(hipGetLastError(), "pwg_layer persistent launch");
    }
    // measured on MI355X, 64 x 800 frames, ms per layer: 128-row tiles + 3-deep ring + W_os in LDS 7.07 (default); the same with a 4-deep ring and
    // W_os fragments from L2 8.4 (the main loop alone is 4.05 either way: not bound by bytes in flight); 64-row tiles, two workgroups per CU 7.5
    static const int cfg = tunable("PWG_CFG", 0);  // 1: 128 rows, 4-deep ring; 3: 64-row tiles, 3-deep; 4: 64-row tiles, 4-deep
    if (cfg == 1) return hi ? launch_pwg_cfg<4, 4, true>(a, L.m, flops, s) : launch_pwg_cfg<4, 4, false>(a, L.m, flops, s);
    if (cfg == 3) return hi ? launch_pwg_cfg<2, 3, true>(a, L.m, flops, s) : launch_pwg_cfg<2, 3, false>(a, L.m, flops, s);
    if (cfg == 4) return hi ? launch_pwg_cfg<2, 4, true>(a, L.m, flops, s) : launch_pwg_cfg<2, 4, false>(a, L.m, flops, s);
    return hi ? launch_pwg_cfg<4, 3, true>(a, L.m, flops, s) : launch_pwg_cfg<4, 3, false>(a, L.m, flops, s);
}

// ---- the generator's last_conv_layers in one launch: wav[m] = relu(relu(skips[m] * scale) W1^T + b1) . w2 + b2  (64 skip channels).
// A 128-sample tile of the skip accumulator is read ONCE (the kernel's only HBM traffic besides 4 bytes per sample out): ReLU * scale and the
// bf16x3 split go straight into LDS in the fragment layout (swizzled 128-byte lines, one 32-column chunk after the other), W1's planes are
// staged once per workgroup, the 64 x 64 GEMM runs on the MFMA pipe, and the 64 -> 1 projection is an in-register dot + a 16-lane reduction.
template <bool HI>
__global__ __launch_bounds__(256) void pwg_last_kernel(const float* __restrict__ skips, float scale, const u16* __restrict__ w1p, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, float b2, float* __restrict__ wav, int M, int ntiles) {
    constexpr int TM = 2, TN = 4, AB = 0, WB = 32768;
    __shared__ __attribute__((aligned(1024))) u8 smem[32768 + 16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 256) {  // W1 [64, 64] planes -> LDS, chunk-major
        const int n = i >> 4, c = (i >> 3) & 1, p = i & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(w1p + ((size_t)(n * 2 + c) * 64 + p * 8));
        *reinterpret_cast<uint4*>(smem + WB + c * 8192 + n * 128 + ((p ^ ((n >> 1) & 7)) << 4)) = v;
    }
    const int col = lane & 15, rq = lane >> 4, r16 = lane & 15, kq = lane >> 4, sw = r16 >> 1;
    float bb[TN], ww[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        bb[tn] = b1[tn * 16 + col];
        ww[tn] = w2[tn * 16 + col];
    }
    const int ar = (wave * TM * 16 + r16) * 128, br = r16 * 128;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int m0 = t * 128;
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j, row = i >> 4;
            v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (m0 + row < M) v[j] = *reinterpret_cast<const f32x4*>(skips + (size_t)(m0 + row) * 64 + (i & 15) * 4);
        }
        __syncthreads();  // the previous tile's fragments have been read (first pass: W1 is staged)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j, row = i >> 4, c4 = (i & 15) * 4, k = c4 & 31;
            f32x4_t y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(v[j][e] * scale, 0.f);
            uint2 h, l;
            split4(y, h, l);
            u8* line = smem + AB + (c4 >> 5) * 16384 + row * 128 + (k & 7) * 2;
            const int sws = (row >> 1) & 7;
            *reinterpret_cast<uint2*>(line + (((k >> 3) ^ sws) << 4)) = h;
            if (!HI) *reinterpret_cast<uint2*>(line + (((4 + (k >> 3)) ^ sws) << 4)) = l;
        }
        __syncthreads();
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c)
            pchunk_mma<TM, TN, HI>(smem, AB + c * 16384 + ar + ((kq ^ sw) << 4), AB + c * 16384 + ar + (((4 + kq) ^ sw) << 4),
                                   WB + c * 8192 + br + ((kq ^ sw) << 4), WB + c * 8192 + br + (((4 + kq) ^ sw) << 4), acc);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sum = 0.f;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) sum += fmaxf(acc[tm][tn][r] + bb[tn], 0.f) * ww[tn];
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
                const int m = m0 + (wave * TM + tm) * 16 + rq * 4 + r;
                if (col == 0 && m < M) wav[m] = sum + b2;
            }
    }
}

int launch_pwg_last_fused(const float* skips, float scale, const u16* w1p, const float* b1, const float* w2, float b2, float* wav, long long m, hipStream_t s) {
    const bool hi = gemm_mode() == FCL_GEMM_BF16;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int ntiles = (int)((m + 127) / 128);
    const unsigned grid = (unsigned)std::min(ntiles, cus * 3);  // 48 KB of LDS: three workgroups per CU
    ProfScope ps(hi ? "pwg_last_kernel/bf16" : "pwg_last_kernel", 2.0 * (double)m * 64 * 65, (int)m, s);
    if (hi) hipLaunchKernelGGL(pwg_last_kernel<true>, dim3(grid), dim3(256), 0, s, skips, scale, w1p, b1, w2, b2, wav, (int)m, ntiles);
    else hipLaunchKernelGGL(pwg_last_kernel<false>, dim3(grid), dim3(256), 0, s, skips, scale, w1p, b1, w2, b2, wav, (int)m, ntiles);
    return check_hip(hipGetLastError(), "pwg_last launch");
}

// --------------------------------------------------------------------------------------------------------------------------------------
__global__ void pack_planes_kernel(const float* __restrict__ x, int ld, int rows, int cols, u16* __restrict__ out, int ldp) {
    const long long total = (long long)rows * ldp * 32;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / (ldp * 32)), k = (int)(i - (long long)r * (ldp * 32));
        store_p32(out, ldp, r, k, k < cols ? x[(size_t)r * ld + k] : 0.f);
    }
}

// Transposed planes (P32T) of x [rows, cols]: one plane row per COLUMN, 32 consecutive rows m per 128-byte line (hi | lo) -- the operand layout
// of the weight-gradient GEMM (contraction over rows).  Tap t (grid.z) reads row m + shift0 + t, zero outside [seg_lo[m], seg_hi[m]) (or outside
// [0, rows)) and past `rows`; plane row = t * cols + c.  One 32 x 32 tile per workgroup through LDS: coalesced reads along c, whole lines out.
__global__ __launch_bounds__(256) void pack_planes_t_kernel(const float* __restrict__ x, int ld, int rows, int cols, int shift0,
                                                            const int* __restrict__ seg_lo, const int* __restrict__ seg_hi, u16* __restrict__ out, int ldp) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tap = blockIdx.z, shift = shift0 + tap;
    const int j = threadIdx.x & 31, i0 = threadIdx.x >> 5;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = i0 + 8 * p, m = m0 + i, c = c0 + j;
        float v = 0.f;
        if (m < rows && c < cols) {
            const int src = m + shift;
            const bool ok = seg_lo ? (src >= seg_lo[m] && src < seg_hi[m]) : (src >= 0 && src < rows);
            if (ok) v = x[(size_t)src * ld + c];
        }
        tile[i][j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = i0 + 8 * p, c = c0 + i;  // plane row = column c; lane j = row m0 + j of the source
        if (c >= cols) continue;
        const float v = tile[j][i];
        const __bf16 h = (__bf16)v;
        u16* line = out + (((size_t)tap * cols + c) * ldp + blockIdx.x) * 64 + j;
        line[0] = __builtin_bit_cast(u16, h);
        line[32] = __builtin_bit_cast(u16, (__bf16)(v - (float)h));
    }
}

}  // namespace fcl

using namespace fcl;

extern "C" {

size_t fcl_planes_elems(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)rows * ((cols + 31) / 32) * 64;
}

void* fcl_debug_ptr(void) { return fcl::g_debug_ptr; }

int fcl_pack_planes_t(const float* x, int ld, int rows, int cols, int ntaps, int shift0, const int32_t* seg_lo, const int32_t* seg_hi, uint16_t* out,
                      fcl_stream_t stream) {
    FCL_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols && ntaps >= 1, FCL_ERR_INVALID, "pack_planes_t: bad arguments");
    FCL_REQUIRE((seg_lo == nullptr) == (seg_hi == nullptr), FCL_ERR_INVALID, "pack_planes_t: seg_lo/seg_hi come in pairs");
    FCL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 127u) == 0, FCL_ERR_ALIGN, "pack_planes_t: the plane buffer must be 128-byte aligned");
    const int ldp = (rows + 31) / 32;
    FCL_REQUIRE((cols + 31) / 32 <= 65535 && ntaps <= 65535, FCL_ERR_SHAPE, "pack_planes_t: too many columns / taps for one launch");
    hipLaunchKernelGGL(pack_planes_t_kernel, dim3((unsigned)ldp, (unsigned)((cols + 31) / 32), (unsigned)ntaps), dim3(256), 0, (hipStream_t)stream, x, ld, rows,
                       cols, shift0, seg_lo, seg_hi, out, ldp);
    return check_hip(hipGetLastError(), "pack_planes_t");
}

int fcl_gemm_tn_planes(const uint16_t* ap_t, const uint16_t* bp_t, float* c, int ldc, int m, int n, int k, int nblk, size_t blk_stride,
                       fcl_stream_t stream) {
    FCL_REQUIRE(ap_t && bp_t && c && m > 0 && n > 0 && k > 0, FCL_ERR_INVALID, "gemm_tn_planes: bad arguments");
    FCL_REQUIRE(((reinterpret_cast<uintptr_t>(ap_t) | reinterpret_cast<uintptr_t>(bp_t)) & 127u) == 0, FCL_ERR_ALIGN, "gemm_tn_planes: planes must be 128-byte aligned");
    FCL_REQUIRE(nblk >= 0 && (nblk == 0 || ((nblk & 3) == 0 && k % nblk == 0)), FCL_ERR_SHAPE, "gemm_tn_planes: nblk must be a multiple of 4 dividing k");
    FCL_REQUIRE(tunable("PRECISION", 1) != 0, FCL_ERR_INVALID, "gemm_tn_planes: the pre-split path is off under FCL_PRECISION=0 (use fcl_gemm_tn_fwd)");
    GemmArgs g = {};
    g.nterms = 1;
    g.term[0].Ap = ap_t;
    g.term[0].Wp = bp_t;
    g.term[0].lda_p = g.term[0].ldw_p = (m + 31) / 32;
    g.term[0].K = m;
    g.M = n;
    g.N = k;
    g.Y = c;
    g.ldy = ldc;
    g.accumulate = 1;
    g.nblk = nblk;
    g.blk_stride = (long long)blk_stride;
    return launch_gemm_planes(g, (hipStream_t)stream);
}

int fcl_pack_planes(const float* x, int ld, int rows, int cols, uint16_t* out, fcl_stream_t stream) {
    FCL_REQUIRE(x && out && rows >= 0 && cols > 0 && ld >= cols, FCL_ERR_INVALID, "pack_planes: bad arguments");
    FCL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 127u) == 0, FCL_ERR_ALIGN, "pack_planes: the plane buffer must be 128-byte aligned");
    if (rows == 0) return 0;
    const int ldp = (cols + 31) / 32;
    long long blocks = ((long long)rows * ldp * 32 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, out, ldp);
    return check_hip(hipGetLastError(), "pack_planes");
}

}  // extern "C"
