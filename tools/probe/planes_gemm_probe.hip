// planes_gemm_probe.hip — developer probe for the pre-split-planes GEMM core (gfx950): bf16x3 operands that arrive ALREADY split
// (hi = bf16_rn(x), lo = bf16_rn(x - hi)) in the "P32" layout, staged global -> LDS by direct LDS-DMA (global_load_lds, 16 B per lane)
// through an NST-deep ring with counted vmcnt waits and one raw barrier per 32-k chunk; no VGPR staging and no VALU work in the main loop.
//   P32 layout of X [R, K]: uint16 [R][Kp/32][2][32], Kp = roundup(K, 32): per row and 32-k block one 128-byte line = 64 B hi | 64 B lo.
//   LDS image of a chunk: rows of 128 B; 16-byte piece c (c = plane*4 + k/8) of row r sits at position c ^ ((r >> 1) & 7): the MFMA
//   fragment reads (ds_read_b128, lane (r16, kq) -> row r16, piece kq) are then bank-conflict free.  LDS-DMA writes lane-linear, so the
//   permutation is applied to the per-lane SOURCE address (guide §5.4 rule 21).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/planes_gemm_probe.hip -o tools/probe/planes_gemm_probe -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef unsigned short u16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

struct PTerm {
    const u16* Ap;  // P32 planes of A [rows, K]
    const u16* Wp;  // P32 planes of W [N, K]
    int lda_b, ldw_b;  // row strides in BYTES (= Kp * 4)
    int K, shift;
    int a_zrow;  // index of the all-zero row of the A plane buffer (= its row count); W planes: row N is the zero row
};
struct PArgs {
    PTerm term[9];
    int nterms, M, N;
    const int* seg_lo;
    const int* seg_hi;
    const float* bias;
    float* Y;
    int ldy;
    const u16* zero;  // >= 128 bytes of zeros
};

__global__ void pack_p32_kernel(const float* __restrict__ x, int ld, int rows, int K, u16* __restrict__ out, int ldp) {
    const int Kp = (K + 31) & ~31;
    const long long total = (long long)rows * Kp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / Kp), k = (int)(i - (long long)r * Kp);
        const float v = k < K ? x[(size_t)r * ld + k] : 0.f;
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        u16* line = out + ((size_t)r * ldp + (k >> 5)) * 64;
        line[k & 31] = __builtin_bit_cast(u16, h);
        line[32 + (k & 31)] = __builtin_bit_cast(u16, l);
    }
}

__device__ __forceinline__ void xcd_tile(int& bx, int& by) {
    const int nx = gridDim.x, nwg = gridDim.x * gridDim.y;
    const int orig = blockIdx.y * nx + blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    by = t / nx;
    bx = t - by * nx;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// VAR (ablation): 0 = full kernel; 1 = LDS-DMA pipeline only (no fragment reads / MFMAs); 2 = no loads (MFMAs on whatever LDS holds); 3 = no output stores
template <int WM, int WN, int TM, int TN, int NST, int VAR = 0, int KS = 1>
__global__ __launch_bounds__(64 * WM * WN) void pgemm_kernel(const PArgs a) {
    constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN, NW = WM * WN;
    constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;  // LDS-DMA row groups (8 rows = 1 KB) per wave per chunk
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0 && (NW % 2) == 0, "tile rows must split evenly over the waves");
    constexpr int GPW = (GA + GB) * KS;
    constexpr int SUB = (BM + BN) * 128;  // one 32-k chunk
    constexpr int STAGE = SUB * KS;       // KS chunks per ring stage: one wait + barrier per KS * 32 k
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = by * BM, n0 = bx * BN;

    // ---- loader coordinates: group g = j * NW + wave covers rows g*8 .. g*8+7 of its region; lane -> (row = lane >> 3, piece = lane & 7)
    const unsigned coff = (unsigned)(((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4))) * 16);  // source piece for this lane's LDS slot
    int am[GA], alo[GA];
    unsigned alen[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int m = m0 + (j * NW + wave) * 8 + (lane >> 3);
        am[j] = m;
        alo[j] = 0;
        alen[j] = m < a.M ? (unsigned)a.M : 0u;  // rows past M: empty segment -> zero row
        if (a.seg_lo != nullptr && m < a.M) {
            alo[j] = a.seg_lo[m];
            alen[j] = (unsigned)(a.seg_hi[m] - alo[j]);
        }
    }
    int bn[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int n = n0 + (j * NW + wave) * 8 + (lane >> 3);
        bn[j] = n < a.N ? n : a.N;  // row N of every W plane buffer is the all-zero row
    }
    // issue-side state: 32-bit byte offsets from the term's (uniform) base pointers, advanced by 128 B per chunk
    unsigned oa[GA], ob[GB];
    const unsigned char *baseA = nullptr, *baseW = nullptr;
    int rem = 0, it = 0;
    auto setup_term = [&](int t) {
        const PTerm T = a.term[t];
        baseA = reinterpret_cast<const unsigned char*>(T.Ap);
        baseW = reinterpret_cast<const unsigned char*>(T.Wp);
#pragma unroll
        for (int j = 0; j < GA; ++j) {
            const int src = am[j] + T.shift;
            const int row = (unsigned)(src - alo[j]) < alen[j] ? src : T.a_zrow;
            oa[j] = (unsigned)row * (unsigned)T.lda_b + coff;
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) ob[j] = (unsigned)bn[j] * (unsigned)T.ldw_b + coff;
        rem = (T.K + 31) >> 5;
    };
    auto issue = [&](int stage) {
        if (VAR == 2) return;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {  // (terms are whole multiples of KS chunks in this probe)
            unsigned char* sbase = smem + stage * STAGE + ks * SUB + wave * 1024;
#pragma unroll
            for (int j = 0; j < GA; ++j) {
                glds16(baseA + oa[j], sbase + j * NW * 1024);
                oa[j] += 128;
            }
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                glds16(baseW + ob[j], sbase + BM * 128 + j * NW * 1024);
                ob[j] += 128;
            }
        }
    };
    auto step_term = [&]() {
        rem -= KS;
        if (rem <= 0 && ++it < a.nterms) setup_term(it);  // rare, wave-uniform
    };

    // ---- fragment read offsets (bytes within a stage)
    const int r16 = lane & 15, kq = lane >> 4;
    const int sw = r16 >> 1;
    const int a_hi = (wm * TM * 16 + r16) * 128 + ((kq ^ sw) << 4);
    const int a_lo = (wm * TM * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
    const int b_hi = BM * 128 + (wn * TN * 16 + r16) * 128 + ((kq ^ sw) << 4);
    const int b_lo = BM * 128 + (wn * TN * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int stage) {
        if (VAR == 1) return;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
        const unsigned char* sb = smem + stage * STAGE + ks * SUB;
        s16x8 ah[TM], al[TM], bh[TN], bl[TN];
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
    };

    int nchunks = 0;
    for (int t = 0; t < a.nterms; ++t) nchunks += ((a.term[t].K + 31) >> 5) / KS;
    setup_term(0);
    int issued = 0;
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) {
        if (issued < nchunks) { issue(p); ++issued; step_term(); }
    }
    int cs = 0, is = NST - 1;
    for (int i = 0; i < nchunks; ++i) {  // chunk i is consumed while chunks i+1 .. i+NST-2 stay in flight and chunk i+NST-1 is issued
        const int left = nchunks - 1 - i;
        if (NST >= 4 && left >= 2) wait_vm<2 * GPW>();
        else if (NST >= 3 && left >= 1) wait_vm<GPW>();
        else wait_vm<0>();
        asm volatile("s_barrier" ::: "memory");  // chunk i has landed for every wave; everyone is done reading the buffer refilled now
        if (left >= NST - 1) {
            issue(is);
            step_term();
            is = is + 1 == NST ? 0 : is + 1;
        }
        compute(cs);
        cs = cs + 1 == NST ? 0 : cs + 1;
    }

    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + (wn * TN + tn) * 16 + col;
            if (n >= a.N) continue;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TM + tm) * 16 + rq * 4 + r;
                if (m < a.M && (VAR == 0 || acc[tm][tn][r] == 12345.678f)) a.Y[(size_t)m * a.ldy + n] = acc[tm][tn][r] + bv;
            }
        }
}

// Loader / consumer specialisation: LW extra waves do nothing but LDS-DMA (all (BM+BN)/8 row groups of a chunk), the WM x WN compute waves do
// nothing but fragment reads and MFMAs; one barrier per chunk joins them.  Tests whether a wave's own glds issue time is what serialises the loop.
template <int WM, int WN, int TM, int TN, int NST, int LW>
__global__ __launch_bounds__(64 * (WM * WN + LW)) void pgemm_ls_kernel(const PArgs a) {
    constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN, NW = WM * WN;
    constexpr int GA = BM / 8 / LW, GB = BN / 8 / LW;
    static_assert((BM / 8) % LW == 0 && (BN / 8) % LW == 0 && (LW % 2) == 0, "row groups must split evenly over the loader waves");
    constexpr int GPW = GA + GB;
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_tile(bx, by);
    const int m0 = by * BM, n0 = bx * BN;
    int nchunks = 0;
    for (int t = 0; t < a.nterms; ++t) nchunks += (a.term[t].K + 31) >> 5;

    if (wave >= NW) {  // ---------------- loader wave
        const int lw = wave - NW;
        const unsigned coff = (unsigned)(((lane & 7) ^ (((lw & 1) << 2) | (lane >> 4))) * 16);
        int am[GA], alo[GA];
        unsigned alen[GA];
#pragma unroll
        for (int j = 0; j < GA; ++j) {
            const int m = m0 + (j * LW + lw) * 8 + (lane >> 3);
            am[j] = m;
            alo[j] = 0;
            alen[j] = m < a.M ? (unsigned)a.M : 0u;
            if (a.seg_lo != nullptr && m < a.M) {
                alo[j] = a.seg_lo[m];
                alen[j] = (unsigned)(a.seg_hi[m] - alo[j]);
            }
        }
        int bn[GB];
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const int n = n0 + (j * LW + lw) * 8 + (lane >> 3);
            bn[j] = n < a.N ? n : a.N;
        }
        unsigned oa[GA], ob[GB];
        const unsigned char *baseA = nullptr, *baseW = nullptr;
        int rem = 0, it = 0;
        auto setup_term = [&](int t) {
            const PTerm T = a.term[t];
            baseA = reinterpret_cast<const unsigned char*>(T.Ap);
            baseW = reinterpret_cast<const unsigned char*>(T.Wp);
#pragma unroll
            for (int j = 0; j < GA; ++j) {
                const int src = am[j] + T.shift;
                const int row = (unsigned)(src - alo[j]) < alen[j] ? src : T.a_zrow;
                oa[j] = (unsigned)row * (unsigned)T.lda_b + coff;
            }
#pragma unroll
            for (int j = 0; j < GB; ++j) ob[j] = (unsigned)bn[j] * (unsigned)T.ldw_b + coff;
            rem = (T.K + 31) >> 5;
        };
        auto issue = [&](int stage) {
            unsigned char* sbase = smem + stage * STAGE + lw * 1024;
#pragma unroll
            for (int j = 0; j < GA; ++j) {
                glds16(baseA + oa[j], sbase + j * LW * 1024);
                oa[j] += 128;
            }
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                glds16(baseW + ob[j], sbase + BM * 128 + j * LW * 1024);
                ob[j] += 128;
            }
            if (--rem == 0 && ++it < a.nterms) setup_term(it);
        };
        setup_term(0);
        int issued = 0;
#pragma unroll
        for (int p = 0; p < NST - 1; ++p)
            if (issued < nchunks) { issue(p); ++issued; }
        int is = NST - 1;
        for (int i = 0; i < nchunks; ++i) {
            const int left = nchunks - 1 - i;
            if (NST >= 4 && left >= 2) wait_vm<2 * GPW>();
            else if (left >= 1) wait_vm<GPW>();
            else wait_vm<0>();
            asm volatile("s_barrier" ::: "memory");
            if (left >= NST - 1) {
                issue(is);
                is = is + 1 == NST ? 0 : is + 1;
            }
        }
        return;
    }
    // ---------------- compute waves
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kq = lane >> 4;
    const int sw = r16 >> 1;
    const int a_hi = (wm * TM * 16 + r16) * 128 + ((kq ^ sw) << 4);
    const int a_lo = (wm * TM * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
    const int b_hi = BM * 128 + (wn * TN * 16 + r16) * 128 + ((kq ^ sw) << 4);
    const int b_lo = BM * 128 + (wn * TN * 16 + r16) * 128 + (((4 + kq) ^ sw) << 4);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int cs = 0;
    for (int i = 0; i < nchunks; ++i) {
        asm volatile("s_barrier" ::: "memory");
        const unsigned char* sb = smem + cs * STAGE;
        s16x8 ah[TM], al[TM], bh[TN], bl[TN];
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
        cs = cs + 1 == NST ? 0 : cs + 1;
    }
    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + (wn * TN + tn) * 16 + col;
            if (n >= a.N) continue;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TM + tm) * 16 + rq * 4 + r;
                if (m < a.M) a.Y[(size_t)m * a.ldy + n] = acc[tm][tn][r] + bv;
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------------------
struct Mat {
    std::vector<float> h;
    float* d = nullptr;
    u16* p = nullptr;
    int rows, K, Kp, ldp;  // ldp: row stride in 128-byte blocks
};

static int g_pad = 0;
static Mat make(int rows, int K, unsigned seed, float scale) {
    Mat m;
    m.rows = rows;
    m.K = K;
    m.Kp = (K + 31) & ~31;
    m.ldp = m.Kp / 32 + g_pad;
    m.h.resize((size_t)rows * K);
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& v : m.h) {
        s = s * 1664525u + 1013904223u;
        v = ((int)(s >> 8) % 20001 - 10000) / 10000.0f * scale;
    }
    CK(hipMalloc(&m.d, m.h.size() * 4));
    CK(hipMemcpy(m.d, m.h.data(), m.h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&m.p, (size_t)(rows + 1) * m.ldp * 128 + 256));
    CK(hipMemset(m.p, 0, (size_t)(rows + 1) * m.ldp * 128 + 256));
    hipLaunchKernelGGL(pack_p32_kernel, dim3(1024), dim3(256), 0, 0, m.d, K, rows, K, m.p, m.ldp);
    CK(hipDeviceSynchronize());
    return m;
}

typedef int (*linear_fn)(const float*, int, const float*, int, const float*, float*, int, int, int, int, int, void*);
typedef int (*conv_fn)(const float*, const float*, const float*, const int*, const int*, const float*, float*, int, int, int, int, int, void*);

template <int WM, int WN, int TM, int TN, int NST, int VAR = 0, int KS = 1>
static float run_cfg(const PArgs& a, int iters, const char* tag, double flops) {
    constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN;
    const size_t lds = (size_t)NST * (BM + BN) * 128 * KS;
    auto k = pgemm_kernel<WM, WN, TM, TN, NST, VAR, KS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(64 * WM * WN), lds, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(64 * WM * WN), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1e3f / iters;
    printf("  %-28s tile %3dx%3d x%d stages, %4d wg: %8.2f us  %7.1f TF(fp32-eq)\n", tag, BM, BN, NST, grid.x * grid.y, us, flops / us / 1e6);
    return us;
}

template <int WM, int WN, int TM, int TN, int NST, int LW>
static float run_ls(const PArgs& a, int iters, const char* tag, double flops) {
    constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN;
    const size_t lds = (size_t)NST * (BM + BN) * 128;
    auto k = pgemm_ls_kernel<WM, WN, TM, TN, NST, LW>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM);
    const int threads = 64 * (WM * WN + LW);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(threads), lds, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(threads), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1e3f / iters;
    printf("  %-28s tile %3dx%3d x%d stages, %4d wg: %8.2f us  %7.1f TF(fp32-eq)\n", tag, BM, BN, NST, grid.x * grid.y, us, flops / us / 1e6);
    return us;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    g_pad = argc > 2 ? atoi(argv[2]) : 0;
    printf("row pad: %d x 128 B\n", g_pad);
    void* lib = dlopen("fcl-taco2_amd/libfcl_hip.so", RTLD_NOW);
    linear_fn lin = lib ? (linear_fn)dlsym(lib, "fcl_linear_fwd") : nullptr;
    conv_fn conv = lib ? (conv_fn)dlsym(lib, "fcl_conv1d_fwd") : nullptr;
    if (!lin) printf("(libfcl_hip.so not found: no old-kernel comparison)\n");
    u16* zero;
    CK(hipMalloc(&zero, 4096));
    CK(hipMemset(zero, 0, 4096));

    struct Shape { const char* name; int M, N, K, taps; };
    const Shape shapes[] = {{"lstm S  (2 x K256)", 2501, 1024, 256, 2}, {"hoist S", 2501, 1024, 256, 1}, {"enc conv S (5 taps)", 3200, 256, 256, 5},
                            {"postnet S (5 taps)", 25026, 128, 128, 5}, {"postnet S last (N=80)", 25026, 80, 128, 5}, {"pred conv (3 taps, 384)", 3200, 384, 384, 3},
                            {"lstm T (2 x K1024)", 2501, 4096, 1024, 2}, {"lstm S small M", 600, 1024, 256, 2}};
    for (const Shape& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K, taps = sh.taps;
        const bool is_conv = taps > 2;
        printf("%s: M=%d N=%d K=%d x %d terms\n", sh.name, M, N, K, taps);
        // A: one activation matrix (conv: shared by all taps; lstm: two different matrices); W: taps matrices [N, K]
        std::vector<Mat> A, W;
        for (int t = 0; t < (is_conv ? 1 : taps); ++t) A.push_back(make(M, K, 7 + t, 1.0f));
        for (int t = 0; t < taps; ++t) W.push_back(make(N, K, 100 + t, 1.0f / sqrtf((float)K * taps)));
        std::vector<int> lo(M), hi(M);
        for (int m = 0; m < M; ++m) { lo[m] = m / 100 * 100; hi[m] = std::min(M, lo[m] + 100); }
        int *dlo, *dhi;
        CK(hipMalloc(&dlo, M * 4));
        CK(hipMalloc(&dhi, M * 4));
        CK(hipMemcpy(dlo, lo.data(), M * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dhi, hi.data(), M * 4, hipMemcpyHostToDevice));
        float* Y;
        CK(hipMalloc(&Y, (size_t)M * N * 4));
        CK(hipMemset(Y, 0, (size_t)M * N * 4));
        PArgs a = {};
        a.nterms = taps; a.M = M; a.N = N; a.Y = Y; a.ldy = N; a.zero = zero;
        for (int t = 0; t < taps; ++t) {
            const Mat& am = is_conv ? A[0] : A[t];
            a.term[t] = PTerm{am.p, W[t].p, am.ldp * 128, W[t].ldp * 128, K, is_conv ? t - taps / 2 : 0, M};
        }
        if (is_conv) { a.seg_lo = dlo; a.seg_hi = dhi; }
        const double flops = 2.0 * M * N * (double)K * taps;
        // correctness on sampled entries (fp64 reference)
        run_cfg<2, 2, 2, 4, 3>(a, 1, "check", flops);
        std::vector<float> y((size_t)M * N);
        CK(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        unsigned s = 99;
        for (int trial = 0; trial < 4000; ++trial) {
            s = s * 1664525u + 1013904223u;
            int m = (s >> 8) % M;
            s = s * 1664525u + 1013904223u;
            int n = (s >> 8) % N;
            if (trial < 64) { m = trial < 32 ? trial : M - 1 - (trial - 32); n = (trial * 37) % N; }
            double ref = 0;
            for (int t = 0; t < taps; ++t) {
                const Mat& am = is_conv ? A[0] : A[t];
                const int src = m + (is_conv ? t - taps / 2 : 0);
                if (is_conv && (src < lo[m] || src >= hi[m])) continue;
                for (int k = 0; k < K; ++k) ref += (double)am.h[(size_t)src * K + k] * (double)W[t].h[(size_t)n * K + k];
            }
            worst = fmax(worst, fabs(ref - y[(size_t)m * N + n]));
            scale = fmax(scale, fabs(ref));
        }
        printf("  max |err| over 4000 samples: %.3e (max |ref| %.3f)  %s\n", worst, scale, worst < 2e-5 * fmax(1.0, scale) ? "OK" : "**** MISMATCH ****");
        // timing: new configurations
        run_cfg<2, 2, 2, 4, 3>(a, iters, "new <2,2,2,4> 3st", flops);
        run_cfg<2, 2, 2, 4, 3, 1>(a, iters, "  ... DMA pipeline only", flops);
        run_cfg<2, 2, 2, 4, 3, 2>(a, iters, "  ... no loads", flops);
        run_cfg<2, 2, 2, 4, 3, 3>(a, iters, "  ... no stores", flops);
        run_cfg<2, 2, 2, 2, 4>(a, iters, "new <2,2,2,2> 4st", flops);
        run_cfg<2, 2, 2, 2, 4, 1>(a, iters, "  ... DMA pipeline only", flops);
        run_cfg<2, 2, 2, 2, 4, 2>(a, iters, "  ... no loads", flops);
        run_cfg<4, 2, 2, 4, 3>(a, iters, "new <4,2,2,4> 3st 8 waves", flops);
        run_cfg<4, 2, 2, 4, 3, 1>(a, iters, "  ... DMA pipeline only", flops);
        run_cfg<4, 2, 2, 4, 3, 2>(a, iters, "  ... no loads", flops);
        run_cfg<4, 2, 2, 4, 3, 3>(a, iters, "  ... no stores", flops);
        run_cfg<2, 2, 2, 2, 3, 0, 2>(a, iters, "BK64 <2,2,2,2> 3st", flops);
        run_cfg<2, 2, 2, 2, 4, 0, 2>(a, iters, "BK64 <2,2,2,2> 4st", flops);
        run_cfg<2, 2, 2, 4, 3, 0, 2>(a, iters, "BK64 <2,2,2,4> 3st", flops);
        run_cfg<4, 2, 2, 4, 2, 0, 2>(a, iters, "BK64 <4,2,2,4> 2st", flops);
        run_ls<4, 2, 2, 4, 3, 2>(a, iters, "LS <4,2,2,4> 3st +2 loaders", flops);
        run_ls<4, 2, 2, 4, 4, 2>(a, iters, "LS <4,2,2,4> 4st +2 loaders", flops);
        run_ls<2, 2, 2, 4, 3, 2>(a, iters, "LS <2,2,2,4> 3st +2 loaders", flops);
        run_ls<2, 2, 4, 4, 3, 2>(a, iters, "LS <2,2,4,4> 3st +2 loaders", flops);
        run_ls<2, 2, 4, 4, 3, 4>(a, iters, "LS <2,2,4,4> 3st +4 loaders", flops);
        {   // correctness of the specialised kernel on a few entries
            CK(hipMemset(Y, 0, (size_t)M * N * 4));
            run_ls<4, 2, 2, 4, 3, 2>(a, 1, "LS check", flops);
            std::vector<float> y2((size_t)M * N);
            CK(hipMemcpy(y2.data(), Y, y2.size() * 4, hipMemcpyDeviceToHost));
            double w2 = 0;
            for (size_t i = 0; i < y2.size(); i += 97) w2 = fmax(w2, fabs((double)y2[i] - (double)y[i]));
            printf("  LS vs base max diff %.3e %s\n", w2, w2 == 0 ? "OK" : "**** MISMATCH ****");
        }
        // old kernel through the library
        if (lin && conv) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            float* wp = nullptr;
            if (is_conv) {  // tap-major [taps, N, K]
                CK(hipMalloc(&wp, (size_t)taps * N * K * 4));
                for (int t = 0; t < taps; ++t) CK(hipMemcpy(wp + (size_t)t * N * K, W[t].d, (size_t)N * K * 4, hipMemcpyDeviceToDevice));
            }
            auto call = [&]() {
                if (is_conv) return conv(A[0].d, wp, nullptr, dlo, dhi, nullptr, Y, M, K, N, taps, 0, nullptr);
                return lin(A[0].d, K, W[0].d, K, nullptr, Y, N, M, N, K, 0, nullptr);  // one term only: K-matched below
            };
            if (!is_conv && taps == 2) {
                // the library's linear has one term: time K = 2K by concatenation is not available here -> time two launches of K (upper bound)
            }
            for (int i = 0; i < 3; ++i) call();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) call();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double fl_old = is_conv ? flops : 2.0 * M * N * (double)K;
            printf("  %-28s %42s %8.2f us  %7.1f TF(fp32-eq)%s\n", "old library kernel", "", ms * 1e3 / iters, fl_old / (ms * 1e3 / iters) / 1e6,
                   (!is_conv && taps == 2) ? "  (ONE K-term only)" : "");
            if (wp) CK(hipFree(wp));
        }
        for (auto& m : A) { CK(hipFree(m.d)); CK(hipFree(m.p)); }
        for (auto& m : W) { CK(hipFree(m.d)); CK(hipFree(m.p)); }
        CK(hipFree(dlo)); CK(hipFree(dhi)); CK(hipFree(Y));
    }
    return 0;
}
