// fcl_common.h — shared declarations for libfcl_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fcl_hip.h"

namespace fcl {

// ---- error plumbing: never throw across the C boundary ------------------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define FCL_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            ::fcl::set_error(__VA_ARGS__);      \
            return (code);                      \
        }                                       \
    } while (0)

#define FCL_HIP(expr)                                         \
    do {                                                      \
        int _rc = ::fcl::check_hip((expr), #expr);            \
        if (_rc != 0) return _rc;                             \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- fused multi-term GEMM (gemm_f32.hip) ---------------------------------------------------------
// Y[m, n] = epi( sum_t sum_k A_t[row_t(m), k] * W_t[n, k] )    A_t, W_t are K-contiguous (row-major)
// row_t(m) = (a_row_map ? a_row_map[m] : m) + shift_t, contributing 0 outside [seg_lo[m], seg_hi[m]).
typedef fcl_gemm_term_t GemmTerm;  // public C struct (include/fcl_hip.h)

enum { FCL_MAX_TERMS = 9 };

// d(loss)/d(a) of w1 * mean|d| + w2 * mean d^2 at d = a - b (one definition: the term-table loss kernel and the GEMM's MSE epilogue must agree bit for bit)
__device__ __forceinline__ float loss_grad1(float d, float w1, float w2, float inv_count) {
    return (w1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + 2.f * w2 * d) * inv_count;
}

struct GemmArgs {
    GemmTerm term[FCL_MAX_TERMS];
    int nterms;
    int M, N;
    const int* seg_lo;  // [M] or null (required when any shift != 0)
    const int* seg_hi;
    const float* bias;  // [N] or null
    const float* rank1_a;  // optional: acc += rank1_a[m * rank1_lda] * rank1_w[n]
    int rank1_lda;
    const float* rank1_w;
    const float* C0;  // optional [M, N] added before the activation
    int ldc0;
    int act;  // FCL_ACT_*
    const uint8_t* keep;  // optional {0,1} keep mask [M, N] applied after act, times keep_scale
    int ldkeep;
    float keep_scale;
    unsigned int rng_seed;  // drop_mode 2: counter-hash Bernoulli(keep 1-p) instead of a mask
    const unsigned int* seed_dev;  // optional device word added to the seed
    float drop_p;
    int drop_mode;  // 0 none, 1 mask, 2 rng
    const float* R;  // optional residual added after act/dropout
    int ldr;
    float* Y;
    int ldy;
    float* Y2;  // optional scattered copy: row = y2_row_base[m] + y2_row_add
    int ldy2;
    const int* y2_row_base;
    int y2_row_add;
    unsigned short* Yp;  // optional P32 planes of the output (ldyp lines per row, written up to the padded width); Y may then be null
    int ldyp;
    int hi_only;  // set by the launchers from gemm_mode(): operands rounded to bf16 (one MFMA per product) instead of the bf16x3 split
    // planes kernels only (the weight-gradient GEMM, fcl_gemm_tn_planes): split of the contraction over gridDim.z (chunks of 32 per slice, single
    // term), atomic accumulation into Y, and an output made of column blocks (column n -> Y + (n / nblk) * blk_stride + m * ldy + n % nblk)
    int conv_k;  // != 0 (set by fcl_conv1d_planes_fwd): the terms are the conv_k taps of ONE Conv1d -- same A planes, shifts -(k-1)/2 .. (k-1)/2, W tap-major
                 // and contiguous -- so the stencil kernel (pconv_kernel) may load each A tile once and reuse it for every tap
    int ksplit_chunks;
    int accumulate;
    int nblk;
    long long blk_stride;
    // grouped Conv1d (fcl_conv1d_planes_group_fwd, pconv_kernel only): gridDim.z independent problems of the same shape; group g reads / writes at
    // base + g * stride (A / W / Yp in uint16 elements, bias / Y in floats)
    long long g_a, g_w, g_bias, g_y, g_yp;
    int dbg_phase;  // developer timing aid (FCL_PGEMM_DBG): 1 = return after the main loop (results are then garbage)
    const int* m_dev;  // optional DEVICE row count (planes kernels): tiles at or beyond *m_dev exit at once -- the frame buffers of a capacity graph
                       // are sized with slack and nothing reads the rows past the batch's real total (fcl_conv1d_planes_rows_fwd)
    // MSE epilogue (planes kernels only, fcl_linear_planes_mse_fwd; round 6): with d = y - loss_t over the valid rows, Y / Yp receive the GRADIENT
    // 2 d * loss_gscale (zero on the other rows) instead of y, and sum |d|, sum d^2 and the element count are added to loss_sums[0 .. 2] (fp64 atomics):
    // the projection of a KD term, its loss and its gradient in one launch -- y itself never reaches memory
    // train-mode BatchNorm statistics in the epilogue (planes kernels only, fcl_conv1d_planes_bn_fwd; round 6): every workgroup adds the column sums and sums of
    // squares of its tile of outputs to bn_ws[0 : N] / bn_ws[N : 2 N] (fp64, zero on entry); the LAST row tile of a column tile to finish (bn_tickets[column tile], zero
    // on entry) turns them into mean, 1 / sqrt(biased var + eps) and the running statistics and leaves sums and ticket zero again -- bn_stats_kernel's protocol without
    // bn_stats_kernel's pass over the outputs
    double* bn_ws;
    unsigned int* bn_tickets;
    float bn_eps, bn_momentum;
    float *bn_mean, *bn_invstd, *bn_rmean, *bn_rvar;
    const float* loss_t;  // [M, N] target (ld_lt floats per row, 16-byte aligned rows), or null
    int ld_lt;
    const uint8_t* loss_valid;  // optional [M]
    float loss_gscale;  // 1 / count
    double* loss_sums;
};

// ---- fused LSTM step (gemm_f32.hip / decoder_step.hip): the argument block is the public fcl_lstm_step_t ----------
typedef fcl_lstm_step_t LstmStepArgs;

// ---- profiling hook (capi.hip) ----------------------------------------------------------------------
extern bool g_prof_on;
void prof_begin(const char* name, double flops, double rows, hipStream_t s, double fill_bytes = 0.0);
void prof_end(hipStream_t s);
struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(const char* name, double flops, double rows, hipStream_t st, double fill_bytes = 0.0) : s(st), on(g_prof_on) {
        if (on) prof_begin(name, flops, rows, s, fill_bytes);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};

// ---- fused feat_out(t-1) -> prenet(t) row-tile kernel (decoder_step.hip) -----------------------------
struct FeatPrenetArgs {
    int M_feat;  // rows live at step t-1 (feat_out part);  h1 == null => no feat part (t = 0, prev_out = 0)
    int M_pre;   // rows live at step t (prenet part);       w0 == null => feat only (after the last step)
    int U, O, P;
    const float* h1;    // [M_feat, U]
    const float* wf_h;  // [O, U]
    const float* F0;    // [*, O] hoisted att_c share of feat_out
    float* before;      // [F, O] frame-major output; row = frame_off[m] + t_prev
    const int* frame_off;
    int t_prev, t_cur;
    const float* teacher_in;  // optional [*, teacher_ld]: prenet input rows (teacher forcing)
    int teacher_ld;
    const float *w0, *b0, *w1, *b1;
    const unsigned short *wf_hi, *wf_lo, *w0_hi, *w0_lo, *w1_hi, *w1_lo;  // optional bf16x3 planes (all or none)
    const float *wf_ff, *w0_ff, *w1_ff;  // optional fragment-major fp32 forms (fcl_pack_frag_f32; exact-fp32 mode: all or none)
    int drop_mode;  // FCL_DROP_*
    const uint8_t *keep0, *keep1;  // [*, P] masks of the two layers
    float keep_scale, drop_p;
    unsigned int seed0, seed1;
    const unsigned int* seed_dev;  // optional device word added to both seeds
    float* pre_out;     // [M_pre, P] (optional when pre_out_p is given)
    float* tap_prenet;  // optional [F, P]; row = frame_off[m] + t_cur
    unsigned short* pre_out_p;  // optional P32 planes of pre_out (ceil(P/32) lines per row): the next LSTM step's pre-split operand
    unsigned short* before_p;   // optional P32 planes of `before` (ceil(O/32) lines per row, zero past O): the postnet's pre-split operand
    int dbg_phase;              // developer timing aid (FCL_FP_DBG): 1..3 = return after the loads / feat / prenet-0 phase (results are then garbage)
    const int* live;            // optional DEVICE live-row counts [*]: M_feat := min(M_feat, live[t_prev]), M_pre := min(M_pre, live[t_cur])
    unsigned int* status;       // with `live`: FCL_STATUS_ROWS_CAP when live[t_cur] exceeds the host's bound M_pre
    int out_act;                // FCL_ACT_*: output_activation_fn on the fed-back frame (decoder_sa.py:614-617); `before` stays raw
};

int row_maps_check(const fcl_row_maps_t* a, bool* fusable);  // pointwise.hip
// fcl_lstm_cell_bwd with row strides for the incoming / outgoing hidden-state carries (backward.hip); the pair form runs two independent problems
// of the same width in one launch
struct CellBwdArgs {
    const float *gates, *c_old, *c_new, *dh_out, *dh_out2, *dc_out;
    const uint8_t *zk_h, *zk_c;
    const int32_t* row_len;
    float *dgates, *dh_old, *dc_old;
    unsigned short* dgates_p;
    int ld_dh, ld_dho, ld_dh2, step, m, u;
    float zoneout;
};
int launch_lstm_cell_bwd(const CellBwdArgs& a, hipStream_t stream);
int launch_lstm_cell_bwd_pair(const CellBwdArgs& a0, const CellBwdArgs& a1, hipStream_t stream);
int launch_lstm_cell_bwd(const float* gates, const float* c_old, const float* c_new, const float* dh_out, int ld_dh, const float* dh_out2, int ld_dh2,
                         const float* dc_out, float zoneout, const uint8_t* zone_keep_h, const uint8_t* zone_keep_c, const int32_t* row_len, int step,
                         float* dgates, float* dh_old, int ld_dho, float* dc_old, uint16_t* dgates_p, int m, int u, hipStream_t stream);

// rows a step kernel processes: the host's count, or (device-driven loops) the smaller of the host's bound and the device's count
__device__ __forceinline__ int live_rows_of(int m_host, const int* m_dev) { return m_dev ? min(m_host, *m_dev) : m_host; }

// ---- bf16x3 operand split shared by the big-tile GEMMs (gemm_f32.hip) and the weight-gradient GEMM (backward.hip) ------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(const f32x4_t v, uint2& hi, uint2& lo) {
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[0], v[1]}, bf16x2_t));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2], v[3]}, bf16x2_t));
    const float r0 = v[0] - __builtin_bit_cast(float, h01 << 16), r1 = v[1] - __builtin_bit_cast(float, h01 & 0xFFFF0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h23 << 16), r3 = v[3] - __builtin_bit_cast(float, h23 & 0xFFFF0000u);
    hi = make_uint2(h01, h23);
    lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){r2, r3}, bf16x2_t)));
}


// ---- persistent BiLSTM training kernels (bilstm.hip) ---------------------------------------------------
// training forward: what bilstm_bptt_persistent_kernel needs, t-major ([T, B, .]); only live cells are written
struct BilstmSave {
    float* gates[2];  // per direction: activated i,f,g,o [T, B, 4H]
    float* c_new[2];  // [T, B, H]
    float* c_old[2];
    float* h_old[2];
    int B;
};

struct BilstmBwd {
    const float* gates[2];
    const float* c_new[2];
    const float* c_old[2];
    const float* whh_t[2];  // [H, 4H]
    float* dg[2];           // [T, B, 4H]
    const float* d_out;     // [B*T, ld], direction d in columns [d*H, d*H + H)
    int ld, B;
};

bool launch_bilstm_train_persistent(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r, const int* lens, float* out, int B, int T,
                                    int H, const BilstmSave& sv, hipStream_t s);
bool launch_bilstm_bptt_persistent(const BilstmBwd& a, const int* lens, int B, int T, int H, hipStream_t s);
size_t bilstm_group_workspace_bytes(int B, int H);
bool launch_bilstm_group(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r, const int* lens, float* out, int B, int T, int H,
                         const BilstmSave* sv, void* ws, size_t ws_bytes, unsigned int* status, hipStream_t s);
bool launch_bilstm_bptt_group(const BilstmBwd& a, const int* lens, int B, int T, int H, void* ws, size_t ws_bytes, unsigned int* status, hipStream_t s);

// P32 planes (include/fcl_hip.h "bf16x3 operand planes"): element (row m, column n) of a buffer with ld lines per row
__device__ __forceinline__ void store_p32(unsigned short* __restrict__ p, int ld, int m, int n, float v) {
    const __bf16 h = (__bf16)v;
    unsigned short* line = p + ((size_t)m * ld + (n >> 5)) * 64 + (n & 31);
    line[0] = __builtin_bit_cast(unsigned short, h);
    line[32] = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
}
bool planes_ok(const GemmTerm* t, int n);                  // every term carries A and W planes (and FCL_PLANES != 0)
int launch_gemm_planes(const GemmArgs& a, hipStream_t s);  // gemm_planes.hip
int launch_lstm_planes(const LstmStepArgs& a, hipStream_t s);
int launch_lstm_planes_pair(const LstmStepArgs& a0, const LstmStepArgs& a1, hipStream_t s, bool* handled);  // two independent steps, one launch
int launch_pwg_layer_fused(const fcl_pwg_layer_t& a, hipStream_t s);  // one Parallel WaveGAN residual block in one launch (r = 64, ksize = 3, aux <= 96)
int launch_pwg_last_fused(const float* skips, float scale, const unsigned short* w1p, const float* b1, const float* w2, float b2, float* wav, long long m,
                          hipStream_t s);  // last_conv_layers in one launch (64 skip channels)
int launch_gemm(const GemmArgs& a, hipStream_t s);
// dw_gemm.hip: dW[n, k] += sum_m a[m, n] * b[m + shift, k] on bf16x3 MFMAs with transposing LDS reads; false = shape left to gemm_tn_kernel
bool launch_dw_mfma(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, int shift0, int ntaps, size_t c_tap_stride,
                    const int32_t* seg_lo, const int32_t* seg_hi, int hi_only, hipStream_t stream);
int launch_lstm_step(const LstmStepArgs& a, hipStream_t s);
int validate_lstm_step(const LstmStepArgs& a);    // launch_lstm_step's argument checks alone (callers of the pair launches run them first)
bool lstm_step_on_planes(const LstmStepArgs& a);  // the step would run on pre-split operands (FCL_PRECISION / FCL_PLANES on, planes on every term)
bool lstm_step_is_small(int M, int U);  // M rows at width U go to the 16-row wave-per-gate kernel (fp32 operands) rather than a big-tile kernel
int launch_lstm_small(const LstmStepArgs& a, hipStream_t s);
int launch_lstm_small_pair(const LstmStepArgs& a0, const LstmStepArgs& a1, hipStream_t s);
int launch_lstm_small_pair_any(const LstmStepArgs& a0, const LstmStepArgs& a1, hipStream_t s, bool* handled);  // only where both would run lstm_small_kernel  // two same-shape small steps, one launch (fp32 operands)
int launch_feat_prenet(const FeatPrenetArgs& a, hipStream_t s);
int gemm_mode();  // capi.hip: the calling thread's fcl_set_gemm_mode() value (FCL_GEMM_F32 / FCL_GEMM_BF16)
int tunable(const char* name, int dflt);  // FCL_<NAME> environment override, read once
// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: opt the CURRENT device in for `bytes` of dynamic LDS the first time
// `func` is launched on it (thread-safe; every later call is a map lookup).  Returns 0 or FCL_ERR_HIP.
int ensure_dyn_lds(const void* func, int bytes);


// counter hash shared by the rng-dropout epilogue (and mirrored nowhere on the host: rng mode is the
// production mode and is not bit-reproducible against the reference's torch RNG stream by design).
__host__ __device__ static inline unsigned int hash_u32(unsigned int x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// dropout of four consecutive columns n .. n + 3 of row m (DROP: 0 none, 1 mask bytes, 2 counter hash: 16 bits per decision; idx = m * N + n) --
// shared by feat_prenet_split_kernel (decoder_step.hip) and decoder_tile_kernel (decoder_tile.hip): the same (seed, row, column) draws the same bits
template <int DROP>
__device__ __forceinline__ f32x4_t drop4(f32x4_t v, unsigned int keep4, unsigned int idx, unsigned int seed, unsigned int thr16, float scale) {
    if (DROP == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((keep4 >> (8 * r)) & 0xFFu) ? v[r] * scale : 0.f;
    }
    if (DROP == 2) {
        const unsigned int h0 = hash_u32(idx ^ seed), h1 = hash_u32((idx + 2u) ^ seed);
        v[0] = (h0 & 0xFFFFu) >= thr16 ? v[0] * scale : 0.f;
        v[1] = (h0 >> 16) >= thr16 ? v[1] * scale : 0.f;
        v[2] = (h1 & 0xFFFFu) >= thr16 ? v[2] * scale : 0.f;
        v[3] = (h1 >> 16) >= thr16 ? v[3] * scale : 0.f;
    }
    return v;
}

// ---- the persistent row-tile decoder loop (decoder_tile.hip) ---------------------------------------------------------------------------
bool decoder_tile_shape_ok(const fcl_decoder_weights_t* w);
size_t decoder_stream_bytes(const fcl_decoder_weights_t* w);
int decoder_stream_pack(const fcl_decoder_weights_t* w, void* out, hipStream_t s);
int launch_decoder_tile(const fcl_decoder_weights_t* w, const fcl_decoder_io_t* io, const float* G0, const float* F0, float* c0, float* c1, int drop_mode,
                        int t_start, const float* h0_init, const float* h1_init, hipStream_t s);

}  // namespace fcl
