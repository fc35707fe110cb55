/* fcl_hip.h — C ABI of libfcl_hip.so: the MI355X (gfx950) FCL-taco2 mel-synthesis hot path.
 *
 * The reference (Wendison/FCL-taco2) has no FFI: its plug-in point is the Python class path
 * `--model-module pkg.mod:Class` (tts_train.py:103-109).  This header is the boundary this build puts
 * UNDER that class: every entry point replaces a block of stock torch ops inside the reference's
 * nets/ modules (cited per function; paths relative to the reference root).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless the name ends in `_host`;
 *   - all activations are fp32, row-major, "channels-last": a sequence tensor is [rows, C] with
 *     rows = (utterance, time) flattened;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return 0 on success, a negative FCL_ERR_* otherwise; fcl_last_error() gives the message for the
 *     calling thread.  Nothing throws across the boundary.  Calls are re-entrant; the only global
 *     state is the thread-local error string.  The caller owns every buffer, workspaces included.
 */
#ifndef FCL_HIP_H_
#define FCL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fcl_stream_t;

enum { FCL_OK = 0, FCL_ERR_INVALID = -1, FCL_ERR_SHAPE = -2, FCL_ERR_ALIGN = -3, FCL_ERR_HIP = -4, FCL_ERR_WORKSPACE = -5 };
enum { FCL_ACT_NONE = 0, FCL_ACT_RELU = 1, FCL_ACT_TANH = 2, FCL_ACT_SIGMOID = 3 };
enum { FCL_DROP_NONE = 0, FCL_DROP_MASK = 1, FCL_DROP_RNG = 2 };
/* Bits of a DEVICE status word (uint32, caller-owned, zeroed by the caller): failures a kernel can only detect while it runs are OR-ed into
 * it instead of being lost.  fcl_adam_step refuses to update the parameters while the word is non-zero; the host reads it back next to the
 * losses (one 4-byte copy) and raises. */
enum {
    FCL_STATUS_GROUP_TIMEOUT = 1, /* a cooperating-workgroup BiLSTM kernel gave up waiting for a group member: its outputs are partial */
    /* device-built row maps (fcl_row_maps_build) and the decoder loop driven by them: */
    FCL_STATUS_ZERO_DURATION = 2, /* a non-padded phoneme has duration 0: the reference's `assert ds_nonzeros.shape[0] == hs.shape[0]`
                                   * (decoder_sa_kd.py:739) — the pass produced nothing valid */
    FCL_STATUS_LMAX_CAP = 4,      /* max duration > lmax_cap: the loop would have needed more steps than were launched (nothing was decoded) */
    FCL_STATUS_FRAMES_CAP = 8,    /* sum of durations > frames_cap: the frame-major buffers are too small (nothing was decoded) */
    FCL_STATUS_ROWS_CAP = 16      /* live rows of a decoder step > the host's bound for that step (rows beyond the bound were not computed) */
};

/* Arithmetic of the MFMA contractions (Linear / Conv1d / LSTM-step GEMMs and the weight-gradient GEMM) launched by the CALLING THREAD:
 *   FCL_GEMM_F32  (default) fp32-equivalent: both operands split into bf16 hi + lo, three MFMAs per product (or exact fp32 MFMAs under FCL_PRECISION=0);
 *   FCL_GEMM_BF16 the autocast form of mixed-precision training (the reference trains under apex AMP O1, tts.py:414-416; bf16 replaces its fp16 and
 *                 needs no loss scaling): both operands ROUNDED to bf16 (round to nearest even = the hi plane), one MFMA per product, fp32
 *                 accumulation, fp32 outputs.  Everything that is not a big-tile contraction keeps fp32: norms, activations, losses, the
 *                 optimizer, and the latency-bound small-row kernels (BiLSTM recurrences, decoder steps with few live rows).
 * The mode is thread-local, like torch.autocast.  Returns FCL_ERR_INVALID for an unknown mode, or for FCL_GEMM_BF16 under FCL_PRECISION=0. */
enum { FCL_GEMM_F32 = 0, FCL_GEMM_BF16 = 1 };

const char* fcl_last_error(void);
/* ABI revision of this header: bumped whenever a struct layout or a signature changes (100 = round 1; 200 = round 2: fcl_gemm_term_t.a_chunk_stride,
 * fcl_pwg_layer_t, the round-2 entry points).  A binding compares it with fcl_version() of the library it loaded before passing any struct. */
#define FCL_ABI_VERSION 422
int fcl_version(void);
void* fcl_debug_ptr(void); /* developer aid: device buffer of the last instrumented launch (FCL_PWG_TS), NULL otherwise */
int fcl_set_gemm_mode(int mode);
int fcl_get_gemm_mode(void);

/* ---- plan-time weight packing (run once per checkpoint) ------------------------------------------ */

/* torch Conv1d weight [Cout, Cin, k] -> tap-major [k, Cout, Cin], each output channel optionally
 * multiplied by scale[Cout] (eval BatchNorm folded into the conv; encoder_sa.py:61-78,
 * decoder_sa.py:199-263). */
int fcl_pack_conv1d_weight(const float* w, const float* scale, float* out, int cout, int cin, int k, fcl_stream_t stream);

/* Eval BatchNorm1d -> per-channel scale = gamma/sqrt(var+eps), shift = beta - mean*scale. */
int fcl_fold_batchnorm(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                       float* scale, float* shift, int c, fcl_stream_t stream);

/* dst[r, 0:cols] = src[r, 0:cols] (strided 2-D copy; splits LSTMCell.weight_ih / feat_out.weight into
 * their att_c / prenet / position column blocks, decoder_sa.py:357-369,397-398). */
int fcl_copy2d(float* dst, int ld_dst, const float* src, int ld_src, int rows, int cols, fcl_stream_t stream);

/* *p += v on the stream (a graph-capturable way to advance the dropout seed word). */
int fcl_u32_add(uint32_t* p, uint32_t v, fcl_stream_t stream);

/* bf16x3 operand planes of a weight matrix w [rows, cols] in MFMA-FRAGMENT-MAJOR order: hi = bf16_rn(x), lo = bf16_rn(x - hi),
 * stored as blocks (tile = row/16, step = col/32) of 64 lanes x 8 bf16, lane (r = lane&15, q = lane>>4) holding
 * w[tile*16 + r][step*32 + q*8 .. +7] (zero past rows/cols): element index ((tile*nsteps + step)*64 + lane)*8, nsteps =
 * ceil(cols/32).  One wave-wide 16-byte load then reads 1 KB contiguous = exactly one v_mfma_f32_16x16x32_bf16 B operand.
 * fcl_frag_bf16_elems() gives the plane length in uint16 elements.  Optional plan-time step for the decoder's small tiles. */
size_t fcl_frag_bf16_elems(int rows, int cols);
int fcl_pack_frag_bf16(const float* w, int rows, int cols, uint16_t* hi, uint16_t* lo, fcl_stream_t stream);
/* the same fragment order in fp32 (fcl_frag_bf16_elems(rows, cols) floats): lane (r16, kq) of (16-row tile, 32-k step) holds W[row][32 st + 4 kq + 0..3]
 * and W[row][32 st + 16 + 4 kq + 0..3] -- the operand form of the exact-fp32 (FCL_PRECISION=0) feat/prenet kernel (v_mfma_f32_16x16x4_f32) */
int fcl_pack_frag_f32(const float* w, int rows, int cols, float* out, fcl_stream_t stream);

/* ---- bf16x3 operand planes ("P32" layout) -----------------------------------------------------------------------------------------
 * The default arithmetic computes a.b as a_lo.b_hi + a_hi.b_lo + a_hi.b_hi on the bf16 MFMA pipe with fp32 accumulation, hi = bf16_rn(x),
 * lo = bf16_rn(x - hi).  Operands that are consumed by a GEMM are therefore kept ALREADY SPLIT next to (or instead of) their fp32 form:
 *     P32 planes of X [R, K]: uint16 [R][ld][2][32] — per row and per block of 32 columns one 128-byte line = 32 hi | 32 lo, zeros past K;
 *     ld >= ceil(K / 32) lines per row; the buffer must be 128-byte aligned.
 * Weights are packed once at plan time (fcl_pack_planes); activations are written in this form by the kernel that produces them (the `*_p`
 * outputs below), so the GEMM main loops move whole lines global -> LDS by LDS-DMA with no register staging and no conversion work. */
size_t fcl_planes_elems(int rows, int cols); /* uint16 elements of a dense plane buffer: rows * ceil(cols / 32) * 64 */
int fcl_pack_planes(const float* x, int ld, int rows, int cols, uint16_t* out, fcl_stream_t stream);

/* out = a + b (bias_ih + bias_hh). */
int fcl_add_vec(const float* a, const float* b, float* out, int n, fcl_stream_t stream);

/* ---- H1: Encoder.embed (encoder_sa.py:58,134) ------------------------------------------------------ */
/* out[m, :] = table[ids[m], :]; ids outside [0, V) produce a zero row.  out_p (optional): the same rows as P32 planes (ceil(e/32) lines per
 * row) for the convolution that consumes them; out may then be NULL. */
int fcl_embedding_fwd(const int64_t* ids, const float* table, float* out, uint16_t* out_p, int m, int v, int e, fcl_stream_t stream);

/* ---- H6/H8 + every nn.Linear: y = act(x . w^T + bias) --------------------------------------------- */
/* x [M, K] (lda), w [N, K] (torch layout, ldw), y [M, N] (ldy).  K, lda, ldw multiples of 4. */
int fcl_linear_fwd(const float* x, int lda, const float* w, int ldw, const float* bias, float* y, int ldy,
                   int m, int n, int k, int act, fcl_stream_t stream);

/* ---- H2/H4/H5/H11: Conv1d over channels-last rows (encoder_sa.py:136-140, decoder_sa.py:284-286,
 *      variance_predictor.py:86-88) ------------------------------------------------------------------ */
/* y[m, :] = act( sum_j x[m + j - (k-1)/2, :] . wp[j]^T + bias ) (+ residual), with rows outside
 * [seg_lo[m], seg_hi[m]) contributing zero (zero padding at utterance edges).  wp is the packed
 * [k, Cout, Cin] weight.  residual (optional, [M, Cout]) is added after the activation. */
int fcl_conv1d_fwd(const float* x, const float* wp, const float* bias, const int32_t* seg_lo, const int32_t* seg_hi,
                   const float* residual, float* y, int m, int cin, int cout, int k, int act, fcl_stream_t stream);

/* The same two on pre-split operands: xp = P32 planes of x (ldxp lines per row), wpp = P32 planes of the weight (Linear: [N, K]; Conv1d: the
 * packed taps as one [k*Cout, Cin] matrix, i.e. fcl_pack_planes of fcl_pack_conv1d_weight's output).  y (fp32) and / or yp (P32 planes, dense:
 * ceil(N/32) lines per row, zero past N) receive the result; residual / bias as above. */
int fcl_linear_planes_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const float* bias, float* y, int ldy, uint16_t* yp,
                          int m, int n, int k, int act, fcl_stream_t stream);
/* round 6 (421): fcl_linear_planes_fwd + masked MSE against `target` + its gradient in ONE launch (the GEMM's epilogue) -- one KD term of the student's update
 * (e2e_tts_tacotron2_sa_kd_student.py:134-179: MSE(proj(student tap), teacher tap) over the valid rows).  With d = x . W^T - target on the rows where row_valid is
 * set (all rows when NULL): grad [m, n] (ldg) / grad_p (P32 planes) = 2 d / count, 0 on the other rows; sums[0 .. 2] += sum |d|, sum d^2, element count (fp64
 * atomics: the three slots fcl_loss_terms_batch fills for a term).  The projection itself is never stored (8 bytes per element of HBM traffic less than
 * projection + fcl_loss_terms_batch).  n % 4 == 0, target 16-byte aligned with ld_t % 4 == 0; planes kernels only (FCL_ERR_INVALID under FCL_PLANES=0 /
 * FCL_PRECISION=0). */
int fcl_linear_planes_mse_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const float* target, int ld_t, const uint8_t* row_valid, double count,
                              float* grad, int ldg, uint16_t* grad_p, double* sums, int m, int n, int k, fcl_stream_t stream);
int fcl_conv1d_planes_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const float* bias, const int32_t* seg_lo, const int32_t* seg_hi,
                          const float* residual, float* y, uint16_t* yp, int m, int cin, int cout, int k, int act, fcl_stream_t stream);
/* fcl_conv1d_planes_fwd with a DEVICE row count (round 5): m is then the capacity of the buffers and tiles at or beyond *m_dev are not computed
 * (their output rows are left untouched).  The postnet of a capacity graph: the frame buffers carry slack (decode driver: x 1.3) and the batch's real
 * frame total lives in HBM (fcl_row_maps_t.totals[0]); nothing reads the rows beyond it (/root/reference/nets/modules/decoder_sa.py:199-263). */
int fcl_conv1d_planes_rows_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const float* bias, const int32_t* seg_lo, const int32_t* seg_hi,
                               const float* residual, float* y, uint16_t* yp, int m, int cin, int cout, int k, int act, const int32_t* m_dev,
                               fcl_stream_t stream);

/* G independent Conv1d's of the SAME shape in one launch (the duration / pitch / energy predictors' layers, variance_predictor.py:48-66): group g
 * reads planes xp + g * x_group_stride (uint16 elements; 0 = all groups read the same input), weights wpp [G][k * Cout][Cin planes] (each group
 * packed as for fcl_conv1d_planes_fwd), bias [G][Cout], and writes y / yp GROUP-MAJOR: [G][M][Cout].  Cin <= 384, Cout % 32 == 0, k >= 3. */
/* round 6 (422): Conv1d (no bias) on pre-split operands + the train-mode BatchNorm statistics of its output from the GEMM's epilogue (encoder / postnet blocks of the
 * training forward, encoder_sa.py:61-78, decoder_sa.py:199-263): z [m, cout] as fcl_conv1d_planes_fwd; mean, invstd and the running statistics exactly as
 * fcl_bn_stats_ws_fwd(z, ...) leaves them (fp64 column sums, the last row tile of a column tile finalises; `zero_workspace` = 2 cout doubles + ceil(cout / 64) tickets,
 * zero on entry and on exit) -- one launch and one pass over z less.  Planes kernels only (FCL_ERR_INVALID under FCL_PLANES=0 / FCL_PRECISION=0). */
int fcl_conv1d_planes_bn_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const int32_t* seg_lo, const int32_t* seg_hi, float* z, int m, int cin, int cout, int k,
                             float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var, double* zero_workspace,
                             fcl_stream_t stream);
int fcl_conv1d_planes_group_fwd(const uint16_t* xp, int ldxp, int64_t x_group_stride, const uint16_t* wpp, const float* bias, const int32_t* seg_lo,
                                const int32_t* seg_hi, float* y, uint16_t* yp, int m, int cin, int cout, int k, int act, int groups,
                                fcl_stream_t stream);

/* ---- H4/H5: channel LayerNorm (+ the predictor's Linear(C->1) and masked_fill) ---------------------- */
/* y[m,:] = LN(x[m,:]) * gamma + beta (y may be NULL).  If lin_w != NULL:
 * scalar[m] = pad_mask[m] ? 0 : (y[m,:] . lin_w + lin_b[0])   (variance_predictor.py:90-93).
 * keep != NULL (training): the Dropout after the LayerNorm, y *= keep[m,c] * keep_scale before it is stored / fed to the head. */
int fcl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, uint16_t* yp /* optional P32 planes of y */,
                      const float* lin_w, const float* lin_b, const uint8_t* pad_mask, const uint8_t* keep, float keep_scale,
                      float* scalar, int m, int c, fcl_stream_t stream);
/* G independent LayerNorms of the same shape in one launch: row mi of group g is read at x + g * x_group_stride + mi * ldx (so both a
 * [M][G * C] matrix -- ldx = G * C, stride C: the output of a Conv1d whose output channels are G predictors' stacked -- and a group-major
 * [G][M][C] one -- ldx = C, stride M * C -- work); gamma / beta / lin_w are [G][C], lin_b [G], pad_mask [M] (shared); y / yp / scalar are written
 * group-major ([G][M][C], [G][M]). */
int fcl_layernorm_group_fwd(const float* x, int ldx, int64_t x_group_stride, const float* gamma, const float* beta, float eps, float* y, uint16_t* yp,
                            const float* lin_w, const float* lin_b, const uint8_t* pad_mask, float* scalar, int m, int c, int groups,
                            fcl_stream_t stream);

/* ---- H4: DurationPredictor.inference rounding (ESPnet; call site ..._kd_student.py:825) ------------- */
/* out = pad_mask ? 0 : (int64) max(rint(linear_domain ? x : exp(x) - offset), 0); rint = half-to-even. */
int fcl_duration_round_fwd(const float* x, int64_t* out, int n, int linear_domain, float offset,
                           const uint8_t* pad_mask, fcl_stream_t stream);

/* ---- H5: pitch_embed + energy_embed + the decoder's `hs + p_embs + e_embs`
 *      (..._kd_student.py:837-838, decoder_sa_kd.py:734) ---------------------------------------------- */
/* out[m,c] = hs[m,c] + bp[c] + sum_j wp[c,j]*p[m+j-k/2] + be[c] + sum_j we[c,j]*e[m+j-k/2]; p_emb/e_emb
 * (optional) receive the two embeddings separately. */
int fcl_variance_embed_add_fwd(const float* hs, const float* p, const float* e, const float* wp, const float* bp,
                               const float* we, const float* be, const int32_t* seg_lo, const int32_t* seg_hi,
                               float* out, float* p_emb, float* e_emb, int m, int c, int k, fcl_stream_t stream);

/* ---- speaker embedding (..._sa.py:555-557 forward, :636-638 inference; `--spk-embed-dim`): out [M, C + S] = cat[hs, F.normalize(spemb)] over the
 *      padded [B, T] row layout (row m belongs to utterance m / t; spk [B, S]; F.normalize: x / max(||x||_2, 1e-12)).  out_p (optional, (C + S) % 32
 *      == 0): the same rows as P32 planes; out may then be NULL. */
int fcl_concat_spk_fwd(const float* hs, int ldh, const float* spk, float* out, uint16_t* out_p, int m, int c, int s, int t, fcl_stream_t stream);

/* ---- H10: position table (..._kd_student.py:845-851): pos[n, t] = t < dur[n] ? (float)t/(float)dur[n] : 0 */
int fcl_position_table_fwd(const int32_t* dur, float* pos, int n, int lmax, fcl_stream_t stream);

/* ---- H9: row gather (decoder_sa.py:467: hs[non_zero_lens_mask.eq(1)], plus the duration sort) -------- */
/* dst_p (optional): the gathered rows as P32 planes (ceil(c/32) lines per row); dst may then be NULL. */
int fcl_gather_rows_fwd(const float* src, const int32_t* idx, float* dst, uint16_t* dst_p, int n, int c, fcl_stream_t stream);

/* ---- H3: Encoder.blstm over packed sequences (encoder_sa.py:98-100,143-146) -------------------------- */
/* x [B*T, C]; lens [B] int32 (device); w_ih_* [4H, C], w_hh_* [4H, H], b_* [4H] (= bias_ih + bias_hh);
 * out [B*T, 2H] = fwd | bwd, zero past each length.  algo: 0 auto, 1 per-step launches, 2 persistent
 * register-resident recurrence (H in {8,16,32,64,128}), 3 (H = 256) four cooperating workgroups per (utterance, direction)
 * exchanging h through global memory — fastest on an idle GPU, not the default when other streams are in flight.
 * algo 3 is SINGLE-STREAM ONLY (two such kernels in flight can starve each other's groups) and needs `status`, a device status word
 * (FCL_STATUS_*): a group that times out reports there and its outputs are partial.  It is refused (per-step launches instead) when
 * 8*B workgroups do not fit the device one per CU.  status may be NULL for the other algorithms.
 * Optional P32 planes: out_p receives out as planes (for the predictor convolutions); x_p / w_ih_f_p / w_ih_r_p (all or none) are the
 * pre-split operands of the input projection, in which case x may be NULL.
 * row_maps (optional; fcl_row_maps_t is declared with fcl_row_maps_build below): also build the batch's row / frame maps, exactly as
 * fcl_row_maps_build(row_maps, stream) would.  Forced durations are known before the encoder runs, so with the persistent recurrence at H = 128
 * (the shipped encoders) the map build is one more workgroup of the recurrence's launch and leaves the pass's dependent chain; every other path
 * issues fcl_row_maps_build's launches after the recurrence.  Same contract and status bits either way. */
struct fcl_row_maps;
size_t fcl_bilstm_workspace_bytes(int b, int t, int h);
int fcl_bilstm_fwd(const float* x, const int32_t* lens, const float* w_ih_f, const float* w_hh_f, const float* b_f,
                   const float* w_ih_r, const float* w_hh_r, const float* b_r, float* out, uint16_t* out_p, const uint16_t* x_p,
                   const uint16_t* w_ih_f_p, const uint16_t* w_ih_r_p, int b, int t, int c, int h,
                   int algo, void* workspace, size_t workspace_bytes, uint32_t* status, const struct fcl_row_maps* row_maps, fcl_stream_t stream);

/* ---- H7 as a single step: one LSTMCell (+ zoneout) update of M rows — the building block of the decoder loop, of the per-step
 *      BiLSTM, and of the TRAINING forward, which also saves what the backward pass needs (decoder_sa.py:63-96, 500-504) ------ */
typedef struct {
    const float* A;  /* [M, K] activations (lda) */
    const float* W;  /* [4U (or N), K] weights, torch layout (ldw) */
    int lda, ldw, K; /* multiples of 4 */
    int shift;       /* conv tap row shift (GEMM terms only; 0 here) */
    const uint16_t* Whi; /* optional fragment-major bf16x3 planes of W (fcl_pack_frag_bf16) */
    const uint16_t* Wlo;
    const uint16_t* Ap;  /* optional P32 planes of A and of W (fcl_pack_planes layout; see "bf16x3 operand planes" above): when EVERY term of a */
    const uint16_t* Wp;  /* call carries both, the contraction runs on the LDS-DMA kernels of gemm_planes.hip and A / W may be NULL */
    int lda_p, ldw_p;    /* row strides of the planes in 128-byte lines (>= ceil(K / 32)) */
    int64_t a_chunk_stride; /* 0: Ap is row-major as above.  != 0: CHUNK-MAJOR planes of A -- line (row m, 32-column chunk c) at byte offset
                             * c * a_chunk_stride + m * 128 (lda_p unused): a tile's rows of one chunk are one contiguous stream, which is what
                             * the vocoder's sample-major activations use (DRAM-friendly: [chunk][row] instead of [row][chunk]) */
    const float* Wff;       /* optional (round 4, exact-fp32 mode): fcl_pack_frag_f32 of W -- small LSTM steps (K = 256 per term) keep it in registers */
} fcl_gemm_term_t;

typedef struct {
    fcl_gemm_term_t term[3]; /* gates[m, g*U+u] = sum_t A_t[m,:] . W_t[g*U+u,:] + ... */
    int nterms;
    int M, U;
    const float* G;          /* optional pre-activation init [*, 4U]; row = m*g_row_mul + g_row_add */
    long long g_row_mul, g_row_add;
    const float* bias;       /* optional [4U] */
    const float* rank1_w;    /* optional [4U]: + (step / dur[m]) * rank1_w  (the decoder's position input) */
    const int32_t* dur;
    int step;
    const float* h_in;       /* [M, U] previous hidden state */
    float* h_out;            /* [M, U], must not alias h_in */
    float* c;                /* [M, U] cell state, in place */
    float zoneout;           /* expectation-form rate (0 = plain LSTMCell) */
    const uint8_t* zone_keep_h; /* optional sampled zoneout masks [M, U] (1 keeps the OLD state), in pairs */
    const uint8_t* zone_keep_c;
    const int32_t* row_len;  /* optional: row live iff step < row_len[m] (packed-sequence semantics) */
    float* out2;             /* optional copy of h: row = (out2_row_base ? out2_row_base[m] : m*out2_row_mul) + out2_row_add */
    const int32_t* out2_row_base;
    long long out2_row_mul, out2_row_add;
    int ld2, out2_col_off;
    uint16_t* h_out_p;       /* optional P32 planes of h_out (ld_hp lines per row): the next step's GEMM operand, written by the same epilogue */
    int ld_hp;
    float* save_gates;       /* optional, training: activated gates i,f,g,o [M, 4U] */
    float* save_c_new;       /* optional: raw new cell (before zoneout) [M, U] */
    float* save_c_old;       /* optional: incoming cell / hidden state [M, U] */
    float* save_h_old;
    const int32_t* m_dev;    /* optional DEVICE word: rows live in this step; the kernel processes min(M, *m_dev) rows (M then bounds the grid).
                              * Lets a decoder loop whose durations were computed on the device run without a host round trip (fcl_decoder_io_t.live_rows) */
} fcl_lstm_step_t;

int fcl_lstm_step_fwd(const fcl_lstm_step_t* args, fcl_stream_t stream);

/* ---- H6-H8 (+H9/H10 scatter): the per-phoneme-parallel decoder loop
 *      (Decoder.inference decoder_sa_kd.py:742-790; Decoder.forward :572-655) ------------------------- */
typedef struct {
    int c;     /* att_c width (= eunits) */
    int p;     /* prenet units */
    int u;     /* dunits */
    int odim;  /* mel bins */
    const float* prenet_w0; /* [P, odim] */
    const float* prenet_b0; /* [P] */
    const float* prenet_w1; /* [P, P] */
    const float* prenet_b1; /* [P] */
    const float* w0_att;    /* [4U, C]  lstm.0.cell.weight_ih[:, :C]      */
    const float* w0_pre;    /* [4U, P]  lstm.0.cell.weight_ih[:, C:C+P]   */
    const float* w0_pos;    /* [4U]     lstm.0.cell.weight_ih[:, C+P]     */
    const float* w0_hh;     /* [4U, U] */
    const float* b0;        /* [4U]     bias_ih + bias_hh */
    const float* w1_ih;     /* [4U, U] */
    const float* w1_hh;     /* [4U, U] */
    const float* b1;        /* [4U] */
    const float* wf_h;      /* [odim, U] feat_out.weight[:, :U] */
    const float* wf_att;    /* [odim, C] feat_out.weight[:, U:] */
    float zoneout_rate;     /* eval-form zoneout (decoder_sa.py:96) */
    float prenet_dropout;   /* always-on prenet dropout rate (decoder_sa.py:156-158) */
    /* optional fragment-major bf16x3 planes (fcl_pack_frag_bf16) of the matrices above; all NULL = exact-fp32 small tiles */
    const uint16_t *prenet_w0_hi, *prenet_w0_lo, *prenet_w1_hi, *prenet_w1_lo;
    const uint16_t *w0_pre_hi, *w0_pre_lo, *w0_hh_hi, *w0_hh_lo, *w1_ih_hi, *w1_ih_lo, *w1_hh_hi, *w1_hh_lo;
    const uint16_t *wf_h_hi, *wf_h_lo;
    /* optional P32 planes (fcl_pack_planes) of the GEMM-sized matrices: with all six set (and C, P, U multiples of 32) the hoists and the
     * LSTM steps with more than ~1000 live rows run on the LDS-DMA kernels, states and prenet outputs travelling between them pre-split */
    const uint16_t *w0_att_p, *wf_att_p, *w0_pre_p, *w0_hh_p, *w1_ih_p, *w1_hh_p;
    int out_act;            /* FCL_ACT_*: `output_activation_fn` on the frame fed back to the prenet in the free-running loop
                             * (decoder_sa.py:614-617; `before` keeps the raw feat_out values, the caller activates the final output :635-636) */
    const float *wf_h_ff, *prenet_w0_ff, *prenet_w1_ff; /* optional (round 4, exact-fp32 mode): fcl_pack_frag_f32 of wf_h / prenet_w0 / prenet_w1: with all
                             * three (and U = P = 256, 64 < odim <= 96) the fused feat/prenet launch keeps its weights in registers as the bf16x3 one does */
    const float *w0_pre_ff, *w0_hh_ff, *w1_ih_ff, *w1_hh_ff; /* ... and of the four LSTM matrices: the small steps' register-resident operands in that mode */
    const uint16_t* stream; /* optional (round 4): the step's weights in the consumption order of the persistent row-tile kernel
                             * (fcl_decoder_stream_pack; fcl_decoder_stream_bytes() bytes, 0 = shape not covered).  With it, the P32 planes above and
                             * a free-running loop of >= FCL_DEC_TILE_MIN_ROWS rows (no teacher forcing, taps or injected masks), fcl_decoder_loop_fwd
                             * runs the whole loop as ONE launch: 32 rows per workgroup, states resident in LDS / registers for all their steps */
    /* ---- structure options beyond the shipped recipes (round 5; /root/reference/nets/modules/decoder_sa.py:119-158, 357-369, 500-504) ----
     * prenet_layers: 0 or 2 = the two blocks above; 1 = prenet_w0 / prenet_b0 alone; 3 = a third block prenet_w2 [P, P] / prenet_b2 [P].
     * dlayers: 0 or 2 = the two cells above; 1 = cell 0 alone (w1_* / b1 unused, feat_out reads cell 0); 3 = a third cell w2_ih [4U, U],
     * w2_hh [4U, U], b2 [4U] on cell 1's new state (feat_out reads it).  Anything but (2, 2) runs the loop launch by launch on the fp32
     * operands (one GEMM per prenet block, one LSTM-step launch per cell, feat_out as a GEMM): host row counts only (fcl_decoder_io_t.live_rows
     * and tail_from are refused), prenet_keep is then [Lmax, prenet_layers, N, P], tap_lstm1 the LAST cell's state. */
    int prenet_layers;
    int dlayers;
    const float *prenet_w2, *prenet_b2, *w2_ih, *w2_hh, *b2;
    /* reduction_factor r (0 / 1 = one frame per step; decoder_sa.py:397-398, 512-516, 611-627): a step emits r frames.  wf_h / wf_att are then
     * [r * odim, .] with the rows PERMUTED to frame-major order (row j * odim + o = feat_out.weight row o * r + j), so a step's output is r
     * consecutive rows of the frame-major `before`; the last of them is the next step's prenet input.  `dur` / live rows count STEPS, frame_off
     * and the frame buffers count FRAMES (r per step).  Runs on the launch-by-launch loop (see dlayers). */
    int reduction_factor;
} fcl_decoder_weights_t;

typedef struct {
    int n;                      /* phoneme rows, SORTED by duration descending */
    int lmax;                   /* = dur[0] */
    const float* att_c;         /* [N, C]: hs + p_embs + e_embs, compacted + sorted rows */
    const int32_t* dur;         /* [N] device, > 0 */
    const int32_t* live_rows_host; /* [Lmax] HOST: rows with dur > t (non-increasing).  Exact when live_rows is NULL; otherwise an UPPER BOUND per step,
                                    * used for grid sizes and kernel selection only (never for correctness: see live_rows / status) */
    const int32_t* frame_off;   /* [N] device: output frame index of (row, t = 0) */
    const float* teacher_ys;    /* NULL = free running; else [N, Lmax, odim] teacher-forced inputs */
    int dropout_mode;           /* FCL_DROP_* for the prenet */
    const uint8_t* prenet_keep; /* FCL_DROP_MASK: [Lmax, 2, N, P] keep masks (sorted row order) */
    uint32_t seed;              /* FCL_DROP_RNG */
    const uint32_t* seed_dev;   /* optional device word added to `seed` at run time (lets a captured hipGraph
                                   draw fresh dropout masks on every replay; bump it with fcl_u32_add) */
    float* before;              /* [F, odim] frame-major decoder output (pre-postnet), F = sum(dur) */
    float* tap_prenet;          /* optional [F, P]  (KD taps, decoder_sa_kd.py:627-637) */
    float* tap_lstm0;           /* optional [F, U] */
    float* tap_lstm1;           /* optional [F, U] */
    void* workspace;
    size_t workspace_bytes;
    const uint16_t* att_c_p;    /* optional P32 planes of att_c (C/32 lines per row; fcl_gather_rows_fwd writes them); att_c may then be NULL */
    uint16_t* before_p;         /* optional out: P32 planes of `before` (ceil(odim/32) lines per row), the postnet's pre-split operand */
    const int32_t* live_rows;   /* optional [Lmax + 1] DEVICE: rows with dur > t as fcl_row_maps_build wrote them (entry Lmax and beyond: 0).  With it the
                                 * loop needs nothing from the host about the durations: every step kernel processes min(live_rows_host[t], live_rows[t])
                                 * rows, `lmax` is the number of steps LAUNCHED (>= the true maximum, else FCL_STATUS_LMAX_CAP was raised by the map
                                 * builder), and a step with more live rows than its bound raises FCL_STATUS_ROWS_CAP in `status`.  NULL: host counts are exact */
    uint32_t* status;           /* device status word (FCL_STATUS_*); required with live_rows */
    int tail_from;              /* 0 = off.  Steps tail_from .. lmax - 1 (0 < tail_from < lmax) of the rows still live then run as ONE launch of the persistent
                                 * row-tile kernel (csrc/decoder_tile.hip; needs fcl_decoder_weights_t.stream and a free-running loop without taps / injected
                                 * masks, ignored otherwise), continuing from the loop's own states: in a capacity graph (lmax = a bound with slack) the steps
                                 * beyond the durations actually seen cost one launch in all instead of three each; rows that do reach them pay ~2x per step */
} fcl_decoder_io_t;

size_t fcl_decoder_loop_workspace_bytes(const fcl_decoder_weights_t* w, int n);
/* the weight stream of the persistent row-tile decoder kernel (csrc/decoder_tile.hip): bytes (0: this U / P / odim is not covered: U = P = 256,
 * odim <= 128) and the packing of the fp32 matrices of `w` (hi | lo bf16 lines, tile-major in consumption order, swizzled for the fragment reads) */
size_t fcl_decoder_stream_bytes(const fcl_decoder_weights_t* w);
int fcl_decoder_stream_pack(const fcl_decoder_weights_t* w, void* out, size_t out_bytes, fcl_stream_t stream);
int fcl_decoder_loop_fwd(const fcl_decoder_weights_t* w, const fcl_decoder_io_t* io, fcl_stream_t stream);

/* ---- H10 on the device: the integer row / frame maps of a batch from durations that live in HBM (predicted by fcl_duration_round_fwd, or
 *      forced and uploaded), with no host round trip.  Replaces the reference's host bookkeeping — `ds_nonzeros` filter + assert
 *      (decoder_sa_kd.py:736-739), the per-phoneme position / trim / concat loops (..._kd_student.py:845-851, decoder_sa_kd.py:781-791) — and this
 *      build's own numpy version (engine.build_row_maps), to which it is bit-identical (tests/test_gpu_rowmaps.py).
 *      Rows are the non-padded (utterance, phoneme) pairs in row-major order (N = sum of phoneme counts, known to the host), or the whole padded
 *      [B, t_max] layout (row_src == NULL).
 *        dur_sorted[r], src_rows[r], frame_off[r] : rows STABLY sorted by duration descending; padded row index b*T + t; first output frame
 *                                                   (exclusive cumsum of the durations in compact order = utterance base + offset inside it)
 *        live_rows[t], t = 0 .. lmax_cap          : rows with duration > t
 *        utt_frame0[b], b = 0 .. B                : first frame of utterance b; [B] = total frames
 *        frame_lo / frame_hi [frames_cap]         : utterance frame range of every frame row (0, 0 beyond the total): the postnet's segment bounds
 *        totals[0] = total frames, totals[1] = max duration, totals[2] = zero-duration rows
 *      Violations are reported in *status (FCL_STATUS_ZERO_DURATION / LMAX_CAP / FRAMES_CAP) and live_rows is then zeroed, so a decoder loop
 *      driven by these maps does nothing instead of writing out of bounds.  Two launches (none of its own through fcl_bilstm_fwd's row_maps). */
typedef struct fcl_row_maps {
    int b, n;                   /* utterances; rows of the row universe (see row_src) */
    int lmax_cap, frames_cap;
    int t_max;                  /* with utt_row0 == NULL: utterance b owns rows [b * t_max, (b + 1) * t_max) */
    const int32_t* row_src;     /* [N] device: padded row index b*T + t of compact row i (a function of the phoneme counts alone), or NULL: the row
                                 * universe IS the padded [B, t_max] layout (n = B * t_max); padding rows carry duration 0, sort behind every real
                                 * row and add nothing to the prefix sums, so the real rows get exactly the maps of the compact form -- this is the
                                 * form whose every size is a capacity, i.e. the one a captured hipGraph can replay for any batch that fits */
    const int32_t* utt_row0;    /* [B + 1] device: row range of every utterance, or NULL (t_max) */
    const uint8_t* pad;         /* optional [N]: 1 = padding row (a duration of 0 there is not an error) */
    const int64_t* dur_i64;     /* durations as int64 (fcl_duration_round_fwd's output), row i at dur[row_src ? row_src[i] : i]; or NULL */
    const int32_t* dur_i32;     /* durations as int32 per ROW OF THE UNIVERSE (forced durations); exactly one of the two is set */
    int32_t* src_rows;          /* [N] */
    int32_t* dur_sorted;        /* [N] */
    int32_t* frame_off;         /* [N] */
    int32_t* order;             /* optional [N]: compact index of sorted row r (mask re-ordering, tests) */
    int32_t* live_rows;         /* [lmax_cap + 1] */
    int32_t* utt_frame0;        /* [B + 1] */
    int32_t* frame_lo;          /* [frames_cap] */
    int32_t* frame_hi;          /* [frames_cap] */
    int32_t* totals;            /* [4] */
    uint32_t* status;           /* device status word, OR-ed */
    int32_t* scratch;           /* optional [2 N] workspace: with it and lmax_cap <= 254 the maps are built by a one-workgroup COUNTING sort (O(N):
                                 * wave-ballot ranks per duration value, block scans), a few microseconds; without it by the O(N^2 / threads) form */
} fcl_row_maps_t;
int fcl_row_maps_build(const fcl_row_maps_t* maps, fcl_stream_t stream);

/* ---- H12: masked L1 / MSE loss sums (Tacotron2Loss ..._sa.py:26-82, Knowledge_loss ..._kd_student.py:134-179,
 *      DurationPredictorLoss) ------------------------------------------------------------------------ */
/* Over the rows with row_valid[m] != 0 (NULL = every row) and all c columns:
 *   out[0] += sum |a - b'|, out[1] += sum (a - b')^2, out[2] += number of elements,
 * with b' = b, or log(b + b_log_offset) when b_log != 0 (the duration target).  out: 3 doubles in device memory,
 * accumulated with atomics (the caller zeroes them); the masked means are out[0]/out[2] and out[1]/out[2]. */
int fcl_masked_l1_mse_fwd(const float* a, int lda, const float* b, int ldb, const uint8_t* row_valid, int m, int c,
                          int b_log, float b_log_offset, double* out, fcl_stream_t stream);

/* ---- H13: gradient primitives of the teacher-forced training step (tts.py:137-179; losses ..._sa.py:601-613).
 *      dX of every Linear / Conv1d is the forward GEMM on transposed weights (fcl_transpose2d + fcl_linear_fwd / fcl_conv1d_fwd
 *      with the taps reversed); the entries below are what the backward pass needs beyond that. ------------------------------- */
/* dW: c[n, k] += sum_m a[m, n] * b[m + shift, k]  (rows of b outside [seg_lo[m], seg_hi[m]) — or outside [0, M) when the
 * bounds are NULL — contribute zero).  Linear: shift 0; Conv1d tap j: shift = j - (k-1)/2, c = packed dW[j] ([Cout, Cin]).
 * Accumulates with fp32 atomics (the caller zeroes c once per step: gradient accumulation is the natural mode).
 * Arithmetic follows FCL_PRECISION like the forward GEMMs: bf16x3-split operands with fp32 accumulation (default) or exact fp32 MFMA (0). */
int fcl_gemm_tn_fwd(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, int shift,
                    const int32_t* seg_lo, const int32_t* seg_hi, fcl_stream_t stream);
/* The same for NTAPS consecutive shifts in one launch: c + j*c_tap_stride gets shift0 + j (all taps of a Conv1d weight gradient, tap-major). */
int fcl_gemm_tn_taps_fwd(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int m, int n, int k, int shift0, int ntaps,
                         size_t c_tap_stride, const int32_t* seg_lo, const int32_t* seg_hi, fcl_stream_t stream);
/* The same contraction on pre-split operands: the LDS-DMA GEMM kernels (fcl_linear_planes_fwd's) with the roles of rows and columns exchanged.
 * fcl_pack_planes_t writes TRANSPOSED planes (P32T) of x [rows, cols]: one plane row per column, ceil(rows/32) lines of 32 consecutive rows
 * (hi | lo, zero past `rows`); with ntaps > 1 plane row t*cols + c holds column c read at row m + shift0 + t (zero outside
 * [seg_lo[m], seg_hi[m]), or outside [0, rows) when the bounds are NULL): the shifted inputs of all taps of a Conv1d weight gradient.
 * out: ntaps * cols * ceil(rows/32) * 64 uint16, 128-byte aligned.
 * fcl_gemm_tn_planes: c[n_, k_] += sum_m a[m, n_] * b[m, k_] with ap_t = P32T of a [m, n] and bp_t = P32T of b [m, k] (k = ntaps * cols for
 * a tap-stacked b).  The contraction is split over workgroup slices and accumulated with fp32 atomics (the caller zeroes c once per update).
 * nblk > 0: output column k_ lands at c + (k_ / nblk) * blk_stride + n_ * ldc + k_ % nblk (nblk = Cin, blk_stride = Cout * Cin: tap-major
 * [ntaps, Cout, Cin] like fcl_gemm_tn_taps_fwd); nblk = 0: plain [n, k] with row stride ldc.  Honours fcl_set_gemm_mode. */
int fcl_pack_planes_t(const float* x, int ld, int rows, int cols, int ntaps, int shift0, const int32_t* seg_lo, const int32_t* seg_hi, uint16_t* out,
                      fcl_stream_t stream);
int fcl_gemm_tn_planes(const uint16_t* ap_t, const uint16_t* bp_t, float* c, int ldc, int m, int n, int k, int nblk, size_t blk_stride,
                       fcl_stream_t stream);
/* out[c] += sum_m x[m,c] (mode 0) | x*y (mode 1) | x*(y - b[c])/g[c] (mode 2: gamma gradient of a folded eval BatchNorm)
 * | x*(y - b[c])*g[c] (mode 3: b = batch mean, g = invstd: gamma gradient of a train-mode BatchNorm). */
int fcl_colsum_fwd(const float* x, const float* y, const float* g, const float* b, float* out, int m, int c, int mode, fcl_stream_t stream);
/* the same pass with a second output: out_x[c] += sum_m x[m,c] (optional) -- train-mode BatchNorm's dbeta beside its dgamma (mode 3), dy read once */
int fcl_colsum2_fwd(const float* x, const float* y, const float* g, const float* b, float* out, float* out_x, int m, int c, int mode, fcl_stream_t stream);
/* Weight / bias gradients of a Conv1d with ONE input channel and odd k <= 16 (the pitch / energy embeddings, ..._sa.py:435-443,
 * ..._kd_student.py:570-596): dw[c, j] += sum_m dy[m, c] * x[m + j - (k-1)/2] over the positions inside row m's utterance [seg_lo[m], seg_hi[m])
 * (null: [0, m)), db[c] += sum_m dy[m, c] (db may be null).  dw is the [C, 1, k] weight gradient, contiguous. */
int fcl_conv1d_in1_dw(const float* dy, int ldy, const float* x, const int32_t* seg_lo, const int32_t* seg_hi, float* dw, float* db, int m, int c, int k,
                      fcl_stream_t stream);
/* dst[r, 0:cols] += alpha * src[r, 0:cols] on rows with row_valid[r] != 0 (null: every row).  Strided on both sides: accumulates a
 * gradient block into a column range of weight_ih / feat_out.weight, adds residual-path gradients, masks padded rows. */
int fcl_add2d(float* dst, int ld_dst, const float* src, int ld_src, int rows, int cols, float alpha, const uint8_t* row_valid, fcl_stream_t stream);
/* y = act(x) [* keep * keep_scale]   (elementwise; the training forward keeps pre-activations for the BatchNorm gradients).
 * yp (optional): the result as P32 planes of the [n / cols, cols] matrix (cols % 32 == 0); y may then be NULL. */
int fcl_act_fwd(const float* x, const uint8_t* keep, float keep_scale, float* y, uint16_t* yp, int cols, size_t n, int act, fcl_stream_t stream);
/* dw[co, ci, j] += dwp[j, co, ci] * (scale ? scale[co] : 1): packed tap-major conv gradient back to the torch Conv1d layout. */
int fcl_unpack_conv1d_grad(const float* dwp, const float* scale, float* dw, int cout, int cin, int k, fcl_stream_t stream);
/* dz = dy * act'(y) [* keep * keep_scale]   (y = the activation's OUTPUT before dropout; act = FCL_ACT_*). */
int fcl_act_bwd(const float* dy, const float* y, const uint8_t* keep, float keep_scale, float* dz, uint16_t* dzp /* optional P32 planes of dz */,
                int cols /* row width for the planes, % 32 == 0 */, size_t n, int act, fcl_stream_t stream);
/* da (+)= (w_l1 * sign(a - b') + 2 * w_mse * (a - b')) / count on the valid rows, 0 elsewhere  (b' as in fcl_masked_l1_mse_fwd). */
int fcl_l1_mse_grad(const float* a, const float* b, const uint8_t* row_valid, int m, int c, int b_log, float b_log_offset, float w_l1,
                    float w_mse, double count, float* da, int accumulate, fcl_stream_t stream);
/* fcl_masked_l1_mse_fwd (sums[0:3] += sum|d|, sum d^2, count) and fcl_l1_mse_grad in ONE pass over a and b (C % 4 == 0).  da_planes (optional,
 * C % 32 == 0): the gradient also as P32 planes [m][C/32][2][32] for the input-gradient GEMM that consumes it (the KD projections). */
int fcl_l1_mse_loss_grad(const float* a, const float* b, const uint8_t* row_valid, int m, int c, int b_log, float b_log_offset, float w_l1,
                         float w_mse, double count, float* da, int accumulate, double* sums, uint16_t* da_planes, fcl_stream_t stream);
/* ---- round 6: the small-launch tail of the training update (csrc/fused_small.hip) ------------------------------------------------------------------
 * Every ELEMENT-WISE loss term of a step in one launch (..._kd_student.py:759-802, ..._sa.py:60-70,122-126): term k computes, over the rows of a [m, c]
 * with valid[r] != 0 (all rows when NULL),
 *     sums[0:3]  += sum |a - b'|, sum (a - b')^2, count          b' = b_log ? log(b + b_log_offset) : b          (fcl_masked_l1_mse_fwd)
 *     da          = (w_l1 sign(a - b') + 2 w_mse (a - b')) / count                                                (fcl_l1_mse_grad)
 * and, when b2 is given (the teacher's output for the same tensor: the output-KD and prosody-KD terms), the same against b2 with valid2 / w_*_2 /
 * count2 into sums2, its gradient ADDED to da -- what a second, accumulating fcl_l1_mse_loss_grad launch did.  da is written (never accumulated
 * into); da_planes (optional, c % 32 == 0): the gradient as P32 planes too.  `terms` is a HOST array (passed to the kernel by value). */
#define FCL_LOSS_MAX_TERMS 12
typedef struct {
    const float* a;
    const float* b;
    const float* b2;         /* optional second target */
    const uint8_t* valid;    /* [m] or NULL */
    const uint8_t* valid2;   /* [m] or NULL (second target) */
    float* da;               /* [m, c] */
    uint16_t* da_planes;     /* optional */
    double* sums;            /* [3] */
    double* sums2;           /* [3], required with b2 */
    int32_t m, c;
    int32_t b_log;
    float b_log_offset;
    float w_l1, w_mse;
    double count;
    float w_l1_2, w_mse_2;
    double count2;
} fcl_loss_term_t;
int fcl_loss_terms_batch(const fcl_loss_term_t* terms, int n_terms, fcl_stream_t stream);
/* dst[r, :] = row_valid[r] ? srcs[0][r, :] + ... + srcs[n_src - 1][r, :] : 0 over dense [rows, cols] matrices (cols % 4 == 0); dst_p (optional, cols % 32
 * == 0): the result as P32 planes; dst may be NULL then, and may alias a source.  `srcs` is a HOST array of n_src <= FCL_SUM_ROWS_MAX device pointers.
 * Replaces the chains of fcl_add2d that assemble a gradient from its sources (pad_packed_sequence's zero rows: row_valid). */
#define FCL_SUM_ROWS_MAX 6
int fcl_sum_rows(const float* const* srcs, int n_src, const uint8_t* row_valid, float* dst, uint16_t* dst_p, int rows, int cols, fcl_stream_t stream);
/* Train-mode BatchNorm backward, first pass, with the activation / dropout backward as its prologue: dz = (dy [+ dy2]) [* keep * keep_scale] * act'(y_act)
 * is written, and dbeta[c] += sum_m dz, dgamma[c] += sum_m dz * (z - mean[c]) * invstd[c] (fp64 accumulation) -- fcl_act_bwd + fcl_colsum2_fwd(mode 3)
 * in one pass; fcl_bn_bwd consumes dz and the two sums. */
int fcl_bn_bwd_sums(const float* dy, const float* dy2, const float* y_act, const uint8_t* keep, float keep_scale, int act, const float* z, const float* mean,
                    const float* invstd, float* dz, float* dgamma, float* dbeta, int m, int c, fcl_stream_t stream);
/* fcl_act_bwd of dy + dy2 (a gradient that arrives from two consumers); dz may be NULL when only the planes are wanted. */
int fcl_act_bwd_sum(const float* dy, const float* dy2, const float* y, const uint8_t* keep, float keep_scale, float* dz, uint16_t* dzp, int cols, size_t n, int act,
                    fcl_stream_t stream);
/* fcl_gather_rows_fwd of src + src2 + src3 (src2 / src3 optional; c % 4 == 0): dst[i, :] = sum_k src_k[idx[i], :], zero rows for idx[i] < 0. */
int fcl_gather_rows_sum_fwd(const float* src, const float* src2, const float* src3, const int32_t* idx, float* dst, uint16_t* dst_p, int n, int c,
                            fcl_stream_t stream);
/* y = act(x . w^T [+ x2 . w2^T] + bias) [+ residual]: nn.Linear on fp32 operands with an optional second (x2 [M, k2], w2 [N, k2]) pair contracted into the
 * same output and an optional residual [M, N] (row stride ldr) -- the input gradients that reach one tensor through two weights, and a gradient
 * injected at the GEMM's output, without a separate addition pass. */
int fcl_linear2_fwd(const float* x, int lda, const float* w, int ldw, int k, const float* x2, int lda2, const float* w2, int ldw2, int k2, const float* bias,
                    const float* residual, int ldr, float* y, int ldy, int m, int n, int act, fcl_stream_t stream);

/* Channel LayerNorm backward (+ the predictor's scalar head: ds = gradient of scalar[m]).  dgamma/dbeta/dlin_w/dlin_b accumulate. */
int fcl_layernorm_bwd(const float* x, const float* gamma, const float* beta, float eps, const float* dy, const float* lin_w, const float* ds,
                      const uint8_t* pad_mask, const uint8_t* keep, float keep_scale, float* dx, float* dgamma, float* dbeta, float* dlin_w,
                      float* dlin_b, int m, int c, fcl_stream_t stream);
/* Train-mode BatchNorm1d over the rows of z [M, C] (encoder_sa.py:61-78, decoder_sa.py:199-263 under model.train(): statistics over every
 * position of the padded batch).  stats: mean / 1/sqrt(biased var + eps) accumulated in fp64, running statistics updated as torch (momentum 0.1,
 * unbiased variance) when given; ONE kernel (the last workgroup of a column group to finish finalises it).  workspace: 2*C + (C+63)/64 doubles;
 * fcl_bn_stats_fwd clears it first (any content on entry), fcl_bn_stats_ws_fwd requires it ZERO on entry and leaves it zero on return (a
 * workspace allocated once per stream serves every call: no clearing launch).  act: y_act = act(gamma*zhat + beta), y_drop = y_act*keep*keep_scale.
 * bwd: dz = gamma*invstd*(dy - dbeta/M - zhat*dgamma/M) with THIS batch's dbeta = sum dy, dgamma = sum dy*zhat (fcl_colsum_fwd modes 0 / 3). */
int fcl_bn_stats_fwd(const float* z, int m, int c, float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                     double* workspace, fcl_stream_t stream);
int fcl_bn_stats_ws_fwd(const float* z, int m, int c, float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                     double* zero_workspace, fcl_stream_t stream);
/* (round 6) y_act may be NULL when y_drop / yp receive the result (a forward that keeps nothing for a backward). */
int fcl_bn_act_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, const uint8_t* keep, float keep_scale,
                   float* y_act, float* y_drop, uint16_t* yp /* optional P32 planes of the block output (after dropout), C % 32 == 0 */, int m, int c, int act,
                   fcl_stream_t stream);
int fcl_bn_bwd(const float* dy, const float* z, const float* mean, const float* invstd, const float* gamma, const float* dbeta, const float* dgamma,
               float* dz, uint16_t* dzp /* optional P32 planes of dz, C % 32 == 0 */, int m, int c,
               float* acc_dbeta, float* acc_dgamma /* optional pair: acc += this batch's dbeta / dgamma (the parameters' gradient accumulators) */,
               fcl_stream_t stream);
/* ---- the training step's time loops, enqueued by ONE call each (H13; decoder_sa.py:472-515, encoder_sa.py:143-146) ----------------------------
 * Cells are step-major: rows sorted by duration descending, cell (t, m) at offset(t) + m with offset(t) = sum_{t' < t} live_rows[t'].
 * s0 / s1 (decoder layers) and s (BiLSTM direction) are the saved tensors {gates [.,4U], c_new, c_old, h_old [.,U]} fcl_lstm_cell_bwd needs.
 * fcl_decoder_train_fwd (round 6): s0 and s1 may ALL be NULL -- a forward that keeps nothing for a backward (the frozen KD teacher) then skips those stores. */
typedef struct {
    int n, lmax, u, p;              /* sorted rows, steps, dunits, prenet units */
    const int32_t* live_rows_host;  /* [lmax] HOST */
    const float* p1d;               /* [F, P] prenet output per cell (after dropout) */
    const float* g0;                /* [N, 4U] att_c . W0_att^T + b0 (hoisted out of the loop) */
    const float *w0_pre, *w0_hh, *w0_pos;
    const int32_t* dur;             /* [N] device */
    const float *w1_ih, *w1_hh, *b1;
    float zoneout;
    const uint8_t *zk_h0, *zk_c0, *zk_h1, *zk_c1; /* optional sampled zoneout masks [F, U] (1 keeps the old state), all or none */
    float* s0[4];
    float* s1[4];
    float *h0_all, *h1_all;         /* [F, U] zoneout-ed outputs of both layers */
    void* workspace;
    size_t workspace_bytes;         /* fcl_decoder_train_workspace_bytes(n, u) */
    /* optional P32 planes (all five or none; P and U multiples of 32): the prenet output per cell [F, P/32 lines] and the four weight matrices;
     * the steps with more than ~1000 live rows (~250 at U >= 512) then run on the LDS-DMA kernels, the states travelling pre-split */
    const uint16_t *p1d_p, *w0_pre_p, *w0_hh_p, *w1_ih_p, *w1_hh_p;
} fcl_decoder_train_t;
typedef struct {
    int n, lmax, u;
    const int32_t* live_rows_host;
    const float* s0[3];             /* gates, c_new, c_old of layer 0 (as saved by the forward) */
    const float* s1[3];
    float zoneout;
    const uint8_t *zk_h0, *zk_c0, *zk_h1, *zk_c1;
    const float* dh1_all;           /* [F, U] gradient w.r.t. layer-1 outputs (feat_out path + KD tap) */
    const float* dh0_all;           /* optional [F, U] extra gradient w.r.t. layer-0 outputs (KD tap) */
    const float *w1_ih_t, *w1_hh_t, *w0_hh_t; /* transposed weights [U, 4U] */
    float *dg0_all, *dg1_all;       /* out [F, 4U]: gate pre-activation gradients of every cell */
    void* workspace;
    size_t workspace_bytes;
    /* optional P32 planes (all five or none; U % 8 == 0): the transposed weights, and out: the gate gradients of every cell as planes
     * [F, 4U/32 lines].  The recurrence's three GEMMs per step then run on the pre-split-operand kernels for steps with enough live rows. */
    const uint16_t *w1_ih_t_p, *w1_hh_t_p, *w0_hh_t_p;
    uint16_t *dg0_all_p, *dg1_all_p;
    /* optional: [W1_hh^T ; W1_ih^T] as one [2U, 4U] matrix (and its planes when the five above are given): the two GEMMs that leave layer 1's gate
     * gradients are then ONE launch per step (four dependent launches per step instead of five) */
    const float* w1_cat_t;
    const uint16_t* w1_cat_t_p;
} fcl_decoder_bptt_t;
typedef struct {
    int b, t, h;
    const int32_t* lens;            /* [B] device */
    const float* gx[2];             /* per direction (0 forward, 1 reverse): x . W_ih^T + bias_ih + bias_hh, [B*T, 4H], rows (b, t) */
    const float* w_hh[2];           /* [4H, H] */
    float* out;                     /* [B*T, 2H]: direction d writes columns [d*H, d*H + H), zeros past each length */
    float* s[2][4];                 /* saved, t-major: gates [T, B, 4H], c_new, c_old, h_old [T, B, H]; only live cells are written */
    void* workspace;                /* fcl_bilstm_train_workspace_bytes(b, h): used by the per-step fallback (H not in {8,16,32,64,128}) */
    size_t workspace_bytes;
    uint32_t* status;               /* device status word (FCL_STATUS_*); NULL = never use the cooperating-workgroup kernel (H = 256) */
} fcl_bilstm_train_t;
typedef struct {
    int b, t, h;
    const int32_t* lens;
    const float* s[2][3];           /* gates, c_new, c_old of each direction */
    const float* d_out;             /* [B*T, ld_dout] gradient w.r.t. the layer output, ZERO on padded rows; direction d in columns [d*H, d*H + H) */
    int ld_dout;
    const float* w_hh_t[2];         /* [H, 4H] */
    float* dg[2];                   /* out, t-major [T, B, 4H]; dead cells are written as 0 */
    void* workspace;
    size_t workspace_bytes;
    uint32_t* status;               /* as in fcl_bilstm_train_t */
} fcl_bilstm_bptt_t;
size_t fcl_decoder_train_workspace_bytes(int n, int u);
int fcl_decoder_train_fwd(const fcl_decoder_train_t* args, fcl_stream_t stream);
int fcl_decoder_bptt(const fcl_decoder_bptt_t* args, fcl_stream_t stream);
size_t fcl_bilstm_train_workspace_bytes(int b, int h);
int fcl_bilstm_train_fwd(const fcl_bilstm_train_t* args, fcl_stream_t stream);
int fcl_bilstm_bptt(const fcl_bilstm_bptt_t* args, fcl_stream_t stream);
/* x *= alpha (gradient averaging after a SUM all-reduce on backends without AVG). */
int fcl_scale(float* x, size_t n, float alpha, fcl_stream_t stream);
/* out[i] = 1 with probability p_one: counter hash of (seed + *seed_dev, i).  The training path's source of dropout keep masks
 * (p_one = 1 - p) and zoneout keep-old masks (p_one = zoneout rate); not bit-compatible with torch's Philox stream by design. */
int fcl_bernoulli_u8(uint8_t* out, size_t n, float p_one, uint32_t seed, const uint32_t* seed_dev, fcl_stream_t stream);
/* Up to FCL_BERNOULLI_MAX_SITES masks in ONE launch (a training forward draws ~22 of them: one launch per site was ~10 % of its launches):
 * site k gets exactly the bytes fcl_bernoulli_u8(out, n, p_one, seed, NULL) would write.  `sites` is a HOST array (passed to the kernel by value). */
#define FCL_BERNOULLI_MAX_SITES 16
typedef struct {
    uint8_t* out;
    int64_t n;
    float p_one;
    uint32_t seed;
} fcl_bernoulli_site_t;
int fcl_bernoulli_batch(const fcl_bernoulli_site_t* sites, int n_sites, fcl_stream_t stream);
/* LSTMCell + zoneout backward of one step from the forward's saved gate activations [M,4U] (i,f,g,o), c_old and c_new (raw):
 * dgates [M,4U] (pre-activation), dh_old (zoneout keep path), dc_old.  The caller adds dgates . W_hh to dh_old.
 * dh_out2 (optional, row stride ld_dh2) is added to dh_out: the per-step output gradient next to the recurrent carry. */
int fcl_lstm_cell_bwd(const float* gates, const float* c_old, const float* c_new, const float* dh_out, const float* dh_out2, int ld_dh2,
                      const float* dc_out, float zoneout,
                      const uint8_t* zone_keep_h, const uint8_t* zone_keep_c, const int32_t* row_len, int step, float* dgates, float* dh_old,
                      float* dc_old, uint16_t* dgates_p /* optional P32 planes of dgates (4U/32 lines per row) */, int m, int u, fcl_stream_t stream);
/* dst[idx[m], :] += src[m, :]  (embedding gradient; idx == skip rows are dropped: padding_idx). */
int fcl_scatter_add_rows(const float* src, const int64_t* idx, float* dst, int m, int c, int64_t skip, fcl_stream_t stream);
int fcl_transpose2d(const float* src, float* dst, int rows, int cols, fcl_stream_t stream);
/* Batched operand forms of the parameters, one launch for any number of matrices (a training step re-derives every packed / transposed /
 * pre-split weight once per optimizer update; the frozen KD teacher once).  Descriptor i produces the [a*b, c] matrix
 *     out[(ia*b + ib), ic] = src[ia*sa + ib*sb + ic*sc]  (+ src2[same offset] when src2 != NULL)        (strides in floats, may be negative)
 * as fp32 (dst, dense, row stride c; optional) and / or as P32 planes (dst_p, ceil(c/32) lines per row, zero past c, 128-byte aligned;
 * optional).  Examples: transpose of W[r, c]: a=1, b=c, c=r, sb=1, sc=c; columns [c0, c0+n) of W: src=W+c0, b=r, c=n, sb=ld, sc=1; Conv1d taps
 * [Cout, Cin, k] -> [k, Cout, Cin]: a=k, b=Cout, c=Cin, sa=1, sb=Cin*k, sc=k; the backward's reversed, transposed taps [k, Cin, Cout]:
 * src=W+k-1, a=k, b=Cin, c=Cout, sa=-1, sb=k, sc=Cin*k; bias_ih + bias_hh: src2.
 * The table lives in DEVICE memory (the caller uploads it once and replays it); first_block = sum of fcl_derive_blocks(a, b, c) of the
 * descriptors before it, total_blocks = the sum over all.  Pointers inside the table cannot be checked here. */
typedef struct {
    const float* src;
    const float* src2;
    float* dst;
    uint16_t* dst_p;
    int32_t a, b, c;
    int32_t sa, sb, sc;
    int32_t first_block;
    int32_t reserved;
} fcl_derive_t;
int fcl_derive_blocks(int a, int b, int c);
int fcl_derive_batch(const fcl_derive_t* descs_dev, int n, int total_blocks, fcl_stream_t stream);
/* Optimizer (tts.py:173-182): *out += sum x^2 ; Adam step with clip_grad_norm_(max_norm) and the NaN guard taken from the
 * device-resident squared gradient norm, so the whole step stays on the stream.
 * step_dev: device int32 = number of updates APPLIED so far (torch's per-parameter `step`); the bias corrections use *step_dev + 1 and the
 * counter advances only when the update is applied, exactly like the reference, which does not call optimizer.step() on a NaN norm
 * (tts.py:173-179).  Deliberate deviation: an INFINITE norm is skipped as well — the reference would scale the gradients by 0, turn the
 * overflowed entries into NaN (inf * 0) and write them into the parameters.  status (optional): the update is also skipped while the device
 * status word is non-zero (a kernel of this step reported partial outputs). */
int fcl_sumsq_accum(const float* x, size_t n, double* out, fcl_stream_t stream);
int fcl_adam_step(float* p, const float* g, float* m, float* v, size_t n, const double* gradnorm_sq, float max_norm, float lr, float beta1,
                  float beta2, float eps, int32_t* step_dev, const uint32_t* status, fcl_stream_t stream);
/* fcl_adam_step with torch.optim.Adam's weight_decay (`--weight-decay`, /root/reference/tts.py:397-399, tts_distill.py:418-420): the L2 term
 * weight_decay * p is added to the clipped gradient inside the step, before the moment updates (torch's non-decoupled form). */
int fcl_adam_step_wd(float* p, const float* g, float* m, float* v, size_t n, const double* gradnorm_sq, float max_norm, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int32_t* step_dev, const uint32_t* status, fcl_stream_t stream);

/* ---- measurement hook (bench.py's live roofline figures) ------------------------------- */
/* While enabled, every GEMM / LSTM-step / BiLSTM launch is bracketed by HIP events on the stream it is
 * launched on.  fcl_prof_collect synchronises those events and returns one entry per kernel instantiation:
 * launches, summed duration, summed executed FLOPs (2*M*N*sum(K); no credit for hoisted or padded work) and
 * summed rows.  Off by default; never enabled inside a timed region. */
typedef struct {
    char name[56];
    int launches;
    double ms;
    double flops;
    double rows;
    double fill_bytes; /* round 6: bytes the launch's workgroups move global -> LDS through the CU's vector-memory path (LDS-DMA lines of the tiled GEMM /
                        * Conv1d / LSTM-step kernels: workgroups x k-chunks x (A lines + W lines) x 128 B); 0 for kernels that do not report it.  Against the
                        * measured per-CU delivery ceiling (tools/stream_probe.hip) this is the BINDING resource of those loops */
} fcl_prof_entry_t;
int fcl_prof_enable(int on);
int fcl_prof_collect(fcl_prof_entry_t* out, int max_entries);

/* ---- §8f N4 / BASELINE configs[4]: Parallel WaveGAN generator (mel -> waveform), the stage the reference delegates to the external
 *      `parallel-wavegan-decode` (inference_student.sh:20-23).  No source in the reference: the architecture is the published
 *      kan-bayashi/ParallelWaveGAN generator (v1, LJSpeech: 80 mels, hop 256 = 4*4*4*4, 30 layers / 3 stacks, 64 residual + skip channels,
 *      128 gate channels, kernel 3), restated in oracle/pwg_oracle.py (parity unpinned).  Rows are SAMPLES, time-major, utterances concatenated;
 *      seg_lo / seg_hi [M] give every row the sample range of its utterance (zero padding at utterance edges).  Needs the pre-split operand
 *      path (refused under FCL_PRECISION=0 / FCL_PLANES=0). ------------------------------------------------------------------------------- */
/* One stage of the upsampling network: nearest-neighbour stretch by `scale` + the 1 x (2*scale+1) smoothing convolution w (no bias), per channel.
 * in [frames * rate_in, c] -> out [frames * rate_in * scale, c] fp32 and / or out_p (P32 planes, ceil(c/32) lines per row, zero past c).
 * frame_utt [frames]: utterance of each mel frame; utt_off [n_utt + 1]: first frame of each utterance.  c % 4 == 0; in / out 16-byte aligned. */
int fcl_pwg_upsample_stage(const float* in, const int32_t* frame_utt, const int32_t* utt_off, int64_t frames, int rate_in, int scale, const float* w,
                           float* out, uint16_t* out_p, int c, int chunk_major /* out_p as [chunk][row] lines (fcl_gemm_term_t.a_chunk_stride) */,
                           fcl_stream_t stream);
/* Coefficient lines of the frame-rate auxiliary term (fcl_pwg_layer_t.kp).  kc [m, 8]: the upsampling network's response to the colour basis
 * e[g][c] = (g mod 5 == c) over the batch's frames (columns 5..7 unused), i.e. kc[m][g mod 5] is the weight of frame g in sample m for the five
 * frames g = f-2 .. f+2 around the sample's own frame f = m / hop.  Line of sample m: that weight at column g - w0(f), zero elsewhere, where the
 * window is w0 = 32 (f >> 5) (read from pt_a) when 2 <= f mod 32 <= 29 and w0 = 32 ((f + 16) >> 5) - 16 (read from pt_b) otherwise. */
int fcl_pwg_aux_coeff(const float* kc, int64_t m, int hop, int64_t frames, uint16_t* kp, fcl_stream_t stream);
/* The generator's input noise z ~ N(0, 1) (ParallelWaveGANGenerator.inference draws torch.randn): counter-based, reproducible per (seed, index). */
int fcl_pwg_noise(float* z, int64_t n, uint32_t seed, fcl_stream_t stream);
/* first_conv (Conv1d1x1 1 -> r): x[m, ch] = w[ch] * z[m] + b[ch], written as fp32 (optional) and as planes (row-major or chunk-major). */
int fcl_pwg_first_conv(const float* z, const float* w, const float* b, float* x, uint16_t* xp, int64_t m, int r, int chunk_major, fcl_stream_t stream);
/* One residual block (ResidualBlock.forward): dilated Conv1d(r -> 2r, ksize, dilation) + conv1x1_aux(aux -> 2r) as ONE GEMM of ksize + 1 K-terms,
 * tanh * sigmoid gate, conv1x1_out / conv1x1_skip as one GEMM, x = (out + x) * sqrt(0.5), skips += skip (= skip when first_layer). */
typedef struct {
    int64_t m;                 /* samples (rows) */
    int32_t r, aux;            /* residual (= skip = gate/2) channels, multiple of 32; auxiliary channels */
    int32_t ksize, dilation;
    int32_t first_layer;       /* != 0: skips is written, not accumulated */
    const int32_t* seg_lo;     /* [m] */
    const int32_t* seg_hi;
    float* x;                  /* [m, r] fp32, in / out */
    uint16_t* xp;              /* its planes (r/32 lines per row), in / out */
    const uint16_t* cp;        /* planes of the upsampled features [m, aux] (ceil(aux/32) lines per row) */
    const uint16_t* w_conv_p;  /* planes of the taps, tap-major [ksize * 2r, r] (fcl_pack_conv1d_weight + fcl_pack_planes) */
    const float* b_conv;       /* [2r] */
    const uint16_t* w_aux_p;   /* planes of conv1x1_aux [2r, aux] */
    const uint16_t* w_os_p;    /* planes of [conv1x1_out ; conv1x1_skip] stacked [2r, r] */
    const float* b_os;         /* [2r] = [b_out ; b_skip] */
    float* skips;              /* [m, r] */
    float* z;                  /* workspaces of the unfused path: [m, 2r] fp32 */
    uint16_t* gp;              /*             planes [m, r] */
    float* o;                  /*             [m, 2r] fp32 */
    uint16_t* xp_out;          /* != NULL: the block runs as ONE launch (r = 64, ksize = 3, aux <= 96 only) that reads x from xp and writes the new
                                * planes to xp_out (a different buffer: neighbouring tiles still read xp for their taps); x, z, gp, o are unused.
                                * In this form xp, xp_out and cp are CHUNK-MAJOR planes (line (chunk c, row m) at c * m_total * 128 + m * 128 bytes) */
    /* one-launch form only, optional (kp != NULL; cp / w_aux_p are then unused): the auxiliary term evaluated at FRAME rate.  The upsampling network
     * is linear and a sample of frame f sees frames f-2 .. f+2 only, so conv1x1_aux(upsample(c))[m] = sum_g k[m][g] * (W_aux c_in[g]): one K-chunk of 32
     * (coefficient line of the sample x the frame window of the projected features) replaces the ceil(aux/32) chunks of upsampled features. */
    const uint16_t* kp;        /* [m] coefficient lines (fcl_pwg_aux_coeff) */
    const uint16_t* pt_a;      /* planes of (W_aux c_in^T) [2r, frames]: line q of a row = frames [32q, 32q + 32) */
    const uint16_t* pt_b;      /* the same shifted by 16 frames: line q = frames [32q - 16, 32q + 16), zero where there is no frame */
    int32_t ld_pt;             /* lines per row of pt_a / pt_b (>= (frames + 16 + 31) / 32) */
    int32_t hop;               /* samples per frame: a multiple of 128 (a 128-sample tile lies inside one frame); with kp, seg_lo / seg_hi may change
                                * only at multiples of hop (utterances are whole frames): the kernel reads one pair of bounds per tile */
} fcl_pwg_layer_t;
int fcl_pwg_layer_fwd(const fcl_pwg_layer_t* a, fcl_stream_t stream);
/* last_conv_layers: wav[m] = relu(relu(skips * scale) W1^T + b1) . w2 + b2.  yp: workspace planes [m, s_ch]; h: workspace fp32 [m, s_ch]
 * (both unused, may be NULL, for s_ch = 64: one launch that reads skips once). */
int fcl_pwg_last_fwd(const float* skips, float scale, const uint16_t* w1p, const float* b1, const float* w2, float b2, uint16_t* yp, float* h, float* wav,
                     int64_t m, int s_ch, fcl_stream_t stream);

/* ---- the input feed of a capacity graph (..._kd_student.py:821-843: what inference() receives per call) -------------------------------------
 * fcl_feed_copy: ONE kernel that copies `bytes` (a multiple of 16) from pinned, mapped host memory (`src`: the DEVICE view of the block,
 * fcl_host_device_ptr) to `dst`, then increments *seq_dev, stores the new value to *seq_host (device view of a pinned word) and, when given,
 * adds 1 to *bump (the pass's RNG seed word).  Captured as the first node of a pass's hipGraph it replaces the hipMemcpyAsync in front of every
 * launch (~80 us of host time per pass); the host may repack the block once *seq_host equals the number of launches it has made. */
/* ---- streams restricted to a share of the compute units (round 6) --------------------------------------------------------------------------------
 * The KD update (tts_distill.py:143-182) runs the frozen teacher's forward one batch ahead on its own stream beside the student's update.  Both
 * are chains of dependent launches; the teacher's LSTM-step workgroups (72 - 96 KB of LDS, ~30 us each) fill every CU, so each of the student's
 * ~260 dependent launches first waits for one of them to retire (kernel trace: median 8.8 us between two launches of the student's stream, 0.1 us
 * on the teacher's).  A stream created here only dispatches to `n_cus` compute units, spread evenly over the XCDs (the queue's CU mask is
 * interleaved across XCCs by the driver: bit i -> XCD i % 8), which leaves the other CUs to the unrestricted streams.  n_cus <= 0 or >= the
 * device's CU count: an ordinary non-blocking stream. */
int fcl_stream_create_cus(int n_cus, fcl_stream_t* out);
int fcl_stream_destroy(fcl_stream_t stream);
/* Compute pipes (round 6).  The queue behind a HIP stream sits on one of MI355X's four compute pipes (queue index mod 4, in creation order within the process) and
 * a pipe advances one of its queues at a time: two chains of dependent launches take 1.0x the time of one when their streams are on different pipes, 1.43x on
 * the same pipe, 2.0x on the same hardware queue (more streams than GPU_MAX_HW_QUEUES).  HIP does not report the pipe; these two entries measure it (two chains
 * of 16 dependent ~20 us launches, alone and together, ~2 ms per pair; the streams must be idle).
 *   fcl_streams_share_pipe: *shared = 1 when a and b contend; *ratio (optional) = pair time / alone.
 *   fcl_stream_create_apart: a new stream apart from every others[k] (<= 24 candidates; n <= 3 can be met on four pipes; FCL_ERR_HIP when none fits, e.g. in a process holding dozens of streams); *tried (optional) = candidates created. */
int fcl_streams_share_pipe(fcl_stream_t a, fcl_stream_t b, int* shared, double* ratio);
int fcl_stream_create_apart(const fcl_stream_t* others, int n, fcl_stream_t* out, int* tried);
void* fcl_host_device_ptr(void* pinned_host);
int fcl_feed_copy(void* dst, const void* src, size_t bytes, uint32_t* seq_dev, uint32_t* seq_host, uint32_t* bump, fcl_stream_t stream);

/* ---- N3: the wire format handed to the vocoder (tts.py:652,674: kaldiio.WriteHelper("ark,scp:..."); inference_student.sh:20-23) --------------
 * HOST function (no device work): appends n Kaldi binary FloatMatrix records -- <key> ' ' "\0BFM " '\4' <int32 rows> '\4' <int32 cols> <float32 data>
 * -- to the open file descriptor `fd` with writev, straight from `data` (the utterances' matrices back to back, rows[i] x cols each; e.g. the
 * pinned landing buffer of a decoded batch).  file_pos: the file offset the first byte lands at; offsets[i] receives the offset of utterance i's
 * "\0B" marker (what its scp line points at).  Returns the new file offset (>= file_pos), or a negative FCL_ERR_* (fcl_last_error()). */
long long fcl_kaldi_ark_append(int fd, long long file_pos, int n, const char* const* keys, const float* data, const int* rows, int cols,
                               long long* offsets);

/* ---- H13 as ONE native routine: the teacher-forced training step orchestrated in C++ (round 4) -------------------------------------------------
 * Replaces, per update, the reference's `[teacher_knowledge = teacher(**x)]; loss = model(**x); loss.backward()` (tts_distill.py:143-182,
 * tts.py:137-179; forward()s: ..._kd_teacher.py:521-603, ..._kd_student.py:673-802, ..._sa.py:520-622) with a handful of host calls: the ~530
 * kernel launches of a KD update were issued one by one through Python / ctypes (16 us each: 8.7 ms of host time per 11 ms update);
 * here the same launches -- the very entry points above, in the same order, on the same three streams -- are issued from C++.
 * An engine (fcl_te_t) is bound to the flat parameter / gradient buffers of ONE model; it owns the operand forms of the parameters (packed taps,
 * transposes, column blocks, P32 planes: fcl_derive_batch, refreshed after every fcl_te_params_changed), a work arena and a zero arena in device
 * memory (sized by a dry run of the step before anything is launched; grown between steps), its weight-gradient stream and its events.
 * Scope: model.train() form (batch-statistics BatchNorm, sampled dropout / zoneout drawn on the device with the seeds the Python engine uses,
 * so both engines produce the same masks), the shipped-recipe structure (use_batch_norm, use_concate, no speaker embedding, no residual encoder,
 * no output activation), channel widths that are multiples of 32 (odim: of 4).  Everything else stays on fcl_taco2_amd.training.TrainEngine's
 * per-launch path, which is also the reference implementation this routine is tested against (equal losses / gradients on the same batch). */
typedef struct fcl_te fcl_te_t;
enum { FCL_TE_TEACHER = 0, FCL_TE_KD_TEACHER = 1, FCL_TE_STUDENT = 2 };
enum { FCL_TE_MAX_SITES = 48, FCL_TE_MAX_LOSSES = 48 };
typedef struct {
    int32_t role;                 /* FCL_TE_* */
    int32_t idim, odim, embed_dim, econv_layers, econv_chans, econv_filts, eunits, dunits, prenet_units, postnet_layers, postnet_chans, postnet_filts;
    int32_t dp_layers, dp_chans, dp_kernel;   /* duration predictor (ESPnet DurationPredictor) */
    int32_t vp_layers, vp_chans, vp_kernel;   /* pitch / energy predictors (variance_predictor.py) */
    int32_t ve_kernel;                         /* pitch / energy embedding Conv1d(1 -> C, k) */
    float dropout_rate, zoneout_rate, dp_dropout, vp_dropout, ve_dropout;
    int32_t use_masking;
    /* student: the teacher's widths (targets of the KD projections) and the distillation switches (..._kd_student.py:438-456) */
    int32_t t_embed_dim, t_econv_chans, t_eunits, t_prenet_units, t_dunits, t_postnet_chans;
    int32_t share_proj, distill_output, distill_encoder, distill_decoder, distill_prosody;
    int32_t accum_grad;
    uint32_t seed;                /* the engine's RNG seed (TrainEngine.seed) */
    uint32_t site_tag[FCL_TE_MAX_SITES]; /* per mask site: crc32(repr(site name)) & 0x7fffffff, order of fcl_te_site_name() */
    int32_t dw_planes_min;        /* weight gradients with at least this many outputs run on transposed planes */
    int32_t pred_stream;          /* predictors' forward / backward beside the decoder's on the weight-gradient stream (FCL_PRED_STREAM) */
    int32_t late_losses;          /* late KD terms on the weight-gradient stream (FCL_KD_LATE_LOSSES) */
} fcl_te_config_t;
/* one batch in the converter's layout (tts.py:215-306) with the integer maps of fcl_taco2_amd.training.build_maps_host, all in DEVICE memory
 * except live_rows_host */
typedef struct {
    int32_t B, T, L, N, F, lmax;
    const int64_t* xs;            /* [B*T] phoneme ids, 0 = padding */
    const float* ys;              /* [B*L, odim] target mels, zero padded */
    const float *f0, *energy, *ds;/* [B*T] ground-truth pitch / energy, durations as floats */
    const int32_t *lens, *e_lo, *e_hi, *f_lo, *f_hi, *src_sorted, *row_of_enc, *cell_frame, *frame_cell, *prev_frame, *cell_row, *dur, *perm_tb;
    const int64_t* cell_row_i64;
    const uint8_t *enc_pad, *enc_valid, *frame_valid, *cell_valid;
    const float* pos4;            /* [F, 4]: t / d in column 0 */
    const int32_t* live_rows_host;/* [lmax] HOST */
    double n_enc, n_frames;       /* valid phoneme positions / valid frames */
} fcl_te_batch_t;
/* the KD teacher's knowledge (..._kd_teacher.py:597-603) as device pointers.  dec[0..2] (prenet, LSTM-0, LSTM-1 taps): CELL-major [F, .] when
 * dec_cell_major != 0 (the native hand-over between two engines that share the batch's maps: no frame round trip), else frame-major [B*L, .] like
 * the reference's tuple. */
typedef struct {
    const float *after, *before;  /* [B*L, odim] */
    const float* enc[5];          /* embed, conv x3, blstm: [B*T, C_t] */
    const float* dec[8];          /* prenet, lstm0, lstm1 taps; postnet layer outputs x5 */
    const float* pro[5];          /* d_outs, p_outs, e_outs [B*T]; p_embs, e_embs [B*T, C_t] */
    int32_t dec_cell_major;
} fcl_te_knowledge_t;
int fcl_te_create(const fcl_te_config_t* cfg, fcl_te_t** out);
void fcl_te_destroy(fcl_te_t* te);
/* names of the mask sites (for site_tag) and of the loss slots (rows of the [FCL_TE_MAX_LOSSES][3] sums), NULL past the end */
const char* fcl_te_site_name(int i);
const char* fcl_te_loss_name(int i);
/* bind a parameter (value + gradient accumulator inside the caller's flat buffers) or a buffer (BatchNorm running statistics) by its state_dict
 * name; fcl_te_finalize checks that everything the configuration needs is bound */
int fcl_te_bind_param(fcl_te_t* te, const char* name, float* value, float* grad, int64_t numel);
int fcl_te_bind_buffer(fcl_te_t* te, const char* name, float* value);
int fcl_te_finalize(fcl_te_t* te, uint32_t* status_word);
/* the parameters were written (optimizer step, load_state_dict): operand forms are re-derived before the next forward */
int fcl_te_params_changed(fcl_te_t* te);
/* the engine's weight-gradient stream (created by the engine): the caller issues bucketed all-reduces from it between backward stages */
fcl_stream_t fcl_te_side_stream(fcl_te_t* te);
/* before the first pass (round 6): should the weight-gradient stream share a compute pipe with `main_stream` (measured, see fcl_streams_share_pipe), it is
 * replaced by a stream that does not; *moved (optional) = 1 then (a handle obtained from fcl_te_side_stream before the call is invalid), -1 when it contends but
 * no replacement could be placed (the stream is kept), 0 when it was fine */
int fcl_te_place_streams(fcl_te_t* te, fcl_stream_t main_stream, int* moved);
/* forward only (frozen KD teacher, train-mode statistics): *know points into one of the engine's two alternating arenas: valid until the
 * second next fcl_te_knowledge call on this engine */
int fcl_te_knowledge(fcl_te_t* te, const fcl_te_batch_t* batch, uint32_t draw, fcl_te_knowledge_t* know, fcl_stream_t stream);
/* forward + losses + backward stage 0 (postnet; gradient bucket 0 is final when it returns, in stream order); then fcl_te_backward_stage 1 (decoder:
 * bucket 1), 2 (predictors / embeddings: bucket 2), 3 (encoder: bucket 3 + the join of the weight-gradient stream).  know: required for the student.
 * loss_sums_host: pinned [FCL_TE_MAX_LOSSES][3] doubles receiving (sum |d|, sum d^2, count) per loss slot, status_host: pinned uint32, both copied
 * asynchronously on `stream` at the end of stage 0.  draw: the forward's ordinal (seeds the masks with cfg.seed and the site tags). */
int fcl_te_forward_backward(fcl_te_t* te, const fcl_te_batch_t* batch, const fcl_te_knowledge_t* know, uint32_t draw, double* loss_sums_host,
                            uint32_t* status_host, fcl_stream_t stream);
int fcl_te_backward_stage(fcl_te_t* te, int stage, fcl_stream_t stream);
/* the end-of-backward join: `stream` waits for everything on the weight-gradient stream (after the caller issued its last bucket's collective there) */
int fcl_te_join(fcl_te_t* te, fcl_stream_t stream);
/* diagnostics: launches issued by the last forward_backward (all stages) / knowledge call, and the arena bytes it used */
int64_t fcl_te_last_launches(fcl_te_t* te);
/* developer aid (FCL_TE_STAMPS=1): ms from the start of the last fcl_te_forward_backward to its phase boundaries on the main stream: [0] start,
 * [1] encoder, [2] prenet + hoists, [3] decoder cells, [4] forward end, [5] losses, [6..9] backward stages 0..3, [10] join; -1 = not recorded */
int fcl_te_phase_ms(fcl_te_t* te, float* out12);
int64_t fcl_te_arena_bytes(fcl_te_t* te);

#ifdef __cplusplus
}
#endif
#endif /* FCL_HIP_H_ */
