// lstm_epilogue.h — the LSTMCell + zoneout epilogue shared by every LSTM-step kernel.
// Reference maths: torch.nn.LSTMCell (gate order i,f,g,o) inside ZoneOutCell (decoder_sa.py:63-96).
#pragma once
#include "fcl_common.h"

namespace fcl {

// v_exp_f32 / v_rcp_f32 based gates: ~1e-7 absolute error (1-2 ulp each), an order of magnitude below the fp32
// summation-order noise of the gate pre-activations, and ~10x fewer VALU issues than ocml expf/tanhf in the
// latency-critical epilogue.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float act_apply(float v, int act) {  // FCL_ACT_*
    return act == FCL_ACT_RELU ? fmaxf(v, 0.f) : (act == FCL_ACT_TANH ? tanh_f(v) : (act == FCL_ACT_SIGMOID ? sigmoid_f(v) : v));
}

// everything one (row m, unit u) needs besides the MFMA partial sums, fetched early to hide latency
struct CellIn {
    float add[4];  // bias + G + pos*w  per gate
    float h_old, c_old;
    bool live;
};

// MODE -1: every optional input tested at run time; MODE 0: decoder layer 0 (G + position, no bias);
// MODE 1: decoder layer 1 (bias only).  The fixed modes issue their loads unconditionally, so hipcc keeps
// them all in flight instead of branching + waiting around each one (guide §5 trap (c)).
template <int MODE = -1>
__device__ __forceinline__ CellIn cell_prefetch(const LstmStepArgs& a, int m, int u) {
    CellIn ci;
    const bool has_bias = MODE < 0 ? a.bias != nullptr : MODE == 1;
    const bool has_g = MODE < 0 ? a.G != nullptr : MODE == 0;
    const bool has_pos = MODE < 0 ? a.rank1_w != nullptr : MODE == 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) ci.add[g] = has_bias ? a.bias[g * a.U + u] : 0.f;
    if (has_g) {
        const float* gr = a.G + (size_t)((long long)m * a.g_row_mul + a.g_row_add) * (4 * a.U);
#pragma unroll
        for (int g = 0; g < 4; ++g) ci.add[g] += gr[g * a.U + u];
    }
    if (has_pos) {
        const float pos = (float)a.step / (float)a.dur[m];  // reference: arange(d).float() / d  (a reciprocal instead: no measurable gain, r3)
#pragma unroll
        for (int g = 0; g < 4; ++g) ci.add[g] += pos * a.rank1_w[g * a.U + u];
    }
    const size_t off = (size_t)m * a.U + u;
    ci.h_old = a.h_in[off];
    ci.c_old = a.c[off];
    ci.live = a.row_len ? (a.step < a.row_len[m]) : true;
    return ci;
}

// the cell + zoneout update of one (row m, unit u): new hidden / cell state in h_w / c_w; the training-only side outputs (saved gates, out2 tap)
// are stored here, the states themselves by the caller (directly: cell_finish; or staged through LDS for row-wise stores: plstm_kernel)
// MODE >= 0 (the launchers pick it only when no train-form mask, no row_len, no saved gates and no out2 tap is set): those options are compiled
// out -- the element code of the big-tile step is instruction-bound (8 cells per lane, ~150 VALU instructions each with every option tested per
// cell: ~4 us per workgroup of the 17 us launch, r3 ISA count), and the synthesis loop uses none of them.
template <int MODE = -1>
__device__ __forceinline__ void cell_math(const LstmStepArgs& a, int m, int u, const float (&acc)[4], const CellIn& ci, float& h_w, float& c_w) {
    const float ig = sigmoid_f(acc[0] + ci.add[0]), fg = sigmoid_f(acc[1] + ci.add[1]);
    const float gg = tanh_f(acc[2] + ci.add[2]), og = sigmoid_f(acc[3] + ci.add[3]);
    const float c_new = fg * ci.c_old + ig * gg;
    const float h_new = og * tanh_f(c_new);
    const size_t off = (size_t)m * a.U + u;
    float h_o, c_o;
    if (MODE < 0 && a.zone_keep_h) {  // train-form zoneout: mask = 1 keeps the OLD state
        h_o = a.zone_keep_h[off] ? ci.h_old : h_new;
        c_o = a.zone_keep_c[off] ? ci.c_old : c_new;
    } else {  // eval form (rate 0 => plain cell): rate*old + (1-rate)*new
        h_o = a.zoneout * ci.h_old + (1.0f - a.zoneout) * h_new;
        c_o = a.zoneout * ci.c_old + (1.0f - a.zoneout) * c_new;
    }
    if (MODE >= 0) {  // (no row_len: every row of the launch is live; no side outputs)
        h_w = h_o;
        c_w = c_o;
        return;
    }
    h_w = ci.live ? h_o : ci.h_old;
    c_w = ci.live ? c_o : ci.c_old;
    if (a.save_gates) {  // training forward: what fcl_lstm_cell_bwd needs
        float* sg = a.save_gates + (size_t)m * 4 * a.U;
        sg[u] = ig; sg[a.U + u] = fg; sg[2 * a.U + u] = gg; sg[3 * a.U + u] = og;
        a.save_c_new[off] = c_new;
        a.save_c_old[off] = ci.c_old;
        a.save_h_old[off] = ci.h_old;
    }
    if (a.out2) {
        const long long row = (a.out2_row_base ? (long long)a.out2_row_base[m] : (long long)m * a.out2_row_mul) + a.out2_row_add;
        a.out2[(size_t)row * a.ld2 + a.out2_col_off + u] = ci.live ? h_o : 0.f;
    }
}

__device__ __forceinline__ void cell_finish(const LstmStepArgs& a, int m, int u, const float (&acc)[4], const CellIn& ci) {
    float h_w, c_w;
    cell_math(a, m, u, acc, ci, h_w, c_w);
    const size_t off = (size_t)m * a.U + u;
    a.h_out[off] = h_w;
    if (a.h_out_p) store_p32(a.h_out_p, a.ld_hp, m, u, h_w);  // the same state as the next step's pre-split GEMM operand
    a.c[off] = c_w;
}

}  // namespace fcl
