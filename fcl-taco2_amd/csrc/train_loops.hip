// train_loops.hip — the time loops of the training step, run on the host side of the C ABI so that one call enqueues a whole recurrence
// (hundreds of launches) instead of one Python round trip per kernel.  Every kernel is one of the library's own entry points.
//   fcl_decoder_train_fwd  : teacher-forced decoder LSTM stack over step-major cells, saving what BPTT needs (decoder_sa.py:472-515)
//   fcl_decoder_bptt       : its reverse-time pass (gate gradients of both layers for every cell)
//   fcl_bilstm_train_fwd   : one direction of the packed encoder BiLSTM with saved gates (encoder_sa.py:98-100,143-146)
//   fcl_bilstm_bptt        : its reverse pass
#include <vector>

#include "fcl_common.h"

using namespace fcl;

namespace {

inline GemmArgs lin(const float* x, int lda, const float* w, int ldw, int k, float* y, int ldy, int m, int n, const float* residual, int ldr) {
    GemmArgs g = {};
    g.term[0] = GemmTerm{x, w, lda, ldw, k, 0, nullptr, nullptr};
    g.nterms = 1;
    g.M = m;
    g.N = n;
    g.act = FCL_ACT_NONE;
    g.R = residual;
    g.ldr = ldr;
    g.Y = y;
    g.ldy = ldy;
    return g;
}

}  // namespace

extern "C" {

// 6 fp32 state buffers [N, U] + 4 plane buffers of the same byte size (h0 / h1 ping-pong as P32 planes), all 128-byte aligned when N*U*4 is
size_t fcl_decoder_train_workspace_bytes(int n, int u) { return (n > 0 && u > 0) ? sizeof(float) * 10 * (size_t)n * u + 128 : 0; }

int fcl_decoder_train_fwd(const fcl_decoder_train_t* a, fcl_stream_t stream) {
    FCL_REQUIRE(a && a->live_rows_host && a->p1d && a->g0 && a->w0_pre && a->w0_hh && a->w0_pos && a->dur && a->w1_ih && a->w1_hh && a->b1,
                FCL_ERR_INVALID, "decoder_train_fwd: null argument");
    FCL_REQUIRE(a->n > 0 && a->lmax > 0 && a->u > 0 && a->p > 0, FCL_ERR_SHAPE, "decoder_train_fwd: bad sizes");
    FCL_REQUIRE(a->h0_all && a->h1_all, FCL_ERR_INVALID, "decoder_train_fwd: outputs missing");
    // (round 6) s0 / s1 all NULL: a forward that keeps nothing for a backward (the frozen KD teacher: 1.5 GB of gate / state stores per pass at FCL-taco2-T width that
    // nobody read); otherwise all eight tensors
    {
        int have = 0;
        for (int q = 0; q < 4; ++q) have += (a->s0[q] != nullptr) + (a->s1[q] != nullptr);
        FCL_REQUIRE(have == 0 || have == 8, FCL_ERR_INVALID, "decoder_train_fwd: the saved tensors s0 / s1 come all together or not at all");
    }
    const bool keep = a->s0[0] != nullptr;
    FCL_REQUIRE((a->zk_h0 == nullptr) == (a->zk_c0 == nullptr) && (a->zk_h0 == nullptr) == (a->zk_h1 == nullptr) &&
                    (a->zk_h0 == nullptr) == (a->zk_c1 == nullptr), FCL_ERR_INVALID, "decoder_train_fwd: the four zoneout masks come together");
    FCL_REQUIRE(a->workspace && a->workspace_bytes >= fcl_decoder_train_workspace_bytes(a->n, a->u), FCL_ERR_WORKSPACE, "decoder_train_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t NU = (size_t)a->n * a->u;
    const int U = a->u;
    float* ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(a->workspace) + 127) & ~(uintptr_t)127);
    const bool planes = a->p1d_p && a->w0_pre_p && a->w0_hh_p && a->w1_ih_p && a->w1_hh_p && !(a->p & 31) && !(U & 31) && !((NU * 4) & 127);
    FCL_HIP(hipMemsetAsync(ws, 0, sizeof(float) * (planes ? 10 : 6) * NU, s));
    float *h0[2] = {ws, ws + NU}, *h1[2] = {ws + 2 * NU, ws + 3 * NU}, *c0 = ws + 4 * NU, *c1 = ws + 5 * NU;
    unsigned short* h0p[2] = {reinterpret_cast<unsigned short*>(ws + 6 * NU), reinterpret_cast<unsigned short*>(ws + 7 * NU)};
    unsigned short* h1p[2] = {reinterpret_cast<unsigned short*>(ws + 8 * NU), reinterpret_cast<unsigned short*>(ws + 9 * NU)};
    const int ldp = a->p / 32, ldu = U / 32;
    // step-major cell offsets; h0 of step t lives in h0[(t + 1) & 1] (step t reads h0[t & 1]), the same for h1
    std::vector<size_t> offs((size_t)a->lmax + 1, 0);
    {
        int prev = a->n;
        for (int t = 0; t < a->lmax; ++t) {
            const int n = a->live_rows_host[t];
            FCL_REQUIRE(n > 0 && n <= prev, FCL_ERR_SHAPE, "decoder_train_fwd: live_rows must be positive and non-increasing");
            prev = n;
            offs[t + 1] = offs[t] + (size_t)n;
        }
    }
    auto is_big = [&](int t) { return planes && !lstm_step_is_small(a->live_rows_host[t], U); };  // pre-split operands for the big-tile steps (their h planes feed the next big step)
    auto layer0 = [&](int t) {
        const int n = a->live_rows_host[t], cur = t & 1;
        const size_t off = offs[t];
        LstmStepArgs l0 = {};
        l0.term[0] = GemmTerm{a->p1d + off * a->p, a->w0_pre, a->p, a->p, a->p, 0, nullptr, nullptr};
        l0.term[1] = GemmTerm{h0[cur], a->w0_hh, U, U, U, 0, nullptr, nullptr};
        if (is_big(t)) {
            l0.term[0].Ap = a->p1d_p + off * (size_t)ldp * 64; l0.term[0].Wp = a->w0_pre_p; l0.term[0].lda_p = l0.term[0].ldw_p = ldp;
            l0.term[1].Ap = h0p[cur]; l0.term[1].Wp = a->w0_hh_p; l0.term[1].lda_p = l0.term[1].ldw_p = ldu;
            l0.h_out_p = h0p[cur ^ 1]; l0.ld_hp = ldu;
        }
        l0.nterms = 2;
        l0.M = n;
        l0.U = U;
        l0.G = a->g0;
        l0.g_row_mul = 1;
        l0.rank1_w = a->w0_pos;
        l0.dur = a->dur;
        l0.step = t;
        l0.h_in = h0[cur];
        l0.h_out = h0[cur ^ 1];
        l0.c = c0;
        l0.zoneout = a->zoneout;
        if (a->zk_h0) { l0.zone_keep_h = a->zk_h0 + off * U; l0.zone_keep_c = a->zk_c0 + off * U; }
        l0.out2 = a->h0_all + off * U;
        l0.out2_row_mul = 1;
        l0.ld2 = U;
        if (keep) {
            l0.save_gates = a->s0[0] + off * 4 * U;
            l0.save_c_new = a->s0[1] + off * U;
            l0.save_c_old = a->s0[2] + off * U;
            l0.save_h_old = a->s0[3] + off * U;
        }
        return l0;
    };
    auto layer1 = [&](int t) {
        const int n = a->live_rows_host[t], cur = t & 1;
        const size_t off = offs[t];
        LstmStepArgs l1 = {};
        l1.term[0] = GemmTerm{h0[cur ^ 1], a->w1_ih, U, U, U, 0, nullptr, nullptr};
        l1.term[1] = GemmTerm{h1[cur], a->w1_hh, U, U, U, 0, nullptr, nullptr};
        if (is_big(t)) {
            l1.term[0].Ap = h0p[cur ^ 1]; l1.term[0].Wp = a->w1_ih_p; l1.term[0].lda_p = l1.term[0].ldw_p = ldu;
            l1.term[1].Ap = h1p[cur]; l1.term[1].Wp = a->w1_hh_p; l1.term[1].lda_p = l1.term[1].ldw_p = ldu;
            l1.h_out_p = h1p[cur ^ 1]; l1.ld_hp = ldu;
        }
        l1.nterms = 2;
        l1.M = n;
        l1.U = U;
        l1.bias = a->b1;
        l1.step = t;
        l1.h_in = h1[cur];
        l1.h_out = h1[cur ^ 1];
        l1.c = c1;
        l1.zoneout = a->zoneout;
        if (a->zk_h1) { l1.zone_keep_h = a->zk_h1 + off * U; l1.zone_keep_c = a->zk_c1 + off * U; }
        l1.out2 = a->h1_all + off * U;
        l1.out2_row_mul = 1;
        l1.ld2 = U;
        if (keep) {
            l1.save_gates = a->s1[0] + off * 4 * U;
            l1.save_c_new = a->s1[1] + off * U;
            l1.save_c_old = a->s1[2] + off * U;
            l1.save_h_old = a->s1[3] + off * U;
        }
        return l1;
    };
    // Under teacher forcing layer 0 never waits for layer 1 (its input is the ground-truth frame's prenet output and its own state), so the two
    // recurrences run as a WAVEFRONT: [L0(0)] -> [L0(1) | L1(0)] -> [L0(2) | L1(1)] -> ... -> [L1(lmax - 1)].  L0(t + 1) reads h0(t) from
    // h0[(t + 1) & 1] and writes h0[t & 1], whose last readers -- L0(t), L1(t - 1) -- ran in earlier launches; L1(t) reads h0(t) beside it.
    // One launch per pair where both steps run the same kernel family (round 5): lmax + 1 dependent launches instead of 2 lmax, and twice the
    // workgroups per launch (fewer part-filled rounds of 128-row tiles at FCL-taco2-T width).  FCL_TRAIN_WAVEFRONT=0: the step-by-step order.
    static const int wavefront = tunable("TRAIN_WAVEFRONT", 1);
    if (!wavefront) {
        for (int t = 0; t < a->lmax; ++t) {
            int rc = launch_lstm_step(layer0(t), s);
            if (rc) return rc;
            rc = launch_lstm_step(layer1(t), s);
            if (rc) return rc;
        }
        return 0;
    }
    int rc = launch_lstm_step(layer0(0), s);
    if (rc) return rc;
    for (int t = 0; t < a->lmax; ++t) {
        const LstmStepArgs l1 = layer1(t);
        if (t + 1 == a->lmax) {
            rc = launch_lstm_step(l1, s);
            if (rc) return rc;
            break;
        }
        const LstmStepArgs l0 = layer0(t + 1);
        bool done = false;
        const bool sm1 = lstm_step_is_small(l1.M, U), sm0 = lstm_step_is_small(l0.M, U);
        //  the pair launches bypass launch_lstm_step: its argument checks run here, and the plane pair is only taken where launch_lstm_step itself would
        // run both steps on the pre-split operands (FCL_PRECISION=0 / FCL_PLANES=0 keep their fp32 kernels for direct callers of the C ABI too)
        rc = validate_lstm_step(l1);
        if (rc) return rc;
        rc = validate_lstm_step(l0);
        if (rc) return rc;
        if (is_big(t) && is_big(t + 1) && lstm_step_on_planes(l1) && lstm_step_on_planes(l0)) {
            rc = launch_lstm_planes_pair(l1, l0, s, &done);  // (l1 first: it has the larger row count)
            if (rc) return rc;
        } else if (sm1 && sm0 && l1.nterms == l0.nterms) {
            rc = launch_lstm_small_pair_any(l1, l0, s, &done);
            if (rc) return rc;
        }
        if (!done) {
            rc = launch_lstm_step(l1, s);
            if (rc) return rc;
            rc = launch_lstm_step(l0, s);
            if (rc) return rc;
        }
    }
    return 0;
}

int fcl_decoder_bptt(const fcl_decoder_bptt_t* a, fcl_stream_t stream) {
    FCL_REQUIRE(a && a->live_rows_host && a->s0[0] && a->s0[1] && a->s0[2] && a->s1[0] && a->s1[1] && a->s1[2] && a->dh1_all && a->w1_ih_t &&
                    a->w1_hh_t && a->w0_hh_t && a->dg0_all && a->dg1_all, FCL_ERR_INVALID, "decoder_bptt: null argument");
    FCL_REQUIRE(a->n > 0 && a->lmax > 0 && a->u > 0, FCL_ERR_SHAPE, "decoder_bptt: bad sizes");
    FCL_REQUIRE(a->workspace && a->workspace_bytes >= fcl_decoder_train_workspace_bytes(a->n, a->u), FCL_ERR_WORKSPACE, "decoder_bptt: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t NU = (size_t)a->n * a->u;
    const int U = a->u, G4 = 4 * a->u;
    float* ws = (float*)a->workspace;
    FCL_HIP(hipMemsetAsync(ws, 0, sizeof(float) * 6 * NU, s));  // rows that only become live at earlier steps must see zero carries
    float *ch0 = ws, *ch1 = ws + NU, *cc0 = ws + 2 * NU, *cc1 = ws + 3 * NU, *tmp_h = ws + 4 * NU, *tmp_c = ws + 5 * NU;
    size_t total = 0;
    for (int t = 0; t < a->lmax; ++t) total += (size_t)a->live_rows_host[t];
    const bool planes = a->w1_ih_t_p && a->w1_hh_t_p && a->w0_hh_t_p && a->dg0_all_p && a->dg1_all_p && !(G4 & 31);
    const int ldg = G4 / 32;  // plane lines per gate-gradient row
    static const int planes_min_rows = tunable("BPTT_PLANES_MIN_M", 128);  // below: the split-K small-M kernel on the fp32 operands is faster (round 6 re-scan: 256 -> 128: teacher update -0.75 %, KD neutral)
    auto linp = [&](const float* dg, const unsigned short* dgp, const float* wt, const unsigned short* wtp, float* y, int n, const float* residual) {
        GemmArgs g = lin(dg, G4, wt, G4, G4, y, U, n, U, residual, U);
        if (planes && n > planes_min_rows) { g.term[0].Ap = dgp; g.term[0].Wp = wtp; g.term[0].lda_p = g.term[0].ldw_p = ldg; }
        return g;
    };
    size_t off = total;
    if (a->w1_cat_t && (!planes || a->w1_cat_t_p) && tunable("BPTT_FUSE", 1)) {
        // Four launches per step instead of five: the two GEMMs that leave layer 1's gate gradients (dg1 . W1_hh -> layer 1's carry, dg1 . W1_ih ->
        // layer 0's carry) are ONE GEMM against [W1_hh^T ; W1_ih^T] into X [N, 2U] = [carry of layer 1 | carry of layer 0], which both cell
        // kernels read and write in place (their "keep" path is the GEMMs' residual): 38 fewer dependent launches on a KD update's critical path
        float* X = ws;  // (the ch0 | ch1 region of the workspace, zeroed above)
        const int U2 = 2 * U;
        float* tmp_c1 = tmp_h;  // (the keep paths live in X: the old tmp_h buffer is layer 1's second cell-carry buffer)
        // ... and THREE once the two cell kernels that follow that GEMM -- layer 0 of step t and layer 1 of step t - 1, both waiting for nothing
        // else -- share a launch: per step  [W1 GEMM(t)] -> [cell L0(t) | cell L1(t - 1)] -> [W0_hh GEMM(t)]
        std::vector<size_t> offs((size_t)a->lmax);
        for (int t = 0; t < a->lmax; ++t) { offs[(size_t)t] = t ? offs[(size_t)t - 1] + (size_t)a->live_rows_host[t - 1] : 0; }
        auto cell1 = [&](int t) {  // layer 1 of step t: carry X[:, :U] in place, cell carry cc1 -> tmp_c1 (swapped by the caller)
            const size_t o = offs[(size_t)t];
            return CellBwdArgs{a->s1[0] + o * G4, a->s1[2] + o * U, a->s1[1] + o * U, X, a->dh1_all + o * U, cc1, a->zk_h1 ? a->zk_h1 + o * U : nullptr,
                               a->zk_c1 ? a->zk_c1 + o * U : nullptr, nullptr, a->dg1_all + o * G4, X, tmp_c1,
                               planes ? a->dg1_all_p + o * (size_t)ldg * 64 : nullptr, U2, U2, U, t, a->live_rows_host[t], U, a->zoneout};
        };
        auto cell0 = [&](int t) {
            const size_t o = offs[(size_t)t];
            return CellBwdArgs{a->s0[0] + o * G4, a->s0[2] + o * U, a->s0[1] + o * U, X + U, a->dh0_all ? a->dh0_all + o * U : nullptr, cc0,
                               a->zk_h0 ? a->zk_h0 + o * U : nullptr, a->zk_c0 ? a->zk_c0 + o * U : nullptr, nullptr, a->dg0_all + o * G4, X + U, tmp_c,
                               planes ? a->dg0_all_p + o * (size_t)ldg * 64 : nullptr, U2, U2, U, t, a->live_rows_host[t], U, a->zoneout};
        };
        int rc = launch_lstm_cell_bwd(cell1(a->lmax - 1), s);
        if (rc) return rc;
        std::swap(cc1, tmp_c1);
        for (int t = a->lmax - 1; t >= 0; --t) {
            const int n = a->live_rows_host[t];
            const size_t o = offs[(size_t)t];
            unsigned short* dg1p = planes ? a->dg1_all_p + o * (size_t)ldg * 64 : nullptr;
            unsigned short* dg0p = planes ? a->dg0_all_p + o * (size_t)ldg * 64 : nullptr;
            GemmArgs g = lin(a->dg1_all + o * G4, G4, a->w1_cat_t, G4, G4, X, U2, n, U2, X, U2);
            if (planes && n > planes_min_rows) { g.term[0].Ap = dg1p; g.term[0].Wp = a->w1_cat_t_p; g.term[0].lda_p = g.term[0].ldw_p = ldg; }
            rc = launch_gemm(g, s);
            if (rc) return rc;
            if (t > 0) {
                rc = launch_lstm_cell_bwd_pair(cell0(t), cell1(t - 1), s);
                std::swap(cc1, tmp_c1);
            } else {
                rc = launch_lstm_cell_bwd(cell0(t), s);
            }
            if (rc) return rc;
            std::swap(cc0, tmp_c);
            GemmArgs g0 = lin(a->dg0_all + o * G4, G4, a->w0_hh_t, G4, G4, X + U, U2, n, U, X + U, U2);
            if (planes && n > planes_min_rows) { g0.term[0].Ap = dg0p; g0.term[0].Wp = a->w0_hh_t_p; g0.term[0].lda_p = g0.term[0].ldw_p = ldg; }
            rc = launch_gemm(g0, s);
            if (rc) return rc;
        }
        return 0;
    }
    for (int t = a->lmax - 1; t >= 0; --t) {
        const int n = a->live_rows_host[t];
        off -= (size_t)n;
        unsigned short* dg1p = planes ? a->dg1_all_p + off * (size_t)ldg * 64 : nullptr;
        unsigned short* dg0p = planes ? a->dg0_all_p + off * (size_t)ldg * 64 : nullptr;
        // layer 1
        int rc = fcl_lstm_cell_bwd(a->s1[0] + off * G4, a->s1[2] + off * U, a->s1[1] + off * U, ch1, a->dh1_all + off * U, U, cc1, a->zoneout,
                                   a->zk_h1 ? a->zk_h1 + off * U : nullptr, a->zk_c1 ? a->zk_c1 + off * U : nullptr, nullptr, t, a->dg1_all + off * G4,
                                   tmp_h, tmp_c, dg1p, n, U, stream);
        if (rc) return rc;
        std::swap(cc1, tmp_c);
        rc = launch_gemm(linp(a->dg1_all + off * G4, dg1p, a->w1_hh_t, a->w1_hh_t_p, ch1, n, tmp_h), s);  // ch1 = dg1 . W1_hh + zoneout keep path
        if (rc) return rc;
        rc = launch_gemm(linp(a->dg1_all + off * G4, dg1p, a->w1_ih_t, a->w1_ih_t_p, ch0, n, ch0), s);  // ch0 += dg1 . W1_ih
        if (rc) return rc;
        // layer 0
        rc = fcl_lstm_cell_bwd(a->s0[0] + off * G4, a->s0[2] + off * U, a->s0[1] + off * U, ch0, a->dh0_all ? a->dh0_all + off * U : nullptr, U, cc0,
                               a->zoneout, a->zk_h0 ? a->zk_h0 + off * U : nullptr, a->zk_c0 ? a->zk_c0 + off * U : nullptr, nullptr, t,
                               a->dg0_all + off * G4, tmp_h, tmp_c, dg0p, n, U, stream);
        if (rc) return rc;
        std::swap(cc0, tmp_c);
        rc = launch_gemm(linp(a->dg0_all + off * G4, dg0p, a->w0_hh_t, a->w0_hh_t_p, ch0, n, tmp_h), s);
        if (rc) return rc;
    }
    return 0;
}

size_t fcl_bilstm_train_workspace_bytes(int b, int h) {
    return (b > 0 && h > 0) ? sizeof(float) * 4 * (size_t)b * h + (h == 256 ? bilstm_group_workspace_bytes(b, h) : 0) : 0;
}

int fcl_bilstm_train_fwd(const fcl_bilstm_train_t* a, fcl_stream_t stream) {
    FCL_REQUIRE(a && a->lens && a->gx[0] && a->gx[1] && a->w_hh[0] && a->w_hh[1] && a->out, FCL_ERR_INVALID, "bilstm_train_fwd: null argument");
    for (int d = 0; d < 2; ++d)
        for (int i = 0; i < 4; ++i) FCL_REQUIRE(a->s[d][i], FCL_ERR_INVALID, "bilstm_train_fwd: save buffers missing");
    FCL_REQUIRE(a->b > 0 && a->t > 0 && a->h > 0 && (a->h & 3) == 0, FCL_ERR_SHAPE, "bilstm_train_fwd: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int B = a->b, T = a->t, H = a->h;
    if (!tunable("BILSTM_TRAIN_STEPS", 0)) {  // one persistent launch for both directions (H in {8,16,32,64,128})
        BilstmSave sv;
        for (int d = 0; d < 2; ++d) { sv.gates[d] = a->s[d][0]; sv.c_new[d] = a->s[d][1]; sv.c_old[d] = a->s[d][2]; sv.h_old[d] = a->s[d][3]; }
        sv.B = B;
        if (launch_bilstm_train_persistent(a->gx[0], a->gx[1], a->w_hh[0], a->w_hh[1], a->lens, a->out, B, T, H, sv, s))
            return check_hip(hipGetLastError(), "bilstm_train_fwd persistent launch");
        if (launch_bilstm_group(a->gx[0], a->gx[1], a->w_hh[0], a->w_hh[1], a->lens, a->out, B, T, H, &sv, a->workspace, a->workspace_bytes, a->status, s))
            return check_hip(hipGetLastError(), "bilstm_train_fwd group launch");
    }
    FCL_REQUIRE(a->workspace && a->workspace_bytes >= fcl_bilstm_train_workspace_bytes(B, H), FCL_ERR_WORKSPACE, "bilstm_train_fwd: workspace too small");
    const size_t BH = (size_t)B * H;
    float* ws = (float*)a->workspace;
    for (int d = 0; d < 2; ++d) {
        FCL_HIP(hipMemsetAsync(ws, 0, sizeof(float) * 3 * BH, s));
        float *h[2] = {ws, ws + BH}, *c = ws + 2 * BH;
        int cur = 0;
        for (int i = 0; i < T; ++i) {
            const int t = d ? T - 1 - i : i;
            LstmStepArgs l = {};
            l.term[0] = GemmTerm{h[cur], a->w_hh[d], H, H, H, 0, nullptr, nullptr};
            l.nterms = 1;
            l.M = B;
            l.U = H;
            l.G = a->gx[d];
            l.g_row_mul = T;
            l.g_row_add = t;
            l.step = t;
            l.h_in = h[cur];
            l.h_out = h[cur ^ 1];
            l.c = c;
            l.row_len = a->lens;
            l.out2 = a->out;
            l.out2_row_mul = T;
            l.out2_row_add = t;
            l.ld2 = 2 * H;
            l.out2_col_off = d * H;
            l.save_gates = a->s[d][0] + (size_t)t * B * 4 * H;
            l.save_c_new = a->s[d][1] + (size_t)t * BH;
            l.save_c_old = a->s[d][2] + (size_t)t * BH;
            l.save_h_old = a->s[d][3] + (size_t)t * BH;
            int rc = launch_lstm_step(l, s);
            if (rc) return rc;
            cur ^= 1;
        }
    }
    return 0;
}

int fcl_bilstm_bptt(const fcl_bilstm_bptt_t* a, fcl_stream_t stream) {
    FCL_REQUIRE(a && a->lens && a->d_out && a->w_hh_t[0] && a->w_hh_t[1] && a->dg[0] && a->dg[1], FCL_ERR_INVALID, "bilstm_bptt: null argument");
    for (int d = 0; d < 2; ++d)
        for (int i = 0; i < 3; ++i) FCL_REQUIRE(a->s[d][i], FCL_ERR_INVALID, "bilstm_bptt: saved tensors missing");
    FCL_REQUIRE(a->b > 0 && a->t > 0 && a->h > 0 && (a->h & 3) == 0 && a->ld_dout >= 2 * a->h, FCL_ERR_SHAPE, "bilstm_bptt: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int B = a->b, T = a->t, H = a->h, G4 = 4 * a->h;
    if (!tunable("BILSTM_TRAIN_STEPS", 0)) {
        BilstmBwd bw;
        for (int d = 0; d < 2; ++d) { bw.gates[d] = a->s[d][0]; bw.c_new[d] = a->s[d][1]; bw.c_old[d] = a->s[d][2]; bw.whh_t[d] = a->w_hh_t[d]; bw.dg[d] = a->dg[d]; }
        bw.d_out = a->d_out;
        bw.ld = a->ld_dout;
        bw.B = B;
        if (launch_bilstm_bptt_persistent(bw, a->lens, B, T, H, s)) return check_hip(hipGetLastError(), "bilstm_bptt persistent launch");
        if (launch_bilstm_bptt_group(bw, a->lens, B, T, H, a->workspace, a->workspace_bytes, a->status, s)) return check_hip(hipGetLastError(), "bilstm_bptt group launch");
    }
    FCL_REQUIRE(a->workspace && a->workspace_bytes >= fcl_bilstm_train_workspace_bytes(B, H), FCL_ERR_WORKSPACE, "bilstm_bptt: workspace too small");
    const size_t BH = (size_t)B * H;
    float* ws = (float*)a->workspace;
    for (int d = 0; d < 2; ++d) {
        FCL_HIP(hipMemsetAsync(ws, 0, sizeof(float) * 4 * BH, s));
        float *dh = ws, *tmp_h = ws + BH, *dc = ws + 2 * BH, *tmp_c = ws + 3 * BH;
        for (int i = T - 1; i >= 0; --i) {  // reverse of the forward visiting order
            const int t = d ? T - 1 - i : i;
            float* dg = a->dg[d] + (size_t)t * B * G4;
            // output gradient of step t: rows (b, t) of d_out, i.e. row stride T * ld_dout; dead cells (row_len) pass the carries through with dg = 0
            int rc = fcl_lstm_cell_bwd(a->s[d][0] + (size_t)t * B * G4, a->s[d][2] + (size_t)t * BH, a->s[d][1] + (size_t)t * BH, dh,
                                       a->d_out + (size_t)t * a->ld_dout + d * H, T * a->ld_dout, dc, 0.f, nullptr, nullptr, a->lens, t, dg, tmp_h, tmp_c,
                                       nullptr, B, H, stream);
            if (rc) return rc;
            std::swap(dc, tmp_c);
            rc = launch_gemm(lin(dg, G4, a->w_hh_t[d], G4, G4, dh, H, B, H, tmp_h, H), s);  // dh = dgates . W_hh + pass-through of dead rows
            if (rc) return rc;
        }
    }
    return 0;
}

}  // extern "C"
