// train_engine.hip — the teacher-forced training step orchestrated in C++: fcl_te_* of include/fcl_hip.h.
//
// What it replaces per update (reference, paths relative to /root/reference): `teacher_knowledge = teacher(**x)` (tts_distill.py:159,
// nets/knowledge_distillation/e2e_tts_tacotron2_sa_kd_teacher.py:521-603), `loss = model(**x)` (..._kd_student.py:673-802 / nets/teacher_training/
// e2e_tts_tacotron2_sa.py:520-622) and `loss.backward()` (tts_distill.py:165-170, tts.py:160-167).  Every FLOP is a launch of an entry point of
// this library (the same ones, in the same order and on the same streams as fcl_taco2_amd/training.py issues them one ctypes call at a time);
// this file is HOST code: index arithmetic, an arena, a table of operand forms, stream forks and joins.  training.py's TrainEngine stays the
// reference implementation (pinned to the real reference's losses / gradients by the goldens G5 - G13) and the path of every option this routine
// does not cover; tests/test_gpu_train_native.py holds the two against each other on the same batches and masks.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <string>
#include <unordered_map>
#include <vector>

#include "fcl_common.h"

namespace fcl {
namespace te {

static const float BN_EPS = 1e-5f, LN_EPS = 1e-12f, BN_MOMENTUM = 0.1f;

struct Param {
    float* p = nullptr;
    float* g = nullptr;
    long long numel = 0;
};

struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0, dirty = 0;  // dirty: high-water mark of real allocations since the last clear (the zero arena's next memset)
    bool dry = false;
    bool overflow = false;  // a real allocation ran past the capacity (a pass that was not sized): TE_L refuses every further launch of the pass
    void reset() { used = 0; overflow = false; }
    void* take(size_t bytes) {
        const size_t o = (used + 255) & ~(size_t)255;
        used = o + bytes;
        if (dry) return reinterpret_cast<char*>((size_t)1 << 30) + o;  // never dereferenced: the dry run launches nothing
        if (used > cap) {  // never hand out memory beyond the arena: the pointer stays inside it and the pass fails with FCL_ERR_WORKSPACE
            overflow = true;
            return base;
        }
        if (used > dirty) dirty = used;
        return base + o;
    }
};

// operand forms of the parameters: ops.DerivedForms (fcl_derive_batch), native
struct Form {
    const float* src = nullptr;
    const float* src2 = nullptr;
    int a = 0, b = 0, c = 0, sa = 0, sb = 0, sc = 0;
    float* dst = nullptr;
    uint16_t* dst_p = nullptr;
    int blocks = 0;
};

struct ConvBn {  // cache of one Conv1d -> BatchNorm -> act -> Dropout block (train form)
    const float* x = nullptr;
    float *z = nullptr, *y_act = nullptr, *mean = nullptr, *invstd = nullptr;
    std::string prefix;
    int act = 0, m = 0, cin = 0, cout = 0, k = 0;
    const int32_t *lo = nullptr, *hi = nullptr;
    const uint8_t* keep = nullptr;
    float ks = 1.f;
};
struct ConvRelu {  // predictor block: Conv1d(bias) -> ReLU
    const float* x = nullptr;
    float* y = nullptr;
    std::string prefix;
    int m = 0, cin = 0, cout = 0, k = 0;
    const int32_t *lo = nullptr, *hi = nullptr;
};
struct PredLayer {
    ConvRelu cc;
    const uint8_t* keep = nullptr;
    float ks = 1.f;
    bool last = false;
    int i = 0;
};
struct Pred {
    std::string name;
    std::vector<PredLayer> layers;
    float* out = nullptr;  // [m] scalar head
};
struct Bilstm {
    const float* x = nullptr;
    float* s[2][4] = {{nullptr}};
    int B = 0, T = 0;
};

struct Ctx {  // one forward's tensors (training.py's _Ctx)
    fcl_te_batch_t b{};
    bool save = true;
    uint32_t draw = 0;
    float *emb = nullptr, *hs = nullptr;
    std::vector<ConvBn> conv_c, post_c;
    std::vector<float*> enc_taps, post_taps;  // [econv_layers + 1], [postnet_layers]
    Bilstm bl;
    Pred dur, pit, en;
    float *p_embs = nullptr, *e_embs = nullptr;
    const uint8_t* emb_keep[2] = {nullptr, nullptr};
    float emb_ks = 1.f;
    float *att_c = nullptr, *pre_in = nullptr, *p0 = nullptr, *p0d = nullptr, *p1 = nullptr, *p1d = nullptr;
    const uint8_t *k0 = nullptr, *k1 = nullptr;
    const uint8_t* zk[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    float pks = 1.f;
    float* S0[4] = {nullptr};
    float* S1[4] = {nullptr};
    float *h0_all = nullptr, *h1_all = nullptr, *before = nullptr, *after = nullptr;
    // losses
    double* sums = nullptr;
    std::unordered_map<std::string, float*> inj;
    // backward carries between stages
    float *d_post_in = nullptr, *d_att_c = nullptr, *d_att = nullptr;
    float* d_preds[3] = {nullptr, nullptr, nullptr};
    // P32 planes the forward already wrote for a tap (embedding, conv blocks, BiLSTM output, prenet): the KD projections read them instead of
    // packing the fp32 tap once more (round 5: 9 of 14 fcl_pack_planes launches of a KD update)
    std::unordered_map<const float*, const uint16_t*> planes_of;
};

}  // namespace te
}  // namespace fcl

using namespace fcl;
using namespace fcl::te;

static const char* const SITE_NAMES[] = {
    // (name, index) pairs as training.py spells them: ("enc.convs", i) ... -- listed with a fixed maximum of 8 layers per stack
    "enc.convs/0", "enc.convs/1", "enc.convs/2", "enc.convs/3", "enc.convs/4", "enc.convs/5", "enc.convs/6", "enc.convs/7",
    "duration_predictor/0", "duration_predictor/1", "duration_predictor/2", "duration_predictor/3",
    "pitch_predictor/0", "pitch_predictor/1", "pitch_predictor/2", "pitch_predictor/3",
    "energy_predictor/0", "energy_predictor/1", "energy_predictor/2", "energy_predictor/3",
    "pitch_embed", "energy_embed", "prenet/0", "prenet/1", "zoneout/0/0", "zoneout/0/1", "zoneout/1/0", "zoneout/1/1",
    "postnet/0", "postnet/1", "postnet/2", "postnet/3", "postnet/4", "postnet/5", "postnet/6", "postnet/7"};
enum { SITE_ENC = 0, SITE_DUR = 8, SITE_PIT = 12, SITE_EN = 16, SITE_PEMB = 20, SITE_EEMB = 21, SITE_PRE = 22, SITE_ZONE = 24, SITE_POST = 28, N_SITES = 36 };
static_assert(sizeof(SITE_NAMES) / sizeof(SITE_NAMES[0]) == N_SITES && N_SITES <= FCL_TE_MAX_SITES, "site table");

// loss slots: (sum |d|, sum d^2, count) rows, named as training.py's LossReport expects them
static const char* const LOSS_NAMES[] = {"after", "before", "dur", "pitch", "energy", "o_after", "o_before", "enc0", "enc1", "enc2", "enc3", "enc4",
                                         "dec0", "dec1", "dec2", "dec3", "dec4", "dec5", "dec6", "dec7", "pro0", "pro1", "pro2", "pro3", "pro4"};
enum { N_LOSSES = 25 };
static int loss_slot(const char* name) {
    for (int i = 0; i < N_LOSSES; ++i)
        if (!strcmp(LOSS_NAMES[i], name)) return i;
    return -1;
}

struct fcl_te {
    fcl_te_config_t cfg{};
    std::unordered_map<std::string, Param> P;
    std::unordered_map<std::string, float*> B;
    bool finalized = false;
    uint32_t* status = nullptr;
    // streams / events
    hipStream_t main = nullptr, side = nullptr, cur = nullptr;
    std::vector<hipEvent_t> evpool;
    size_t evnext = 0;
    hipEvent_t pred_ev = nullptr, late_ev = nullptr;
    bool pred_pending = false, late_pending = false, dw_pending = false;
    bool forms_new = false;  // a sizing run registered operand forms: the real pass starts by deriving them (forms_refresh)
    // arenas: work (two alternating for the frozen teacher: its knowledge outlives the call), zero
    Arena work[2], zero[2];
    int cur_arena = 0, n_arenas = 1;
    double* bn_ws[2] = {nullptr, nullptr};  // zero workspaces of fcl_bn_stats_ws_fwd: [0] main stream, [1] weight-gradient stream
    // operand forms
    std::unordered_map<std::string, Form> forms;
    std::vector<std::string> form_order;
    fcl_derive_t* table_dev = nullptr;
    int table_n = 0, table_blocks = 0;
    bool table_stale = true, params_dirty = true;
    std::vector<void*> form_allocs;
    float* cat_f = nullptr;  // [W1_hh^T ; W1_ih^T] (fcl_decoder_bptt's w1_cat_t) and its planes
    uint16_t* cat_p = nullptr;
    std::vector<std::array<long long, 7>> sized[2];  // per arena: (B, T, L, N, F, lmax, variant) of passes a dry run has verified to fit it
    bool dry = false;
    int64_t launches = 0, last_launches = 0;
    Ctx c;
    int stage_done = -1;
    // developer aid (FCL_TE_STAMPS=1): timing events on the main stream at the phase boundaries of an update (fcl_te_phase_ms)
    bool stamps = false;
    hipEvent_t stamp_ev[12] = {};
    bool stamp_set[12] = {};

    Param& Pm(const std::string& k) { return P.at(k); }
};

#define TE_L(expr)                                                                                                        \
    do {                                                                                                                  \
        ++E.launches;                                                                                                     \
        if (!E.dry) {                                                                                                     \
            if (E.work[E.cur_arena].overflow || E.zero[E.cur_arena].overflow)                                             \
                FCL_REQUIRE(false, FCL_ERR_WORKSPACE, "fcl_te: the pass needs more arena memory than its sizing run found"); \
            const int rc_ = (expr);                                                                                       \
            if (rc_) return rc_;                                                                                          \
        }                                                                                                                 \
    } while (0)
#define TE_TRY(expr)                \
    do {                            \
        const int rc_ = (expr);     \
        if (rc_) return rc_;        \
    } while (0)

namespace {

typedef fcl_te E_t;

inline Arena& WA(E_t& E) { return E.work[E.cur_arena]; }
inline Arena& ZA(E_t& E) { return E.zero[E.cur_arena]; }
inline float* f32(E_t& E, long long rows, long long cols = 1) { return static_cast<float*>(WA(E).take((size_t)rows * cols * 4)); }
inline uint8_t* u8(E_t& E, long long n) { return static_cast<uint8_t*>(WA(E).take((size_t)n)); }
inline size_t planes_elems(long long rows, int cols) { return (size_t)(rows > 0 ? rows : 1) * ((cols + 31) / 32) * 64; }
inline uint16_t* pl16(E_t& E, long long rows, int cols) { return static_cast<uint16_t*>(WA(E).take(planes_elems(rows, cols) * 2)); }
inline float* zf32(E_t& E, long long n) { return static_cast<float*>(ZA(E).take((size_t)n * 4)); }

inline void stamp(E_t& E, int i) {  // FCL_TE_STAMPS=1: 0 start | 1 encoder | 2 prenet + hoists | 3 decoder cells | 4 forward end | 5 losses | 6 .. 9 backward stages 0 .. 3 | 10 join
    if (!E.stamps || E.dry || i < 0 || i >= 12) return;
    if (!E.stamp_ev[i] && hipEventCreate(&E.stamp_ev[i]) != hipSuccess) return;
    E.stamp_set[i] = hipEventRecord(E.stamp_ev[i], E.main) == hipSuccess;
}

int ev_wait(E_t& E, hipStream_t waiter, hipStream_t on) {  // waiter waits for everything enqueued on `on` so far
    if (E.dry || waiter == on) return 0;
    hipEvent_t ev = E.evpool[E.evnext++ % E.evpool.size()];
    FCL_HIP(hipEventRecord(ev, on));
    FCL_HIP(hipStreamWaitEvent(waiter, ev, 0));
    return 0;
}

// ---- operand forms -------------------------------------------------------------------------------------------------------------------------
int form_get(E_t& E, const std::string& key0, const float* src, const float* src2, int a, int b, int c, int sa, int sb, int sc, bool want_f32, bool want_p,
             Form** out, float* dst_fixed = nullptr, uint16_t* dstp_fixed = nullptr) {
    const std::string key = key0 + (want_f32 ? (want_p ? "#fp" : "#f") : "#p");  // one entry per (matrix, output flavour)
    auto it = E.forms.find(key);
    if (it != E.forms.end()) {
        *out = &it->second;
        return 0;
    }
    // Round 6: a form first asked for during the SIZING run of a pass is registered there -- its memory allocated, its descriptor queued for the batched
    // fcl_derive_batch launch that opens the real pass (forms_refresh) -- instead of on the fly in the middle of the step (hipMalloc + a synchronous descriptor
    // upload + a one-entry launch per form: a first step used to carry ~60 of those between its kernels, on whichever stream was current).
    Form f;
    f.src = src; f.src2 = src2; f.a = a; f.b = b; f.c = c; f.sa = sa; f.sb = sb; f.sc = sc;
    f.blocks = fcl_derive_blocks(a, b, c);
    FCL_REQUIRE(f.blocks > 0, FCL_ERR_SHAPE, "fcl_te: bad derived-form geometry for %s", key.c_str());
    if (want_f32) {
        if (dst_fixed) f.dst = dst_fixed;
        else {
            void* p = nullptr;
            FCL_HIP(hipMalloc(&p, (size_t)a * b * c * 4 + 256));
            E.form_allocs.push_back(p);
            f.dst = static_cast<float*>(p);
        }
    }
    if (want_p) {
        if (dstp_fixed) f.dst_p = dstp_fixed;
        else {
            void* p = nullptr;
            FCL_HIP(hipMalloc(&p, planes_elems((long long)a * b, c) * 2 + 256));
            E.form_allocs.push_back(p);
            f.dst_p = static_cast<uint16_t*>(p);
        }
    }
    if (E.dry) {
        E.forms[key] = f;
        E.form_order.push_back(key);
        E.table_stale = true;
        E.forms_new = true;
        *out = &E.forms[key];
        return 0;
    }
    // computed on the spot by a one-entry table (synchronous upload: first use only); part of the batched table from the next refresh on
    fcl_derive_t d{};
    d.src = f.src; d.src2 = f.src2; d.dst = f.dst; d.dst_p = f.dst_p; d.a = a; d.b = b; d.c = c; d.sa = sa; d.sb = sb; d.sc = sc; d.first_block = 0;
    void* one = nullptr;
    FCL_HIP(hipMalloc(&one, sizeof(d)));
    E.form_allocs.push_back(one);
    FCL_HIP(hipMemcpy(one, &d, sizeof(d), hipMemcpyHostToDevice));
    ++E.launches;
    TE_TRY(fcl_derive_batch(static_cast<const fcl_derive_t*>(one), 1, f.blocks, E.cur));
    E.forms[key] = f;
    E.form_order.push_back(key);
    E.table_stale = true;
    *out = &E.forms[key];
    return 0;
}

int forms_refresh(E_t& E) {  // once per parameter update: every registered form in ONE launch
    if (E.dry || !E.params_dirty) return 0;
    E.params_dirty = false;
    if (E.form_order.empty()) return 0;
    if (E.table_stale) {
        std::vector<fcl_derive_t> t(E.form_order.size());
        int first = 0;
        for (size_t i = 0; i < t.size(); ++i) {
            const Form& f = E.forms[E.form_order[i]];
            t[i] = fcl_derive_t{f.src, f.src2, f.dst, f.dst_p, f.a, f.b, f.c, f.sa, f.sb, f.sc, first, 0};
            first += f.blocks;
        }
        if (E.table_dev) E.form_allocs.push_back(E.table_dev);  // freed with the engine (a launch may still read the old table)
        void* p = nullptr;
        FCL_HIP(hipMalloc(&p, t.size() * sizeof(fcl_derive_t)));
        FCL_HIP(hipMemcpy(p, t.data(), t.size() * sizeof(fcl_derive_t), hipMemcpyHostToDevice));
        E.table_dev = static_cast<fcl_derive_t*>(p);
        E.table_n = (int)t.size();
        E.table_blocks = first;
        E.table_stale = false;
    }
    ++E.launches;
    return fcl_derive_batch(E.table_dev, E.table_n, E.table_blocks, E.cur);
}

// shapes of the parameters the forms are cut from
struct Shape3 { int a, b, c; };

// P32 planes of a whole 2-D parameter [rows, cols]
int w_planes(E_t& E, const std::string& name, int rows, int cols, const uint16_t** out) {
    Form* f;
    TE_TRY(form_get(E, "p/" + name, E.Pm(name).p, nullptr, 1, rows, cols, 0, cols, 1, false, true, &f));
    *out = f->dst_p;
    return 0;
}
// transpose of a 2-D parameter block: src[r, c] (row stride ld, starting at column col0) -> [c, r]; fp32 and / or planes
int w_t(E_t& E, const std::string& key, const float* src, int rows, int cols, int ld, bool want_f32, bool want_p, const float** out, const uint16_t** outp) {
    Form* f;
    TE_TRY(form_get(E, "t/" + key, src, nullptr, 1, cols, rows, 0, 1, ld, want_f32, want_p, &f));
    if (out) *out = f->dst;
    if (outp) *outp = f->dst_p;
    return 0;
}
// contiguous copy of the column block w[:, col0 : col0 + n] of a [rows, ld] parameter
int w_cols(E_t& E, const std::string& name, int rows, int ld, int col0, int n, bool want_p, const float** out, const uint16_t** outp) {
    Form* f;
    char k[32];
    snprintf(k, sizeof(k), "/%d/%d", col0, n);
    TE_TRY(form_get(E, "c/" + name + k, E.Pm(name).p + col0, nullptr, 1, rows, n, 0, ld, 1, true, want_p, &f));
    if (out) *out = f->dst;
    if (outp) *outp = f->dst_p;
    return 0;
}
int w_bsum(E_t& E, const std::string& n1, const std::string& n2, int n, const float** out) {
    Form* f;
    TE_TRY(form_get(E, "bs/" + n1, E.Pm(n1).p, E.Pm(n2).p, 1, 1, n, 0, 0, 1, true, false, &f));
    *out = f->dst;
    return 0;
}
// Conv1d weight [cout, cin, k]: packed taps [k, cout, cin] (forward) and reversed, transposed taps [k, cin, cout] (input gradient)
int conv_cp(E_t& E, const std::string& name, int cout, int cin, int k, bool want_f32, bool want_p, const float** out, const uint16_t** outp) {
    Form* f;
    TE_TRY(form_get(E, "cp/" + name, E.Pm(name).p, nullptr, k, cout, cin, 1, cin * k, k, want_f32, want_p, &f));
    if (out) *out = f->dst;
    if (outp) *outp = f->dst_p;
    return 0;
}
int conv_ct(E_t& E, const std::string& name, int cout, int cin, int k, bool want_f32, bool want_p, const float** out, const uint16_t** outp) {
    Form* f;
    TE_TRY(form_get(E, "ct/" + name, E.Pm(name).p + (k - 1), nullptr, k, cin, cout, -1, k, cin * k, want_f32, want_p, &f));
    if (out) *out = f->dst;
    if (outp) *outp = f->dst_p;
    return 0;
}

// ---- forks and joins (training.py _dw / _join_dw / _pred_fork / _late_fork) ----------------------------------------------------------------------
struct SideScope {  // everything inside runs on the weight-gradient stream, ordered behind the main stream's position at entry
    E_t& E;
    hipStream_t saved;
    int rc;
    explicit SideScope(E_t& e) : E(e), saved(e.cur), rc(0) {
        rc = ev_wait(E, E.side, E.main);
        E.cur = E.side;
    }
    ~SideScope() { E.cur = saved; }
};
int join_dw(E_t& E) {
    if (E.dw_pending) {
        TE_TRY(ev_wait(E, E.main, E.side));
        E.dw_pending = false;
    }
    return 0;
}
int pred_join(E_t& E) {
    if (E.pred_pending) {
        if (!E.dry) FCL_HIP(hipStreamWaitEvent(E.main, E.pred_ev, 0));
        E.pred_pending = false;
    }
    return 0;
}
int late_join(E_t& E) {
    if (E.late_pending) {
        if (!E.dry) FCL_HIP(hipStreamWaitEvent(E.main, E.late_ev, 0));
        E.late_pending = false;
    }
    return 0;
}
inline double* bnws(E_t& E) { return E.bn_ws[E.cur == E.side ? 1 : 0]; }

// ---- masks ------------------------------------------------------------------------------------------------------------------------------------
struct Site { int id; long long n; float p_one; uint8_t* out; };
int draw_masks(E_t& E, std::vector<Site>& sites) {
    if (sites.empty()) return 0;
    size_t tot = 0;
    std::vector<size_t> off(sites.size());
    for (size_t i = 0; i < sites.size(); ++i) {
        off[i] = tot;
        tot += ((size_t)sites[i].n + 255) / 256 * 256;
    }
    uint8_t* buf = u8(E, (long long)tot);
    for (size_t k0 = 0; k0 < sites.size(); k0 += FCL_BERNOULLI_MAX_SITES) {
        fcl_bernoulli_site_t arr[FCL_BERNOULLI_MAX_SITES];
        const int n = (int)std::min<size_t>(FCL_BERNOULLI_MAX_SITES, sites.size() - k0);
        for (int j = 0; j < n; ++j) {
            Site& s = sites[k0 + j];
            s.out = buf + off[k0 + j];
            arr[j].out = s.out;
            arr[j].n = s.n;
            arr[j].p_one = s.p_one;
            arr[j].seed = (uint32_t)((uint64_t)E.cfg.seed * 7919u + (uint64_t)E.c.draw * 104729u + (uint64_t)E.cfg.site_tag[s.id]);
        }
        TE_L(fcl_bernoulli_batch(arr, n, E.cur));
    }
    return 0;
}

// ---- layers ---------------------------------------------------------------------------------------------------------------------------------------
// Conv1d(no bias) -> BatchNorm(batch statistics) -> act -> Dropout: training.py _conv_bn_fwd, train form, pre-split operands
int conv_bn_fwd(E_t& E, const float* x, const uint16_t* xp, int m, const std::string& prefix, int cout, int cin, int k, const int32_t* lo, const int32_t* hi,
                int act, const uint8_t* keep, float p_drop, bool want_planes, float** y_out, uint16_t** yp_out, ConvBn* cc) {
    const std::string wn = prefix + ".0.weight";
    const uint16_t* wpp;
    TE_TRY(conv_cp(E, wn, cout, cin, k, false, true, nullptr, &wpp));
    float* z = f32(E, m, cout);
    float *mean = f32(E, cout), *invstd = f32(E, cout);
    // FCL_TE_BN_FUSED=1 (round 6, VERDICT r5 #1b): the batch statistics come out of the convolution's epilogue (fcl_conv1d_planes_bn_fwd: same fp64 sums, same ticketed
    // finalize) -- one launch and one pass over z [m, cout] less per block
    static const int bn_fused = tunable("TE_BN_FUSED", 1);
    if (bn_fused && tunable("PRECISION", 1) != 0 && tunable("PLANES", 1) != 0) {
        TE_L(fcl_conv1d_planes_bn_fwd(xp, (cin + 31) / 32, wpp, lo, hi, z, m, cin, cout, k, BN_EPS, BN_MOMENTUM, mean, invstd, E.B.at(prefix + ".1.running_mean"),
                                      E.B.at(prefix + ".1.running_var"), bnws(E), E.cur));
    } else {
        TE_L(fcl_conv1d_planes_fwd(xp, (cin + 31) / 32, wpp, nullptr, lo, hi, nullptr, z, nullptr, m, cin, cout, k, FCL_ACT_NONE, E.cur));
        TE_L(fcl_bn_stats_ws_fwd(z, m, cout, BN_EPS, BN_MOMENTUM, mean, invstd, E.B.at(prefix + ".1.running_mean"), E.B.at(prefix + ".1.running_var"), bnws(E), E.cur));
    }
    const float ks = keep ? 1.0f / (1.0f - p_drop) : 1.0f;
    // (round 6) a forward that saves nothing for a backward (the frozen KD teacher) and drops: the pre-dropout activation has no reader -- one [m, cout] write less per block
    float* y_act = (E.c.save || !keep) ? f32(E, m, cout) : nullptr;
    float* y_drop = keep ? f32(E, m, cout) : nullptr;
    uint16_t* yp = (want_planes && cout % 32 == 0) ? pl16(E, m, cout) : nullptr;
    TE_L(fcl_bn_act_fwd(z, mean, invstd, E.Pm(prefix + ".1.weight").p, E.Pm(prefix + ".1.bias").p, keep, ks, y_act, y_drop, yp, m, cout, act, E.cur));
    cc->x = x; cc->z = z; cc->y_act = y_act; cc->mean = mean; cc->invstd = invstd; cc->prefix = prefix; cc->act = act; cc->m = m; cc->cin = cin; cc->cout = cout;
    cc->k = k; cc->lo = lo; cc->hi = hi; cc->keep = keep; cc->ks = ks;
    *y_out = keep ? y_drop : y_act;
    *yp_out = yp;
    return 0;
}

// out += dz^T x for every pair; Conv1d: one x and a tap-major out [k, n, kk] (training.py _dw_gemm)
struct DwPair { const float* x; int kk; float* out; int ldc; };
int dw_gemm(E_t& E, const float* dz, int m, int n, const std::vector<DwPair>& pairs, int ntaps = 0, const int32_t* lo = nullptr, const int32_t* hi = nullptr) {
    long long outs = 0;
    bool mult4 = true;
    for (const DwPair& p : pairs) {
        outs += (long long)n * p.kk * (ntaps ? ntaps : 1);
        mult4 = mult4 && (p.kk % 4 == 0);
    }
    if (outs >= E.cfg.dw_planes_min && mult4 && m >= 512) {
        uint16_t* ap = static_cast<uint16_t*>(WA(E).take((size_t)n * ((m + 31) / 32) * 64 * 2));
        TE_L(fcl_pack_planes_t(dz, n, m, n, 1, 0, nullptr, nullptr, ap, E.cur));
        for (const DwPair& p : pairs) {
            const int nt = ntaps ? ntaps : 1;
            uint16_t* bp = static_cast<uint16_t*>(WA(E).take((size_t)nt * p.kk * ((m + 31) / 32) * 64 * 2));
            TE_L(fcl_pack_planes_t(p.x, p.kk, m, p.kk, nt, ntaps ? -((ntaps - 1) / 2) : 0, ntaps ? lo : nullptr, ntaps ? hi : nullptr, bp, E.cur));
            if (ntaps) TE_L(fcl_gemm_tn_planes(ap, bp, p.out, p.kk, m, n, nt * p.kk, p.kk, (size_t)n * p.kk, E.cur));
            else TE_L(fcl_gemm_tn_planes(ap, bp, p.out, p.ldc, m, n, p.kk, 0, 0, E.cur));
        }
        return 0;
    }
    for (const DwPair& p : pairs) {
        if (ntaps) TE_L(fcl_gemm_tn_taps_fwd(dz, n, p.x, p.kk, p.out, p.kk, m, n, p.kk, -((ntaps - 1) / 2), ntaps, (size_t)n * p.kk, lo, hi, E.cur));
        else TE_L(fcl_gemm_tn_fwd(dz, n, p.x, p.kk, p.out, p.ldc, m, n, p.kk, 0, nullptr, nullptr, E.cur));
    }
    return 0;
}

// input gradient of a Conv1d = the forward conv of dz with the taps reversed and transposed
int conv_dx(E_t& E, const float* dz, const uint16_t* dzp, int m, const std::string& wname, int cout, int cin, int k, const int32_t* lo, const int32_t* hi, float** dx) {
    *dx = f32(E, m, cin);
    if (dzp) {
        const uint16_t* wtp;
        TE_TRY(conv_ct(E, wname, cout, cin, k, false, true, nullptr, &wtp));
        TE_L(fcl_conv1d_planes_fwd(dzp, (cout + 31) / 32, wtp, nullptr, lo, hi, nullptr, *dx, nullptr, m, cout, cin, k, FCL_ACT_NONE, E.cur));
    } else {
        const float* wt;
        TE_TRY(conv_ct(E, wname, cout, cin, k, true, false, &wt, nullptr));
        TE_L(fcl_conv1d_fwd(dz, wt, nullptr, lo, hi, nullptr, *dx, m, cout, cin, k, FCL_ACT_NONE, E.cur));
    }
    return 0;
}

// dy2 (optional): a second source of the block's output gradient (a KD term injected at this tap), added on the fly
int conv_bn_bwd(E_t& E, const float* dy, const float* dy2, const ConvBn& cc, float** dx) {
    const std::string& pre = cc.prefix;
    const int m = cc.m, cout = cc.cout, cin = cc.cin, k = cc.k;
    const bool pl = cout % 32 == 0;
    const float* dz = dy;
    float *dbeta = zf32(E, cout), *dgamma = zf32(E, cout);
    if (cc.act != FCL_ACT_NONE || cc.keep || dy2) {  // round 6: activation / dropout backward as the prologue of the column sums (one pass, dz written once)
        float* t = f32(E, m, cout);
        TE_L(fcl_bn_bwd_sums(dy, dy2, cc.y_act, cc.keep, cc.ks, cc.act, cc.z, cc.mean, cc.invstd, t, dgamma, dbeta, m, cout, E.cur));
        dz = t;
    } else {
        TE_L(fcl_colsum2_fwd(dz, cc.z, cc.invstd, cc.mean, dgamma, dbeta, m, cout, 3, E.cur));  // both sums from one pass over dz
    }
    float* dz2 = f32(E, m, cout);
    uint16_t* dzp = pl ? pl16(E, m, cout) : nullptr;
    TE_L(fcl_bn_bwd(dz, cc.z, cc.mean, cc.invstd, E.Pm(pre + ".1.weight").p, dbeta, dgamma, dz2, dzp, m, cout, E.Pm(pre + ".1.bias").g, E.Pm(pre + ".1.weight").g, E.cur));
    {
        SideScope sc(E);
        TE_TRY(sc.rc);
        float* dwp = zf32(E, (long long)k * cout * cin);
        TE_TRY(dw_gemm(E, dz2, m, cout, {{cc.x, cin, dwp, cin}}, k, cc.lo, cc.hi));
        TE_L(fcl_unpack_conv1d_grad(dwp, nullptr, E.Pm(pre + ".0.weight").g, cout, cin, k, E.cur));
        E.dw_pending = true;
    }
    return conv_dx(E, dz2, dzp, m, pre + ".0.weight", cout, cin, k, cc.lo, cc.hi, dx);
}

int conv_relu_fwd(E_t& E, const float* x, const uint16_t* xp, int m, const std::string& prefix, int cout, int cin, int k, const int32_t* lo, const int32_t* hi,
                  ConvRelu* cc) {
    const uint16_t* wpp;
    TE_TRY(conv_cp(E, prefix + ".weight", cout, cin, k, false, true, nullptr, &wpp));
    float* y = f32(E, m, cout);
    TE_L(fcl_conv1d_planes_fwd(xp, (cin + 31) / 32, wpp, E.Pm(prefix + ".bias").p, lo, hi, nullptr, y, nullptr, m, cin, cout, k, FCL_ACT_RELU, E.cur));
    cc->x = x; cc->y = y; cc->prefix = prefix; cc->m = m; cc->cin = cin; cc->cout = cout; cc->k = k; cc->lo = lo; cc->hi = hi;
    return 0;
}

int conv_relu_bwd(E_t& E, const float* dy, const ConvRelu& cc, float** dx) {
    const int m = cc.m, cout = cc.cout, cin = cc.cin, k = cc.k;
    const bool pl = cout % 32 == 0;
    float* dz = f32(E, m, cout);
    uint16_t* dzp = pl ? pl16(E, m, cout) : nullptr;
    TE_L(fcl_act_bwd(dy, cc.y, nullptr, 1.0f, dz, dzp, pl ? cout : 0, (size_t)m * cout, FCL_ACT_RELU, E.cur));
    {
        SideScope sc(E);
        TE_TRY(sc.rc);
        TE_L(fcl_colsum2_fwd(dz, nullptr, nullptr, nullptr, E.Pm(cc.prefix + ".bias").g, nullptr, m, cout, 0, E.cur));
        float* dwp = zf32(E, (long long)k * cout * cin);
        TE_TRY(dw_gemm(E, dz, m, cout, {{cc.x, cin, dwp, cin}}, k, cc.lo, cc.hi));
        TE_L(fcl_unpack_conv1d_grad(dwp, nullptr, E.Pm(cc.prefix + ".weight").g, cout, cin, k, E.cur));
        E.dw_pending = true;
    }
    return conv_dx(E, dz, dzp, m, cc.prefix + ".weight", cout, cin, k, cc.lo, cc.hi, dx);
}

// one predictor (ESPnet DurationPredictor / variance_predictor.py): layers x {Conv1d -> ReLU -> LayerNorm -> Dropout}, Linear -> 1, masked_fill
int predictor_fwd(E_t& E, const float* hs, const uint16_t* hs_p, int m, int cin0, const std::string& name, int layers, int chans, int ksz, float p_drop,
                  const int32_t* lo, const int32_t* hi, const uint8_t* pad, const uint8_t* const* keeps, Pred* out) {
    out->name = name;
    out->layers.clear();
    const float* x = hs;
    const uint16_t* xp = hs_p;
    for (int i = 0; i < layers; ++i) {
        PredLayer pl_;
        char pre[96];
        snprintf(pre, sizeof(pre), "%s.conv.%d.0", name.c_str(), i);
        TE_TRY(conv_relu_fwd(E, x, xp, m, pre, chans, i == 0 ? cin0 : chans, ksz, lo, hi, &pl_.cc));
        const bool last = i == layers - 1;
        const uint8_t* keep = keeps ? keeps[i] : nullptr;
        const float ks = keep ? 1.0f / (1.0f - p_drop) : 1.0f;
        char g[96], b[96];
        snprintf(g, sizeof(g), "%s.conv.%d.2.weight", name.c_str(), i);
        snprintf(b, sizeof(b), "%s.conv.%d.2.bias", name.c_str(), i);
        const bool wantp = !last && chans % 32 == 0;
        float* ln = (!last && (E.c.save || !wantp)) ? f32(E, m, chans) : nullptr;
        uint16_t* lnp = wantp ? pl16(E, m, chans) : nullptr;
        float* scalar = last ? f32(E, m) : nullptr;
        TE_L(fcl_layernorm_fwd(pl_.cc.y, E.Pm(g).p, E.Pm(b).p, LN_EPS, ln, lnp, last ? E.Pm(name + ".linear.weight").p : nullptr,
                               last ? E.Pm(name + ".linear.bias").p : nullptr, last ? pad : nullptr, keep, ks, scalar, m, chans, E.cur));
        pl_.keep = keep; pl_.ks = ks; pl_.last = last; pl_.i = i;
        out->layers.push_back(pl_);
        x = ln;
        xp = lnp;
        if (last) out->out = scalar;
    }
    return 0;
}

int predictor_bwd(E_t& E, const float* d_out, const Pred& pr, const uint8_t* pad, float** dx_out) {
    float* dx = nullptr;
    const std::string& name = pr.name;
    for (int j = (int)pr.layers.size() - 1; j >= 0; --j) {
        const PredLayer& L_ = pr.layers[j];
        char g[96], b[96];
        snprintf(g, sizeof(g), "%s.conv.%d.2.weight", name.c_str(), L_.i);
        snprintf(b, sizeof(b), "%s.conv.%d.2.bias", name.c_str(), L_.i);
        const int m = L_.cc.m, c = L_.cc.cout;
        float* dy = f32(E, m, c);
        if (L_.last)
            TE_L(fcl_layernorm_bwd(L_.cc.y, E.Pm(g).p, E.Pm(b).p, LN_EPS, nullptr, E.Pm(name + ".linear.weight").p, d_out, pad, L_.keep, L_.ks, dy, E.Pm(g).g,
                                   E.Pm(b).g, E.Pm(name + ".linear.weight").g, E.Pm(name + ".linear.bias").g, m, c, E.cur));
        else
            TE_L(fcl_layernorm_bwd(L_.cc.y, E.Pm(g).p, E.Pm(b).p, LN_EPS, dx, nullptr, nullptr, nullptr, L_.keep, L_.ks, dy, E.Pm(g).g, E.Pm(b).g, nullptr, nullptr,
                                   m, c, E.cur));
        TE_TRY(conv_relu_bwd(E, dy, L_.cc, &dx));
    }
    *dx_out = dx;
    return 0;
}

}  // namespace

// ================================================================================================================================================
// forward (training.py _forward)
// ================================================================================================================================================
static int te_forward(fcl_te& E) {
    const fcl_te_config_t& cf = E.cfg;
    Ctx& c = E.c;
    const fcl_te_batch_t& b = c.b;
    const int B = b.B, T = b.T, L = b.L, N = b.N, F = b.F;
    const int O = cf.odim, U = cf.dunits, Pn = cf.prenet_units, C = cf.eunits, Ee = cf.embed_dim, Cc = cf.econv_chans;
    const int BT = B * T, BL = B * L;
    const float p_conv = cf.dropout_rate;
    const bool drop_conv = p_conv > 0.f;
    TE_TRY(forms_refresh(E));
    // ---- encoder
    c.emb = f32(E, BT, Ee);
    uint16_t* xp = pl16(E, BT, Ee);
    TE_L(fcl_embedding_fwd(b.xs, E.Pm("enc.embed.weight").p, c.emb, xp, BT, cf.idim, Ee, E.cur));
    c.planes_of.clear();
    c.planes_of[c.emb] = xp;
    const float* x = c.emb;
    c.conv_c.assign(cf.econv_layers, ConvBn());
    c.enc_taps.assign(1, c.emb);
    std::vector<Site> sites;
    if (drop_conv) {
        for (int i = 0; i < cf.econv_layers; ++i) sites.push_back(Site{SITE_ENC + i, (long long)BT * Cc, 1.0f - p_conv, nullptr});
        TE_TRY(draw_masks(E, sites));
    }
    for (int i = 0; i < cf.econv_layers; ++i) {
        char pre[64];
        snprintf(pre, sizeof(pre), "enc.convs.%d", i);
        float* y;
        uint16_t* yp;
        TE_TRY(conv_bn_fwd(E, x, xp, BT, pre, Cc, i == 0 ? Ee : Cc, cf.econv_filts, b.e_lo, b.e_hi, FCL_ACT_RELU, drop_conv ? sites[i].out : nullptr, p_conv, true, &y,
                           &yp, &c.conv_c[i]));
        x = y;
        xp = yp;
        if (yp) c.planes_of[y] = yp;
        c.enc_taps.push_back(y);
    }
    // ---- BiLSTM (packed sequences)
    const int H = C / 2;
    uint16_t* hs_p = nullptr;
    {
        const uint16_t *wip_f, *wip_r;
        TE_TRY(w_planes(E, "enc.blstm.weight_ih_l0", 4 * H, Cc, &wip_f));
        TE_TRY(w_planes(E, "enc.blstm.weight_ih_l0_reverse", 4 * H, Cc, &wip_r));
        const float *bf, *br;
        TE_TRY(w_bsum(E, "enc.blstm.bias_ih_l0", "enc.blstm.bias_hh_l0", 4 * H, &bf));
        TE_TRY(w_bsum(E, "enc.blstm.bias_ih_l0_reverse", "enc.blstm.bias_hh_l0_reverse", 4 * H, &br));
        c.hs = f32(E, BT, 2 * H);
        if (!c.save) {  // forward only (the frozen KD teacher): the persistent / cooperating-workgroup recurrence of the synthesis path
            const size_t nb = fcl_bilstm_workspace_bytes(B, T, H);
            void* ws = WA(E).take(nb);
            hs_p = pl16(E, BT, 2 * H);
            TE_L(fcl_bilstm_fwd(nullptr, b.lens, E.Pm("enc.blstm.weight_ih_l0").p, E.Pm("enc.blstm.weight_hh_l0").p, bf, E.Pm("enc.blstm.weight_ih_l0_reverse").p,
                                E.Pm("enc.blstm.weight_hh_l0_reverse").p, br, c.hs, hs_p, xp, wip_f, wip_r, B, T, Cc, H, H == 256 ? 3 : 0, ws, nb, E.status, nullptr,
                                E.cur));
        } else {
            fcl_bilstm_train_t a{};
            a.b = B; a.t = T; a.h = H; a.lens = b.lens; a.out = c.hs; a.status = E.status;
            for (int d = 0; d < 2; ++d) {
                float* gx = f32(E, BT, 4 * H);
                TE_L(fcl_linear_planes_fwd(xp, (Cc + 31) / 32, d ? wip_r : wip_f, d ? br : bf, gx, 4 * H, nullptr, BT, 4 * H, Cc, FCL_ACT_NONE, E.cur));
                a.gx[d] = gx;
                a.w_hh[d] = E.Pm(d ? "enc.blstm.weight_hh_l0_reverse" : "enc.blstm.weight_hh_l0").p;
                // gates, c_new, c_old, h_old (t-major); zero-filled: dead cells are never written but are read by the batched weight-gradient GEMM
                a.s[d][0] = zf32(E, (long long)T * B * 4 * H);
                for (int q = 1; q < 4; ++q) a.s[d][q] = zf32(E, (long long)T * B * H);
                for (int q = 0; q < 4; ++q) c.bl.s[d][q] = a.s[d][q];
            }
            a.workspace_bytes = fcl_bilstm_train_workspace_bytes(B, H);
            a.workspace = WA(E).take(a.workspace_bytes);
            TE_L(fcl_bilstm_train_fwd(&a, E.cur));
            c.bl.x = x; c.bl.B = B; c.bl.T = T;
            hs_p = pl16(E, BT, 2 * H);
            TE_L(fcl_pack_planes(c.hs, 2 * H, BT, 2 * H, hs_p, E.cur));
        }
    }
    if (hs_p) c.planes_of[c.hs] = hs_p;
    c.enc_taps.push_back(c.hs);
    stamp(E, 1);
    // ---- predictors + embeds: their dropout masks in one launch, their forward beside the decoder's (weight-gradient stream) when a backward follows
    const float p_emb = cf.ve_dropout;
    std::vector<Site> ps;
    auto add_pred_sites = [&](int base, int layers, int chans, float pd) {
        if (pd > 0.f)
            for (int i = 0; i < layers; ++i) ps.push_back(Site{base + i, (long long)BT * chans, 1.0f - pd, nullptr});
    };
    add_pred_sites(SITE_DUR, cf.dp_layers, cf.dp_chans, cf.dp_dropout);
    add_pred_sites(SITE_PIT, cf.vp_layers, cf.vp_chans, cf.vp_dropout);
    add_pred_sites(SITE_EN, cf.vp_layers, cf.vp_chans, cf.vp_dropout);
    if (p_emb > 0.f) {
        ps.push_back(Site{SITE_PEMB, (long long)BT * C, 1.0f - p_emb, nullptr});
        ps.push_back(Site{SITE_EEMB, (long long)BT * C, 1.0f - p_emb, nullptr});
    }
    TE_TRY(draw_masks(E, ps));
    auto keep_of = [&](int id) -> const uint8_t* {
        for (const Site& s : ps)
            if (s.id == id) return s.out;
        return nullptr;
    };
    {
        const bool fork = cf.pred_stream && c.save;
        hipStream_t saved = E.cur;
        if (fork) {
            TE_TRY(ev_wait(E, E.side, E.main));
            E.cur = E.side;
        }
        const uint8_t* kd_[4];
        for (int i = 0; i < cf.dp_layers; ++i) kd_[i] = keep_of(SITE_DUR + i);
        TE_TRY(predictor_fwd(E, c.hs, hs_p, BT, C, "duration_predictor", cf.dp_layers, cf.dp_chans, cf.dp_kernel, cf.dp_dropout, b.e_lo, b.e_hi, b.enc_pad,
                             cf.dp_dropout > 0.f ? kd_ : nullptr, &c.dur));
        for (int i = 0; i < cf.vp_layers; ++i) kd_[i] = keep_of(SITE_PIT + i);
        TE_TRY(predictor_fwd(E, c.hs, hs_p, BT, C, "pitch_predictor", cf.vp_layers, cf.vp_chans, cf.vp_kernel, cf.vp_dropout, b.e_lo, b.e_hi, b.enc_pad,
                             cf.vp_dropout > 0.f ? kd_ : nullptr, &c.pit));
        for (int i = 0; i < cf.vp_layers; ++i) kd_[i] = keep_of(SITE_EN + i);
        TE_TRY(predictor_fwd(E, c.hs, hs_p, BT, C, "energy_predictor", cf.vp_layers, cf.vp_chans, cf.vp_kernel, cf.vp_dropout, b.e_lo, b.e_hi, b.enc_pad,
                             cf.vp_dropout > 0.f ? kd_ : nullptr, &c.en));
        if (fork) {
            if (!E.dry) FCL_HIP(hipEventRecord(E.pred_ev, E.side));  // (the join waits for the predictors, not for weight gradients queued behind them)
            E.pred_pending = true;
            E.cur = saved;
        }
    }
    float* att = f32(E, BT, C);
    float *pe = f32(E, BT, C), *ee = f32(E, BT, C);
    const int kk = cf.ve_kernel;
    TE_L(fcl_variance_embed_add_fwd(c.hs, b.f0, b.energy, E.Pm("pitch_embed.0.weight").p, E.Pm("pitch_embed.0.bias").p, E.Pm("energy_embed.0.weight").p,
                                    E.Pm("energy_embed.0.bias").p, b.e_lo, b.e_hi, att, pe, ee, BT, C, kk, E.cur));
    c.emb_keep[0] = c.emb_keep[1] = nullptr;
    c.emb_ks = 1.f;
    if (p_emb > 0.f) {
        c.emb_keep[0] = keep_of(SITE_PEMB);
        c.emb_keep[1] = keep_of(SITE_EEMB);
        c.emb_ks = 1.0f / (1.0f - p_emb);
        float *pe2 = f32(E, BT, C), *ee2 = f32(E, BT, C);
        TE_L(fcl_act_fwd(pe, c.emb_keep[0], c.emb_ks, pe2, nullptr, 0, (size_t)BT * C, FCL_ACT_NONE, E.cur));
        TE_L(fcl_act_fwd(ee, c.emb_keep[1], c.emb_ks, ee2, nullptr, 0, (size_t)BT * C, FCL_ACT_NONE, E.cur));
        pe = pe2;
        ee = ee2;
        att = f32(E, BT, C);
        TE_L(fcl_copy2d(att, C, c.hs, C, BT, C, E.cur));
        TE_L(fcl_add2d(att, C, pe, C, BT, C, 1.0f, nullptr, E.cur));
        TE_L(fcl_add2d(att, C, ee, C, BT, C, 1.0f, nullptr, E.cur));
    }
    c.p_embs = pe;
    c.e_embs = ee;
    // ---- decoder, teacher forced, step-major cells
    c.att_c = f32(E, N, C);
    uint16_t* att_p = pl16(E, N, C);
    TE_L(fcl_gather_rows_fwd(att, b.src_sorted, c.att_c, att_p, N, C, E.cur));
    c.pre_in = f32(E, F, O);
    uint16_t* pre_in_p = pl16(E, F, O);
    TE_L(fcl_gather_rows_fwd(b.ys, b.prev_frame, c.pre_in, pre_in_p, F, O, E.cur));  // idx -1 -> zero row
    c.k0 = c.k1 = nullptr;
    c.pks = 1.f;
    for (int l = 0; l < 2; ++l) c.zk[l][0] = c.zk[l][1] = nullptr;
    const float zr = cf.zoneout_rate;
    if (cf.dropout_rate > 0.f) {  // the prenet's dropout is on in BOTH modes (decoder_sa.py:156-158)
        c.pks = 1.0f / (1.0f - cf.dropout_rate);
        std::vector<Site> ds;
        ds.push_back(Site{SITE_PRE + 0, (long long)F * Pn, 1.0f - cf.dropout_rate, nullptr});
        ds.push_back(Site{SITE_PRE + 1, (long long)F * Pn, 1.0f - cf.dropout_rate, nullptr});
        if (zr > 0.f)
            for (int l = 0; l < 2; ++l)
                for (int j = 0; j < 2; ++j) ds.push_back(Site{SITE_ZONE + 2 * l + j, (long long)F * U, zr, nullptr});
        TE_TRY(draw_masks(E, ds));
        c.k0 = ds[0].out;
        c.k1 = ds[1].out;
        if (ds.size() == 6)
            for (int l = 0; l < 2; ++l)
                for (int j = 0; j < 2; ++j) c.zk[l][j] = ds[2 + 2 * l + j].out;
    } else if (zr > 0.f) {
        std::vector<Site> ds;
        for (int l = 0; l < 2; ++l)
            for (int j = 0; j < 2; ++j) ds.push_back(Site{SITE_ZONE + 2 * l + j, (long long)F * U, zr, nullptr});
        TE_TRY(draw_masks(E, ds));
        for (int l = 0; l < 2; ++l)
            for (int j = 0; j < 2; ++j) c.zk[l][j] = ds[2 * l + j].out;
    }
    const char *w0n = "dec.prenet.prenet.0.0.weight", *b0n = "dec.prenet.prenet.0.0.bias", *w1n = "dec.prenet.prenet.1.0.weight", *b1n = "dec.prenet.prenet.1.0.bias";
    const uint16_t *w0p, *w1p;
    TE_TRY(w_planes(E, w0n, Pn, O, &w0p));
    TE_TRY(w_planes(E, w1n, Pn, Pn, &w1p));
    c.p0 = f32(E, F, Pn);
    TE_L(fcl_linear_planes_fwd(pre_in_p, (O + 31) / 32, w0p, E.Pm(b0n).p, c.p0, Pn, nullptr, F, Pn, O, FCL_ACT_RELU, E.cur));  // pre-dropout activations are kept
    c.p0d = f32(E, F, Pn);
    uint16_t* p0d_p = pl16(E, F, Pn);
    TE_L(fcl_act_fwd(c.p0, c.k0, c.pks, c.p0d, p0d_p, Pn, (size_t)F * Pn, FCL_ACT_NONE, E.cur));
    c.p1 = f32(E, F, Pn);
    TE_L(fcl_linear_planes_fwd(p0d_p, Pn / 32, w1p, E.Pm(b1n).p, c.p1, Pn, nullptr, F, Pn, Pn, FCL_ACT_RELU, E.cur));
    c.p1d = f32(E, F, Pn);
    uint16_t* p1d_p = pl16(E, F, Pn);
    TE_L(fcl_act_fwd(c.p1, c.k1, c.pks, c.p1d, p1d_p, Pn, (size_t)F * Pn, FCL_ACT_NONE, E.cur));
    c.planes_of[c.p1d] = p1d_p;
    const std::string wih0 = "dec.lstm.0.cell.weight_ih";
    const int ld0 = C + Pn + 1;
    const float *w0_att, *w0_pre, *w0_pos, *b0s, *b1s, *wf_h, *wf_att;
    const uint16_t *w0_att_p, *w0_pre_p, *wf_h_p, *wf_att_p, *w0_hh_p, *w1_ih_p, *w1_hh_p;
    TE_TRY(w_cols(E, wih0, 4 * U, ld0, 0, C, true, &w0_att, &w0_att_p));
    TE_TRY(w_cols(E, wih0, 4 * U, ld0, C, Pn, true, &w0_pre, &w0_pre_p));
    TE_TRY(w_cols(E, wih0, 4 * U, ld0, C + Pn, 1, false, &w0_pos, nullptr));
    TE_TRY(w_bsum(E, "dec.lstm.0.cell.bias_ih", "dec.lstm.0.cell.bias_hh", 4 * U, &b0s));
    TE_TRY(w_bsum(E, "dec.lstm.1.cell.bias_ih", "dec.lstm.1.cell.bias_hh", 4 * U, &b1s));
    TE_TRY(w_cols(E, "dec.feat_out.weight", O, U + C, 0, U, true, &wf_h, &wf_h_p));
    TE_TRY(w_cols(E, "dec.feat_out.weight", O, U + C, U, C, true, &wf_att, &wf_att_p));
    TE_TRY(w_planes(E, "dec.lstm.0.cell.weight_hh", 4 * U, U, &w0_hh_p));
    TE_TRY(w_planes(E, "dec.lstm.1.cell.weight_ih", 4 * U, U, &w1_ih_p));
    TE_TRY(w_planes(E, "dec.lstm.1.cell.weight_hh", 4 * U, U, &w1_hh_p));
    float* G0 = f32(E, N, 4 * U);
    TE_L(fcl_linear_planes_fwd(att_p, C / 32, w0_att_p, b0s, G0, 4 * U, nullptr, N, 4 * U, C, FCL_ACT_NONE, E.cur));  // hoisted att_c share of the layer-0 gates
    float* F0 = f32(E, N, O);
    TE_L(fcl_linear_planes_fwd(att_p, C / 32, wf_att_p, nullptr, F0, O, nullptr, N, O, C, FCL_ACT_NONE, E.cur));
    for (int q = 0; q < 4; ++q) {  // gates, c_new, c_old, h_old: what the BPTT reads -- nothing of it for a forward without a backward (the frozen KD teacher)
        c.S0[q] = c.save ? f32(E, F, q == 0 ? 4 * U : U) : nullptr;
        c.S1[q] = c.save ? f32(E, F, q == 0 ? 4 * U : U) : nullptr;
    }
    c.h0_all = f32(E, F, U);
    c.h1_all = f32(E, F, U);
    {
        fcl_decoder_train_t a{};
        a.n = N; a.lmax = b.lmax; a.u = U; a.p = Pn; a.live_rows_host = b.live_rows_host; a.p1d = c.p1d; a.g0 = G0; a.w0_pre = w0_pre;
        a.w0_hh = E.Pm("dec.lstm.0.cell.weight_hh").p; a.w0_pos = w0_pos; a.dur = b.dur; a.w1_ih = E.Pm("dec.lstm.1.cell.weight_ih").p;
        a.w1_hh = E.Pm("dec.lstm.1.cell.weight_hh").p; a.b1 = b1s; a.zoneout = zr;
        a.zk_h0 = c.zk[0][0]; a.zk_c0 = c.zk[0][1]; a.zk_h1 = c.zk[1][0]; a.zk_c1 = c.zk[1][1];
        for (int q = 0; q < 4; ++q) { a.s0[q] = c.S0[q]; a.s1[q] = c.S1[q]; }
        a.h0_all = c.h0_all; a.h1_all = c.h1_all;
        if (((long long)N * U * 4) % 128 == 0) { a.p1d_p = p1d_p; a.w0_pre_p = w0_pre_p; a.w0_hh_p = w0_hh_p; a.w1_ih_p = w1_ih_p; a.w1_hh_p = w1_hh_p; }
        a.workspace_bytes = fcl_decoder_train_workspace_bytes(N, U);
        a.workspace = WA(E).take(a.workspace_bytes);
        stamp(E, 2);
        TE_L(fcl_decoder_train_fwd(&a, E.cur));
        stamp(E, 3);
    }
    uint16_t* h1_p = pl16(E, F, U);
    TE_L(fcl_pack_planes(c.h1_all, U, F, U, h1_p, E.cur));
    float* out_cells = f32(E, F, O);
    TE_L(fcl_linear_planes_fwd(h1_p, U / 32, wf_h_p, nullptr, out_cells, O, nullptr, F, O, U, FCL_ACT_NONE, E.cur));
    float* f0c = f32(E, F, O);
    TE_L(fcl_gather_rows_fwd(F0, b.cell_row, f0c, nullptr, F, O, E.cur));
    TE_L(fcl_add2d(out_cells, O, f0c, O, F, O, 1.0f, nullptr, E.cur));
    c.before = f32(E, BL, O);
    xp = pl16(E, BL, O);
    TE_L(fcl_gather_rows_fwd(out_cells, b.frame_cell, c.before, xp, BL, O, E.cur));  // zero where no cell maps (padding)
    // ---- postnet
    x = c.before;
    const int n_post = cf.postnet_layers, Cp = cf.postnet_chans;
    c.post_c.assign(n_post, ConvBn());
    c.post_taps.clear();
    std::vector<Site> pk;
    if (drop_conv) {
        for (int i = 0; i < n_post; ++i) pk.push_back(Site{SITE_POST + i, (long long)BL * (i == n_post - 1 ? O : Cp), 1.0f - p_conv, nullptr});
        TE_TRY(draw_masks(E, pk));
    }
    for (int i = 0; i < n_post; ++i) {
        char pre[64];
        snprintf(pre, sizeof(pre), "dec.postnet.postnet.%d", i);
        float* y;
        uint16_t* yp;
        const int cout = i == n_post - 1 ? O : Cp, cin = i == 0 ? O : Cp;
        TE_TRY(conv_bn_fwd(E, x, xp, BL, pre, cout, cin, cf.postnet_filts, b.f_lo, b.f_hi, i == n_post - 1 ? FCL_ACT_NONE : FCL_ACT_TANH, drop_conv ? pk[i].out : nullptr,
                           p_conv, i < n_post - 1, &y, &yp, &c.post_c[i]));
        x = y;
        xp = yp;
        if (yp) c.planes_of[y] = yp;
        c.post_taps.push_back(y);
    }
    c.after = f32(E, BL, O);
    TE_L(fcl_add_vec(c.before, x, c.after, BL * O, E.cur));
    TE_TRY(pred_join(E));
    return 0;
}

// ================================================================================================================================================
// losses and the gradient every term injects at its tap (training.py _losses)
// ================================================================================================================================================
// Round 6: every element-wise term of a step is ONE fcl_loss_terms_batch launch (a term may carry the ground truth AND the teacher's output as targets),
// and a KD projection term is split into its forward (s = s_in . W^T) and its backward (dW on the weight-gradient stream, ds_in = ds . W) around that
// launch: the student's stream issues 9 launches for its loss phase instead of 35, each of which waited for a compute unit beside the frozen teacher.
struct LossBatchBuilder {
    fcl_te& E;
    std::vector<fcl_loss_term_t> terms;
    explicit LossBatchBuilder(fcl_te& e) : E(e) {}
    // one target; returns the gradient buffer (allocated here unless da is given)
    int add(const char* name, const float* a, const float* bb, const uint8_t* valid, int m, int cc, double count, float w_l1, float w_mse, bool b_log, bool want_planes,
            float** da_out, uint16_t** dap_out) {
        const int slot = loss_slot(name);
        FCL_REQUIRE(slot >= 0, FCL_ERR_INVALID, "fcl_te: unknown loss %s", name);
        fcl_loss_term_t t{};
        t.a = a; t.b = bb; t.valid = valid; t.m = m; t.c = cc; t.b_log = b_log ? 1 : 0; t.b_log_offset = b_log ? 1.0f : 0.0f; t.w_l1 = w_l1; t.w_mse = w_mse;
        t.count = count * E.cfg.accum_grad;
        t.da = f32(E, m, cc);
        t.da_planes = (want_planes && cc % 32 == 0) ? pl16(E, m, cc) : nullptr;
        t.sums = E.c.sums + 3 * slot;
        terms.push_back(t);
        if (da_out) *da_out = t.da;
        if (dap_out) *dap_out = t.da_planes;
        return 0;
    }
    // a second target for the term added last (its gradient joins the first's)
    int second(const char* name, const float* b2, const uint8_t* valid2, double count2, float w_l1, float w_mse) {
        const int slot = loss_slot(name);
        FCL_REQUIRE(slot >= 0 && !terms.empty(), FCL_ERR_INVALID, "fcl_te: unknown loss %s", name);
        fcl_loss_term_t& t = terms.back();
        t.b2 = b2; t.valid2 = valid2; t.count2 = count2 * E.cfg.accum_grad; t.w_l1_2 = w_l1; t.w_mse_2 = w_mse; t.sums2 = E.c.sums + 3 * slot;
        return 0;
    }
    int flush() {
        for (size_t k0 = 0; k0 < terms.size(); k0 += FCL_LOSS_MAX_TERMS)
            TE_L(fcl_loss_terms_batch(terms.data() + k0, (int)std::min<size_t>(FCL_LOSS_MAX_TERMS, terms.size() - k0), E.cur));
        terms.clear();
        return 0;
    }
};

// MSE(s_in . W^T, t) over the valid rows, in two halves around the loss launch
struct KdTerm {
    std::string lname, proj, key;
    const float* s_in;
    int rows, n, k;
    const float* t;
    const uint8_t* valid;
    double nvalid;
    bool planes;
    float* ds;
    uint16_t* ds_p;
};
static int te_kd_fwd(fcl_te& E, LossBatchBuilder& lb, std::vector<KdTerm>& pend, const char* lname, const char* key, const float* s_in, int rows, const std::string& proj,
                     int n, int k, const float* t, const uint8_t* valid, double nvalid) {
    KdTerm q;
    q.lname = lname; q.proj = proj; q.key = key; q.s_in = s_in; q.rows = rows; q.n = n; q.k = k; q.t = t; q.valid = valid; q.nvalid = nvalid;
    q.planes = n % 32 == 0 && k % 32 == 0 && rows >= 4096;
    Param& W = E.Pm(proj + ".weight");
    // FCL_TE_KD_FUSED=1 (round 6): a term on the planes kernels is ONE launch -- projection, masked MSE and gradient in the GEMM's epilogue
    // (fcl_linear_planes_mse_fwd); the projected tensor [rows, n] is never written or read back (8 of the 20 bytes per element the term moved)
    static const int fused = tunable("TE_KD_FUSED", 1);
    const bool fuse = fused && q.planes && (reinterpret_cast<uintptr_t>(t) & 15u) == 0 && (n & 3) == 0;
    float* s = fuse ? nullptr : f32(E, rows, n);
    if (q.planes) {
        const uint16_t* sp;
        auto hit = E.c.planes_of.find(s_in);
        if (hit != E.c.planes_of.end()) sp = hit->second;
        else {
            uint16_t* spk = pl16(E, rows, k);
            TE_L(fcl_pack_planes(s_in, k, rows, k, spk, E.cur));
            sp = spk;
        }
        const uint16_t* wp;
        TE_TRY(w_planes(E, proj + ".weight", n, k, &wp));
        if (fuse) {
            const int slot = loss_slot(lname);
            FCL_REQUIRE(slot >= 0, FCL_ERR_INVALID, "fcl_te: unknown loss %s", lname);
            q.ds = f32(E, rows, n);
            q.ds_p = pl16(E, rows, n);
            TE_L(fcl_linear_planes_mse_fwd(sp, k / 32, wp, t, n, valid, nvalid * n * E.cfg.accum_grad, q.ds, n, q.ds_p, E.c.sums + 3 * slot, rows, n, k, E.cur));
            pend.push_back(q);
            return 0;
        }
        TE_L(fcl_linear_planes_fwd(sp, k / 32, wp, nullptr, s, n, nullptr, rows, n, k, FCL_ACT_NONE, E.cur));
    } else {
        TE_L(fcl_linear_fwd(s_in, k, W.p, k, nullptr, s, n, rows, n, k, FCL_ACT_NONE, E.cur));
    }
    TE_TRY(lb.add(lname, s, t, valid, rows, n, nvalid * n, 0.0f, 1.0f, false, q.planes, &q.ds, &q.ds_p));
    pend.push_back(q);
    return 0;
}
static int te_kd_bwd(fcl_te& E, const KdTerm& q) {
    Param& W = E.Pm(q.proj + ".weight");
    {
        SideScope sc(E);
        TE_TRY(sc.rc);
        if (q.planes) TE_TRY(dw_gemm(E, q.ds, q.rows, q.n, {{q.s_in, q.k, W.g, q.k}}));
        else TE_L(fcl_gemm_tn_fwd(q.ds, q.n, q.s_in, q.k, W.g, q.k, q.rows, q.n, q.k, 0, nullptr, nullptr, E.cur));
        E.dw_pending = true;
    }
    float* ds_in = f32(E, q.rows, q.k);
    if (q.planes) {
        const uint16_t* wtp;
        TE_TRY(w_t(E, q.proj + ".weight", W.p, q.n, q.k, q.k, false, true, nullptr, &wtp));
        TE_L(fcl_linear_planes_fwd(q.ds_p, q.n / 32, wtp, nullptr, ds_in, q.k, nullptr, q.rows, q.k, q.n, FCL_ACT_NONE, E.cur));
    } else {
        const float* wt;
        TE_TRY(w_t(E, q.proj + ".weight", W.p, q.n, q.k, q.k, true, false, &wt, nullptr));
        TE_L(fcl_linear_fwd(q.ds, q.n, wt, q.n, nullptr, ds_in, q.k, q.rows, q.k, q.n, FCL_ACT_NONE, E.cur));
    }
    E.c.inj[q.key] = ds_in;
    return 0;
}

static int te_losses(fcl_te& E, const fcl_te_knowledge_t* know) {
    const fcl_te_config_t& cf = E.cfg;
    Ctx& c = E.c;
    const fcl_te_batch_t& b = c.b;
    const int BT = b.B * b.T, BL = b.B * b.L, F = b.F, O = cf.odim, C = cf.eunits;
    const double nf = b.n_frames * O, ne = b.n_enc;
    c.sums = reinterpret_cast<double*>(ZA(E).take(FCL_TE_MAX_LOSSES * 3 * sizeof(double)));
    c.inj.clear();
    const bool um = !cf.use_masking;
    const uint8_t* fv = um ? nullptr : b.frame_valid;
    const double nfm = um ? (double)BL * O : nf;
    const uint8_t* ev = um ? nullptr : b.enc_valid;
    const double nem = um ? (double)BT : ne;
    const bool student = cf.role == FCL_TE_STUDENT;
    FCL_REQUIRE(!student || know != nullptr, FCL_ERR_INVALID, "fcl_te: the student step needs the teacher's knowledge (tts_distill.py:159-161)");
    const int U = cf.dunits, Pn = cf.prenet_units, Cp = cf.postnet_chans;
    const std::string cp[3] = {cf.share_proj ? "enc.convs_proj.0" : "enc.convs_proj.0", cf.share_proj ? "enc.convs_proj.0" : "enc.convs_proj.1",
                               cf.share_proj ? "enc.convs_proj.0" : "enc.convs_proj.2"};
    const std::string lp[2] = {cf.share_proj ? "dec.lstm_proj" : "dec.lstm0_proj", cf.share_proj ? "dec.lstm_proj" : "dec.lstm1_proj"};
    auto pp = [&](int i) { return cf.share_proj ? std::string("dec.post_proj") : ("dec.post" + std::to_string(i) + "_proj"); };
    // FCL_TE_R6: bit 0 = the late group is enqueued BEFORE the student's stream's own terms, so that it runs beside them (-0.15 ms per KD update, same-box A/B); bit 1 =
    // the prosody embedding terms join the late group (their gradients are wanted at backward stage 2).  (Bit 0 was off for most of round 6: beside the late group's
    // GEMMs the loss kernel's packed-FP32 arithmetic came out stale in lanes 48 - 63 in ~10 % of fresh engines' first updates -- DESIGN 4c; the library is built
    // without packed-FP32 instructions since.)
    static const int r6 = tunable("TE_R6", 3);
    auto late_group = [&]() -> int {  // the KD terms whose gradients the backward needs LATE run on the weight-gradient stream beside the frame-level terms and the postnet's backward
        const bool fork = cf.pred_stream && c.save && cf.late_losses;
        hipStream_t saved = E.cur;
        if (fork) {
            TE_TRY(ev_wait(E, E.side, E.main));
            E.cur = E.side;
        }
        LossBatchBuilder lb(E);
        std::vector<KdTerm> pend;
        if (cf.distill_decoder) {
            const float* tc[3];
            const int wd[3] = {cf.t_prenet_units, cf.t_dunits, cf.t_dunits};
            for (int i = 0; i < 3; ++i) {
                if (know->dec_cell_major) tc[i] = know->dec[i];
                else {
                    float* t = f32(E, F, wd[i]);
                    TE_L(fcl_gather_rows_fwd(know->dec[i], b.cell_frame, t, nullptr, F, wd[i], E.cur));
                    tc[i] = t;
                }
            }
            TE_TRY(te_kd_fwd(E, lb, pend, "dec2", "h1", c.h1_all, F, lp[1], cf.t_dunits, U, tc[2], b.cell_valid, b.n_frames));
            TE_TRY(te_kd_fwd(E, lb, pend, "dec1", "h0", c.h0_all, F, lp[0], cf.t_dunits, U, tc[1], b.cell_valid, b.n_frames));
            TE_TRY(te_kd_fwd(E, lb, pend, "dec0", "p1d", c.p1d, F, "dec.prenet_proj", cf.t_prenet_units, Pn, tc[0], b.cell_valid, b.n_frames));
        }
        if (cf.distill_encoder) {
            TE_TRY(te_kd_fwd(E, lb, pend, "enc0", "enc0", c.enc_taps[0], BT, "enc.embed_proj", cf.t_embed_dim, cf.embed_dim, know->enc[0], b.enc_valid, ne));
            for (int i = 0; i < 3; ++i) {
                char nm[8];
                snprintf(nm, sizeof(nm), "enc%d", i + 1);
                TE_TRY(te_kd_fwd(E, lb, pend, nm, nm, c.enc_taps[1 + i], BT, cp[i], cf.t_econv_chans, cf.econv_chans, know->enc[1 + i], b.enc_valid, ne));
            }
            TE_TRY(te_kd_fwd(E, lb, pend, "enc4", "hs", c.hs, BT, "enc.blstm_proj", cf.t_eunits, C, know->enc[4], b.enc_valid, ne));
        }
        if (cf.distill_prosody && (r6 & 2)) {  // the embedding taps' gradients are wanted at backward stage 2: late as well
            TE_TRY(te_kd_fwd(E, lb, pend, "pro3", "p_embs", c.p_embs, BT, "pemb_proj", cf.t_eunits, C, know->pro[3], b.enc_valid, ne));
            TE_TRY(te_kd_fwd(E, lb, pend, "pro4", "e_embs", c.e_embs, BT, "eemb_proj", cf.t_eunits, C, know->pro[4], b.enc_valid, ne));
        }
        TE_TRY(lb.flush());
        for (const KdTerm& q : pend) TE_TRY(te_kd_bwd(E, q));
        if (fork) {
            if (!E.dry) FCL_HIP(hipEventRecord(E.late_ev, E.side));
            E.late_pending = true;
            E.cur = saved;
        }
        return 0;
    };
    if (student && (r6 & 1)) TE_TRY(late_group());
    // the student's stream: the postnet taps' projections, ONE loss launch (their terms + every element-wise term), their backward halves
    LossBatchBuilder lb(E);
    std::vector<KdTerm> pend;
    if (student && cf.distill_prosody && !(r6 & 2)) {
        TE_TRY(te_kd_fwd(E, lb, pend, "pro3", "p_embs", c.p_embs, BT, "pemb_proj", cf.t_eunits, C, know->pro[3], b.enc_valid, ne));
        TE_TRY(te_kd_fwd(E, lb, pend, "pro4", "e_embs", c.e_embs, BT, "eemb_proj", cf.t_eunits, C, know->pro[4], b.enc_valid, ne));
    }
    if (student && cf.distill_decoder)
        for (int i = 0; i < 4; ++i) {
            char nm[8];
            snprintf(nm, sizeof(nm), "dec%d", 3 + i);
            TE_TRY(te_kd_fwd(E, lb, pend, nm, ("post" + std::to_string(i)).c_str(), c.post_taps[i], BL, pp(i), cf.t_postnet_chans, Cp, know->dec[3 + i], b.frame_valid,
                             b.n_frames));
        }
    float* g;
    TE_TRY(lb.add("after", c.after, b.ys, fv, BL, O, nfm, 1.f, 1.f, false, false, &g, nullptr));
    if (student && cf.distill_output) TE_TRY(lb.second("o_after", know->after, fv, nfm, 1.f, 1.f));
    c.inj["after"] = g;
    TE_TRY(lb.add("before", c.before, b.ys, fv, BL, O, nfm, 1.f, 1.f, false, false, &g, nullptr));
    if (student && cf.distill_output) TE_TRY(lb.second("o_before", know->before, fv, nfm, 1.f, 1.f));
    c.inj["before"] = g;
    TE_TRY(lb.add("dur", c.dur.out, b.ds, b.enc_valid, BT, 1, ne, 0.f, 1.f, true, false, &g, nullptr));
    if (student && cf.distill_prosody) TE_TRY(lb.second("pro0", know->pro[0], b.enc_valid, ne, 0.f, 1.f));
    c.inj["d_outs"] = g;
    TE_TRY(lb.add("pitch", c.pit.out, b.f0, ev, BT, 1, nem, 0.f, 1.f, false, false, &g, nullptr));
    if (student && cf.distill_prosody) TE_TRY(lb.second("pro1", know->pro[1], b.enc_valid, ne, 0.f, 1.f));
    c.inj["p_outs"] = g;
    TE_TRY(lb.add("energy", c.en.out, b.energy, ev, BT, 1, nem, 0.f, 1.f, false, false, &g, nullptr));
    if (student && cf.distill_prosody) TE_TRY(lb.second("pro2", know->pro[2], b.enc_valid, ne, 0.f, 1.f));
    c.inj["e_outs"] = g;
    if (student && cf.distill_decoder) {
        TE_TRY(lb.add("dec7", c.post_taps[4], know->dec[7], b.frame_valid, BL, O, nf, 0.f, 1.f, false, false, &g, nullptr));
        c.inj["post4"] = g;
    }
    TE_TRY(lb.flush());
    for (int i = (int)pend.size() - 1; i >= 0; --i) TE_TRY(te_kd_bwd(E, pend[i]));  // the postnet's backward consumes post3 first
    if (student && !(r6 & 1)) TE_TRY(late_group());
    return 0;
}

// ================================================================================================================================================
// backward (training.py _backward), cut at the points where a gradient bucket becomes final
// ================================================================================================================================================
static int te_backward_stage0(fcl_te& E) {  // predictors' backward forked; postnet
    const fcl_te_config_t& cf = E.cfg;
    Ctx& c = E.c;
    const fcl_te_batch_t& b = c.b;
    const int BL = b.B * b.L, O = cf.odim;
    {
        const bool fork = cf.pred_stream && c.save;
        hipStream_t saved = E.cur;
        if (fork) {
            TE_TRY(ev_wait(E, E.side, E.main));
            E.cur = E.side;
        }
        TE_TRY(predictor_bwd(E, c.inj["d_outs"], c.dur, b.enc_pad, &c.d_preds[0]));
        TE_TRY(predictor_bwd(E, c.inj["p_outs"], c.pit, b.enc_pad, &c.d_preds[1]));
        TE_TRY(predictor_bwd(E, c.inj["e_outs"], c.en, b.enc_pad, &c.d_preds[2]));
        if (fork) {
            if (!E.dry) FCL_HIP(hipEventRecord(E.pred_ev, E.side));
            E.pred_pending = true;
            E.dw_pending = true;
            E.cur = saved;
        }
    }
    float* dx = c.inj["after"];
    for (int i = (int)c.post_c.size() - 1; i >= 0; --i) {
        auto it = c.inj.find("post" + std::to_string(i));
        float* nx;
        TE_TRY(conv_bn_bwd(E, dx, it != c.inj.end() ? it->second : nullptr, c.post_c[i], &nx));
        dx = nx;
    }
    c.d_post_in = dx;  // d before = inj[before] + inj[after] + the postnet's input gradient: summed by the gather of stage 1 (round 6)
    (void)BL;
    (void)O;
    TE_TRY(late_join(E));  // the KD gradients at the prenet / LSTM / encoder taps
    return 0;
}

// round 5, measured and NOT adopted (FCL_TE_DX_PLANES=1 turns it on): four input-gradient GEMMs whose A operand is a gradient tensor nobody has planes of (K = odim
// = 80 or K = 4 U / 4 H contractions) run on the fp32-operand kernel (`gemm_kernel<...>/bf16x3`, 25 - 60 TFLOP/s: 0.5 ms of kernel time per teacher update).  With
// planes from the producer (the gathers write them) or one packing pass and the LDS-DMA kernels instead: teacher update 9.82 vs 9.77 ms, KD 9.03 vs 9.09 ms on one
// box (`gpurun_out/r6i`) -- they are off the critical path and the packing passes cost what the faster GEMMs save.
static bool dx_planes() {
    static const int on = tunable("TE_DX_PLANES", 0);
    return on != 0 && tunable("PRECISION", 1) != 0 && tunable("PLANES", 1) != 0;
}
// Y[m, n] = A[m, k] . Wt^T for a weight kept as the transposed form `key` ([n, k] rows): planes when ap / a packing pass can provide them
static int dx_linear(fcl_te& E, const float* a, const uint16_t* ap, int m, int k, const std::string& key, const float* wsrc, int rows, int cols, int ld, float* y, int n) {
    if (dx_planes() && n % 4 == 0) {
        if (!ap) {
            uint16_t* t = pl16(E, m, k);
            TE_L(fcl_pack_planes(a, k, m, k, t, E.cur));
            ap = t;
        }
        const uint16_t* wtp;
        TE_TRY(w_t(E, key, wsrc, rows, cols, ld, false, true, nullptr, &wtp));
        TE_L(fcl_linear_planes_fwd(ap, (k + 31) / 32, wtp, nullptr, y, n, nullptr, m, n, k, FCL_ACT_NONE, E.cur));
        return 0;
    }
    const float* wt;
    TE_TRY(w_t(E, key, wsrc, rows, cols, ld, true, false, &wt, nullptr));
    TE_L(fcl_linear_fwd(a, k, wt, k, nullptr, y, n, m, n, k, FCL_ACT_NONE, E.cur));
    return 0;
}

static int te_backward_stage1(fcl_te& E) {  // decoder BPTT + prenet
    const fcl_te_config_t& cf = E.cfg;
    Ctx& c = E.c;
    const fcl_te_batch_t& b = c.b;
    const int N = b.N, F = b.F, O = cf.odim, U = cf.dunits, Pn = cf.prenet_units, C = cf.eunits;
    float* d_out_cells = f32(E, F, O);
    uint16_t* d_out_cells_p = dx_planes() ? pl16(E, F, O) : nullptr;
    TE_L(fcl_gather_rows_sum_fwd(c.inj["before"], c.inj["after"], c.d_post_in, b.cell_frame, d_out_cells, d_out_cells_p, F, O, E.cur));
    Param& WF = E.Pm("dec.feat_out.weight");
    {
        SideScope sc(E);
        TE_TRY(sc.rc);
        TE_L(fcl_gemm_tn_fwd(d_out_cells, O, c.h1_all, U, WF.g, U + C, F, O, U, 0, nullptr, nullptr, E.cur));  // column block [:, :U] written in place
        E.dw_pending = true;
    }
    const float *wf_h, *wf_att;
    TE_TRY(w_cols(E, "dec.feat_out.weight", O, U + C, 0, U, true, &wf_h, nullptr));
    TE_TRY(w_cols(E, "dec.feat_out.weight", O, U + C, U, C, true, &wf_att, nullptr));
    float* dh1_all = f32(E, F, U);
    if (!dx_planes()) {  // round 6: the KD gradient at the h1 tap joins as the GEMM's residual
        const float* wt;
        TE_TRY(w_t(E, "dec.feat_out.weight/h", WF.p, O, U, U + C, true, false, &wt, nullptr));
        TE_L(fcl_linear2_fwd(d_out_cells, O, wt, O, O, nullptr, 0, nullptr, 0, 0, nullptr, c.inj.count("h1") ? c.inj["h1"] : nullptr, U, dh1_all, U, F, U, FCL_ACT_NONE, E.cur));
    } else {
        TE_TRY(dx_linear(E, d_out_cells, d_out_cells_p, F, O, "dec.feat_out.weight/h", WF.p, O, U, U + C, dh1_all, U));
        if (c.inj.count("h1")) TE_L(fcl_add2d(dh1_all, U, c.inj["h1"], U, F, U, 1.0f, nullptr, E.cur));
    }
    float* dF0 = zf32(E, (long long)N * O);
    TE_L(fcl_scatter_add_rows(d_out_cells, b.cell_row_i64, dF0, F, O, -1, E.cur));
    {
        SideScope sc(E);
        TE_TRY(sc.rc);
        TE_L(fcl_gemm_tn_fwd(dF0, O, c.att_c, C, WF.g + U, U + C, N, O, C, 0, nullptr, nullptr, E.cur));
        E.dw_pending = true;
    }
    c.d_att_c = f32(E, N, C);
    if (dx_planes()) TE_TRY(dx_linear(E, dF0, nullptr, N, O, "dec.feat_out.weight/att", WF.p + U, O, C, U + C, c.d_att_c, C));
    float *dg0_all = f32(E, F, 4 * U), *dg1_all = f32(E, F, 4 * U);
    const std::string w1ih = "dec.lstm.1.cell.weight_ih", w1hh = "dec.lstm.1.cell.weight_hh", w0hh = "dec.lstm.0.cell.weight_hh", wih0 = "dec.lstm.0.cell.weight_ih";
    const int ld0 = C + Pn + 1;
    // [W1_hh^T ; W1_ih^T] as ONE [2U, 4U] matrix (both GEMMs that leave layer 1's gate gradients in one launch per step): two forms, one buffer
    const float *w1_ih_t, *w1_hh_t, *w0_hh_t;
    const uint16_t *w1_ih_t_p, *w1_hh_t_p, *w0_hh_t_p;
    TE_TRY(w_t(E, w1ih, E.Pm(w1ih).p, 4 * U, U, U, true, true, &w1_ih_t, &w1_ih_t_p));
    TE_TRY(w_t(E, w1hh, E.Pm(w1hh).p, 4 * U, U, U, true, true, &w1_hh_t, &w1_hh_t_p));
    TE_TRY(w_t(E, w0hh, E.Pm(w0hh).p, 4 * U, U, U, true, true, &w0_hh_t, &w0_hh_t_p));
    const float* cat_f = nullptr;
    const uint16_t* cat_p = nullptr;
    {
        if (!E.cat_f) {  // one [2U, 4U] buffer (fp32 and planes), filled by two forms: rows [0, U) = W1_hh^T, rows [U, 2U) = W1_ih^T
            void *pf = nullptr, *pp_ = nullptr;
            FCL_HIP(hipMalloc(&pf, (size_t)2 * U * 4 * U * 4 + 256));
            FCL_HIP(hipMalloc(&pp_, planes_elems(2 * U, 4 * U) * 2 + 256));
            E.form_allocs.push_back(pf);
            E.form_allocs.push_back(pp_);
            E.cat_f = static_cast<float*>(pf);
            E.cat_p = static_cast<uint16_t*>(pp_);
        }
        Form *fa, *fb;
        TE_TRY(form_get(E, "cat/w1/a", E.Pm(w1hh).p, nullptr, 1, U, 4 * U, 0, 1, U, true, true, &fa, E.cat_f, E.cat_p));
        TE_TRY(form_get(E, "cat/w1/b", E.Pm(w1ih).p, nullptr, 1, U, 4 * U, 0, 1, U, true, true, &fb, E.cat_f + (size_t)U * 4 * U, E.cat_p + planes_elems(U, 4 * U)));
        cat_f = E.cat_f;
        cat_p = E.cat_p;
    }
    uint16_t *dg0_p = pl16(E, F, 4 * U), *dg1_p = pl16(E, F, 4 * U);
    {
        fcl_decoder_bptt_t a{};
        a.n = N; a.lmax = b.lmax; a.u = U; a.live_rows_host = b.live_rows_host; a.zoneout = cf.zoneout_rate; a.dh1_all = dh1_all;
        a.dh0_all = c.inj.count("h0") ? c.inj["h0"] : nullptr;
        a.w1_ih_t = w1_ih_t; a.w1_hh_t = w1_hh_t; a.w0_hh_t = w0_hh_t; a.dg0_all = dg0_all; a.dg1_all = dg1_all;
        a.zk_h0 = c.zk[0][0]; a.zk_c0 = c.zk[0][1]; a.zk_h1 = c.zk[1][0]; a.zk_c1 = c.zk[1][1];
        for (int q = 0; q < 3; ++q) { a.s0[q] = c.S0[q]; a.s1[q] = c.S1[q]; }
        a.w1_ih_t_p = w1_ih_t_p; a.w1_hh_t_p = w1_hh_t_p; a.w0_hh_t_p = w0_hh_t_p; a.dg0_all_p = dg0_p; a.dg1_all_p = dg1_p;
        a.w1_cat_t = cat_f; a.w1_cat_t_p = cat_p;
        a.workspace_bytes = fcl_decoder_train_workspace_bytes(N, U);
        a.workspace = WA(E).take(a.workspace_bytes);
        TE_L(fcl_decoder_bptt(&a, E.cur));
    }
    const float* w0_pre;
    TE_TRY(w_cols(E, wih0, 4 * U, ld0, C, Pn, true, &w0_pre, nullptr));
    const uint16_t* w0_pre_t_p;
    Form* fpt;
    TE_TRY(form_get(E, "t/" + wih0 + "/pre", E.Pm(wih0).p + C, nullptr, 1, Pn, 4 * U, 0, 1, ld0, false, true, &fpt));
    w0_pre_t_p = fpt->dst_p;
    float* dp1_all = f32(E, F, Pn);
    TE_L(fcl_linear_planes_fwd(dg0_p, 4 * U / 32, w0_pre_t_p, nullptr, dp1_all, Pn, nullptr, F, Pn, 4 * U, FCL_ACT_NONE, E.cur));  // gradient w.r.t. the prenet output of every cell
    float* g_ih0 = E.Pm(wih0).g;  // [4U, C + P + 1] = [att_c | prenet | position]
    {
        SideScope sc(E);  // weight gradients of the two cells from the saved step-major tensors
        TE_TRY(sc.rc);
        TE_TRY(dw_gemm(E, dg1_all, F, 4 * U, {{c.h0_all, U, E.Pm(w1ih).g, U}, {c.S1[3], U, E.Pm(w1hh).g, U}}));
        TE_L(fcl_colsum2_fwd(dg0_all, nullptr, nullptr, nullptr, E.Pm("dec.lstm.0.cell.bias_ih").g, E.Pm("dec.lstm.0.cell.bias_hh").g, F, 4 * U, 0, E.cur));
        TE_L(fcl_colsum2_fwd(dg1_all, nullptr, nullptr, nullptr, E.Pm("dec.lstm.1.cell.bias_ih").g, E.Pm("dec.lstm.1.cell.bias_hh").g, F, 4 * U, 0, E.cur));
        TE_TRY(dw_gemm(E, dg0_all, F, 4 * U, {{c.S0[3], U, E.Pm(w0hh).g, U}, {c.p1d, Pn, g_ih0 + C, ld0}}));
        float* dw0_pos4 = zf32(E, (long long)4 * U * 4);
        TE_L(fcl_gemm_tn_fwd(dg0_all, 4 * U, b.pos4, 4, dw0_pos4, 4, F, 4 * U, 4, 0, nullptr, nullptr, E.cur));
        TE_L(fcl_add2d(g_ih0 + C + Pn, ld0, dw0_pos4, 4, 4 * U, 1, 1.0f, nullptr, E.cur));
        E.dw_pending = true;
    }
    float* dG0 = zf32(E, (long long)N * 4 * U);
    TE_L(fcl_scatter_add_rows(dg0_all, b.cell_row_i64, dG0, F, 4 * U, -1, E.cur));
    {
        SideScope sc(E);
        TE_TRY(sc.rc);
        TE_L(fcl_gemm_tn_fwd(dG0, 4 * U, c.att_c, C, g_ih0, ld0, N, 4 * U, C, 0, nullptr, nullptr, E.cur));
        E.dw_pending = true;
    }
    if (!dx_planes()) {  // round 6: the two input gradients that reach att_c (through feat_out and through LSTM 0's input block) in ONE two-term GEMM
        const float *wt_f, *wt_g;
        TE_TRY(w_t(E, "dec.feat_out.weight/att", WF.p + U, O, C, U + C, true, false, &wt_f, nullptr));
        TE_TRY(w_t(E, wih0 + "/att", E.Pm(wih0).p, 4 * U, C, ld0, true, false, &wt_g, nullptr));
        TE_L(fcl_linear2_fwd(dF0, O, wt_f, O, O, dG0, 4 * U, wt_g, 4 * U, 4 * U, nullptr, nullptr, 0, c.d_att_c, C, N, C, FCL_ACT_NONE, E.cur));
    } else {
        float* t = f32(E, N, C);
        TE_TRY(dx_linear(E, dG0, nullptr, N, 4 * U, wih0 + "/att", E.Pm(wih0).p, 4 * U, C, ld0, t, C));
        TE_L(fcl_add2d(c.d_att_c, C, t, C, N, C, 1.0f, nullptr, E.cur));
    }
    // prenet (batched over all cells)
    const std::string w0n = "dec.prenet.prenet.0.0.weight", b0n = "dec.prenet.prenet.0.0.bias", w1n = "dec.prenet.prenet.1.0.weight", b1n = "dec.prenet.prenet.1.0.bias";
    float* dz1 = f32(E, F, Pn);
    uint16_t* dz1_p = pl16(E, F, Pn);
    if (c.inj.count("p1d")) TE_L(fcl_act_bwd_sum(dp1_all, c.inj["p1d"], c.p1, c.k1, c.pks, dz1, dz1_p, Pn, (size_t)F * Pn, FCL_ACT_RELU, E.cur));
    else TE_L(fcl_act_bwd(dp1_all, c.p1, c.k1, c.pks, dz1, dz1_p, Pn, (size_t)F * Pn, FCL_ACT_RELU, E.cur));
    {
        SideScope sc(E);
        TE_TRY(sc.rc);
        TE_L(fcl_gemm_tn_fwd(dz1, Pn, c.p0d, Pn, E.Pm(w1n).g, Pn, F, Pn, Pn, 0, nullptr, nullptr, E.cur));
        TE_L(fcl_colsum2_fwd(dz1, nullptr, nullptr, nullptr, E.Pm(b1n).g, nullptr, F, Pn, 0, E.cur));
        E.dw_pending = true;
    }
    const uint16_t* w1t_p;
    TE_TRY(w_t(E, w1n, E.Pm(w1n).p, Pn, Pn, Pn, false, true, nullptr, &w1t_p));
    float* dp0 = f32(E, F, Pn);
    TE_L(fcl_linear_planes_fwd(dz1_p, Pn / 32, w1t_p, nullptr, dp0, Pn, nullptr, F, Pn, Pn, FCL_ACT_NONE, E.cur));
    float* dz0 = f32(E, F, Pn);
    TE_L(fcl_act_bwd(dp0, c.p0, c.k0, c.pks, dz0, nullptr, 0, (size_t)F * Pn, FCL_ACT_RELU, E.cur));
    {
        SideScope sc(E);
        TE_TRY(sc.rc);
        TE_L(fcl_gemm_tn_fwd(dz0, Pn, c.pre_in, O, E.Pm(w0n).g, O, F, Pn, O, 0, nullptr, nullptr, E.cur));
        TE_L(fcl_colsum2_fwd(dz0, nullptr, nullptr, nullptr, E.Pm(b0n).g, nullptr, F, Pn, 0, E.cur));
        E.dw_pending = true;
    }
    return 0;
}

static int te_backward_stage2(fcl_te& E) {  // att = hs + p_embs + e_embs; predictors joined
    const fcl_te_config_t& cf = E.cfg;
    Ctx& c = E.c;
    const fcl_te_batch_t& b = c.b;
    const int BT = b.B * b.T, C = cf.eunits;
    c.d_att = f32(E, BT, C);
    TE_L(fcl_gather_rows_fwd(c.d_att_c, b.row_of_enc, c.d_att, nullptr, BT, C, E.cur));  // back to (b, t) rows; rows without a phoneme get 0
    const int kk = cf.ve_kernel;
    const char* nm[2] = {"pitch", "energy"};
    const char* tap[2] = {"p_embs", "e_embs"};
    const float* sig[2] = {b.f0, b.energy};
    for (int q = 0; q < 2; ++q) {
        float* d_e = c.d_att;
        const float* inj_q = c.inj.count(tap[q]) ? c.inj[tap[q]] : nullptr;
        if (c.emb_keep[q]) {  // round 6: the KD gradient at the embedding tap joins inside the dropout backward
            float* t = f32(E, BT, C);
            if (inj_q) TE_L(fcl_act_bwd_sum(d_e, inj_q, nullptr, c.emb_keep[q], c.emb_ks, t, nullptr, 0, (size_t)BT * C, FCL_ACT_NONE, E.cur));
            else TE_L(fcl_act_bwd(d_e, nullptr, c.emb_keep[q], c.emb_ks, t, nullptr, 0, (size_t)BT * C, FCL_ACT_NONE, E.cur));
            d_e = t;
        } else if (inj_q) {
            float* t = f32(E, BT, C);
            const float* srcs[2] = {c.d_att, inj_q};
            TE_L(fcl_sum_rows(srcs, 2, nullptr, t, nullptr, BT, C, E.cur));
            d_e = t;
        }
        SideScope sc(E);  // Conv1d(1 -> C, k): weight and bias gradients in one launch
        TE_TRY(sc.rc);
        TE_L(fcl_conv1d_in1_dw(d_e, C, sig[q], b.e_lo, b.e_hi, E.Pm(std::string(nm[q]) + "_embed.0.weight").g, E.Pm(std::string(nm[q]) + "_embed.0.bias").g, BT, C, kk,
                               E.cur));
        E.dw_pending = true;
    }
    TE_TRY(pred_join(E));  // the predictors' backward was enqueued at the top of stage 0, on the weight-gradient stream; stage 3 sums their input gradients
    return 0;
}

static int te_backward_stage3(fcl_te& E) {  // encoder
    const fcl_te_config_t& cf = E.cfg;
    Ctx& c = E.c;
    const fcl_te_batch_t& b = c.b;
    const int B = b.B, T = b.T, BT = B * T, C = cf.eunits, H = C / 2, Cc = cf.econv_chans;
    // d hs = d att + the three predictors' input gradients (+ the KD term at the hs tap), zero on padded positions (pad_packed_sequence: padded outputs are
    // constants): one launch (round 6; was copy + 4 adds + a masked add into a zeroed buffer)
    float* d_hs_live = f32(E, BT, C);
    {
        const float* srcs[FCL_SUM_ROWS_MAX] = {c.d_att, c.d_preds[0], c.d_preds[1], c.d_preds[2], nullptr, nullptr};
        int ns = 4;
        if (c.inj.count("hs")) srcs[ns++] = c.inj["hs"];
        TE_L(fcl_sum_rows(srcs, ns, b.enc_valid, d_hs_live, nullptr, BT, C, E.cur));
    }
    // BiLSTM backward
    float* dx = dx_planes() ? zf32(E, (long long)BT * Cc) : f32(E, BT, Cc);
    float* dgx2[2] = {nullptr, nullptr};
    {
        fcl_bilstm_bptt_t a{};
        a.b = B; a.t = T; a.h = H; a.lens = b.lens; a.d_out = d_hs_live; a.ld_dout = C; a.status = E.status;
        float* dgs[2];
        const char* sfx[2] = {"", "_reverse"};
        for (int d = 0; d < 2; ++d) {
            dgs[d] = f32(E, (long long)T * B, 4 * H);
            a.dg[d] = dgs[d];
            for (int q = 0; q < 3; ++q) a.s[d][q] = c.bl.s[d][q];
            const std::string wn = std::string("enc.blstm.weight_hh_l0") + sfx[d];
            TE_TRY(w_t(E, wn, E.Pm(wn).p, 4 * H, H, H, true, false, &a.w_hh_t[d], nullptr));
        }
        a.workspace_bytes = fcl_bilstm_train_workspace_bytes(B, H);
        a.workspace = WA(E).take(a.workspace_bytes);
        TE_L(fcl_bilstm_bptt(&a, E.cur));
        for (int d = 0; d < 2; ++d) {
            float* dgx = f32(E, BT, 4 * H);
            uint16_t* dgx_p = dx_planes() ? pl16(E, BT, 4 * H) : nullptr;
            TE_L(fcl_gather_rows_fwd(dgs[d], b.perm_tb, dgx, dgx_p, BT, 4 * H, E.cur));  // back to (b, t) rows like x
            const std::string s_ = sfx[d];
            {
                SideScope sc(E);
                TE_TRY(sc.rc);
                TE_L(fcl_gemm_tn_fwd(dgs[d], 4 * H, c.bl.s[d][3], H, E.Pm("enc.blstm.weight_hh_l0" + s_).g, H, T * B, 4 * H, H, 0, nullptr, nullptr, E.cur));
                TE_L(fcl_gemm_tn_fwd(dgx, 4 * H, c.bl.x, Cc, E.Pm("enc.blstm.weight_ih_l0" + s_).g, Cc, BT, 4 * H, Cc, 0, nullptr, nullptr, E.cur));
                TE_L(fcl_colsum2_fwd(dgx, nullptr, nullptr, nullptr, E.Pm("enc.blstm.bias_ih_l0" + s_).g, E.Pm("enc.blstm.bias_hh_l0" + s_).g, BT, 4 * H, 0, E.cur));
                E.dw_pending = true;
            }
            dgx2[d] = dgx;
            if (dx_planes()) {
                const std::string wn = "enc.blstm.weight_ih_l0" + s_;
                float* t = f32(E, BT, Cc);
                TE_TRY(dx_linear(E, dgx, dgx_p, BT, 4 * H, wn, E.Pm(wn).p, 4 * H, Cc, Cc, t, Cc));
                TE_L(fcl_add2d(dx, Cc, t, Cc, BT, Cc, 1.0f, nullptr, E.cur));
            }
        }
        if (!dx_planes()) {  // round 6: both directions' input gradients in ONE two-term GEMM
            const std::string wf = "enc.blstm.weight_ih_l0", wr = "enc.blstm.weight_ih_l0_reverse";
            const float *wt_f, *wt_r;
            TE_TRY(w_t(E, wf, E.Pm(wf).p, 4 * H, Cc, Cc, true, false, &wt_f, nullptr));
            TE_TRY(w_t(E, wr, E.Pm(wr).p, 4 * H, Cc, Cc, true, false, &wt_r, nullptr));
            TE_L(fcl_linear2_fwd(dgx2[0], 4 * H, wt_f, 4 * H, 4 * H, dgx2[1], 4 * H, wt_r, 4 * H, 4 * H, nullptr, nullptr, 0, dx, Cc, BT, Cc, FCL_ACT_NONE, E.cur));
        }
    }
    for (int i = (int)c.conv_c.size() - 1; i >= 0; --i) {
        char key[8];
        snprintf(key, sizeof(key), "enc%d", i + 1);
        float* nx;
        TE_TRY(conv_bn_bwd(E, dx, c.inj.count(key) ? c.inj[key] : nullptr, c.conv_c[i], &nx));
        dx = nx;
    }
    if (c.inj.count("enc0")) TE_L(fcl_add2d(dx, cf.embed_dim, c.inj["enc0"], cf.embed_dim, BT, cf.embed_dim, 1.0f, nullptr, E.cur));
    {
        SideScope sc(E);
        TE_TRY(sc.rc);
        TE_L(fcl_scatter_add_rows(dx, b.xs, E.Pm("enc.embed.weight").g, BT, cf.embed_dim, 0, E.cur));  // padding_idx = 0 gets no gradient
        E.dw_pending = true;
    }
    return 0;
}

// ================================================================================================================================================
// C entry points
// ================================================================================================================================================
extern "C" {

const char* fcl_te_site_name(int i) { return (i >= 0 && i < N_SITES) ? SITE_NAMES[i] : nullptr; }
const char* fcl_te_loss_name(int i) { return (i >= 0 && i < N_LOSSES) ? LOSS_NAMES[i] : nullptr; }

int fcl_te_create(const fcl_te_config_t* cfg, fcl_te_t** out) {
    FCL_REQUIRE(cfg && out, FCL_ERR_INVALID, "fcl_te_create: null argument");
    const fcl_te_config_t& c = *cfg;
    FCL_REQUIRE(c.role >= FCL_TE_TEACHER && c.role <= FCL_TE_STUDENT, FCL_ERR_INVALID, "fcl_te_create: bad role %d", c.role);
    FCL_REQUIRE(tunable("PRECISION", 1) != 0 && tunable("PLANES", 1) != 0, FCL_ERR_INVALID, "fcl_te: the native step runs on pre-split operands (FCL_PRECISION=0 / FCL_PLANES=0 are set)");
    FCL_REQUIRE(c.embed_dim % 32 == 0 && c.econv_chans % 32 == 0 && c.eunits % 64 == 0 && c.dunits % 32 == 0 && c.prenet_units % 32 == 0 && c.postnet_chans % 32 == 0 &&
                    c.dp_chans % 32 == 0 && c.vp_chans % 32 == 0 && c.odim % 4 == 0 && c.odim > 0,
                FCL_ERR_SHAPE, "fcl_te: channel widths must be multiples of 32 (eunits: 64, odim: 4)");
    FCL_REQUIRE(c.embed_dim == c.econv_chans && c.eunits == c.econv_chans, FCL_ERR_SHAPE, "fcl_te: embed_dim / econv_chans / eunits must agree");
    FCL_REQUIRE(c.econv_layers >= 1 && c.econv_layers <= 8 && c.postnet_layers >= 2 && c.postnet_layers <= 8 && c.dp_layers >= 1 && c.dp_layers <= 4 && c.vp_layers >= 1 &&
                    c.vp_layers <= 4,
                FCL_ERR_SHAPE, "fcl_te: layer counts out of range");
    FCL_REQUIRE(c.dropout_rate >= 0.f && c.dropout_rate < 1.f && c.zoneout_rate >= 0.f && c.zoneout_rate < 1.f, FCL_ERR_INVALID, "fcl_te: bad dropout / zoneout rate");
    FCL_REQUIRE(c.accum_grad >= 1, FCL_ERR_INVALID, "fcl_te: accum_grad must be >= 1");
    if (c.role == FCL_TE_STUDENT) {
        FCL_REQUIRE(c.t_embed_dim > 0 && c.t_econv_chans > 0 && c.t_eunits > 0 && c.t_prenet_units > 0 && c.t_dunits > 0 && c.t_postnet_chans > 0, FCL_ERR_SHAPE,
                    "fcl_te: the student needs the teacher's widths");
        FCL_REQUIRE(!c.distill_decoder || c.postnet_layers == 5, FCL_ERR_SHAPE, "fcl_te: decoder distillation taps five postnet layers (..._kd_student.py:778-797)");
        FCL_REQUIRE(!c.distill_encoder || c.econv_layers == 3, FCL_ERR_SHAPE, "fcl_te: encoder distillation taps three encoder convolutions");
    }
    fcl_te* E = new fcl_te();
    E->cfg = c;
    E->n_arenas = c.role == FCL_TE_KD_TEACHER ? 2 : 1;
    *out = E;
    return 0;
}

void fcl_te_destroy(fcl_te_t* E) {
    if (!E) return;
    (void)hipDeviceSynchronize();
    for (void* p : E->form_allocs) (void)hipFree(p);
    if (E->table_dev) (void)hipFree(E->table_dev);
    for (int i = 0; i < 2; ++i) {
        if (E->work[i].base) (void)hipFree(E->work[i].base);
        if (E->zero[i].base) (void)hipFree(E->zero[i].base);
        if (E->bn_ws[i]) (void)hipFree(E->bn_ws[i]);
    }
    for (hipEvent_t ev : E->evpool) (void)hipEventDestroy(ev);
    if (E->pred_ev) (void)hipEventDestroy(E->pred_ev);
    if (E->late_ev) (void)hipEventDestroy(E->late_ev);
    if (E->side) (void)hipStreamDestroy(E->side);
    delete E;
}

int fcl_te_bind_param(fcl_te_t* E, const char* name, float* value, float* grad, int64_t numel) {
    FCL_REQUIRE(E && name && value && numel > 0, FCL_ERR_INVALID, "fcl_te_bind_param: bad argument");
    FCL_REQUIRE(aligned16(value) && (!grad || aligned16(grad)), FCL_ERR_ALIGN, "fcl_te_bind_param: %s is not 16-byte aligned", name);
    Param p;
    p.p = value; p.g = grad; p.numel = numel;
    E->P[name] = p;
    E->finalized = false;
    return 0;
}

int fcl_te_bind_buffer(fcl_te_t* E, const char* name, float* value) {
    FCL_REQUIRE(E && name && value, FCL_ERR_INVALID, "fcl_te_bind_buffer: bad argument");
    E->B[name] = value;
    return 0;
}

int fcl_te_finalize(fcl_te_t* E, uint32_t* status_word) {
    FCL_REQUIRE(E && status_word, FCL_ERR_INVALID, "fcl_te_finalize: null argument");
    const fcl_te_config_t& c = E->cfg;
    const bool train = c.role != FCL_TE_KD_TEACHER;
    auto need = [&](const std::string& n, long long numel) -> int {
        auto it = E->P.find(n);
        FCL_REQUIRE(it != E->P.end(), FCL_ERR_INVALID, "fcl_te_finalize: parameter %s is not bound", n.c_str());
        FCL_REQUIRE(it->second.numel == numel, FCL_ERR_SHAPE, "fcl_te_finalize: parameter %s has %lld elements, expected %lld", n.c_str(), it->second.numel, numel);
        FCL_REQUIRE(!train || it->second.g, FCL_ERR_INVALID, "fcl_te_finalize: parameter %s has no gradient buffer", n.c_str());
        return 0;
    };
    const int C = c.eunits, U = c.dunits, Pn = c.prenet_units, O = c.odim, H = C / 2;
    TE_TRY(need("enc.embed.weight", (long long)c.idim * c.embed_dim));
    auto bn = [&](const std::string& pre, int ch) -> int {
        TE_TRY(need(pre + ".1.weight", ch));
        TE_TRY(need(pre + ".1.bias", ch));
        FCL_REQUIRE(E->B.count(pre + ".1.running_mean") && E->B.count(pre + ".1.running_var"), FCL_ERR_INVALID, "fcl_te_finalize: running statistics of %s are not bound",
                    pre.c_str());
        return 0;
    };
    for (int i = 0; i < c.econv_layers; ++i) {
        const std::string pre = "enc.convs." + std::to_string(i);
        TE_TRY(need(pre + ".0.weight", (long long)c.econv_chans * (i == 0 ? c.embed_dim : c.econv_chans) * c.econv_filts));
        TE_TRY(bn(pre, c.econv_chans));
    }
    for (const char* sfx : {"", "_reverse"}) {
        TE_TRY(need(std::string("enc.blstm.weight_ih_l0") + sfx, (long long)4 * H * c.econv_chans));
        TE_TRY(need(std::string("enc.blstm.weight_hh_l0") + sfx, (long long)4 * H * H));
        TE_TRY(need(std::string("enc.blstm.bias_ih_l0") + sfx, 4 * H));
        TE_TRY(need(std::string("enc.blstm.bias_hh_l0") + sfx, 4 * H));
    }
    auto pred = [&](const std::string& nm, int layers, int chans, int k) -> int {
        for (int i = 0; i < layers; ++i) {
            const std::string pre = nm + ".conv." + std::to_string(i);
            TE_TRY(need(pre + ".0.weight", (long long)chans * (i == 0 ? C : chans) * k));
            TE_TRY(need(pre + ".0.bias", chans));
            TE_TRY(need(pre + ".2.weight", chans));
            TE_TRY(need(pre + ".2.bias", chans));
        }
        TE_TRY(need(nm + ".linear.weight", chans));
        TE_TRY(need(nm + ".linear.bias", 1));
        return 0;
    };
    TE_TRY(pred("duration_predictor", c.dp_layers, c.dp_chans, c.dp_kernel));
    TE_TRY(pred("pitch_predictor", c.vp_layers, c.vp_chans, c.vp_kernel));
    TE_TRY(pred("energy_predictor", c.vp_layers, c.vp_chans, c.vp_kernel));
    for (const char* nm : {"pitch", "energy"}) {
        TE_TRY(need(std::string(nm) + "_embed.0.weight", (long long)C * c.ve_kernel));
        TE_TRY(need(std::string(nm) + "_embed.0.bias", C));
    }
    TE_TRY(need("dec.prenet.prenet.0.0.weight", (long long)Pn * O));
    TE_TRY(need("dec.prenet.prenet.0.0.bias", Pn));
    TE_TRY(need("dec.prenet.prenet.1.0.weight", (long long)Pn * Pn));
    TE_TRY(need("dec.prenet.prenet.1.0.bias", Pn));
    TE_TRY(need("dec.lstm.0.cell.weight_ih", (long long)4 * U * (C + Pn + 1)));
    TE_TRY(need("dec.lstm.0.cell.weight_hh", (long long)4 * U * U));
    TE_TRY(need("dec.lstm.1.cell.weight_ih", (long long)4 * U * U));
    TE_TRY(need("dec.lstm.1.cell.weight_hh", (long long)4 * U * U));
    for (int l = 0; l < 2; ++l)
        for (const char* bnm : {"bias_ih", "bias_hh"}) TE_TRY(need("dec.lstm." + std::to_string(l) + ".cell." + bnm, 4 * U));
    TE_TRY(need("dec.feat_out.weight", (long long)O * (U + C)));
    for (int i = 0; i < c.postnet_layers; ++i) {
        const std::string pre = "dec.postnet.postnet." + std::to_string(i);
        const int cout = i == c.postnet_layers - 1 ? O : c.postnet_chans, cin = i == 0 ? O : c.postnet_chans;
        TE_TRY(need(pre + ".0.weight", (long long)cout * cin * c.postnet_filts));
        TE_TRY(bn(pre, cout));
    }
    if (c.role == FCL_TE_STUDENT) {
        if (c.distill_encoder) {
            TE_TRY(need("enc.embed_proj.weight", (long long)c.t_embed_dim * c.embed_dim));
            for (int i = 0; i < (c.share_proj ? 1 : 3); ++i) TE_TRY(need("enc.convs_proj." + std::to_string(i) + ".weight", (long long)c.t_econv_chans * c.econv_chans));
            TE_TRY(need("enc.blstm_proj.weight", (long long)c.t_eunits * C));
        }
        if (c.distill_decoder) {
            TE_TRY(need("dec.prenet_proj.weight", (long long)c.t_prenet_units * Pn));
            if (c.share_proj) {
                TE_TRY(need("dec.lstm_proj.weight", (long long)c.t_dunits * U));
                TE_TRY(need("dec.post_proj.weight", (long long)c.t_postnet_chans * c.postnet_chans));
            } else {
                TE_TRY(need("dec.lstm0_proj.weight", (long long)c.t_dunits * U));
                TE_TRY(need("dec.lstm1_proj.weight", (long long)c.t_dunits * U));
                for (int i = 0; i < 4; ++i) TE_TRY(need("dec.post" + std::to_string(i) + "_proj.weight", (long long)c.t_postnet_chans * c.postnet_chans));
            }
        }
        if (c.distill_prosody) {
            TE_TRY(need("pemb_proj.weight", (long long)c.t_eunits * C));
            TE_TRY(need("eemb_proj.weight", (long long)c.t_eunits * C));
        }
    }
    if (!E->side) {
        {   // FCL_TE_SIDE_CUS=n: the weight-gradient stream dispatches to n compute units only (fcl_stream_create_cus; 0 = all)
            fcl_stream_t st = nullptr;
            TE_TRY(fcl_stream_create_cus(tunable("TE_SIDE_CUS", 0), &st));
            E->side = (hipStream_t)st;
        }
        E->evpool.resize(64);
        for (hipEvent_t& ev : E->evpool) FCL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        FCL_HIP(hipEventCreateWithFlags(&E->pred_ev, hipEventDisableTiming));
        FCL_HIP(hipEventCreateWithFlags(&E->late_ev, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) {
            void* p = nullptr;
            FCL_HIP(hipMalloc(&p, 4096 * sizeof(double)));
            FCL_HIP(hipMemset(p, 0, 4096 * sizeof(double)));
            E->bn_ws[i] = static_cast<double*>(p);
        }
    }
    E->status = status_word;
    E->finalized = true;
    E->params_dirty = true;
    return 0;
}

int fcl_te_params_changed(fcl_te_t* E) {
    FCL_REQUIRE(E, FCL_ERR_INVALID, "fcl_te_params_changed: null engine");
    E->params_dirty = true;
    return 0;
}

fcl_stream_t fcl_te_side_stream(fcl_te_t* E) { return E ? (fcl_stream_t)E->side : nullptr; }

/* Before the first pass: if the engine's weight-gradient stream shares a compute pipe with the stream the caller will run the passes on (measured,
 * fcl_streams_share_pipe), it is replaced by one that does not.  *moved (optional) = 1 when it was replaced.  The old handle is invalid afterwards. */
int fcl_te_place_streams(fcl_te_t* E, fcl_stream_t main_stream, int* moved) {
    FCL_REQUIRE(E && E->finalized && E->side, FCL_ERR_INVALID, "fcl_te_place_streams: the engine is not finalized");
    if (moved) *moved = 0;
    if (tunable("TE_SIDE_CUS", 0) > 0) return 0;  // a CU-masked queue contends with every other queue whatever its pipe (12 of 12 candidates, round 6): left where it is
    int shared = 0;
    TE_TRY(fcl_streams_share_pipe(main_stream, (fcl_stream_t)E->side, &shared, nullptr));
    if (!shared) return 0;
    fcl_stream_t others[1] = {main_stream}, fresh = nullptr;
    if (fcl_stream_create_apart(others, 1, &fresh, nullptr) != 0) {  // no candidate fits (a process holding dozens of streams): keep what there is, say so
        if (moved) *moved = -1;
        return 0;
    }
    (void)hipStreamDestroy(E->side);
    E->side = (hipStream_t)fresh;
    if (moved) *moved = 1;
    return 0;
}
int64_t fcl_te_last_launches(fcl_te_t* E) { return E ? E->last_launches : -1; }
int64_t fcl_te_arena_bytes(fcl_te_t* E) { return E ? (int64_t)(E->work[0].cap + E->zero[0].cap + E->work[1].cap + E->zero[1].cap) : -1; }

}  // extern "C"

static int te_check_batch(const fcl_te_batch_t* b) {
    FCL_REQUIRE(b, FCL_ERR_INVALID, "fcl_te: null batch");
    FCL_REQUIRE(b->B > 0 && b->T > 0 && b->L > 0 && b->N > 0 && b->F > 0 && b->lmax > 0, FCL_ERR_SHAPE, "fcl_te: empty batch (B=%d T=%d L=%d N=%d F=%d lmax=%d)", b->B, b->T,
                b->L, b->N, b->F, b->lmax);
    FCL_REQUIRE(b->xs && b->ys && b->f0 && b->energy && b->ds && b->lens && b->e_lo && b->e_hi && b->f_lo && b->f_hi && b->src_sorted && b->row_of_enc && b->cell_frame &&
                    b->frame_cell && b->prev_frame && b->cell_row && b->dur && b->perm_tb && b->cell_row_i64 && b->enc_pad && b->enc_valid && b->frame_valid &&
                    b->cell_valid && b->pos4 && b->live_rows_host,
                FCL_ERR_INVALID, "fcl_te: null batch pointer");
    long long f = 0;
    int prev = b->N;
    FCL_REQUIRE(b->live_rows_host[0] == b->N, FCL_ERR_INVALID, "fcl_te: live_rows_host[0] must equal N");
    for (int t = 0; t < b->lmax; ++t) {
        FCL_REQUIRE(b->live_rows_host[t] > 0 && b->live_rows_host[t] <= prev, FCL_ERR_INVALID, "fcl_te: live_rows_host must be positive and non-increasing");
        prev = b->live_rows_host[t];
        f += prev;
    }
    FCL_REQUIRE(f == b->F, FCL_ERR_SHAPE, "fcl_te: live_rows_host sums to %lld cells, F = %d", f, b->F);
    return 0;
}

// size the arenas by a dry run of `body` (no launches), grow them if needed (rare: the first steps), then clear what the previous pass used of the
// zero arena -- ordered behind everything the weight-gradient stream still holds
// variant: what else gates an allocation besides the six counts -- 1 = forward only (fcl_te_knowledge), 2 = the whole update, 3 = the whole update
// fed a FRAME-major knowledge tuple (te_losses gathers three extra cell-major copies).  Each form allocates a superset of the one below it, so a
// verified pass covers every pass with a variant and counts that are no larger (the key used to be the six counts alone).
template <typename Body>
static int te_run_sized(fcl_te& E, int variant, Body body) {
    Arena &W = E.work[E.cur_arena], &Z = E.zero[E.cur_arena];
    const fcl_te_batch_t& b = E.c.b;
    const std::array<long long, 7> dims = {b.B, b.T, b.L, b.N, b.F, b.lmax, variant};
    std::vector<std::array<long long, 7>>& ok = E.sized[E.cur_arena];
    bool fits = false;
    if (W.base != nullptr && Z.base != nullptr)
        for (const auto& v : ok) {
            bool le = true;
            for (int i = 0; i < 7; ++i) le = le && dims[i] <= v[i];
            fits = fits || le;
        }
    if (!fits) {  // every allocation of a pass is a product of these six counts and configuration constants: a batch that is no larger in any
                  // of them than one a dry run has verified fits as well -- the dry run (all the host bookkeeping of a step, once more) is skipped
        E.dry = true;
        W.dry = Z.dry = true;
        W.reset();
        Z.reset();
        const int64_t l0 = E.launches;
        const bool pp = E.pred_pending, lp = E.late_pending, dp = E.dw_pending, pd = E.params_dirty;
        int rc = body();
        E.dry = false;
        W.dry = Z.dry = false;
        E.launches = l0;
        E.pred_pending = pp; E.late_pending = lp; E.dw_pending = dp; E.params_dirty = pd || E.forms_new;
        E.forms_new = false;
        if (rc) return rc;
        const size_t need_w = W.used + 4096, need_z = Z.used + 4096;
        if (need_w > W.cap || need_z > Z.cap) {
            FCL_HIP(hipStreamSynchronize(E.main));
            FCL_HIP(hipStreamSynchronize(E.side));
            if (need_w > W.cap) {
                if (W.base) FCL_HIP(hipFree(W.base));
                W.base = nullptr;
                W.cap = need_w + need_w / 4;
                void* p = nullptr;
                FCL_HIP(hipMalloc(&p, W.cap));
                W.base = static_cast<char*>(p);
            }
            if (need_z > Z.cap) {
                if (Z.base) FCL_HIP(hipFree(Z.base));
                Z.base = nullptr;
                Z.cap = need_z + need_z / 4;
                void* p = nullptr;
                FCL_HIP(hipMalloc(&p, Z.cap));
                Z.base = static_cast<char*>(p);
                FCL_HIP(hipMemsetAsync(Z.base, 0, Z.cap, E.main));
                Z.dirty = 0;
            }
        }
        if (ok.size() >= 32) ok.erase(ok.begin());
        ok.push_back(dims);  // verified: the arenas (which only ever grow) hold this geometry
    }
    W.reset();
    Z.reset();
    return 0;
}

// clear what earlier passes dirtied of the zero arena (one launch for every accumulation target of a pass)
static int te_clear_zero(fcl_te& E) {
    Arena& Z = E.zero[E.cur_arena];
    if (Z.dirty) FCL_HIP(hipMemsetAsync(Z.base, 0, Z.dirty, E.main));
    Z.dirty = 0;
    return 0;
}

extern "C" {

int fcl_te_knowledge(fcl_te_t* Ep, const fcl_te_batch_t* batch, uint32_t draw, fcl_te_knowledge_t* know, fcl_stream_t stream) {
    FCL_REQUIRE(Ep && know, FCL_ERR_INVALID, "fcl_te_knowledge: null argument");
    fcl_te& E = *Ep;
    FCL_REQUIRE(E.finalized, FCL_ERR_INVALID, "fcl_te_knowledge: fcl_te_finalize has not run");
    TE_TRY(te_check_batch(batch));
    E.main = E.cur = (hipStream_t)stream;
    E.cur_arena = (E.cur_arena + 1) % E.n_arenas;
    E.c.b = *batch;
    E.c.save = false;
    E.c.draw = draw;
    const int64_t l0 = E.launches;
    // the frozen teacher enqueues nothing on its weight-gradient stream: nothing to order the clear behind
    TE_TRY(te_run_sized(E, 1, [&]() { return te_forward(E); }));
    TE_TRY(te_clear_zero(E));
    TE_TRY(te_forward(E));
    Ctx& c = E.c;
    memset(know, 0, sizeof(*know));
    know->after = c.after;
    know->before = c.before;
    for (int i = 0; i < 5 && i < (int)c.enc_taps.size(); ++i) know->enc[i] = c.enc_taps[i];
    know->dec[0] = c.p1d;
    know->dec[1] = c.h0_all;
    know->dec[2] = c.h1_all;
    for (int i = 0; i < 5 && i < (int)c.post_taps.size(); ++i) know->dec[3 + i] = c.post_taps[i];
    know->pro[0] = c.dur.out;
    know->pro[1] = c.pit.out;
    know->pro[2] = c.en.out;
    know->pro[3] = c.p_embs;
    know->pro[4] = c.e_embs;
    know->dec_cell_major = 1;
    E.last_launches = E.launches - l0;
    return 0;
}

int fcl_te_forward_backward(fcl_te_t* Ep, const fcl_te_batch_t* batch, const fcl_te_knowledge_t* know, uint32_t draw, double* loss_sums_host, uint32_t* status_host,
                            fcl_stream_t stream) {
    FCL_REQUIRE(Ep && loss_sums_host, FCL_ERR_INVALID, "fcl_te_forward_backward: null argument");
    fcl_te& E = *Ep;
    FCL_REQUIRE(E.finalized, FCL_ERR_INVALID, "fcl_te_forward_backward: fcl_te_finalize has not run");
    FCL_REQUIRE(E.cfg.role != FCL_TE_KD_TEACHER, FCL_ERR_INVALID, "fcl_te_forward_backward: the frozen KD teacher computes no loss");
    TE_TRY(te_check_batch(batch));
    E.main = E.cur = (hipStream_t)stream;
    E.cur_arena = 0;
    E.c.b = *batch;
    E.c.save = true;
    E.c.draw = draw;
    E.stage_done = -1;
    const int64_t l0 = E.launches;
    fcl_te_knowledge_t kcopy{};
    if (know) kcopy = *know;
    const fcl_te_knowledge_t* kp = know ? &kcopy : nullptr;
    auto whole = [&]() -> int {
        TE_TRY(te_forward(E));
        TE_TRY(te_losses(E, kp));
        TE_TRY(te_backward_stage0(E));
        TE_TRY(te_backward_stage1(E));
        TE_TRY(te_backward_stage2(E));
        TE_TRY(te_backward_stage3(E));
        return 0;
    };
    TE_TRY(te_run_sized(E, (kp && !kp->dec_cell_major) ? 3 : 2, whole));
    // the arena's clear must not depend on the previous step having ended with its join: order this step behind everything the weight-gradient
    // stream still holds, then clear what the previous step used of the zero arena (one launch for every accumulation target of the step)
    TE_TRY(ev_wait(E, E.main, E.side));
    E.pred_pending = E.late_pending = E.dw_pending = false;
    TE_TRY(te_clear_zero(E));
    static const int stamps_on = tunable("TE_STAMPS", 0);
    E.stamps = stamps_on != 0;
    for (bool& b : E.stamp_set) b = false;
    stamp(E, 0);
    TE_TRY(te_forward(E));
    stamp(E, 4);
    TE_TRY(te_losses(E, kp));
    stamp(E, 5);
    TE_TRY(te_backward_stage0(E));
    stamp(E, 6);
    // the named losses: complete once the late terms are joined (stage 0 ends with that join)
    // (after an arena overflow take() hands out the arena's base -- TE_L refuses every launch from then on, and so must the copies below: c.sums
    // would alias live memory)
    FCL_REQUIRE(!E.work[E.cur_arena].overflow && !E.zero[E.cur_arena].overflow, FCL_ERR_WORKSPACE, "fcl_te: the pass needs more arena memory than its sizing run found");
    FCL_HIP(hipMemcpyAsync(loss_sums_host, E.c.sums, FCL_TE_MAX_LOSSES * 3 * sizeof(double), hipMemcpyDeviceToHost, E.main));
    if (status_host) FCL_HIP(hipMemcpyAsync(status_host, E.status, sizeof(uint32_t), hipMemcpyDeviceToHost, E.main));
    E.stage_done = 0;
    E.last_launches = E.launches - l0;
    return 0;
}

int fcl_te_backward_stage(fcl_te_t* Ep, int stage, fcl_stream_t stream) {
    FCL_REQUIRE(Ep, FCL_ERR_INVALID, "fcl_te_backward_stage: null engine");
    fcl_te& E = *Ep;
    FCL_REQUIRE(stage >= 1 && stage <= 3 && stage == E.stage_done + 1, FCL_ERR_INVALID, "fcl_te_backward_stage: stage %d after stage %d", stage, E.stage_done);
    FCL_REQUIRE((hipStream_t)stream == E.main, FCL_ERR_INVALID, "fcl_te_backward_stage: the stages of one step run on one stream");
    const int64_t l0 = E.launches;
    E.cur = E.main;
    if (stage == 1) TE_TRY(te_backward_stage1(E));
    if (stage == 2) TE_TRY(te_backward_stage2(E));
    if (stage == 3) TE_TRY(te_backward_stage3(E));
    stamp(E, 6 + stage);
    E.stage_done = stage;
    E.last_launches += E.launches - l0;
    return 0;
}

/* the join of the weight-gradient stream at the end of backward (after the caller has issued its last bucket's collective from that stream) */
int fcl_te_join(fcl_te_t* Ep, fcl_stream_t stream) {
    FCL_REQUIRE(Ep, FCL_ERR_INVALID, "fcl_te_join: null engine");
    fcl_te& E = *Ep;
    E.main = (hipStream_t)stream;
    E.dw_pending = true;
    const int rc = join_dw(E);
    stamp(E, 10);
    return rc;
}

/* developer aid (FCL_TE_STAMPS=1): milliseconds from the start of the last fcl_te_forward_backward to each phase boundary on the main stream
 * (see stamp(); -1 = not recorded); synchronises on the last recorded event */
int fcl_te_phase_ms(fcl_te_t* E, float* out12) {
    FCL_REQUIRE(E && out12, FCL_ERR_INVALID, "fcl_te_phase_ms: null argument");
    for (int i = 0; i < 12; ++i) {
        out12[i] = -1.f;
        if (i == 0 || !E->stamp_set[i] || !E->stamp_set[0]) continue;
        if (hipEventSynchronize(E->stamp_ev[i]) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, E->stamp_ev[0], E->stamp_ev[i]) == hipSuccess) out12[i] = ms;
    }
    if (E->stamp_set[0]) out12[0] = 0.f;
    return 0;
}

}  // extern "C"
