// capi.hip — extern "C" surface of libfcl_hip.so: error plumbing, GEMM-backed ops, the decoder loop.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "fcl_common.h"

namespace fcl {

static thread_local char g_err[512] = "";
static thread_local int g_gemm_mode = FCL_GEMM_F32;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    static const int sync_debug = getenv("FCL_SYNC_DEBUG") ? atoi(getenv("FCL_SYNC_DEBUG")) : 0;  // developer aid: name every launch, then wait for it
    if (sync_debug && e == hipSuccess) {
        fprintf(stderr, "[fcl] %s\n", what);
        fflush(stderr);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) return 0;
    set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    return FCL_ERR_HIP;
}

int gemm_mode() { return g_gemm_mode; }

int tunable(const char* name, int dflt) {
    char key[64];
    snprintf(key, sizeof(key), "FCL_%s", name);
    const char* v = getenv(key);
    return v && *v ? atoi(v) : dflt;
}

int ensure_dyn_lds(const void* func, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> granted;  // (kernel, device) -> bytes already opted in
    int dev = 0;
    FCL_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    int& have = granted[std::make_pair(func, dev)];
    if (have >= bytes) return 0;
    FCL_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
    return 0;
}

// ---- profiling records ------------------------------------------------------------------------------
bool g_prof_on = false;
struct ProfRec { char name[56]; double flops, rows, fill; hipEvent_t a, b; };
static std::vector<ProfRec> g_prof;

void prof_begin(const char* name, double flops, double rows, hipStream_t s, double fill_bytes) {
    ProfRec r{};
    strncpy(r.name, name, sizeof(r.name) - 1);
    r.flops = flops;
    r.rows = rows;
    r.fill = fill_bytes;
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
}
void prof_end(hipStream_t s) { (void)hipEventRecord(g_prof.back().b, s); }

__global__ void zero_kernel(float* p, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.f;
}

static inline size_t align_up(size_t x) { return (x + 127) & ~(size_t)127; }  // P32 planes want 128-byte lines

struct DecoderWs {
    float *G0, *F0, *pre_a, *pre_b, *h0[2], *c0, *h1[2], *c1, *prev;
    float *h2[2], *c2;  // third cell (fcl_decoder_weights_t.dlayers == 3)
    unsigned short *h0_p[2], *h1_p[2], *pre_p;  // P32 planes of the recurrent states / the prenet output (the LSTM steps' pre-split operands)
    size_t bytes, state_bytes;                  // state_bytes: h0 .. h1_p, zeroed before the loop
};

// P32 planes are used by the decoder loop when the plan provides them and the widths are whole 32-column lines
static bool decoder_planes(const fcl_decoder_weights_t* w) {
    return w->w0_att_p && w->wf_att_p && w->w0_pre_p && w->w0_hh_p && w->w1_ih_p && w->w1_hh_p && !(w->c & 31) && !(w->p & 31) && !(w->u & 31);
}

static DecoderWs carve(const fcl_decoder_weights_t* w, int n, void* base) {
    DecoderWs ws;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += align_up(floats * sizeof(float));
        return p;
    };
    const size_t N = (size_t)n;
    const size_t RF = w->reduction_factor > 1 ? (size_t)w->reduction_factor : 1;  // frames per step: F0 / prev hold a step's r frames
    ws.G0 = take(N * 4 * w->u);
    ws.F0 = take(N * w->odim * RF);
    ws.pre_a = take(N * w->p);
    ws.pre_b = take(N * w->p);
    // what the loop READS before it writes — the states entering step 0 (fp32 and, for the pre-split path, planes) and prev_out — is contiguous, so
    // one kernel zeroes exactly that; the ping-pong partners are written by step 0 before anything reads them
    ws.h0[0] = take(N * w->u);
    ws.c0 = take(N * w->u);
    ws.h1[0] = take(N * w->u);
    ws.c1 = take(N * w->u);
    ws.h2[0] = ws.h2[1] = ws.c2 = nullptr;
    if (w->dlayers == 3) {  // inside the zeroed region
        ws.h2[0] = take(N * w->u);
        ws.c2 = take(N * w->u);
    }
    ws.h0_p[0] = reinterpret_cast<unsigned short*>(take(N * w->u));  // as many bytes as the fp32 form (2 x 2 bytes per element)
    ws.h1_p[0] = reinterpret_cast<unsigned short*>(take(N * w->u));
    ws.prev = take(N * w->odim * RF);
    ws.state_bytes = off - (size_t)(reinterpret_cast<char*>(ws.h0[0]) - reinterpret_cast<char*>(base));
    ws.h0[1] = take(N * w->u);
    ws.h1[1] = take(N * w->u);
    ws.h0_p[1] = reinterpret_cast<unsigned short*>(take(N * w->u));
    ws.h1_p[1] = reinterpret_cast<unsigned short*>(take(N * w->u));
    ws.pre_p = reinterpret_cast<unsigned short*>(take(N * w->p));
    if (w->dlayers == 3) ws.h2[1] = take(N * w->u);
    ws.bytes = off;
    return ws;
}

}  // namespace fcl

using namespace fcl;

extern "C" {

const char* fcl_last_error(void) { return g_err; }
int fcl_version(void) { return FCL_ABI_VERSION; }

int fcl_set_gemm_mode(int mode) {
    FCL_REQUIRE(mode == FCL_GEMM_F32 || mode == FCL_GEMM_BF16, FCL_ERR_INVALID, "set_gemm_mode: unknown mode %d", mode);
    FCL_REQUIRE(mode == FCL_GEMM_F32 || tunable("PRECISION", 1) != 0, FCL_ERR_INVALID,
                "set_gemm_mode: FCL_GEMM_BF16 needs the bf16 MFMA path (FCL_PRECISION=0 selects exact fp32 MFMAs)");
    g_gemm_mode = mode;
    return 0;
}
int fcl_get_gemm_mode(void) { return g_gemm_mode; }

int fcl_linear_fwd(const float* x, int lda, const float* w, int ldw, const float* bias, float* y, int ldy, int m, int n,
                   int k, int act, fcl_stream_t stream) {
    FCL_REQUIRE(x && w && y, FCL_ERR_INVALID, "linear_fwd: null argument");
    FCL_REQUIRE(act >= FCL_ACT_NONE && act <= FCL_ACT_TANH, FCL_ERR_INVALID, "linear_fwd: bad act %d", act);
    FCL_REQUIRE(lda >= k && ldw >= k && ldy >= n, FCL_ERR_SHAPE, "linear_fwd: leading dimensions too small");
    GemmArgs g = {};
    g.term[0] = GemmTerm{x, w, lda, ldw, k, 0};
    g.nterms = 1;
    g.M = m;
    g.N = n;
    g.bias = bias;
    g.act = act;
    g.Y = y;
    g.ldy = ldy;
    return launch_gemm(g, (hipStream_t)stream);
}

int fcl_linear2_fwd(const float* x, int lda, const float* w, int ldw, int k, const float* x2, int lda2, const float* w2, int ldw2, int k2, const float* bias,
                    const float* residual, int ldr, float* y, int ldy, int m, int n, int act, fcl_stream_t stream) {
    FCL_REQUIRE(x && w && y, FCL_ERR_INVALID, "linear2_fwd: null argument");
    FCL_REQUIRE((x2 == nullptr) == (w2 == nullptr), FCL_ERR_INVALID, "linear2_fwd: x2 and w2 come in pairs");
    FCL_REQUIRE(act >= FCL_ACT_NONE && act <= FCL_ACT_TANH, FCL_ERR_INVALID, "linear2_fwd: bad act %d", act);
    FCL_REQUIRE(lda >= k && ldw >= k && ldy >= n && (!x2 || (lda2 >= k2 && ldw2 >= k2)) && (!residual || ldr >= n), FCL_ERR_SHAPE,
                "linear2_fwd: leading dimensions too small");
    GemmArgs g = {};
    g.term[0] = GemmTerm{x, w, lda, ldw, k, 0};
    g.nterms = 1;
    if (x2) {
        g.term[1] = GemmTerm{x2, w2, lda2, ldw2, k2, 0};
        g.nterms = 2;
    }
    g.M = m;
    g.N = n;
    g.bias = bias;
    g.act = act;
    g.R = residual;
    g.ldr = ldr;
    g.Y = y;
    g.ldy = ldy;
    return launch_gemm(g, (hipStream_t)stream);
}

int fcl_conv1d_fwd(const float* x, const float* wp, const float* bias, const int32_t* seg_lo, const int32_t* seg_hi,
                   const float* residual, float* y, int m, int cin, int cout, int k, int act, fcl_stream_t stream) {
    FCL_REQUIRE(x && wp && y && seg_lo && seg_hi, FCL_ERR_INVALID, "conv1d_fwd: null argument");
    FCL_REQUIRE(k >= 1 && (k & 1) && k <= FCL_MAX_TERMS, FCL_ERR_SHAPE, "conv1d_fwd: kernel size %d must be odd and <= %d", k, FCL_MAX_TERMS);
    FCL_REQUIRE(act >= FCL_ACT_NONE && act <= FCL_ACT_TANH, FCL_ERR_INVALID, "conv1d_fwd: bad act %d", act);
    FCL_REQUIRE(y != x, FCL_ERR_INVALID, "conv1d_fwd: in-place convolution is not supported");
    GemmArgs g = {};
    for (int j = 0; j < k; ++j) g.term[j] = GemmTerm{x, wp + (size_t)j * cout * cin, cin, cin, cin, j - (k - 1) / 2};
    g.nterms = k;
    g.conv_k = k >= 3 ? k : 0;  // one Conv1d: same input, tap-major contiguous weights -> the stencil kernel may load each input tile once (exact-fp32 lines, round 5)
    g.M = m;
    g.N = cout;
    g.seg_lo = seg_lo;
    g.seg_hi = seg_hi;
    g.bias = bias;
    g.act = act;
    g.R = residual;
    g.ldr = cout;
    g.Y = y;
    g.ldy = cout;
    return launch_gemm(g, (hipStream_t)stream);
}

int fcl_linear_planes_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const float* bias, float* y, int ldy, uint16_t* yp,
                          int m, int n, int k, int act, fcl_stream_t stream) {
    FCL_REQUIRE(xp && wpp && (y || yp), FCL_ERR_INVALID, "linear_planes_fwd: null argument");
    FCL_REQUIRE(act >= FCL_ACT_NONE && act <= FCL_ACT_TANH, FCL_ERR_INVALID, "linear_planes_fwd: bad act %d", act);
    FCL_REQUIRE(ldxp * 32 >= k && (!y || ldy >= n), FCL_ERR_SHAPE, "linear_planes_fwd: leading dimensions too small");
    GemmArgs g = {};
    g.term[0].K = k;
    g.term[0].Ap = xp; g.term[0].lda_p = ldxp;
    g.term[0].Wp = wpp; g.term[0].ldw_p = (k + 31) / 32;
    g.nterms = 1;
    g.M = m; g.N = n; g.bias = bias; g.act = act;
    g.Y = y; g.ldy = ldy;
    g.Yp = yp; g.ldyp = (n + 31) / 32;
    return launch_gemm(g, (hipStream_t)stream);
}

/* y = x . W^T as fcl_linear_planes_fwd, the masked MSE against `target` and its gradient in ONE launch (the GEMM's epilogue): with d = y - target on the rows
 * where row_valid is set (all rows when NULL), grad / grad_p (planes) receive 2 d / count (0 on the other rows) and sums[0 .. 2] += sum |d|, sum d^2, number of
 * elements (fp64 atomics; the slots fcl_loss_terms_batch fills).  y itself is never stored.  Planes kernels only (refused under FCL_PLANES=0). */
int fcl_linear_planes_mse_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const float* target, int ld_t, const uint8_t* row_valid, double count,
                              float* grad, int ldg, uint16_t* grad_p, double* sums, int m, int n, int k, fcl_stream_t stream) {
    FCL_REQUIRE(xp && wpp && target && sums && (grad || grad_p) && count > 0.0, FCL_ERR_INVALID, "linear_planes_mse_fwd: null argument / count <= 0");
    FCL_REQUIRE(ldxp * 32 >= k && (!grad || ldg >= n) && ld_t >= n, FCL_ERR_SHAPE, "linear_planes_mse_fwd: leading dimensions too small");
    FCL_REQUIRE((n & 3) == 0 && (ld_t & 3) == 0 && (reinterpret_cast<uintptr_t>(target) & 15u) == 0, FCL_ERR_ALIGN,
                "linear_planes_mse_fwd: n %% 4 == 0 and a 16-byte aligned target with ld_t %% 4 == 0 required");
    GemmArgs g = {};
    g.term[0].K = k;
    g.term[0].Ap = xp; g.term[0].lda_p = ldxp;
    g.term[0].Wp = wpp; g.term[0].ldw_p = (k + 31) / 32;
    g.nterms = 1;
    g.M = m; g.N = n; g.act = FCL_ACT_NONE;
    g.Y = grad; g.ldy = ldg;
    g.Yp = grad_p; g.ldyp = (n + 31) / 32;
    g.loss_t = target; g.ld_lt = ld_t; g.loss_valid = row_valid; g.loss_gscale = (float)(1.0 / count); g.loss_sums = sums;
    FCL_REQUIRE(tunable("PRECISION", 1) != 0 && planes_ok(g.term, 1), FCL_ERR_INVALID, "linear_planes_mse_fwd: the planes kernels are off (FCL_PLANES=0 / FCL_PRECISION=0)");
    if (m == 0) return 0;
    return launch_gemm(g, (hipStream_t)stream);
}

int fcl_conv1d_planes_rows_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const float* bias, const int32_t* seg_lo, const int32_t* seg_hi,
                               const float* residual, float* y, uint16_t* yp, int m, int cin, int cout, int k, int act, const int32_t* m_dev,
                               fcl_stream_t stream) {
    FCL_REQUIRE(xp && wpp && (y || yp) && seg_lo && seg_hi, FCL_ERR_INVALID, "conv1d_planes_fwd: null argument");
    FCL_REQUIRE(k >= 1 && (k & 1) && k <= FCL_MAX_TERMS, FCL_ERR_SHAPE, "conv1d_planes_fwd: kernel size %d must be odd and <= %d", k, FCL_MAX_TERMS);
    FCL_REQUIRE(act >= FCL_ACT_NONE && act <= FCL_ACT_TANH, FCL_ERR_INVALID, "conv1d_planes_fwd: bad act %d", act);
    FCL_REQUIRE(ldxp * 32 >= cin, FCL_ERR_SHAPE, "conv1d_planes_fwd: input planes narrower than Cin");
    const int ldw = (cin + 31) / 32;
    GemmArgs g = {};
    for (int j = 0; j < k; ++j) {
        g.term[j].K = cin;
        g.term[j].shift = j - (k - 1) / 2;
        g.term[j].Ap = xp; g.term[j].lda_p = ldxp;
        g.term[j].Wp = wpp + (size_t)j * cout * ldw * 64; g.term[j].ldw_p = ldw;
    }
    g.nterms = k;
    g.conv_k = k;
    g.M = m; g.N = cout;
    g.seg_lo = seg_lo; g.seg_hi = seg_hi;
    g.bias = bias; g.act = act;
    g.R = residual; g.ldr = cout;
    g.Y = y; g.ldy = cout;
    g.Yp = yp; g.ldyp = (cout + 31) / 32;
    g.m_dev = m_dev;
    return launch_gemm(g, (hipStream_t)stream);
}

int fcl_conv1d_planes_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const float* bias, const int32_t* seg_lo, const int32_t* seg_hi,
                          const float* residual, float* y, uint16_t* yp, int m, int cin, int cout, int k, int act, fcl_stream_t stream) {
    return fcl_conv1d_planes_rows_fwd(xp, ldxp, wpp, bias, seg_lo, seg_hi, residual, y, yp, m, cin, cout, k, act, nullptr, stream);
}

/* Conv1d (no bias) on pre-split operands with train-mode BatchNorm statistics from the GEMM's epilogue (round 6): z [m, cout] as fcl_conv1d_planes_fwd, and mean /
 * invstd / running statistics exactly as fcl_bn_stats_ws_fwd(z, ...) would leave them (same fp64 sums, same ticketed finalize, `zero_workspace` = the same 2 cout
 * doubles + tickets, zero on entry and on exit) -- without that kernel's pass over z.  Planes kernels only (FCL_ERR_INVALID under FCL_PLANES=0 / FCL_PRECISION=0). */
int fcl_conv1d_planes_bn_fwd(const uint16_t* xp, int ldxp, const uint16_t* wpp, const int32_t* seg_lo, const int32_t* seg_hi, float* z, int m, int cin, int cout, int k,
                             float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var, double* zero_workspace,
                             fcl_stream_t stream) {
    FCL_REQUIRE(xp && wpp && z && seg_lo && seg_hi && mean && invstd && zero_workspace && m > 0, FCL_ERR_INVALID, "conv1d_planes_bn_fwd: null argument / m <= 0");
    FCL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), FCL_ERR_INVALID, "conv1d_planes_bn_fwd: running statistics come in pairs");
    FCL_REQUIRE(k >= 1 && (k & 1) && k <= FCL_MAX_TERMS, FCL_ERR_SHAPE, "conv1d_planes_bn_fwd: kernel size %d must be odd and <= %d", k, FCL_MAX_TERMS);
    FCL_REQUIRE(ldxp * 32 >= cin, FCL_ERR_SHAPE, "conv1d_planes_bn_fwd: input planes narrower than Cin");
    const int ldw = (cin + 31) / 32;
    GemmArgs g = {};
    for (int j = 0; j < k; ++j) {
        g.term[j].K = cin;
        g.term[j].shift = j - (k - 1) / 2;
        g.term[j].Ap = xp; g.term[j].lda_p = ldxp;
        g.term[j].Wp = wpp + (size_t)j * cout * ldw * 64; g.term[j].ldw_p = ldw;
    }
    g.nterms = k;
    g.conv_k = k >= 3 ? k : 0;
    g.M = m; g.N = cout;
    g.seg_lo = seg_lo; g.seg_hi = seg_hi;
    g.act = FCL_ACT_NONE;
    g.Y = z; g.ldy = cout;
    g.bn_ws = zero_workspace;
    g.bn_tickets = reinterpret_cast<unsigned int*>(zero_workspace + 2 * (size_t)cout);
    g.bn_eps = eps; g.bn_momentum = momentum;
    g.bn_mean = mean; g.bn_invstd = invstd; g.bn_rmean = running_mean; g.bn_rvar = running_var;
    FCL_REQUIRE(tunable("PRECISION", 1) != 0 && planes_ok(g.term, g.nterms), FCL_ERR_INVALID, "conv1d_planes_bn_fwd: the planes kernels are off (FCL_PLANES=0 / FCL_PRECISION=0)");
    return launch_gemm(g, (hipStream_t)stream);
}

int fcl_conv1d_planes_group_fwd(const uint16_t* xp, int ldxp, int64_t x_group_stride, const uint16_t* wpp, const float* bias, const int32_t* seg_lo,
                                const int32_t* seg_hi, float* y, uint16_t* yp, int m, int cin, int cout, int k, int act, int groups, fcl_stream_t stream) {
    FCL_REQUIRE(xp && wpp && (y || yp) && seg_lo && seg_hi && groups >= 1 && groups <= 65535, FCL_ERR_INVALID, "conv1d_planes_group_fwd: bad arguments");
    FCL_REQUIRE(k >= 3 && (k & 1) && k <= FCL_MAX_TERMS, FCL_ERR_SHAPE, "conv1d_planes_group_fwd: kernel size %d must be odd, 3 .. %d", k, FCL_MAX_TERMS);
    FCL_REQUIRE(act >= FCL_ACT_NONE && act <= FCL_ACT_TANH, FCL_ERR_INVALID, "conv1d_planes_group_fwd: bad act %d", act);
    FCL_REQUIRE(ldxp * 32 >= cin && cin <= 384 && !(cout & 31), FCL_ERR_SHAPE, "conv1d_planes_group_fwd: needs Cin <= 384 (the stencil kernel) and Cout %% 32 == 0");
    FCL_REQUIRE(tunable("PRECISION", 1) != 0 && tunable("PLANES", 1) != 0 && tunable("PCONV", 1) != 0, FCL_ERR_INVALID,
                "conv1d_planes_group_fwd: the pre-split stencil path is off (FCL_PRECISION / FCL_PLANES / FCL_PCONV)");
    const int ldw = (cin + 31) / 32;
    GemmArgs g = {};
    for (int j = 0; j < k; ++j) {
        g.term[j].K = cin;
        g.term[j].shift = j - (k - 1) / 2;
        g.term[j].Ap = xp; g.term[j].lda_p = ldxp;
        g.term[j].Wp = wpp + (size_t)j * cout * ldw * 64; g.term[j].ldw_p = ldw;
    }
    g.nterms = k;
    g.conv_k = k;
    g.M = m; g.N = cout;
    g.seg_lo = seg_lo; g.seg_hi = seg_hi;
    g.bias = bias; g.act = act;
    g.Y = y; g.ldy = cout;
    g.Yp = yp; g.ldyp = cout / 32;
    // group g: x planes at xp + g * x_group_stride (uint16 elements; 0 = every group reads the same input), weights [G][k * Cout][Cin planes], bias
    // [G][Cout], outputs group-major [G][M][Cout]
    g.nblk = groups;
    g.g_a = x_group_stride;
    g.g_w = (long long)k * cout * ldw * 64;
    g.g_bias = cout;
    g.g_y = (long long)m * cout;
    g.g_yp = (long long)m * (cout / 32) * 64;
    return launch_gemm(g, (hipStream_t)stream);
}

int fcl_lstm_step_fwd(const fcl_lstm_step_t* args, fcl_stream_t stream) {
    FCL_REQUIRE(args, FCL_ERR_INVALID, "lstm_step_fwd: null argument");
    FCL_REQUIRE(!args->save_gates || (args->save_c_new && args->save_c_old && args->save_h_old), FCL_ERR_INVALID,
                "lstm_step_fwd: save_gates needs save_c_new / save_c_old / save_h_old");
    return launch_lstm_step(*args, (hipStream_t)stream);
}

int fcl_prof_enable(int on) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = on != 0;
    return 0;
}

int fcl_prof_collect(fcl_prof_entry_t* out, int max_entries) {
    FCL_REQUIRE(out && max_entries > 0, FCL_ERR_INVALID, "prof_collect: bad arguments");
    int n = 0;
    for (auto& r : g_prof) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        int i = 0;
        for (; i < n; ++i)
            if (strcmp(out[i].name, r.name) == 0) break;
        if (i == n) {
            if (n == max_entries) continue;
            memset(&out[n], 0, sizeof(out[n]));
            strncpy(out[n].name, r.name, sizeof(out[n].name) - 1);
            ++n;
        }
        out[i].launches += 1;
        out[i].ms += ms;
        out[i].flops += r.flops;
        out[i].rows += r.rows;
        out[i].fill_bytes += r.fill;
    }
    return n;
}


// The loop for a decoder whose cell / prenet block counts are not the shipped (2, 2) (fcl_decoder_weights_t.dlayers / prenet_layers): the same
// arithmetic launch by launch on the fp32 operands -- hoists as in the main form, then per step feat_out(t - 1) as a GEMM with the frame scatter in
// its epilogue, one GEMM per prenet block (bias + ReLU + always-on dropout in the epilogue), one LSTM-step launch per cell.
// /root/reference/nets/modules/decoder_sa.py:119-158 (Prenet), :357-369 (the cell stack), :590-617 (the inference loop), :472-515 (teacher forcing).
static int decoder_loop_generic(const fcl_decoder_weights_t* w, const fcl_decoder_io_t* io, hipStream_t s) {
    const int PL = w->prenet_layers ? w->prenet_layers : 2, DL = w->dlayers ? w->dlayers : 2;
    FCL_REQUIRE(PL >= 1 && PL <= 3 && DL >= 1 && DL <= 3, FCL_ERR_SHAPE, "decoder_loop_fwd: prenet_layers %d / dlayers %d (1 .. 3 are implemented)", PL, DL);
    FCL_REQUIRE(PL < 3 || (w->prenet_w2 && w->prenet_b2), FCL_ERR_INVALID, "decoder_loop_fwd: prenet_layers 3 needs prenet_w2 / prenet_b2");
    FCL_REQUIRE(PL < 2 || (w->prenet_w1 && w->prenet_b1), FCL_ERR_INVALID, "decoder_loop_fwd: prenet_layers >= 2 needs prenet_w1 / prenet_b1");
    FCL_REQUIRE(DL < 2 || (w->w1_ih && w->w1_hh && w->b1), FCL_ERR_INVALID, "decoder_loop_fwd: dlayers >= 2 needs w1_ih / w1_hh / b1");
    FCL_REQUIRE(DL < 3 || (w->w2_ih && w->w2_hh && w->b2), FCL_ERR_INVALID, "decoder_loop_fwd: dlayers 3 needs w2_ih / w2_hh / b2");
    FCL_REQUIRE(!io->live_rows && !io->tail_from, FCL_ERR_INVALID, "decoder_loop_fwd: device row counts / tail_from need the (2, 2) decoder structure");
    FCL_REQUIRE(io->att_c, FCL_ERR_INVALID, "decoder_loop_fwd: this decoder structure runs on the fp32 operands: att_c is required");
    const int N = io->n, C = w->c, P = w->p, U = w->u, O = w->odim;
    const int R = w->reduction_factor > 1 ? w->reduction_factor : 1, OR = O * R;  // a step emits R frames (wf_h / wf_att: [R * odim, .], frame-major rows)
    FCL_REQUIRE(R == 1 || (!io->tap_prenet && !io->tap_lstm0 && !io->tap_lstm1), FCL_ERR_INVALID, "decoder_loop_fwd: step-level taps are frame-indexed: reduction_factor 1 only");
    DecoderWs ws = carve(w, N, io->workspace);
    hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, s, ws.h0[0], (long long)(ws.state_bytes / 4));
    FCL_HIP(hipGetLastError());
    {
        GemmArgs g = {};
        g.term[0] = GemmTerm{io->att_c, w->w0_att, C, C, C, 0};
        g.nterms = 1; g.M = N; g.N = 4 * U; g.bias = w->b0; g.Y = ws.G0; g.ldy = 4 * U;
        int rc = launch_gemm(g, s);
        if (rc) return rc;
        GemmArgs f = {};
        f.term[0] = GemmTerm{io->att_c, w->wf_att, C, C, C, 0};
        f.nterms = 1; f.M = N; f.N = OR; f.Y = ws.F0; f.ldy = OR;
        rc = launch_gemm(f, s);
        if (rc) return rc;
    }
    const float keep_scale = 1.0f / (1.0f - w->prenet_dropout);
    const int drop_mode = (w->prenet_dropout > 0.f) ? io->dropout_mode : FCL_DROP_NONE;
    const float* pw[3] = {w->prenet_w0, w->prenet_w1, w->prenet_w2};
    const float* pb[3] = {w->prenet_b0, w->prenet_b1, w->prenet_b2};
    const float* wih[3] = {nullptr, w->w1_ih, w->w2_ih};
    const float* whh[3] = {w->w0_hh, w->w1_hh, w->w2_hh};
    const float* bl[3] = {nullptr, w->b1, w->b2};
    float* hh[3][2] = {{ws.h0[0], ws.h0[1]}, {ws.h1[0], ws.h1[1]}, {ws.h2[0], ws.h2[1]}};
    float* cc[3] = {ws.c0, ws.c1, ws.c2};
    int cur = 0;
    for (int t = 0; t <= io->lmax; ++t) {
        const int n = t < io->lmax ? io->live_rows_host[t] : 0, n_prev = t > 0 ? io->live_rows_host[t - 1] : 0;
        // teacher forcing: step t reads the target frame that closes step t - 1's group (decoder_sa.py:487-489, 513: ys[:, r - 1 :: r]); the caller's
        // teacher_ys holds one frame per FRAME index: [N, lmax * R, odim]
        const float* teacher_in = (io->teacher_ys && t > 0) ? io->teacher_ys + (size_t)(t * R - 1) * O : nullptr;
        int rc;
        if (t > 0) {  // feat_out(t - 1) on the last cell's new state: R frames = R * odim consecutive floats of the frame-major `before`, starting at
                      // frame frame_off[m] + (t - 1) R; the fed-back frame (the last of them) is activated (decoder_sa.py:614-617), `before` is raw
            GemmArgs f = {};
            f.term[0] = GemmTerm{hh[DL - 1][cur], w->wf_h, U, U, U, 0};
            f.nterms = 1; f.M = n_prev; f.N = OR; f.C0 = ws.F0; f.ldc0 = OR; f.Y = ws.prev; f.ldy = OR;
            f.Y2 = io->before; f.ldy2 = O; f.y2_row_base = io->frame_off; f.y2_row_add = (t - 1) * R;
            rc = launch_gemm(f, s);
            if (rc) return rc;
            if (w->out_act != FCL_ACT_NONE && t < io->lmax && !teacher_in) {
                rc = fcl_act_fwd(ws.prev, nullptr, 1.0f, ws.prev, nullptr, 0, (size_t)n_prev * OR, w->out_act, (fcl_stream_t)s);
                if (rc) return rc;
            }
        }
        if (t == io->lmax) break;
        // prenet: PL x {Linear -> ReLU -> dropout (always on)}; the last block's output lands in pre_b
        const float* x = teacher_in ? teacher_in : ws.prev + (size_t)(R - 1) * O;
        int ldx = teacher_in ? io->lmax * R * O : OR, kx = O;
        for (int l = 0; l < PL; ++l) {
            float* y = ((PL - 1 - l) & 1) ? ws.pre_a : ws.pre_b;
            GemmArgs p0 = {};
            p0.term[0] = GemmTerm{x, pw[l], ldx, kx, kx, 0};
            p0.nterms = 1; p0.M = n; p0.N = P; p0.bias = pb[l]; p0.act = FCL_ACT_RELU; p0.Y = y; p0.ldy = P;
            p0.drop_mode = drop_mode; p0.keep_scale = keep_scale; p0.drop_p = w->prenet_dropout;
            p0.keep = drop_mode == FCL_DROP_MASK ? io->prenet_keep + ((size_t)(t * PL + l) * N) * P : nullptr;
            p0.ldkeep = P; p0.rng_seed = io->seed * 2654435761u + (unsigned)(t * PL + l); p0.seed_dev = io->seed_dev;
            if (l == PL - 1 && io->tap_prenet) { p0.Y2 = io->tap_prenet; p0.ldy2 = P; p0.y2_row_base = io->frame_off; p0.y2_row_add = t; }
            rc = launch_gemm(p0, s);
            if (rc) return rc;
            x = y; ldx = P; kx = P;
        }
        for (int l = 0; l < DL; ++l) {
            LstmStepArgs a = {};
            if (l == 0) {
                a.term[0] = GemmTerm{ws.pre_b, w->w0_pre, P, P, P, 0};
                a.G = ws.G0; a.g_row_mul = 1; a.rank1_w = w->w0_pos; a.dur = io->dur;
            } else {
                a.term[0] = GemmTerm{hh[l - 1][cur ^ 1], wih[l], U, U, U, 0};
                a.bias = bl[l];
            }
            a.term[1] = GemmTerm{hh[l][cur], whh[l], U, U, U, 0};
            a.nterms = 2; a.M = n; a.U = U; a.step = t;
            a.h_in = hh[l][cur]; a.h_out = hh[l][cur ^ 1]; a.c = cc[l]; a.zoneout = w->zoneout_rate;
            float* tap = l == 0 ? io->tap_lstm0 : (l == DL - 1 ? io->tap_lstm1 : nullptr);  // (one cell: tap_lstm0 alone is written)
            if (tap) { a.out2 = tap; a.out2_row_base = io->frame_off; a.out2_row_add = t; a.ld2 = U; }
            rc = launch_lstm_step(a, s);
            if (rc) return rc;
        }
        cur ^= 1;
    }
    return 0;
}

size_t fcl_decoder_loop_workspace_bytes(const fcl_decoder_weights_t* w, int n) {
    if (!w || n <= 0) return 0;
    return carve(w, n, nullptr).bytes + 256;
}

size_t fcl_decoder_stream_bytes(const fcl_decoder_weights_t* w) { return w ? decoder_stream_bytes(w) : 0; }

int fcl_decoder_stream_pack(const fcl_decoder_weights_t* w, void* out, size_t out_bytes, fcl_stream_t stream) {
    FCL_REQUIRE(w && out, FCL_ERR_INVALID, "decoder_stream_pack: null argument");
    const size_t need = decoder_stream_bytes(w);
    FCL_REQUIRE(need > 0, FCL_ERR_SHAPE, "decoder_stream_pack: U = %d / P = %d / odim = %d is not covered by the row-tile kernel (U = P = 256, odim <= 128)", w->u, w->p, w->odim);
    FCL_REQUIRE(out_bytes >= need && aligned16(out), FCL_ERR_WORKSPACE, "decoder_stream_pack: the buffer needs %zu bytes, 16-byte aligned", need);
    FCL_REQUIRE(w->prenet_w0 && w->prenet_w1 && w->w0_pre && w->w0_hh && w->w1_ih && w->w1_hh && w->wf_h, FCL_ERR_INVALID, "decoder_stream_pack: null weight pointer");
    return decoder_stream_pack(w, out, (hipStream_t)stream);
}

int fcl_decoder_loop_fwd(const fcl_decoder_weights_t* w, const fcl_decoder_io_t* io, fcl_stream_t stream) {
    FCL_REQUIRE(w && io, FCL_ERR_INVALID, "decoder_loop_fwd: null argument");
    FCL_REQUIRE(w->c > 0 && w->p > 0 && w->u > 0 && w->odim > 0 && !(w->c & 3) && !(w->p & 3) && !(w->u & 3) && !(w->odim & 3),
                FCL_ERR_SHAPE, "decoder_loop_fwd: C/P/U/odim must be positive multiples of 4 (got %d/%d/%d/%d)", w->c, w->p, w->u, w->odim);
    const bool generic = (w->prenet_layers != 0 && w->prenet_layers != 2) || (w->dlayers != 0 && w->dlayers != 2) || w->reduction_factor > 1;
    FCL_REQUIRE(w->prenet_w0 && w->prenet_b0 && w->w0_att && w->w0_pre && w->w0_pos && w->w0_hh && w->b0 && w->wf_h && w->wf_att &&
                    (generic || (w->prenet_w1 && w->prenet_b1 && w->w1_ih && w->w1_hh && w->b1)),
                FCL_ERR_INVALID, "decoder_loop_fwd: null weight pointer");
    FCL_REQUIRE(io->n >= 0 && io->lmax >= 0, FCL_ERR_SHAPE, "decoder_loop_fwd: bad N=%d Lmax=%d", io->n, io->lmax);
    if (io->n == 0 || io->lmax == 0) return 0;
    FCL_REQUIRE((io->att_c || io->att_c_p) && io->dur && io->live_rows_host && io->frame_off && io->before, FCL_ERR_INVALID, "decoder_loop_fwd: null io pointer");
    FCL_REQUIRE(io->dropout_mode >= FCL_DROP_NONE && io->dropout_mode <= FCL_DROP_RNG, FCL_ERR_INVALID, "decoder_loop_fwd: bad dropout_mode");
    FCL_REQUIRE(io->dropout_mode != FCL_DROP_MASK || io->prenet_keep, FCL_ERR_INVALID, "decoder_loop_fwd: FCL_DROP_MASK needs prenet_keep");
    FCL_REQUIRE(w->prenet_dropout >= 0.f && w->prenet_dropout < 1.f, FCL_ERR_INVALID, "decoder_loop_fwd: bad prenet_dropout");
    FCL_REQUIRE(io->workspace && aligned16(io->workspace) && io->workspace_bytes >= fcl_decoder_loop_workspace_bytes(w, io->n), FCL_ERR_WORKSPACE,
                "decoder_loop_fwd: workspace missing, misaligned or smaller than fcl_decoder_loop_workspace_bytes()");
    FCL_REQUIRE(!io->live_rows || io->status, FCL_ERR_INVALID, "decoder_loop_fwd: device live_rows need a device status word");
    {
        int prev = io->n;
        // host counts are exact (every row is live at step 0: durations > 0) unless the device counts drive the loop: then they are upper bounds
        FCL_REQUIRE(io->live_rows || io->live_rows_host[0] == io->n, FCL_ERR_INVALID, "decoder_loop_fwd: live_rows_host[0] must equal N (all durations > 0)");
        for (int t = 0; t < io->lmax; ++t) {
            FCL_REQUIRE(io->live_rows_host[t] > 0 && io->live_rows_host[t] <= prev, FCL_ERR_INVALID,
                        "decoder_loop_fwd: live_rows_host must be positive and non-increasing (rows sorted by duration descending)");
            prev = io->live_rows_host[t];
        }
    }
    hipStream_t s = (hipStream_t)stream;
    if (generic) return decoder_loop_generic(w, io, s);
    const int N = io->n, C = w->c, P = w->p, U = w->u, O = w->odim;
    DecoderWs ws = carve(w, N, io->workspace);

    // zero recurrent state (h0[2], c0, h1[2], c1 are contiguous) and prev_out
    static const bool planes_on = tunable("PRECISION", 1) != 0 && tunable("PLANES", 1) != 0;
    const bool planes = decoder_planes(w) && io->att_c_p != nullptr && planes_on;
    FCL_REQUIRE(planes || io->att_c, FCL_ERR_INVALID, "decoder_loop_fwd: att_c is required when the P32 path is off");
    FCL_REQUIRE(!io->before_p || planes, FCL_ERR_INVALID, "decoder_loop_fwd: before_p needs the P32 weight planes and att_c_p");
    {
        hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, s, ws.h0[0], (long long)(ws.state_bytes / 4));
        FCL_HIP(hipGetLastError());
    }
    const int ldc = C / 32, ldp = P / 32, ldu = U / 32;  // plane strides in 128-byte lines (planes mode: whole lines)
    auto small_step = [&](int m) { return lstm_step_is_small(m, U); };  // launch_lstm_step's own choice: small steps read the fp32 operands
    // loop-invariant hoists: att_c is constant across steps, so its share of the LSTM-0
    // gate pre-activations and of feat_out is one GEMM each instead of Lmax of them.
    {
        GemmArgs g = {};
        g.term[0] = GemmTerm{io->att_c, w->w0_att, C, C, C, 0};
        if (planes) { g.term[0].Ap = io->att_c_p; g.term[0].Wp = w->w0_att_p; g.term[0].lda_p = g.term[0].ldw_p = ldc; }
        g.nterms = 1; g.M = N; g.N = 4 * U; g.bias = w->b0; g.Y = ws.G0; g.ldy = 4 * U;
        int rc = launch_gemm(g, s);
        if (rc) return rc;
        GemmArgs f = {};
        f.term[0] = GemmTerm{io->att_c, w->wf_att, C, C, C, 0};
        if (planes) { f.term[0].Ap = io->att_c_p; f.term[0].Wp = w->wf_att_p; f.term[0].lda_p = f.term[0].ldw_p = ldc; }
        f.nterms = 1; f.M = N; f.N = O; f.Y = ws.F0; f.ldy = O;
        rc = launch_gemm(f, s);
        if (rc) return rc;
    }
    const float keep_scale = 1.0f / (1.0f - w->prenet_dropout);
    const int drop_mode = (w->prenet_dropout > 0.f) ? io->dropout_mode : FCL_DROP_NONE;
    {
        // round 4, opt-in (FCL_DEC_TILE=1): the whole loop as ONE launch of persistent 32-row workgroups (decoder_tile.hip) -- free-running synthesis
        // of the covered shape.  OFF by default: it halves the decoder's CU-time but a step of a tile is the 4.7 MB weight stream through ONE CU
        // (58 - 70 us) where the per-step launches put all 256 CUs on the step (45 us): with four passes in flight the pass is bound by that
        // chain, and the line drops from 45 to 38 M frames/s (HISTORY "round 4")
        static const int tile_on = tunable("DEC_TILE", 0), tile_min = tunable("DEC_TILE_MIN_ROWS", 1024);
        if (tile_on && planes && w->stream && decoder_tile_shape_ok(w) && N >= tile_min && io->lmax <= 64 && !io->teacher_ys && !io->tap_prenet && !io->tap_lstm0 &&
            !io->tap_lstm1 && drop_mode != FCL_DROP_MASK && gemm_mode() != FCL_GEMM_BF16)
            return launch_decoder_tile(w, io, ws.G0, ws.F0, ws.c0, ws.c1, drop_mode, 0, nullptr, nullptr, s);
    }
    // fcl_decoder_io_t.tail_from: from that step on the rows still live continue in ONE launch of the persistent row-tile kernel, from the loop's own
    // fp32 states -- in a capacity graph the steps beyond the longest duration seen so far then cost one launch in all instead of three each
    const bool tile_ok = planes && w->stream && decoder_tile_shape_ok(w) && io->lmax <= 64 && !io->teacher_ys && !io->tap_prenet && !io->tap_lstm0 && !io->tap_lstm1 &&
                         drop_mode != FCL_DROP_MASK && gemm_mode() != FCL_GEMM_BF16;
    const int tail_from = (tile_ok && io->tail_from > 0 && io->tail_from < io->lmax) ? io->tail_from : 0;
    static const int fused = tunable("FUSED_PRENET", 1);
    FCL_REQUIRE(fused || !io->live_rows, FCL_ERR_INVALID, "decoder_loop_fwd: device live_rows need the fused feat/prenet kernel (FCL_FUSED_PRENET=1)");
    FCL_REQUIRE(fused || !tail_from, FCL_ERR_INVALID, "decoder_loop_fwd: tail_from needs the fused feat/prenet kernel (FCL_FUSED_PRENET=1)");
    int cur = 0;
    for (int t = 0; t <= io->lmax; ++t) {
        const bool hand_over = tail_from > 0 && t == tail_from;      // this step and all later ones: the tile kernel (after feat_out(t - 1) of every row live at t - 1)
        const int n = (t < io->lmax && !hand_over) ? io->live_rows_host[t] : 0;        // rows live at step t
        const int n_prev = t > 0 ? io->live_rows_host[t - 1] : 0;      // rows whose feat_out(t-1) is due
        const uint8_t* keep0 = drop_mode == FCL_DROP_MASK && t < io->lmax ? io->prenet_keep + ((size_t)(t * 2 + 0) * N) * P : nullptr;
        const uint8_t* keep1 = drop_mode == FCL_DROP_MASK && t < io->lmax ? io->prenet_keep + ((size_t)(t * 2 + 1) * N) * P : nullptr;
        const unsigned seed0 = io->seed * 2654435761u + (unsigned)(t * 2 + 0), seed1 = seed0 + 1;
        const float* teacher_in = (io->teacher_ys && t > 0) ? io->teacher_ys + (size_t)(t - 1) * O : nullptr;
        int rc;
        if (fused) {
            // H8 feat_out(t-1) [+ H10 scatter] -> H6 prenet(t), one launch
            FeatPrenetArgs fp = {};
            fp.M_feat = n_prev; fp.M_pre = n; fp.U = U; fp.O = O; fp.P = P;
            fp.h1 = t > 0 ? ws.h1[cur] : nullptr; fp.wf_h = w->wf_h; fp.F0 = ws.F0;
            fp.before = io->before; fp.frame_off = io->frame_off; fp.t_prev = t - 1; fp.t_cur = t;
            fp.teacher_in = teacher_in; fp.teacher_ld = io->lmax * O;
            if (t < io->lmax && !hand_over) { fp.w0 = w->prenet_w0; fp.b0 = w->prenet_b0; fp.w1 = w->prenet_w1; fp.b1 = w->prenet_b1; }
            fp.wf_hi = w->wf_h_hi; fp.wf_lo = w->wf_h_lo; fp.w0_hi = w->prenet_w0_hi; fp.w0_lo = w->prenet_w0_lo;
            fp.w1_hi = w->prenet_w1_hi; fp.w1_lo = w->prenet_w1_lo;
            fp.wf_ff = w->wf_h_ff; fp.w0_ff = w->prenet_w0_ff; fp.w1_ff = w->prenet_w1_ff;
            fp.drop_mode = drop_mode; fp.keep0 = keep0; fp.keep1 = keep1; fp.keep_scale = keep_scale; fp.drop_p = w->prenet_dropout;
            fp.seed0 = seed0; fp.seed1 = seed1; fp.seed_dev = io->seed_dev; fp.pre_out = ws.pre_b; fp.tap_prenet = io->tap_prenet;
            fp.live = io->live_rows; fp.status = io->status; fp.out_act = w->out_act;
            if (planes) {
                fp.before_p = io->before_p;
                if (!small_step(n)) { fp.pre_out_p = ws.pre_p; fp.pre_out = nullptr; }  // the big-tile LSTM step reads planes only
            }
            rc = launch_feat_prenet(fp, s);
            if (rc) return rc;
        } else {
            if (t > 0) {  // H8 feat_out(t-1) (+ H10 scatter): out = h1 . Wf_h^T + F0
                GemmArgs f = {};
                f.term[0] = GemmTerm{ws.h1[cur], w->wf_h, U, U, U, 0};
                f.nterms = 1; f.M = n_prev; f.N = O; f.C0 = ws.F0; f.ldc0 = O; f.Y = ws.prev; f.ldy = O;
                f.Y2 = io->before; f.ldy2 = O; f.y2_row_base = io->frame_off; f.y2_row_add = t - 1;
                rc = launch_gemm(f, s);
                if (rc) return rc;
            }
            if (t < io->lmax) {  // H6 prenet: 2 x {Linear -> ReLU -> dropout (always on)}
                GemmArgs p0 = {};
                if (teacher_in) p0.term[0] = GemmTerm{teacher_in, w->prenet_w0, io->lmax * O, O, O, 0};
                else p0.term[0] = GemmTerm{ws.prev, w->prenet_w0, O, O, O, 0};
                p0.nterms = 1; p0.M = n; p0.N = P; p0.bias = w->prenet_b0; p0.act = FCL_ACT_RELU; p0.Y = ws.pre_a; p0.ldy = P;
                p0.drop_mode = drop_mode; p0.keep_scale = keep_scale; p0.drop_p = w->prenet_dropout;
                p0.keep = keep0; p0.ldkeep = P; p0.rng_seed = seed0; p0.seed_dev = io->seed_dev;
                rc = launch_gemm(p0, s);
                if (rc) return rc;
                GemmArgs p1 = p0;
                p1.term[0] = GemmTerm{ws.pre_a, w->prenet_w1, P, P, P, 0};
                p1.bias = w->prenet_b1; p1.Y = ws.pre_b; p1.keep = keep1; p1.rng_seed = seed1;
                if (io->tap_prenet) { p1.Y2 = io->tap_prenet; p1.ldy2 = P; p1.y2_row_base = io->frame_off; p1.y2_row_add = t; }
                rc = launch_gemm(p1, s);
                if (rc) return rc;
            }
        }
        if (hand_over) return launch_decoder_tile(w, io, ws.G0, ws.F0, ws.c0, ws.c1, drop_mode, t, ws.h0[cur], ws.h1[cur], s);
        if (t == io->lmax) break;
        // H7 layer 0: gates = G0 + prenet . W_pre^T + pos * w_pos + h0 . W_hh^T ; cell ; zoneout
        LstmStepArgs l0 = {};
        const bool big = planes && !small_step(n);  // big steps: pre-split operands through the LDS-DMA kernels; their outputs feed the next step's
        const bool next_big = planes && t + 1 < io->lmax && !small_step(io->live_rows_host[t + 1]);
        l0.term[0] = GemmTerm{ws.pre_b, w->w0_pre, P, P, P, 0, w->w0_pre_hi, w->w0_pre_lo};
        l0.term[1] = GemmTerm{ws.h0[cur], w->w0_hh, U, U, U, 0, w->w0_hh_hi, w->w0_hh_lo};
        if (big) {
            l0.term[0].Ap = ws.pre_p; l0.term[0].Wp = w->w0_pre_p; l0.term[0].lda_p = l0.term[0].ldw_p = ldp;
            l0.term[1].Ap = ws.h0_p[cur]; l0.term[1].Wp = w->w0_hh_p; l0.term[1].lda_p = l0.term[1].ldw_p = ldu;
            l0.h_out_p = ws.h0_p[cur ^ 1]; l0.ld_hp = ldu;  // read by layer 1 now and by layer 0 of the next step
        }
        l0.term[0].Wff = w->w0_pre_ff; l0.term[1].Wff = w->w0_hh_ff;
        l0.nterms = 2; l0.M = n; l0.U = U; l0.G = ws.G0; l0.g_row_mul = 1; l0.g_row_add = 0;
        l0.m_dev = io->live_rows ? io->live_rows + t : nullptr;
        l0.rank1_w = w->w0_pos; l0.dur = io->dur; l0.step = t;
        l0.h_in = ws.h0[cur]; l0.h_out = ws.h0[cur ^ 1]; l0.c = ws.c0; l0.zoneout = w->zoneout_rate;
        if (io->tap_lstm0) { l0.out2 = io->tap_lstm0; l0.out2_row_base = io->frame_off; l0.out2_row_add = t; l0.ld2 = U; }
        rc = launch_lstm_step(l0, s);
        if (rc) return rc;
        // H7 layer 1
        LstmStepArgs l1 = {};
        l1.term[0] = GemmTerm{ws.h0[cur ^ 1], w->w1_ih, U, U, U, 0, w->w1_ih_hi, w->w1_ih_lo};
        l1.term[1] = GemmTerm{ws.h1[cur], w->w1_hh, U, U, U, 0, w->w1_hh_hi, w->w1_hh_lo};
        if (big) {
            l1.term[0].Ap = ws.h0_p[cur ^ 1]; l1.term[0].Wp = w->w1_ih_p; l1.term[0].lda_p = l1.term[0].ldw_p = ldu;
            l1.term[1].Ap = ws.h1_p[cur]; l1.term[1].Wp = w->w1_hh_p; l1.term[1].lda_p = l1.term[1].ldw_p = ldu;
            if (next_big) { l1.h_out_p = ws.h1_p[cur ^ 1]; l1.ld_hp = ldu; }
        }
        l1.term[0].Wff = w->w1_ih_ff; l1.term[1].Wff = w->w1_hh_ff;
        l1.nterms = 2; l1.M = n; l1.U = U; l1.bias = w->b1; l1.step = t;
        l1.m_dev = l0.m_dev;
        l1.h_in = ws.h1[cur]; l1.h_out = ws.h1[cur ^ 1]; l1.c = ws.c1; l1.zoneout = w->zoneout_rate;
        if (io->tap_lstm1) { l1.out2 = io->tap_lstm1; l1.out2_row_base = io->frame_off; l1.out2_row_add = t; l1.ld2 = U; }
        rc = launch_lstm_step(l1, s);
        if (rc) return rc;
        cur ^= 1;
    }
    return 0;
}

}  // extern "C"
