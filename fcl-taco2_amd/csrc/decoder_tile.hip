// decoder_tile.hip — the WHOLE free-running decoder loop (H6 prenet, H7 two LSTMCell + zoneout steps, H8 feat_out, H10 frame scatter:
// Decoder.inference, decoder_sa_kd.py:742-790) for a tile of 32 phoneme rows in ONE persistent workgroup (round 4).
//
// Why: the rows of the per-phoneme-parallel decoder are independent of one another for ALL their steps — only the weights are shared.  The per-step
// launches (feat/prenet, LSTM 0, LSTM 1: ~70 launches per pass) are each a 2 400 x 1 024 x 512 problem that is 3 us of MFMA work inside a launch
// floor, a first-operand latency and an epilogue, and their operands (h0, h1, prenet output) make a round trip through HBM / L2 between every two
// launches.  Here a workgroup keeps its 32 rows' recurrent state in LDS (3 x 32 KB of pre-split operand lines: x = prev_out -> prenet 0 -> prenet 1,
// h0, h1) and in registers (c0, c1 and the fp32 h0 / h1: the lanes that own a unit's four gates), and the decoder's weights — the only thing that is
// NOT row-local — stream past it once per step: 4.6 MB from L2 through a 4 x 16 KB LDS-DMA ring, in consumption order, pre-swizzled at plan time
// (fcl_decoder_stream_pack), so the four loader waves issue nothing but linear 1 KB pieces.  tools/stream_probe.hip measured the skeleton first:
// 256 workgroups pulling the SAME 4.6 MB cyclically sustain 35 B/clk per CU (21.8 TB/s over the chip, no L2 thrash up to 18 MB: the workgroups
// stay in loose lockstep, one fetches a line and 31 others on its XCD hit it) = 54 us per step and tile beside the MFMAs.
//
// Geometry: 8 consumer waves + 4 loader waves.  One ring SLOT = one 32-k chunk of one 16-column weight tile per consumer wave (8 x 2 KB: 16 lines
// of 32 hi | 32 lo).  Weight tiles are the MFMA's A operand (transposed accumulators: a lane holds four consecutive columns of one row), activations
// the B operand straight from the LDS state lines.  Stream order per step: feat_out (8 slots; waves 0..ceil(O/16)), prenet 0 (2 passes x ceil(O/32)),
// prenet 1 (2 x 8), LSTM 0 (2 passes x 16 chunks x 4 gates: wave w owns units (pass * 8 + w) * 16 .. + 15, all four gates), LSTM 1 (same).
// One s_barrier per slot (slot i landed / slot i - 1's buffer free) joins all twelve waves; the four in-place state updates of a step (x twice, h0, h1)
// hold their results in registers across one extra barrier.  Products, their order and every epilogue are those of the per-step kernels
// (plstm_kernel / feat_prenet_split_kernel): same chunk order, same three bf16 MFMAs per chunk, same cell math.
#include "lstm_epilogue.h"

namespace fcl {

typedef unsigned short u16;
typedef unsigned char u8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

namespace {

constexpr int DT_ROWS = 32, DT_U = 256, DT_P = 256, DT_SLOT = 16384, DT_NS = 4, DT_BUF = DT_ROWS * DT_U * 4;  // 32 KB per state buffer
constexpr int DT_LDS = 3 * DT_BUF + DT_NS * DT_SLOT, DT_MAX_STEPS = 64, DT_NL = 4, DT_PPL = 16 / DT_NL;

struct DecTileArgs {
    const u8* stream;
    int slots_per_step, oc;  // oc = ceil(O / 32): chunks of the prenet's input
    int n, lmax, O;
    const float *G0, *F0, *w_pos, *b1, *pb0, *pb1;
    const int *dur, *frame_off, *live;
    unsigned int* status;
    float* before;
    u16* before_p;
    float *c0, *c1;  // [n, U] fp32 cell states (the loop workspace: zeroed before the launch, or the per-step loop's own states at a hand-over)
    // hand-over from the per-step loop (fcl_decoder_io_t.tail_from): the rows still live at step t_start continue here from the loop's fp32 states
    int t_start, n_start;          // first step of this launch; the host's bound on the rows live at it (t_start = 0: n_start = n)
    const float *h0_init, *h1_init;  // [n, U] fp32 hidden states entering step t_start (NULL at t_start = 0: zeros)
    float zoneout, keep_scale, drop_p;
    int out_act;
    unsigned int seed;
    const unsigned int* seed_dev;
    long long* ts;  // developer aid (FCL_DEC_TILE_TS): 8 cycle sums per workgroup (wave 0): F loop / F epilogue / prenet loops / prenet epilogues / LSTM loops / LSTM epilogues
    int bound[DT_MAX_STEPS];  // the host's per-step row bounds (fcl_decoder_io_t.live_rows_host)
};

__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void slot_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// stream slot index (inside a step) at which a phase starts
struct DtPhases {
    int p0, p1, l0, l1, total;
    __host__ __device__ explicit DtPhases(int oc) : p0(8), p1(8 + 2 * oc), l0(8 + 2 * oc + 16), l1(8 + 2 * oc + 16 + 128), total(8 + 2 * oc + 16 + 256) {}
};

template <int DROP>
__global__ __launch_bounds__(64 * (8 + DT_NL)) void decoder_tile_kernel(const DecTileArgs a) {
    extern __shared__ __attribute__((aligned(1024))) u8 smem[];
    u8* xb = smem;
    u8* h0b = smem + DT_BUF;
    u8* h1b = smem + 2 * DT_BUF;
    u8* ring = smem + 3 * DT_BUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * DT_ROWS;
    // Rows are sorted by duration, descending: row r is live at step t iff dur[r] > t, and this tile runs as many steps as its first row.  With
    // device-built maps a violated capacity zeroes live_rows (fcl_row_maps_build): nothing may run then.  The host's per-step row bounds only size
    // the per-step launches of the other path; here a step has no grid -- they are checked all the same, so that both paths report alike.
    if (a.live && blockIdx.x == 0 && tid == 0) {
#pragma unroll
        for (int t = 0; t < DT_MAX_STEPS; ++t)
            if (t < a.lmax && a.live[t] > a.bound[t]) atomicOr(a.status, (unsigned int)FCL_STATUS_ROWS_CAP);
    }
    // (values loaded from global memory are per-lane to the compiler: readfirstlane makes every loop bound below a scalar)
    const int t0 = a.t_start;
    const int n_live = __builtin_amdgcn_readfirstlane(a.live ? min(a.live[t0], a.n_start) : a.n_start);
    const int T = __builtin_amdgcn_readfirstlane(r0 < n_live ? min(a.dur[r0], a.lmax) : 0);
    if (T <= t0) return;
    const DtPhases ph(a.oc);
    // steps t0 .. T - 1 run the step body (everything but feat_out), feat_out runs at every t in (max(t0, 1) - 1 .. T] -- at a hand-over (t0 > 0) the
    // first one recomputes the frame of step t0 - 1 as the prenet's input (the per-step loop has stored it)
    const int sps = ph.total, total = (T - t0) * (sps - 8) + (T - max(t0, 1) + 1) * 8;

    if (wave >= 8) {  // ---- loader waves: linear 1 KB pieces in stream order, one counted wait and one barrier per slot --------------------------
        const int lw = wave - 8;
        const u8* src = a.stream + lw * DT_PPL * 1024 + lane * 16;
        int s = t0 > 0 ? 0 : ph.p0, issued = 0;
        auto issue = [&]() {
            u8* dst = ring + (issued & (DT_NS - 1)) * DT_SLOT + lw * DT_PPL * 1024;
            const u8* g = src + (size_t)s * DT_SLOT;
#pragma unroll
            for (int j = 0; j < DT_PPL; ++j) glds16(g + j * 1024, dst + j * 1024);
            s = s + 1 == sps ? 0 : s + 1;
            ++issued;
        };
#pragma unroll
        for (int p = 0; p < DT_NS - 1; ++p) issue();
        int cs = t0 > 0 ? 0 : ph.p0;
        for (int i = 0; i < total; ++i) {
            if (total - 1 - i >= DT_NS - 2) wait_vm<(DT_NS - 2) * DT_PPL>();
            else wait_vm<0>();
            asm volatile("s_barrier" ::: "memory");
            if (issued < total) issue();
            if (cs == 0 || cs == ph.p1 || cs == ph.l0 || cs == ph.l1) asm volatile("s_barrier" ::: "memory");  // the consumers' in-place state update
            cs = cs + 1 == sps ? 0 : cs + 1;
        }
        return;
    }

    // ---- consumer waves ---------------------------------------------------------------------------------------------------------------------------
    const int r16 = lane & 15, kq = lane >> 4, sw = r16 >> 1;
    const int arow = lane & 15, cq = lane >> 4;  // transposed accumulators: activation row arow of the row tile, columns cq * 4 .. + 3 of the weight tile
    const int f_hi = r16 * 128 + ((kq ^ sw) << 4), f_lo = r16 * 128 + (((4 + kq) ^ sw) << 4);  // fragment offsets inside a 16-line tile
    const u8* wslot = ring + wave * 2048;
    int rc = 0;  // ring slot holding the stream slot consumed next
    // LDS state line of (row, column n .. n + 3): byte offset of the hi half (lo: + the xor of piece 4)
    auto st_off = [&](int row, int n) { return (n >> 5) * (DT_ROWS * 128) + row * 128 + (n & 7) * 2; };
    auto st_store = [&](u8* buf, int row, int n, const f32x4 v) {
        uint2 h, l;
        split4(v, h, l);
        const int o = st_off(row, n), s2 = (row >> 1) & 7, pc = (n & 31) >> 3;
        *reinterpret_cast<uint2*>(buf + o + ((pc ^ s2) << 4)) = h;
        *reinterpret_cast<uint2*>(buf + o + (((4 + pc) ^ s2) << 4)) = l;
    };
    // prev_out = 0; h = 0 before step 0, or (hand-over) the hidden states entering step t0 split into the state lines -- every line of h0 / h1 is then
    // written by exactly one thread here and nobody zeroes it (no barrier orders two threads before the first slot's)
    for (int i = tid; i < (a.h0_init ? 1 : 3) * DT_BUF / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    if (a.h0_init) {
        for (int i = tid; i < DT_ROWS * (DT_U / 4); i += 512) {
            const int row = i / (DT_U / 4), col = (i % (DT_U / 4)) * 4;
            const size_t g = (size_t)min(r0 + row, a.n - 1) * DT_U + col;
            st_store(h0b, row, col, *reinterpret_cast<const f32x4*>(a.h0_init + g));
            st_store(h1b, row, col, *reinterpret_cast<const f32x4*>(a.h1_init + g));
        }
    }
    int mrow[2], mc[2], durv[2], fo[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        mrow[tm] = r0 + tm * 16 + arow;
        mc[tm] = min(mrow[tm], a.n - 1);
        durv[tm] = mrow[tm] < n_live ? a.dur[mc[tm]] : 0;  // (rows of the tile that are not live at t0 have nothing left to do: they never store)
        fo[tm] = a.frame_off[mc[tm]];
    }
    const unsigned int thr16 = (unsigned int)(a.drop_p * 65536.0f);
    const unsigned int sbump = (DROP == 2 && a.seed_dev) ? *a.seed_dev * 0x9E3779B9u : 0u;
    const int O = a.O, oc = a.oc, ldbp = oc;

    // one slot: the weight fragment of this wave's tile x the two activation fragments
    auto mma3 = [&](const s16x8 wh, const s16x8 wl, const s16x8 (&ah)[2], const s16x8 (&al)[2], f32x4 (&acc)[2]) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[tm], acc[tm], 0, 0, 0);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[tm], acc[tm], 0, 0, 0);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[tm], acc[tm], 0, 0, 0);
    };
    auto read_a = [&](const u8* buf, int c, s16x8 (&ah)[2], s16x8 (&al)[2]) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            ah[tm] = *reinterpret_cast<const s16x8*>(buf + c * (DT_ROWS * 128) + tm * 2048 + f_hi);
            al[tm] = *reinterpret_cast<const s16x8*>(buf + c * (DT_ROWS * 128) + tm * 2048 + f_lo);
        }
    };
    auto read_w = [&](s16x8& wh, s16x8& wl) {
        const u8* sb = wslot + rc * DT_SLOT;
        wh = *reinterpret_cast<const s16x8*>(sb + f_hi);
        wl = *reinterpret_cast<const s16x8*>(sb + f_lo);
        rc = (rc + 1) & (DT_NS - 1);
    };
    // plain phase: chunks of A from `buf`; `hook` runs between the first slot's barrier and an extra barrier (in-place update of a state buffer)
    auto gemm_phase = [&](const u8* buf, int nch, f32x4 (&acc)[2], auto&& hook, bool hooked) {
        acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nch; ++c) {
            slot_barrier();
            if (hooked && c == 0) {
                hook();
                slot_barrier();
            }
            s16x8 ah[2], al[2], wh, wl;
            read_a(buf, c, ah, al);
            read_w(wh, wl);
            mma3(wh, wl, ah, al, acc);
        }
    };
    // chunks [cb, ce) of an LSTM pass: 16 chunks (8 of bufA, 8 of bufB) x 4 gates
    auto lstm_chunks = [&](const u8* bufA, const u8* bufB, f32x4 (&acc)[4][2], int cb, int ce, auto&& hook, bool hooked) {
        for (int c = cb; c < ce; ++c) {
            s16x8 ah[2], al[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                slot_barrier();
                if (g == 0) {
                    if (hooked && c == 0) {
                        hook();
                        slot_barrier();
                    }
                    read_a(c < 8 ? bufA : bufB, c & 7, ah, al);
                }
                    s16x8 wh, wl;
                read_w(wh, wl);
                mma3(wh, wl, ah, al, acc[g]);
            }
        }
    };
    // the old hidden state of (row, columns n .. n + 3) as the state lines hold it (hi + lo: h to 2^-17 relative; only the zoneout blend reads it)
    auto st_load = [&](const u8* buf, int row, int n) {
        const int o = st_off(row, n), s2 = (row >> 1) & 7, pc = (n & 31) >> 3;
        const uint2 h = *reinterpret_cast<const uint2*>(buf + o + ((pc ^ s2) << 4)), l = *reinterpret_cast<const uint2*>(buf + o + (((4 + pc) ^ s2) << 4));
        f32x4 v;
        v[0] = __builtin_bit_cast(float, h.x << 16) + __builtin_bit_cast(float, l.x << 16);
        v[1] = __builtin_bit_cast(float, h.x & 0xFFFF0000u) + __builtin_bit_cast(float, l.x & 0xFFFF0000u);
        v[2] = __builtin_bit_cast(float, h.y << 16) + __builtin_bit_cast(float, l.y << 16);
        v[3] = __builtin_bit_cast(float, h.y & 0xFFFF0000u) + __builtin_bit_cast(float, l.y & 0xFFFF0000u);
        return v;
    };
    // LSTMCell + eval-form zoneout of four consecutive units of one row (lstm_epilogue.h cell_math): returns the new h, updates c
    auto cell4 = [&](const f32x4 (&acc)[4][2], int tm, const f32x4 (&add)[4], const f32x4 h_old, f32x4& c) {
        f32x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ig = sigmoid_f(acc[0][tm][e] + add[0][e]), fg = sigmoid_f(acc[1][tm][e] + add[1][e]);
            const float gg = tanh_f(acc[2][tm][e] + add[2][e]), og = sigmoid_f(acc[3][tm][e] + add[3][e]);
            const float c_new = fg * c[e] + ig * gg;
            const float h_new = og * tanh_f(c_new);
            h[e] = a.zoneout * h_old[e] + (1.0f - a.zoneout) * h_new;
            c[e] = a.zoneout * c[e] + (1.0f - a.zoneout) * c_new;
        }
        return h;
    };
    auto nohook = [] {};
    long long tsum[6] = {0, 0, 0, 0, 0, 0}, tlast = a.ts ? (long long)__builtin_readcyclecounter() : 0;
    auto tick = [&](int cat) {
        if (a.ts) {
            const long long now = (long long)__builtin_readcyclecounter();
            tsum[cat] += now - tlast;
            tlast = now;
        }
    };

    uint2 pah[2][2], pal[2][2];  // held prenet outputs (split) of the two passes
    uint2 hnh[2][2], hnl[2][2];  // held new hidden state (split) of the layer that ran last, until its buffer may be overwritten
    auto publish = [&](u8* buf, const uint2 (&vh)[2][2], const uint2 (&vl)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int row = tm * 16 + arow, n = (j * 8 + wave) * 16 + cq * 4;
                const int o = st_off(row, n), s2 = (row >> 1) & 7, pc = (n & 31) >> 3;
                *reinterpret_cast<uint2*>(buf + o + ((pc ^ s2) << 4)) = vh[j][tm];
                *reinterpret_cast<uint2*>(buf + o + (((4 + pc) ^ s2) << 4)) = vl[j][tm];
            }
    };
    auto publish_x = [&] { publish(xb, pah, pal); };
    for (int t = t0; t <= T; ++t) {
        if (t > 0) {  // ---- H8 feat_out of step t - 1 (+ H10 scatter); its first slot publishes LSTM 1's new h1 -------------------------------------
            const int fnc = wave * 16 + cq * 4;
            const bool fw = wave * 16 < oc * 32;
            f32x4 f0v[2];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) f0v[tm] = (fw && fnc < O) ? *reinterpret_cast<const f32x4*>(a.F0 + (size_t)mc[tm] * O + fnc) : (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 acc[2];
            gemm_phase(h1b, 8, acc, [&] { if (t > t0) publish(h1b, hnh, hnl); }, true);  // (first step of a hand-over: h1 is the loaded state)
            tick(0);
            if (fw) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const bool live = durv[tm] > t - 1;
                    const long long fr = (long long)fo[tm] + (t - 1);
                    const bool store = live && t > t0;  // (t == t0 > 0: the per-step loop's feat_out launch has stored this frame)
                    if (live && fnc < O) {
                        v = acc[tm] + f0v[tm];
                        if (store) *reinterpret_cast<f32x4*>(a.before + (size_t)fr * O + fnc) = v;
                    }
                    if (store && a.before_p) {  // (columns O .. 32 oc - 1: the zero padding of the frame's last line)
                        uint2 h, l;
                        split4(v, h, l);
                        u16* line = a.before_p + ((size_t)fr * ldbp + (fnc >> 5)) * 64 + (fnc & 31);
                        *reinterpret_cast<uint2*>(line) = h;
                        *reinterpret_cast<uint2*>(line + 32) = l;
                    }
                    if (live && fnc < O && a.out_act) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], a.out_act);
                    }
                    st_store(xb, tm * 16 + arow, fnc, v);  // the prenet's input (x is free: prenet 1's output was consumed by LSTM 0)
                }
            }
        }
        tick(1);
        if (t == T) break;
        const unsigned int sd = a.seed * 2654435761u + (unsigned int)(t * 2);
        const unsigned int seed0 = DROP == 2 ? hash_u32(sd + sbump) : 0u, seed1 = DROP == 2 ? hash_u32(sd + 1u + sbump) : 0u;
        // ---- H6 prenet layer 0: x (prev_out, oc chunks) -> held; layer 1: x (layer 0's output, published at its first slot) -> held -------------------
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nc = (j * 8 + wave) * 16 + cq * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>((layer ? a.pb1 : a.pb0) + nc);
                f32x4 acc[2];
                if (layer == 0) gemm_phase(xb, oc, acc, nohook, false);
                else gemm_phase(xb, 8, acc, publish_x, j == 0);  // (layer 0's held output is dead once published: layer 1's goes into the same registers)
                tick(2);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    f32x4 v = acc[tm] + b;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    v = drop4<DROP>(v, 0u, (unsigned int)mrow[tm] * (unsigned int)DT_P + (unsigned int)nc, layer ? seed1 : seed0, thr16, a.keep_scale);
                    split4(v, pah[j][tm], pal[j][tm]);
                }
                tick(3);
            }
        }
        // ---- H7: layer 0  gates = G0 + pos * w_pos + [x, h0] W^T,  layer 1  gates = b1 + [h0', h1] W^T;  cell;  zoneout ---------------------------
        // The epilogue's operands (G0 / w_pos or the bias, the old cell state) are requested before the pass's last four chunks: their latency hides
        // under 16 slots.  c lives in the loop's workspace (fp32, exact); the old h is read back from the state lines.
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            u8* hb = layer ? h1b : h0b;
            float* cg = layer ? a.c1 : a.c0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 acc[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g][0] = acc[g][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int u4 = (j * 8 + wave) * 16 + cq * 4;
                if (layer == 0) {
                    lstm_chunks(xb, h0b, acc, 0, 8, publish_x, j == 0);
                    f32x4 gv[2][4];  // the hoisted att_c share of the gates: requested here, added four chunks (16 slots) later
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            gv[tm][g] = *reinterpret_cast<const f32x4*>(a.G0 + (size_t)mc[tm] * (4 * DT_U) + g * DT_U + u4);
                    lstm_chunks(xb, h0b, acc, 8, 12, nohook, false);
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int g = 0; g < 4; ++g) acc[g][tm] += gv[tm][g];
                } else {
                    lstm_chunks(h0b, h1b, acc, 0, 12, [&] { publish(h0b, hnh, hnl); }, j == 0);
                }
                f32x4 wp[4], cv[2];  // w_pos / bias of this lane's units and the old cell state: requested before the last four chunks
#pragma unroll
                for (int g = 0; g < 4; ++g) wp[g] = *reinterpret_cast<const f32x4*>((layer ? a.b1 : a.w_pos) + g * DT_U + u4);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) cv[tm] = *reinterpret_cast<const f32x4*>(cg + (size_t)mc[tm] * DT_U + u4);
                if (layer == 0) lstm_chunks(xb, h0b, acc, 12, 16, nohook, false);
                else lstm_chunks(h0b, h1b, acc, 12, 16, nohook, false);
                tick(4);
                uint2 nh[2], nl[2];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const float pos = layer ? 1.0f : (float)t / (float)max(durv[tm], 1);
                    f32x4 add[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) add[g] = pos * wp[g];  // layer 0: pos * w_pos (G0 is already in the accumulators); layer 1: the bias
                    const f32x4 h_old = st_load(hb, tm * 16 + arow, u4);
                    const f32x4 h = cell4(acc, tm, add, h_old, cv[tm]);
                    if (mrow[tm] < a.n) *reinterpret_cast<f32x4*>(cg + (size_t)mrow[tm] * DT_U + u4) = cv[tm];
                    split4(h, nh[tm], nl[tm]);
                }
                // layer 0's held h0' is published at layer 1's first slot, layer 1's h1' at feat_out's: by then the previous holder is dead
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) { hnh[j][tm] = nh[tm]; hnl[j][tm] = nl[tm]; }
                tick(5);
            }
        }
    }
    if (a.ts && tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.ts[blockIdx.x * 8 + i] = tsum[i];
        a.ts[blockIdx.x * 8 + 6] = T;
    }
}

// ---- the weight stream: one thread per 16-byte unit ----------------------------------------------------------------------------------------------
struct DecStreamSrc {
    const float *wf_h, *p_w0, *p_w1, *w0_pre, *w0_hh, *w1_ih, *w1_hh;
    int O, oc;
};

__global__ __launch_bounds__(256) void decoder_stream_pack_kernel(const DecStreamSrc s, u8* __restrict__ out, long long units) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= units) return;
    const DtPhases ph(s.oc);
    const int q = (int)(i & 7), r = (int)((i >> 3) & 15), w = (int)((i >> 7) & 7);
    const int slot = (int)(i >> 10);
    const int piece = q ^ (r >> 1);  // logical 16-byte piece stored at physical position q of line r: 0..3 hi k-groups, 4..7 lo
    const int kk = (piece & 3) * 8;
    const float* W = nullptr;  // row-major [N][K]
    int n = -1, N = 0, K = 0, c = 0;
    if (slot < ph.p0) { W = s.wf_h; N = s.O; K = DT_U; c = slot; n = w * 16 + r; }
    else if (slot < ph.p1) { const int x = slot - ph.p0; W = s.p_w0; N = DT_P; K = s.O; c = x % s.oc; n = ((x / s.oc) * 8 + w) * 16 + r; }
    else if (slot < ph.l0) { const int x = slot - ph.p1; W = s.p_w1; N = DT_P; K = DT_P; c = x & 7; n = ((x >> 3) * 8 + w) * 16 + r; }
    else {
        const bool l1 = slot >= ph.l1;
        const int x = slot - (l1 ? ph.l1 : ph.l0), j = x >> 6, cc = (x >> 2) & 15, g = x & 3;
        n = g * DT_U + (j * 8 + w) * 16 + r;
        N = 4 * DT_U;
        K = 256;
        c = cc & 7;
        W = cc < 8 ? (l1 ? s.w1_ih : s.w0_pre) : (l1 ? s.w1_hh : s.w0_hh);
    }
    u16 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = c * 32 + kk + e;
        const float x = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
        const __bf16 h = (__bf16)x;
        v[e] = piece < 4 ? __builtin_bit_cast(u16, h) : __builtin_bit_cast(u16, (__bf16)(x - (float)h));
    }
    *reinterpret_cast<uint4*>(out + i * 16) = make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16));
}

}  // namespace

// shapes the tile kernel is built for (FCL-taco2-S's decoder): U = P = 256, odim a multiple of 4 up to 128
bool decoder_tile_shape_ok(const fcl_decoder_weights_t* w) { return w->u == DT_U && w->p == DT_P && w->odim > 0 && w->odim <= 128 && (w->odim & 3) == 0; }

size_t decoder_stream_bytes(const fcl_decoder_weights_t* w) {
    if (!decoder_tile_shape_ok(w)) return 0;
    return (size_t)DtPhases((w->odim + 31) >> 5).total * DT_SLOT;
}

int decoder_stream_pack(const fcl_decoder_weights_t* w, void* out, hipStream_t s) {
    DecStreamSrc src = {w->wf_h, w->prenet_w0, w->prenet_w1, w->w0_pre, w->w0_hh, w->w1_ih, w->w1_hh, w->odim, (w->odim + 31) >> 5};
    const long long units = (long long)(decoder_stream_bytes(w) / 16);
    hipLaunchKernelGGL(decoder_stream_pack_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, src, reinterpret_cast<u8*>(out), units);
    return check_hip(hipGetLastError(), "decoder_stream_pack");
}

// the loop of fcl_decoder_loop_fwd from its hoisted terms on (G0, F0 in the workspace): one launch, ceil(n / 32) workgroups
int launch_decoder_tile(const fcl_decoder_weights_t* w, const fcl_decoder_io_t* io, const float* G0, const float* F0, float* c0, float* c1, int drop_mode,
                        int t_start, const float* h0_init, const float* h1_init, hipStream_t s) {
    DecTileArgs a = {};
    a.stream = reinterpret_cast<const u8*>(w->stream);
    a.oc = (w->odim + 31) >> 5;
    a.slots_per_step = DtPhases(a.oc).total;
    a.n = io->n; a.lmax = io->lmax; a.O = w->odim;
    a.G0 = G0; a.F0 = F0; a.w_pos = w->w0_pos; a.b1 = w->b1; a.pb0 = w->prenet_b0; a.pb1 = w->prenet_b1;
    a.dur = io->dur; a.frame_off = io->frame_off; a.live = io->live_rows; a.status = io->status;
    a.before = io->before; a.before_p = io->before_p; a.c0 = c0; a.c1 = c1;
    a.t_start = t_start; a.n_start = t_start > 0 ? io->live_rows_host[t_start] : io->n; a.h0_init = h0_init; a.h1_init = h1_init;
    a.zoneout = w->zoneout_rate; a.keep_scale = 1.0f / (1.0f - w->prenet_dropout); a.drop_p = w->prenet_dropout;
    a.out_act = w->out_act; a.seed = io->seed; a.seed_dev = io->seed_dev;
    // developer aid only (FCL_DEC_TILE_TS=1): phase stamps of workgroup 0, printed after a stream synchronisation -- not usable under graph capture
    // and not thread-safe (one static device buffer on the device of the first call)
    static const int want_ts = tunable("DEC_TILE_TS", 0);
    static long long* ts_dev = nullptr;
    if (want_ts && !ts_dev) (void)hipMalloc(&ts_dev, 4096 * 8 * sizeof(long long));
    a.ts = want_ts ? ts_dev : nullptr;
    FCL_REQUIRE(io->lmax <= DT_MAX_STEPS, FCL_ERR_SHAPE, "decoder tile kernel: lmax %d > %d", io->lmax, DT_MAX_STEPS);
    for (int t = 0; t < io->lmax; ++t) a.bound[t] = io->live_rows_host[t];
    const void* fn = drop_mode == FCL_DROP_RNG ? reinterpret_cast<const void*>(decoder_tile_kernel<2>) : reinterpret_cast<const void*>(decoder_tile_kernel<0>);
    const int rc = ensure_dyn_lds(fn, DT_LDS);
    if (rc) return rc;
    const dim3 grid((a.n_start + DT_ROWS - 1) / DT_ROWS), block(64 * (8 + DT_NL));
    double steps = 0;
    for (int t = t_start; t < io->lmax; ++t) steps += io->live_rows_host[t];
    // algorithmic flops: per live row-step 2 * (4U (P + U) + 4U * 2U + P (O + P) + O U)
    const double per = 2.0 * (4.0 * DT_U * (DT_P + DT_U) + 4.0 * DT_U * 2 * DT_U + (double)DT_P * (w->odim + DT_P) + (double)w->odim * DT_U);
    ProfScope prof(drop_mode == FCL_DROP_RNG ? "decoder_tile_kernel<rng>" : "decoder_tile_kernel<none>", per * steps, steps, s);
    if (drop_mode == FCL_DROP_RNG) hipLaunchKernelGGL(decoder_tile_kernel<2>, grid, block, DT_LDS, s, a);
    else hipLaunchKernelGGL(decoder_tile_kernel<0>, grid, block, DT_LDS, s, a);
    if (want_ts) {  // developer aid: synchronous, prints the first workgroup's per-step phase cycles
        (void)hipStreamSynchronize(s);
        long long h[8];
        (void)hipMemcpy(h, ts_dev, sizeof(h), hipMemcpyDeviceToHost);
        const double T = (double)h[6];
        fprintf(stderr, "decoder_tile ts (cycles per step, workgroup 0, %d steps): F loop %.0f  F epi %.0f  prenet loops %.0f  prenet epi %.0f  LSTM loops %.0f  LSTM epi %.0f  sum %.0f\n", (int)T,
                h[0] / T, h[1] / T, h[2] / T, h[3] / T, h[4] / T, h[5] / T, (h[0] + h[1] + h[2] + h[3] + h[4] + h[5]) / T);
    }
    return check_hip(hipGetLastError(), "decoder_tile_kernel");
}

}  // namespace fcl
