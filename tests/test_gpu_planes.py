"""-m gpu: the pre-split (P32) operand path — plane packing (bit-exact vs a numpy restatement of the bf16 split), the LDS-DMA GEMM / Conv1d /
LSTM-step kernels against fp64 references and against the fp32-operand kernels, the plane outputs of every producer, edge shapes (rows and
columns that do not fill a tile, K that is not a multiple of 32, one-row matrices, ragged conv segments)."""
import numpy as np
import pytest
import torch

from helpers import max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    import fcl_taco2_amd  # noqa: F401
    from fcl_taco2_amd import _lib, ops as _ops

    _lib.load()
    if not _ops.planes_enabled():
        pytest.skip("FCL_PRECISION=0 / FCL_PLANES=0: the pre-split operand path is off")
    return _ops


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.to(DEV).contiguous()


def rnd(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


def bf16_rn(x):
    """numpy restatement of the device's float -> bf16 conversion (round to nearest even), as the uint16 bit pattern."""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((b + 0x7FFF + ((b >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def split_planes_np(x):
    """[R, K] float32 -> P32 planes uint16 [R, ceil(K/32), 2, 32] (include/fcl_hip.h: hi = bf16_rn(x), lo = bf16_rn(x - hi), zeros past K)."""
    r, k = x.shape
    kp = (k + 31) // 32 * 32
    xp = np.zeros((r, kp), np.float32)
    xp[:, :k] = x
    hi = bf16_rn(xp)
    lo = bf16_rn(xp - bf16_to_f32(hi))
    return np.stack([hi.reshape(r, kp // 32, 32), lo.reshape(r, kp // 32, 32)], axis=2)


def unpack(planes, rows, cols):
    """device planes (int16 [rows, lines * 64]) -> hi + lo as float64 [rows, cols] and the raw uint16 array."""
    raw = planes.cpu().numpy().view(np.uint16).reshape(rows, -1, 2, 32)
    val = bf16_to_f32(raw[:, :, 0, :]).astype(np.float64) + bf16_to_f32(raw[:, :, 1, :]).astype(np.float64)
    return val.reshape(rows, -1)[:, :cols], raw


@pytest.mark.parametrize("rows,cols", [(1, 4), (7, 80), (33, 256), (130, 100)])
def test_pack_planes_is_the_documented_layout_bit_for_bit(ops, rows, cols):
    rng = np.random.RandomState(rows * 31 + cols)
    x = rnd(rng, rows, cols) * np.float32(3.0)
    x[0, 0] = 0.0
    if cols > 2:
        x[0, 1], x[0, 2] = np.float32(1e-30), np.float32(-65504.0)
    got = ops.pack_planes(dev(x)).cpu().numpy().view(np.uint16).reshape(rows, -1, 2, 32)
    assert np.array_equal(got, split_planes_np(x))
    from fcl_taco2_amd import _lib

    assert _lib.load().fcl_planes_elems(rows, cols) == got.size
    val, _ = unpack(ops.pack_planes(dev(x)), rows, cols)
    assert np.max(np.abs(val - x.astype(np.float64)) / np.maximum(np.abs(x), 1e-30)) < 2.0 ** -15  # hi + lo carries ~16 mantissa bits


@pytest.mark.parametrize("m,n,k", [(1, 16, 32), (17, 80, 80), (100, 130, 96), (257, 384, 260), (2501, 1024, 256), (3000, 80, 256), (64, 1104, 512)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_on_planes_vs_fp64_and_plane_output(ops, m, n, k, act):
    rng = np.random.RandomState(m + n + k)
    x, w, b = rnd(rng, m, k), rnd(rng, n, k) / np.float32(np.sqrt(k)), rnd(rng, n)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    ref = np.maximum(ref, 0) if act == 1 else np.tanh(ref) if act == 2 else ref
    xp, wp = ops.pack_planes(dev(x)), ops.pack_planes(dev(w))
    y, yp = ops.linear_planes(xp, wp, n, k, dev(b), act, want_f32=True, want_planes=True)
    assert max_abs(y.cpu().double(), ref) < 3e-5  # bf16x3: ~2^-16 relative per product, fp32 accumulation
    val, raw = unpack(yp, m, n)
    assert np.max(np.abs(val - y.cpu().numpy().astype(np.float64))) < 2.0 ** -15 * max(1.0, float(np.abs(ref).max()))
    assert not raw.reshape(m, -1, 2, 32).transpose(0, 1, 3, 2).reshape(m, -1, 2)[:, n:].any()  # zero padding past N in both planes
    yp_only = ops.linear_planes(xp, wp, n, k, dev(b), act, want_f32=False, want_planes=True)[1]
    assert torch.equal(yp_only, yp)  # the planes do not depend on whether the fp32 copy is written


@pytest.mark.parametrize("cin,cout,ksz", [(80, 128, 5), (256, 256, 5), (384, 384, 3), (128, 80, 5), (16, 8, 9)])
def test_conv1d_on_planes_vs_the_fp32_operand_kernel(ops, cin, cout, ksz):
    """Ragged segments (1-row, shorter than the kernel, longer than a tile), taps reading past both segment ends, residual + tanh."""
    rng = np.random.RandomState(cin + cout)
    seg_lens = [3, 1, 50, 17, 129, 300]
    M = sum(seg_lens)
    lo = np.repeat(np.cumsum([0] + seg_lens[:-1]), seg_lens).astype(np.int32)
    hi = (lo + np.repeat(seg_lens, seg_lens)).astype(np.int32)
    x, w, b, res = rnd(rng, M, cin), rnd(rng, cout, cin, ksz) / np.float32(np.sqrt(cin * ksz)), rnd(rng, cout), rnd(rng, M, cout)
    wp = ops.pack_conv1d_weight(dev(w))
    y_ref = ops.conv1d(dev(x), wp, dev(b), dev(lo), dev(hi), ops.ACT_TANH, residual=dev(res))

    class CV(object):
        pass

    cv = CV()
    cv.wpp, cv.bias, cv.cin, cv.cout, cv.k = ops.pack_planes(wp.reshape(ksz * cout, cin)), dev(b), cin, cout, ksz
    y, yp = ops.conv1d_planes(ops.pack_planes(dev(x)), cv, dev(lo), dev(hi), ops.ACT_TANH, residual=dev(res), want_f32=True, want_planes=True)
    assert max_abs(y.cpu(), y_ref.cpu()) < 3e-5
    # independent fp64 reference on two segments (zero padding at the segment ends)
    pad = (ksz - 1) // 2
    for s0, ln in ((0, 3), (3, 1), (71, 129)):
        xs = np.zeros((ln + 2 * pad, cin))
        xs[pad: pad + ln] = x[s0: s0 + ln]
        want = np.stack([sum(xs[t + j] @ w[:, :, j].astype(np.float64).T for j in range(ksz)) for t in range(ln)]) + b
        want = np.tanh(want) + res[s0: s0 + ln]
        assert max_abs(y.cpu().double()[s0: s0 + ln], want) < 5e-5
    val, _ = unpack(yp, M, cout)
    assert np.max(np.abs(val - y.cpu().numpy())) < 2.0 ** -15 * max(1.0, float(y.abs().max()))


@pytest.mark.parametrize("m,u,k0,k1", [(2501, 256, 256, 256), (1100, 256, 256, 256), (300, 1024, 256, 1024), (70, 32, 32, 64),
                                       (2501, 1024, 256, 1024), (1190, 1024, 256, 1024)])  # T-size steps: more tiles than CUs
def test_lstm_step_on_planes_vs_the_fp32_operand_kernels(ops, m, u, k0, k1):
    """Same step through the big-tile fp32-operand kernel (planes off for this call: no plane pointers) and through the LDS-DMA kernel:
    h, c and the h planes agree; decoder layer-0 form (hoisted G + position term) and layer-1 form (bias)."""
    import ctypes as C

    from fcl_taco2_amd import _lib

    rng = np.random.RandomState(m + u)
    A0, A1 = rnd(rng, m, k0), rnd(rng, m, k1) * np.float32(0.5)
    W0, W1 = rnd(rng, 4 * u, k0) / np.float32(np.sqrt(k0 + k1)), rnd(rng, 4 * u, k1) / np.float32(np.sqrt(k0 + k1))
    G, bias, wpos = rnd(rng, m, 4 * u) * np.float32(0.3), rnd(rng, 4 * u) * np.float32(0.3), rnd(rng, 4 * u) * np.float32(0.3)
    dur = rng.randint(1, 30, size=m).astype(np.int32)
    h_in, c_in = rnd(rng, m, u) * np.float32(0.5), rnd(rng, m, u)
    for layer0 in (True, False):
        outs = []
        for planes in (False, True):
            a = _lib.LstmStep()
            a.nterms, a.M, a.U = 2, m, u
            keep = []
            for i, (A, W, K) in enumerate(((A0, W0, k0), (A1, W1, k1))):
                At, Wt = dev(A), dev(W)
                keep += [At, Wt]
                ap = wp = None
                if planes:
                    ap, wp = ops.pack_planes(At), ops.pack_planes(Wt)
                    keep += [ap, wp]
                a.term[i] = _lib.GemmTerm(At.data_ptr(), Wt.data_ptr(), K, K, K, 0, None, None, ap.data_ptr() if planes else None,
                                          wp.data_ptr() if planes else None, (K + 31) // 32, (K + 31) // 32)
            Gt, bt, wt, dt = dev(G), dev(bias), dev(wpos), dev(dur)
            if layer0:
                a.G, a.g_row_mul, a.rank1_w, a.dur, a.step = Gt.data_ptr(), 1, wt.data_ptr(), dt.data_ptr(), 3
            else:
                a.bias = bt.data_ptr()
            hin, c, hout = dev(h_in), dev(c_in.copy()), torch.empty(m, u, device=DEV)
            hp = ops.planes_empty(m, u, DEV) if (planes and u % 32 == 0) else None
            a.h_in, a.h_out, a.c, a.zoneout = hin.data_ptr(), hout.data_ptr(), c.data_ptr(), 0.1
            if hp is not None:
                a.h_out_p, a.ld_hp = hp.data_ptr(), u // 32
            _lib.check(_lib.load().fcl_lstm_step_fwd(C.byref(a), ops._stream()))
            torch.cuda.synchronize()
            outs.append((hout.cpu(), c.cpu(), hp))
        (h0, c0, _), (h1, c1, hp) = outs
        assert max_abs(h1, h0) < 2e-5 and max_abs(c1, c0) < 5e-5
        if hp is not None:
            val, _ = unpack(hp, m, u)
            assert np.max(np.abs(val - h1.numpy())) < 2.0 ** -15


def test_producers_write_the_same_planes_as_pack_planes(ops):
    """embedding / row gather / LayerNorm / act_fwd / bn_act: their P32 outputs equal fcl_pack_planes of their fp32 outputs bit for bit."""
    rng = np.random.RandomState(5)
    table = dev(rnd(rng, 20, 96))
    ids = dev(rng.randint(0, 20, size=300).astype(np.int64))
    y, yp = ops.embedding(ids, table, want_planes=True)
    assert torch.equal(yp, ops.pack_planes(y))
    src = dev(rnd(rng, 500, 80))
    idx = dev(rng.randint(-1, 500, size=333).astype(np.int32))  # -1: zero row
    g, gp = ops.gather_rows(src, idx, want_planes=True)
    assert torch.equal(gp, ops.pack_planes(g)) and torch.equal(ops.gather_rows(src, idx, want_f32=False, want_planes=True)[1], gp)
    x = dev(rnd(rng, 257, 384))
    gam, bet = dev(rnd(rng, 384)), dev(rnd(rng, 384))
    ln, _, lnp = ops.layernorm(x, gam, bet, 1e-12, want_planes=True)
    assert torch.equal(lnp, ops.pack_planes(ln))
    keep = dev((rng.rand(257, 384) > 0.5).astype(np.uint8))
    a, ap = ops.act_fwd(x, ops.ACT_RELU, keep, 2.0, want_planes=True)
    assert torch.equal(ap, ops.pack_planes(a))
    mean, invstd = dev(rnd(rng, 384)), dev(np.abs(rnd(rng, 384)) + np.float32(0.5))
    _, yd, ydp = ops.bn_act(x, mean, invstd, gam, bet, ops.ACT_TANH, keep, 2.0, want_planes=True)
    assert torch.equal(ydp, ops.pack_planes(yd))


def test_synthesis_with_and_without_planes_agree(ops):
    """End to end: the pre-split path (default) and the fp32-operand path (planes switched off in the plan) give the same mel to ~1e-5."""
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r);"
        "import fcl_taco2_amd; from fcl_taco2_amd import engine, hparams as HP, synthetic as SYN; from fcl_taco2_amd.plan import SynthesisPlan;"
        "hp = HP.student_hparams(dropout_rate=0.0); plan = SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, 'cuda:0');"
        "xs, ds = SYN.batch_c2(hp.idim, batch=3, t_lo=20, t_hi=40, seed=11);"
        "m = engine.synthesize(plan, xs, ds); np.save(sys.argv[1], torch.cat(m).cpu().numpy())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for flag in ("1", "0"):
        path = "/tmp/fcl_planes_%s.npy" % flag
        env = dict(os.environ, FCL_PLANES=flag)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env)
        outs.append(np.load(path))
    assert outs[0].shape == outs[1].shape and np.max(np.abs(outs[0] - outs[1])) < 5e-5


def test_derived_forms_batch_is_bit_exact_and_follows_parameter_updates(ops):
    """fcl_derive_batch (one launch for all operand forms of the parameters): every form equals the one-at-a-time kernels bit for bit, both the
    fp32 output and the planes; a refresh after an in-place parameter change reproduces them from the new values with the same buffers."""
    rng = np.random.RandomState(5)
    W = dev(rnd(rng, 96, 77))       # Linear-like [rows, ld], odd ld
    Cw = dev(rnd(rng, 40, 24, 5))   # Conv1d weight [Cout, Cin, k]
    b1, b2 = dev(rnd(rng, 130)), dev(rnd(rng, 130))
    forms = ops.DerivedForms(torch.device(DEV))

    def ask(stamp):
        out = {}
        out["t"] = forms.get(stamp, ("t",), W, (1, 77, 96, 0, 1, 77), f32=True, planes=True)
        out["c"] = forms.get(stamp, ("c",), W, (1, 96, 33, 0, 77, 1), base=40, f32=True, planes=True)
        out["ct"] = forms.get(stamp, ("ct",), W, (1, 33, 96, 0, 1, 77), base=40, f32=True, planes=True)
        out["col"] = forms.get(stamp, ("col",), W, (1, 96, 1, 0, 77, 1), base=76, f32=True, planes=False)
        out["cp"] = forms.get(stamp, ("cp",), Cw, (5, 40, 24, 1, 120, 5), f32=True, planes=True)
        out["cr"] = forms.get(stamp, ("cr",), Cw, (5, 24, 40, -1, 5, 120), base=4, f32=True, planes=True)
        out["bs"] = forms.get(stamp, ("bs",), b1, (1, 1, 130, 0, 0, 1), src2=b2, f32=True, planes=False)
        out["p"] = forms.get(stamp, ("p",), W, (1, 96, 77, 0, 77, 1), f32=False, planes=True)
        return out

    def expect():
        wp = Cw.permute(2, 0, 1).contiguous()  # [k, Cout, Cin]
        return {
            "t": W.t().contiguous(), "c": W[:, 40:73].contiguous(), "ct": W[:, 40:73].t().contiguous(), "col": W[:, 76:77].contiguous(),
            "cp": wp.reshape(5 * 40, 24), "cr": torch.stack([wp[4 - j].t() for j in range(5)]).reshape(5 * 24, 40).contiguous(), "bs": (b1 + b2).reshape(1, -1),
            "p": W,
        }

    def check(out):
        for k, ref in expect().items():
            f32, pl = out[k]
            if f32 is not None:
                assert torch.equal(f32, ref), k
            if pl is not None:
                assert torch.equal(pl, ops.pack_planes(ref)), k

    first = ask(0)  # registered and computed one by one
    check(first)
    ptrs = {k: tuple(None if t is None else t.data_ptr() for t in v) for k, v in first.items()}
    W.mul_(1.5).add_(0.25)
    Cw.neg_()
    b2.add_(1.0)
    second = ask(1)  # one batched launch refreshes all eight
    assert forms.table is not None and forms.table[1] == 8
    assert {k: tuple(None if t is None else t.data_ptr() for t in v) for k, v in second.items()} == ptrs
    check(second)


def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("m,n,k", [(300, 96, 160), (2501, 1024, 256), (37, 40, 72)])
def test_bf16_gemm_mode_rounds_both_operands_and_accumulates_in_fp32(ops, m, n, k):
    """ops.gemm_mode("bf16") (fcl_set_gemm_mode: the autocast form for --use-amp training): the planes GEMM, the fp32-operand GEMM and the
    weight-gradient GEMM equal the fp64 product of the bf16-ROUNDED operands (tolerance = fp32 accumulation only), differ from the fp32-equivalent
    result at bf16 level, and the mode is restored on exit."""
    from fcl_taco2_amd import _lib

    rng = np.random.RandomState(m + n)
    x, w, b = dev(rnd(rng, m, k)), dev(rnd(rng, n, k)), dev(rnd(rng, n))
    ref = _bf16_round(x) @ _bf16_round(w).t() + b.double()
    exact = x.double() @ w.double().t() + b.double()
    scale = float(exact.abs().max())
    xp, wp = ops.pack_planes(x), ops.pack_planes(w)
    with ops.gemm_mode("bf16"):
        assert _lib.load().fcl_get_gemm_mode() == _lib.GEMM_BF16
        y_p = ops.linear_planes(xp, wp, n, k, b, want_f32=True, want_planes=False)[0]
        y_f = ops.linear(x, w, b)
    assert _lib.load().fcl_get_gemm_mode() == _lib.GEMM_F32
    for y in (y_p, y_f):
        assert max_abs(y.double(), ref) < 2e-5 * scale
        assert max_abs(y.double(), exact) > 1e-4 * scale  # really the rounded operands
    y3 = ops.linear_planes(xp, wp, n, k, b, want_f32=True, want_planes=False)[0]  # back to the fp32-equivalent split
    assert max_abs(y3.double(), exact) < 2e-5 * scale
    # weight gradient: c[n, k] += sum_m a[m, n] * b[m, k]
    dy = dev(rnd(rng, m, n))
    g = torch.zeros(n, k, device=DEV)
    with ops.gemm_mode("bf16"):
        ops.gemm_tn(dy, x, g)
    gref = _bf16_round(dy).t() @ _bf16_round(x)
    assert max_abs(g.double(), gref) < 3e-5 * float(gref.abs().max())


def test_bf16_gemm_mode_lstm_step_on_planes(ops):
    """The LDS-DMA LSTM-step kernel under ops.gemm_mode("bf16") = the cell evaluated in fp64 on bf16-rounded x, h and weights."""
    import ctypes as C

    from fcl_taco2_amd import _lib

    rng = np.random.RandomState(11)
    m, u, k0 = 700, 256, 256
    x, h = dev(rnd(rng, m, k0)), dev(0.5 * rnd(rng, m, u))
    c0 = dev(0.5 * rnd(rng, m, u))
    w_ih, w_hh, b = dev(0.1 * rnd(rng, 4 * u, k0)), dev(0.1 * rnd(rng, 4 * u, u)), dev(0.1 * rnd(rng, 4 * u))
    gates = _bf16_round(x) @ _bf16_round(w_ih).t() + _bf16_round(h) @ _bf16_round(w_hh).t() + b.double()
    i, f, g, o = gates.split(u, dim=1)
    c_ref = torch.sigmoid(f) * c0.double() + torch.sigmoid(i) * torch.tanh(g)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    a = _lib.LstmStep()
    a.nterms, a.M, a.U = 2, m, u
    planes = [ops.pack_planes(t) for t in (x, w_ih, h, w_hh)]
    a.term[0] = _lib.GemmTerm(x.data_ptr(), w_ih.data_ptr(), k0, k0, k0, 0, None, None, planes[0].data_ptr(), planes[1].data_ptr(), k0 // 32, k0 // 32)
    a.term[1] = _lib.GemmTerm(h.data_ptr(), w_hh.data_ptr(), u, u, u, 0, None, None, planes[2].data_ptr(), planes[3].data_ptr(), u // 32, u // 32)
    a.bias = b.data_ptr()
    c_io, h_out = c0.clone(), torch.empty(m, u, device=DEV)
    a.h_in, a.h_out, a.c, a.zoneout = h.data_ptr(), h_out.data_ptr(), c_io.data_ptr(), 0.0
    with ops.gemm_mode("bf16"):
        _lib.check(_lib.load().fcl_lstm_step_fwd(C.byref(a), ops._stream()))
    torch.cuda.synchronize()
    assert max_abs(h_out.double(), h_ref) < 1e-5 and max_abs(c_io.double(), c_ref) < 1e-5
    exact = x.double() @ w_ih.double().t() + h.double() @ w_hh.double().t() + b.double()
    assert float((gates - exact).abs().max()) > 1e-3  # the rounding is visible at gate level


@pytest.mark.parametrize("m,n,k", [(1000, 64, 96), (2501, 256, 128), (333, 40, 36), (4100, 1024, 256)])
def test_weight_gradient_gemm_on_transposed_planes(ops, m, n, k):
    """fcl_pack_planes_t + fcl_gemm_tn_planes: c[n, k] += sum_m a[m, n] b[m, k] equals the fp64 product (fp32-equivalent), accumulates into c,
    handles contraction lengths that are not multiples of 32 and outputs that do not fill a tile; strided output rows."""
    rng = np.random.RandomState(m)
    a, b = dev(rnd(rng, m, n)), dev(rnd(rng, m, k))
    ref = a.double().t() @ b.double()
    base = dev(rnd(rng, n, k + 8))
    out = base.clone()
    ops.gemm_tn_planes(ops.pack_planes_t(a), ops.pack_planes_t(b), out[:, 4 : 4 + k], m)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert max_abs((out[:, 4 : 4 + k] - base[:, 4 : 4 + k]).double(), ref) < 3e-5 * scale
    assert torch.equal(out[:, :4], base[:, :4]) and torch.equal(out[:, 4 + k :], base[:, 4 + k :])


def test_conv_weight_gradient_taps_on_transposed_planes(ops):
    """All taps of a Conv1d weight gradient in one GEMM: the shifted, segment-masked inputs stacked as plane rows; tap-major [k, Cout, Cin] output
    equal to fcl_gemm_tn_taps_fwd's."""
    rng = np.random.RandomState(3)
    lens = [70, 41, 5, 1, 64]
    m, cin, cout, ksz = sum(lens), 48, 72, 5
    lo = np.concatenate([np.full(n, s, np.int32) for n, s in zip(lens, np.cumsum([0] + lens[:-1]))])
    hi = np.concatenate([np.full(n, s + n, np.int32) for n, s in zip(lens, np.cumsum([0] + lens[:-1]))])
    dz, x = dev(rnd(rng, m, cout)), dev(rnd(rng, m, cin))
    lo_d, hi_d = dev(lo), dev(hi)
    want = torch.zeros(ksz, cout, cin, device=DEV)
    ops.gemm_tn_taps(dz, x, want, -2, seg_lo=lo_d, seg_hi=hi_d)
    got = torch.zeros(ksz, cout, cin, device=DEV)
    ops.gemm_tn_planes(ops.pack_planes_t(dz), ops.pack_planes_t(x, ntaps=ksz, shift0=-2, seg_lo=lo_d, seg_hi=hi_d), got, m)
    torch.cuda.synchronize()
    assert max_abs(got, want) < 3e-5 * float(want.abs().max())
