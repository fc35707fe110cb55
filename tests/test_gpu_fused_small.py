"""-m gpu: the round-6 fused small launches of the training update (csrc/fused_small.hip) against numpy / torch float64 restatements and against the
kernels they replace (fcl_l1_mse_loss_grad, fcl_masked_l1_mse_fwd + fcl_l1_mse_grad, fcl_add2d chains, fcl_act_bwd + fcl_colsum2_fwd, fcl_gather_rows_fwd,
fcl_linear_fwd): the loss arithmetic is that of ..._kd_student.py:759-802 / ..._sa.py:60-70 (masked L1 + MSE means and their gradients)."""
import os

import numpy as np
import pytest
import torch

from helpers import max_abs
from test_gpu_planes import split_planes_np

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    import fcl_taco2_amd  # noqa: F401
    from fcl_taco2_amd import _lib, ops as _ops

    _lib.load()
    return _ops


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.to(DEV).contiguous()


def rnd(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


def ref_term(a, b, valid, count, w1, w2, log_off=None):
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    if log_off is not None:
        b64 = np.log(b.astype(np.float32) + np.float32(log_off)).astype(np.float64)
    d = a64 - b64
    ok = np.ones(a.shape[0], bool) if valid is None else valid.astype(bool)
    dm = d[ok]
    sums = np.array([np.abs(dm).sum(), (dm * dm).sum(), dm.size], np.float64)
    g = (w1 * np.sign(d) + 2.0 * w2 * d) / count
    g[~ok] = 0.0
    return sums, g


def test_loss_terms_batch_matches_float64_and_the_single_term_kernels(ops):
    rng = np.random.RandomState(0)
    shapes = [(2500, 80), (2500, 80), (300, 1), (300, 1), (2500, 512), (257, 128), (31, 6)]
    terms, refs = [], []
    sums = torch.zeros(16, 3, dtype=torch.float64, device=DEV)
    for i, (m, c) in enumerate(shapes):
        a, b, b2 = rnd(rng, m, c), rnd(rng, m, c), rnd(rng, m, c)
        valid = (rng.rand(m) > 0.2).astype(np.uint8)
        valid2 = (rng.rand(m) > 0.5).astype(np.uint8)
        log_off = 1.0 if i == 2 else None
        if log_off is not None:
            b = np.abs(b) * 3.0
        two = i in (0, 2, 3, 5)
        cnt, cnt2 = float(valid.sum() * c), float(max(valid2.sum(), 1) * c)
        w = (1.0, 1.0) if c == 80 else (0.0, 1.0)
        d = dict(a=dev(a), b=dev(b), valid=dev(valid) if i != 1 else None, slot=2 * i, count=cnt if i != 1 else float(m * c), w_l1=w[0], w_mse=w[1],
                 b_log_offset=log_off, want_planes=(c % 32 == 0))
        s1, g = ref_term(a, b, valid if i != 1 else None, d["count"], w[0], w[1], log_off)
        r = dict(s1=s1, g=g, s2=None)
        if two:
            d.update(b2=dev(b2), valid2=dev(valid2), slot2=2 * i + 1, count2=cnt2, w_l1_2=0.5, w_mse_2=2.0)
            s2, g2 = ref_term(a, b2, valid2, cnt2, 0.5, 2.0)
            r["s2"], r["g"] = s2, g + g2
        terms.append(d)
        refs.append(r)
    outs = ops.loss_terms_batch(terms, sums)
    torch.cuda.synchronize()
    S = sums.cpu().numpy()
    for i, ((da, dap), r, (m, c)) in enumerate(zip(outs, refs, shapes)):
        assert np.allclose(S[2 * i], r["s1"], rtol=1e-6, atol=1e-9), (i, S[2 * i], r["s1"])
        if r["s2"] is not None:
            assert np.allclose(S[2 * i + 1], r["s2"], rtol=1e-6, atol=1e-9), (i, S[2 * i + 1], r["s2"])
        else:
            assert not S[2 * i + 1].any()
        got = da.cpu().numpy().astype(np.float64)
        assert np.abs(got - r["g"]).max() <= 3e-7 * max(1e-30, np.abs(r["g"]).max()) + 1e-12, i
        if dap is not None:  # planes = the bf16 hi / lo split of the fp32 gradient, bit for bit
            raw = dap.cpu().numpy().view(np.uint16).reshape(m, -1, 2, 32)
            assert (raw == split_planes_np(da.cpu().numpy())).all()
    # against the kernels the batch replaces: same fp32 gradient bit for bit (single target), same sums
    a, b = terms[4]["a"], terms[4]["b"]
    s_old = torch.zeros(3, dtype=torch.float64, device=DEV)
    da_old = ops.l1_mse_loss_grad(a, b, terms[4]["valid"], terms[4]["count"], 0.0, 1.0, s_old)
    assert torch.equal(da_old, outs[4][0])
    assert np.allclose(s_old.cpu().numpy(), S[8], rtol=1e-12)
    # two targets == the accumulate pass of the old kernel (up to the contraction of the second term's multiply with the addition: one fp32 rounding)
    t0 = terms[0]
    s_a, s_b = torch.zeros(3, dtype=torch.float64, device=DEV), torch.zeros(3, dtype=torch.float64, device=DEV)
    g = ops.l1_mse_loss_grad(t0["a"], t0["b"], t0["valid"], t0["count"], 1.0, 1.0, s_a)
    g = ops.l1_mse_loss_grad(t0["a"], t0["b2"], t0["valid2"], t0["count2"], 0.5, 2.0, s_b, da=g)
    assert float((g - outs[0][0]).abs().max()) <= 2e-7 * float(g.abs().max())


def test_loss_terms_batch_empty_rows_and_argument_validation(ops):
    from fcl_taco2_amd import _lib

    sums = torch.zeros(4, 3, dtype=torch.float64, device=DEV)
    a = torch.zeros(0, 8, device=DEV)
    outs = ops.loss_terms_batch([dict(a=a, b=a, slot=0, count=1.0, w_l1=1.0, w_mse=1.0)], sums)
    torch.cuda.synchronize()
    assert outs[0][0].shape == (0, 8) and not sums.cpu().numpy().any()
    arr = (_lib.LossTerm * 1)()
    assert _lib.load().fcl_loss_terms_batch(arr, 1, None) != 0  # null operands
    assert _lib.load().fcl_loss_terms_batch(arr, 13, None) != 0  # more than FCL_LOSS_MAX_TERMS


@pytest.mark.parametrize("rows,cols,n", [(300, 256, 5), (2561, 128, 2), (7, 4, 6)])
def test_sum_rows(ops, rows, cols, n):
    rng = np.random.RandomState(rows)
    srcs = [rnd(rng, rows, cols) for _ in range(n)]
    valid = (rng.rand(rows) > 0.3).astype(np.uint8)
    ref = np.zeros((rows, cols), np.float32)
    acc = srcs[0].copy()
    for s in srcs[1:]:
        acc = acc + s  # fp32, left to right: the kernel's order
    ref[valid.astype(bool)] = acc[valid.astype(bool)]
    ds = [dev(s) for s in srcs]
    if cols % 32 == 0:
        out, planes = ops.sum_rows(ds, dev(valid), want_planes=True)
        raw = planes.cpu().numpy().view(np.uint16).reshape(rows, -1, 2, 32)
        assert (raw == split_planes_np(ref)).all()
    else:
        out = ops.sum_rows(ds, dev(valid))
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(ops.sum_rows(ds).cpu().numpy(), acc)
    # in place on the first source
    ops.sum_rows(ds, None, out=ds[0])
    assert np.array_equal(ds[0].cpu().numpy(), acc)


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("m,c,use_keep,use_dy2", [(2500, 128, True, True), (900, 256, True, False), (131, 80, False, True), (9000, 64, False, False)])
def test_bn_bwd_sums_equals_act_bwd_plus_colsum2(ops, act, m, c, use_keep, use_dy2):
    rng = np.random.RandomState(m + act)
    dy, dy2, z = rnd(rng, m, c), rnd(rng, m, c), rnd(rng, m, c)
    y = np.tanh(rnd(rng, m, c)) if act == 2 else np.maximum(rnd(rng, m, c), 0)
    keep = (rng.rand(m, c) > 0.5).astype(np.uint8)
    mean, invstd = rnd(rng, c) * 0.1, np.abs(rnd(rng, c)) + 0.5
    ks = 2.0
    dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    dz = ops.bn_bwd_sums(dev(dy), dev(z), dev(mean), dev(invstd), dg, db, act=act, y_act=dev(y), keep=dev(keep) if use_keep else None, keep_scale=ks,
                         dy2=dev(dy2) if use_dy2 else None)
    # the launches it replaces
    src = dev(dy)
    if use_dy2:
        src = ops.sum_rows([dev(dy), dev(dy2)])
    dz_old = ops.act_bwd(src, dev(y), act, keep=dev(keep) if use_keep else None, keep_scale=ks)
    dg_old, db_old = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    ops.colsum(dz_old, dg_old, y=dev(z), gamma=dev(invstd), beta=dev(mean), mode=3, out_x=db_old)
    assert torch.equal(dz, dz_old)
    assert max_abs(dg, dg_old.cpu()) <= 2e-6 * float(dg_old.abs().max()) + 1e-6
    assert max_abs(db, db_old.cpu()) <= 2e-6 * float(db_old.abs().max()) + 1e-6
    # float64
    g = (dy.astype(np.float64) + (dy2 if use_dy2 else 0)) * ((keep * ks) if use_keep else 1.0)
    if act == 1:
        g = g * (y > 0)
    elif act == 2:
        g = g * (1.0 - y.astype(np.float64) ** 2)
    zh = (z.astype(np.float64) - mean) * invstd
    assert np.abs(dg.cpu().numpy() - (g * zh).sum(0)).max() < 2e-4 * np.sqrt(m)
    assert np.abs(db.cpu().numpy() - g.sum(0)).max() < 2e-4 * np.sqrt(m)


def test_act_bwd_sum_and_gather_rows_sum(ops):
    rng = np.random.RandomState(5)
    m, c = 1300, 256
    dy, dy2, y = rnd(rng, m, c), rnd(rng, m, c), np.maximum(rnd(rng, m, c), 0)
    keep = (rng.rand(m, c) > 0.5).astype(np.uint8)
    for act in (0, 1):
        for k in (None, dev(keep)):
            got, gp = ops.act_bwd_sum(dev(dy), dev(dy2), dev(y), act, keep=k, keep_scale=2.0, want_planes=True)
            old, op = ops.act_bwd(ops.sum_rows([dev(dy), dev(dy2)]), dev(y), act, keep=k, keep_scale=2.0, want_planes=True)
            assert torch.equal(got, old) and torch.equal(gp, op)
    n = 700
    idx = rng.randint(-1, m, size=n).astype(np.int32)
    s1, s2, s3 = rnd(rng, m, 80), rnd(rng, m, 80), rnd(rng, m, 80)
    ref = np.where(idx[:, None] >= 0, (s1 + s2 + s3)[np.maximum(idx, 0)], 0.0).astype(np.float32)
    got, gp = ops.gather_rows_sum(dev(s1), dev(s2), dev(s3), dev(idx), want_planes=True)
    assert np.array_equal(got.cpu().numpy(), ref)
    raw = gp.cpu().numpy().view(np.uint16).reshape(n, -1, 2, 32)
    assert (raw == split_planes_np(ref)).all()
    ref1 = np.where(idx[:, None] >= 0, s1[np.maximum(idx, 0)], 0.0).astype(np.float32)
    assert np.array_equal(ops.gather_rows_sum(dev(s1), None, None, dev(idx)).cpu().numpy(), ref1)


@pytest.mark.parametrize("m,n,k,k2", [(3200, 256, 80, 1024), (2560, 256, 512, 512), (500, 256, 80, 0), (33, 8, 12, 4)])
def test_linear2_two_terms_and_residual(ops, m, n, k, k2):
    rng = np.random.RandomState(m)
    x, w = rnd(rng, m, k), rnd(rng, n, k) * 0.1
    x2, w2 = (rnd(rng, m, k2), rnd(rng, n, k2) * 0.1) if k2 else (None, None)
    res = rnd(rng, m, n)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + res
    if k2:
        ref = ref + x2.astype(np.float64) @ w2.astype(np.float64).T
    got = ops.linear2(dev(x), dev(w), dev(x2) if k2 else None, dev(w2) if k2 else None, residual=dev(res))
    tol = 2e-4 + 3e-5 * np.abs(ref).max()
    assert np.abs(got.cpu().numpy() - ref).max() < tol
    # the launches it replaces: linear + linear + two additions
    old = ops.linear(dev(x), dev(w))
    if k2:
        old = ops.sum_rows([old, ops.linear(dev(x2), dev(w2))]) if n % 4 == 0 else old + ops.linear(dev(x2), dev(w2))
    old = old + dev(res)
    assert np.abs(got.cpu().numpy() - old.cpu().numpy()).max() < tol


def test_bernoulli_batch_draws_byte_for_byte_what_the_single_site_kernel_draws():
    """fcl_bernoulli_batch writes 16 bytes per thread (round 6); ragged ends, tiny sites and a site that is not 16-byte aligned take the byte path: every
    site equals fcl_bernoulli_u8 with the same seed."""
    import ctypes as C

    from fcl_taco2_amd import _lib
    from fcl_taco2_amd import ops as O

    sizes = [1, 15, 16, 17, 4095, 4096, 4097, 100003, 1 << 20]
    sites = [((n,), 0.1 + 0.08 * k, 1000 + k) for k, n in enumerate(sizes)]
    got = O.bernoulli_batch(sites, DEV)
    for (shape, p, seed), g in zip(sites, got):
        want = O.bernoulli_u8(shape, p, seed, DEV)
        assert torch.equal(g, want), shape
    # an output that is NOT 16-byte aligned
    buf = torch.zeros(5000 + 3, device=DEV, dtype=torch.uint8)
    arr = (_lib.BernoulliSite * 1)()
    arr[0].out, arr[0].n, arr[0].p_one, arr[0].seed = buf.data_ptr() + 3, 5000, 0.5, 77
    _lib.check(_lib.load().fcl_bernoulli_batch(arr, 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.equal(buf[3:], O.bernoulli_u8((5000,), 0.5, 77, DEV)) and int(buf[:3].sum()) == 0


@pytest.mark.parametrize("m,n,k,masked", [(300, 64, 32, True), (4099, 512, 128, True), (2500, 1024, 256, False), (129, 96, 64, True)])
def test_linear_planes_mse_is_projection_plus_loss_terms_in_one_launch(m, n, k, masked):
    """fcl_linear_planes_mse_fwd (round 6: a KD term in the GEMM's epilogue) against the two launches it replaces -- fcl_linear_planes_fwd then
    fcl_loss_terms_batch -- gradients and their planes BIT FOR BIT (same y, same d, same gradient expression), sums to fp64 rounding (another summation order);
    and against float64."""
    from fcl_taco2_amd import ops as O

    if os.environ.get("FCL_PRECISION", "1") == "0" or os.environ.get("FCL_PLANES", "1") == "0":
        pytest.skip("planes kernels only")
    rs = np.random.RandomState(m + n)
    x = torch.from_numpy(rs.randn(m, k).astype(np.float32)).to(DEV)
    w = torch.from_numpy((rs.randn(n, k) / np.sqrt(k)).astype(np.float32)).to(DEV)
    t = torch.from_numpy(rs.randn(m, n).astype(np.float32)).to(DEV)
    valid = torch.from_numpy((rs.rand(m) < 0.8).astype(np.uint8)).to(DEV) if masked else None
    count = float((int(valid.sum()) if masked else m) * n)
    xp, wp = O.pack_planes(x), O.pack_planes(w)
    y, _ = O.linear_planes(xp, wp, n, k)
    sums_ref = torch.zeros(2, 3, dtype=torch.float64, device=DEV)
    (da, dap), = O.loss_terms_batch([dict(a=y, b=t, valid=valid, slot=1, count=count, w_l1=0.0, w_mse=1.0, want_planes=True)], sums_ref)
    sums = torch.zeros(2, 3, dtype=torch.float64, device=DEV)
    g, gp = O.linear_planes_mse(xp, wp, t, valid, count, sums[1], m, n, k)
    assert torch.equal(g, da)
    assert torch.equal(gp[:m], dap[:m])
    assert float(sums[0].abs().sum()) == 0.0 and torch.allclose(sums[1], sums_ref[1], rtol=1e-12, atol=0.0), (sums, sums_ref)
    # float64: d = x w^T - t on the valid rows
    d = x.double() @ w.double().t() - t.double()
    if masked:
        d = d * valid.bool().unsqueeze(1)
    assert float((g.double() - 2.0 * d / count).abs().max()) < 2e-5 / count * max(1.0, float(d.abs().max())) + 1e-9
    assert abs(float(sums[1, 1]) - float((d * d).sum())) < 1e-4 * float((d * d).sum())
    assert float(sums[1, 2]) == count
    # planes only / fp32 only
    s2 = torch.zeros(3, dtype=torch.float64, device=DEV)
    g2, gp2 = O.linear_planes_mse(xp, wp, t, valid, count, s2, m, n, k, want_f32=False)
    assert g2 is None and torch.equal(gp2[:m], gp[:m])


@pytest.mark.parametrize("m,cin,cout,k", [(2560, 256, 256, 5), (24301, 128, 128, 5), (1584, 512, 512, 5), (300, 80, 128, 5), (25001, 512, 80, 5)])
def test_conv1d_planes_bn_is_conv_plus_bn_stats_in_one_launch(m, cin, cout, k):
    """fcl_conv1d_planes_bn_fwd (round 6, VERDICT r5 #1b: BatchNorm statistics from the convolution's epilogue) against the two launches it replaces: z bit for bit,
    mean / invstd / running statistics to fp64-summation-order rounding; the workspace is zero again afterwards (a second call gives the same result); the stencil
    (Cin <= 384) and the K-term (Cin = 512) kernels, 64- and 128-row tiles, ragged last tiles."""
    from fcl_taco2_amd import ops as O

    if os.environ.get("FCL_PRECISION", "1") == "0" or os.environ.get("FCL_PLANES", "1") == "0":
        pytest.skip("planes kernels only")
    rs = np.random.RandomState(m + cout)
    x = torch.from_numpy(rs.randn(m, cin).astype(np.float32)).to(DEV)
    w = torch.from_numpy((rs.randn(cout, cin, k) / np.sqrt(cin * k)).astype(np.float32)).to(DEV)
    # three utterances: segment bounds per row
    cuts = [0, m // 3, m // 3 + m // 4, m]
    lo = torch.zeros(m, dtype=torch.int32)
    hi = torch.zeros(m, dtype=torch.int32)
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo[a:b], hi[a:b] = a, b
    lo, hi = lo.to(DEV), hi.to(DEV)
    xp = O.pack_planes(x)
    wpk = O.pack_conv1d_weight(w)  # [k, cout, cin]
    wpp = O.pack_planes(wpk.reshape(k * cout, cin))

    class CV:  # the plan's ConvPack, by hand
        pass

    cv = CV()
    cv.wpp, cv.bias, cv.k, cv.cin, cv.cout = wpp, None, k, cin, cout
    z_ref, _ = O.conv1d_planes(xp, cv, lo, hi, want_f32=True, want_planes=False)
    rm_ref, rv_ref = torch.full((cout,), 0.25, device=DEV), torch.full((cout,), 2.0, device=DEV)
    mean_ref, inv_ref = O.bn_stats(z_ref, 1e-5, 0.1, rm_ref, rv_ref)
    for rep in range(2):  # twice: the workspace (sums, tickets) must be left zero
        rm, rv = torch.full((cout,), 0.25, device=DEV), torch.full((cout,), 2.0, device=DEV)
        z, mean, inv = O.conv1d_planes_bn(xp, wpp, lo, hi, cin, cout, k, 1e-5, 0.1, rm, rv)
        assert torch.equal(z, z_ref), rep
        assert torch.allclose(mean, mean_ref, rtol=1e-6, atol=1e-7) and torch.allclose(inv, inv_ref, rtol=1e-6, atol=0.0), rep
        assert torch.allclose(rm, rm_ref, rtol=1e-6, atol=1e-7) and torch.allclose(rv, rv_ref, rtol=1e-6, atol=0.0), rep
    # float64
    zd = z.double()
    assert torch.allclose(mean.double(), zd.mean(0), rtol=1e-5, atol=1e-6)
    assert torch.allclose(inv.double(), 1.0 / torch.sqrt(zd.var(0, unbiased=False) + 1e-5), rtol=1e-5)
