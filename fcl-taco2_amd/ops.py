"""Tensor-level wrappers over the C ABI.  torch is used for device memory and the stream handle only;
every FLOP runs in libfcl_hip.so.  All tensors must be contiguous fp32 / int32 / int64 / uint8 CUDA(HIP)
tensors; nothing here falls back to torch ops."""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, DROP_MASK, DROP_NONE, DROP_RNG, check  # noqa: F401


try:  # raw handle of the current stream without building a torch.cuda.Stream object per call (7 us -> 0.3 us; a training step issues ~1000 ops)
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:  # pragma: no cover - older/newer torch without the private accessors
    _raw_stream = _cur_device = None


def _stream():
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.FclError("fcl-taco2_amd: expected a device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.FclError("fcl-taco2_amd: expected contiguous %s, got %s contiguous=%s" % (dtype, t.dtype, t.is_contiguous()))
    return t.data_ptr()


def pack_conv1d_weight(w, scale=None):
    cout, cin, k = w.shape
    out = torch.empty(k, cout, cin, device=w.device, dtype=torch.float32)
    check(_lib.load().fcl_pack_conv1d_weight(_p(w), _p(scale), _p(out), cout, cin, k, _stream()))
    return out


def fold_batchnorm(gamma, beta, mean, var, eps=1e-5):
    scale, shift = torch.empty_like(gamma), torch.empty_like(gamma)
    check(_lib.load().fcl_fold_batchnorm(_p(gamma), _p(beta), _p(mean), _p(var), eps, _p(scale), _p(shift), gamma.numel(), _stream()))
    return scale, shift


def copy_cols(src, col0, ncols):
    """Contiguous copy of src[:, col0:col0+ncols]."""
    rows, ld = src.shape
    dst = torch.empty(rows, ncols, device=src.device, dtype=torch.float32)
    check(_lib.load().fcl_copy2d(_p(dst), ncols, src.data_ptr() + 4 * col0, ld, rows, ncols, _stream()))
    return dst


def copy2d(dst, src):
    """dst[:, :] = src[:, :] for 2-D views with unit column stride."""
    rows, cols = src.shape
    assert dst.shape == src.shape and dst.stride(1) == 1 and src.stride(1) == 1 and dst.dtype == src.dtype == torch.float32 and dst.is_cuda
    check(_lib.load().fcl_copy2d(dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0), rows, cols, _stream()))
    return dst


def pack_frag_bf16(w):
    """Fragment-major bf16x3 planes (hi, lo) of a float32 matrix [rows, cols] (see include/fcl_hip.h)."""
    rows, cols = w.shape
    n = _lib.load().fcl_frag_bf16_elems(rows, cols)
    hi = torch.empty(n, device=w.device, dtype=torch.int16)
    lo = torch.empty(n, device=w.device, dtype=torch.int16)
    check(_lib.load().fcl_pack_frag_bf16(_p(w), rows, cols, _p(hi, torch.int16), _p(lo, torch.int16), _stream()))
    return hi, lo


def pack_frag_f32(w):
    """Fragment-major fp32 form of a float32 matrix [rows, cols] (include/fcl_hip.h fcl_pack_frag_f32): the exact-fp32 feat/prenet kernel's operand."""
    rows, cols = w.shape
    out = torch.empty(_lib.load().fcl_frag_bf16_elems(rows, cols), device=w.device, dtype=torch.float32)
    check(_lib.load().fcl_pack_frag_f32(_p(w), rows, cols, _p(out), _stream()))
    return out


class gemm_mode(object):
    """Context manager for the calling thread's contraction arithmetic (include/fcl_hip.h fcl_set_gemm_mode): `with ops.gemm_mode("bf16"):` runs the
    big-tile GEMMs inside on bf16-rounded operands (autocast); "f32" (default) is the fp32-equivalent bf16x3 split."""

    MODES = {"f32": _lib.GEMM_F32, "bf16": _lib.GEMM_BF16, None: _lib.GEMM_F32}

    def __init__(self, mode):
        if mode not in self.MODES:
            raise ValueError("gemm mode must be 'f32' or 'bf16', got %r" % (mode,))
        self.mode = self.MODES[mode]

    def __enter__(self):
        lib = _lib.load()
        self.prev = lib.fcl_get_gemm_mode()
        check(lib.fcl_set_gemm_mode(self.mode))
        return self

    def __exit__(self, *exc):
        check(_lib.load().fcl_set_gemm_mode(self.prev))
        return False


_PLANES_ON = None


def planes_enabled():
    """The pre-split (P32) operand path is on unless exact-fp32 arithmetic (FCL_PRECISION=0) or FCL_PLANES=0 is requested.  Read once per process,
    as the library reads its tunables (a training update asks ~180 times)."""
    global _PLANES_ON
    if _PLANES_ON is None:
        import os

        _PLANES_ON = os.environ.get("FCL_PRECISION", "1") != "0" and os.environ.get("FCL_PLANES", "1") != "0"
    return _PLANES_ON


def planes_empty(rows, cols, device):
    """Uninitialised P32 plane buffer of a [rows, cols] matrix (include/fcl_hip.h): int16 [rows, ceil(cols/32) * 64], 128-byte aligned rows."""
    t = torch.empty(max(rows, 1), (cols + 31) // 32 * 64, device=device, dtype=torch.int16)
    assert t.data_ptr() % 128 == 0
    return t


def pack_planes(x):
    """P32 planes of a float32 matrix [rows, cols] (plan-time weight packing; activations get theirs from their producers)."""
    rows, cols = x.shape
    assert x.stride(1) == 1
    out = planes_empty(rows, cols, x.device)
    check(_lib.load().fcl_pack_planes(x.data_ptr(), x.stride(0), rows, cols, _p(out, torch.int16), _stream()))
    return out


def concat_spk(hs, spk, t, want_f32=True, want_planes=False):
    """cat[hs, F.normalize(spk)[row // t]] over the padded [B, t] rows (..._sa.py:555-557): returns (fp32 [M, C + S] or None, planes or None)."""
    m, c = hs.shape
    s = spk.shape[1]
    assert hs.stride(1) == 1 and spk.is_contiguous() and spk.dtype == torch.float32 and m <= spk.shape[0] * t
    out = torch.empty(m, c + s, device=hs.device, dtype=torch.float32) if want_f32 else None
    outp = planes_empty(m, c + s, hs.device) if want_planes else None
    check(_lib.load().fcl_concat_spk_fwd(_p(hs), hs.stride(0), _p(spk), _p(out), _p(outp, torch.int16), m, c, s, t, _stream()))
    return out, outp


def add_vec(a, b):
    out = torch.empty_like(a)
    check(_lib.load().fcl_add_vec(_p(a), _p(b), _p(out), a.numel(), _stream()))
    return out


def embedding(ids, table, want_f32=True, want_planes=False):
    """Returns the fp32 rows, or (fp32 or None, P32 planes) when want_planes."""
    m = ids.numel()
    v, e = table.shape
    out = torch.empty(m, e, device=table.device, dtype=torch.float32) if want_f32 else None
    outp = planes_empty(m, e, table.device) if want_planes else None
    check(_lib.load().fcl_embedding_fwd(_p(ids, torch.int64), _p(table), _p(out), _p(outp, torch.int16), m, v, e, _stream()))
    return (out, outp) if want_planes else out


def linear(x, w, bias=None, act=ACT_NONE, out=None):
    m, k = x.shape
    n = w.shape[0]
    y = out if out is not None else torch.empty(m, n, device=x.device, dtype=torch.float32)
    check(_lib.load().fcl_linear_fwd(_p(x), k, _p(w), w.shape[1], _p(bias), _p(y), n, m, n, k, act, _stream()))
    return y


def conv1d(x, wp, bias, seg_lo, seg_hi, act=ACT_NONE, residual=None):
    m, cin = x.shape
    k, cout, cin2 = wp.shape
    assert cin2 == cin
    y = torch.empty(m, cout, device=x.device, dtype=torch.float32)
    check(_lib.load().fcl_conv1d_fwd(_p(x), _p(wp), _p(bias), _p(seg_lo, torch.int32), _p(seg_hi, torch.int32), _p(residual), _p(y),
                                     m, cin, cout, k, act, _stream()))
    return y


def conv1d_planes(xp, cv, seg_lo, seg_hi, act=ACT_NONE, residual=None, want_f32=False, want_planes=True, m_dev=None):
    """Conv1d on pre-split operands: xp = P32 planes of x [m, cin]; cv: a plan ConvPack with .wpp (planes of the packed taps).
    Returns (y fp32 or None, y planes or None).  m_dev: device int32 row count -- tiles at or beyond it are skipped (capacity buffers)."""
    m = xp.shape[0]
    ldxp = xp.shape[1] // 64
    y = torch.empty(m, cv.cout, device=xp.device, dtype=torch.float32) if want_f32 else None
    yp = planes_empty(m, cv.cout, xp.device) if want_planes else None
    check(_lib.load().fcl_conv1d_planes_rows_fwd(_p(xp, torch.int16), ldxp, _p(cv.wpp, torch.int16), _p(cv.bias), _p(seg_lo, torch.int32),
                                                 _p(seg_hi, torch.int32), _p(residual), _p(y), _p(yp, torch.int16), m, cv.cin, cv.cout, cv.k, act,
                                                 _p(m_dev, torch.int32), _stream()))
    return y, yp


def layernorm(x, gamma, beta, eps, want_y=True, lin_w=None, lin_b=None, pad_mask=None, keep=None, keep_scale=1.0, want_planes=False):
    """Returns (y, scalar); with want_planes (y or None, scalar, P32 planes of y)."""
    m, c = x.shape
    y = torch.empty_like(x) if want_y else None
    yp = planes_empty(m, c, x.device) if want_planes else None
    scalar = torch.empty(m, device=x.device, dtype=torch.float32) if lin_w is not None else None
    check(_lib.load().fcl_layernorm_fwd(_p(x), _p(gamma), _p(beta), eps, _p(y), _p(yp, torch.int16), _p(lin_w), _p(lin_b), _p(pad_mask, torch.uint8),
                                        _p(keep, torch.uint8), keep_scale, _p(scalar), m, c, _stream()))
    return (y, scalar, yp) if want_planes else (y, scalar)


def conv1d_planes_group(xp, x_group_stride, wpp, bias, seg_lo, seg_hi, m, cin, cout, k, groups, act=ACT_NONE, want_f32=True, want_planes=False):
    """G Conv1d's of one shape in ONE launch (include/fcl_hip.h fcl_conv1d_planes_group_fwd).  xp: P32 planes, group g at + g * x_group_stride uint16
    elements (0: every group reads the same input); wpp [G][k * Cout][Cin planes]; bias [G * Cout].  Returns (y [G * m, Cout] or None, planes or None),
    group-major."""
    ldxp = xp.shape[1] // 64
    y = torch.empty(groups * m, cout, device=xp.device, dtype=torch.float32) if want_f32 else None
    yp = planes_empty(groups * m, cout, xp.device) if want_planes else None
    check(_lib.load().fcl_conv1d_planes_group_fwd(_p(xp, torch.int16), ldxp, int(x_group_stride), _p(wpp, torch.int16), _p(bias), _p(seg_lo, torch.int32),
                                                  _p(seg_hi, torch.int32), _p(y), _p(yp, torch.int16), m, cin, cout, k, act, groups, _stream()))
    return y, yp


def layernorm_group(x, ldx, x_group_stride, gamma, beta, eps, m, c, groups, want_y=False, want_planes=False, lin_w=None, lin_b=None, pad_mask=None):
    """G LayerNorms of one shape in ONE launch (fcl_layernorm_group_fwd): row mi of group g at x + g * x_group_stride + mi * ldx; outputs group-major.
    Returns (y [G * m, c] or None, scalar [G * m] or None, planes or None)."""
    y = torch.empty(groups * m, c, device=x.device, dtype=torch.float32) if want_y else None
    yp = planes_empty(groups * m, c, x.device) if want_planes else None
    scalar = torch.empty(groups * m, device=x.device, dtype=torch.float32) if lin_w is not None else None
    check(_lib.load().fcl_layernorm_group_fwd(_p(x), ldx, int(x_group_stride), _p(gamma), _p(beta), eps, _p(y), _p(yp, torch.int16), _p(lin_w), _p(lin_b),
                                              _p(pad_mask, torch.uint8), _p(scalar), m, c, groups, _stream()))
    return y, scalar, yp


def duration_round(x, linear_domain=False, offset=1.0, pad_mask=None):
    out = torch.empty(x.numel(), device=x.device, dtype=torch.int64)
    check(_lib.load().fcl_duration_round_fwd(_p(x), _p(out, torch.int64), x.numel(), int(linear_domain), offset, _p(pad_mask, torch.uint8), _stream()))
    return out


def variance_embed_add(hs, p, e, wp, bp, we, be, seg_lo, seg_hi, want_embs=False):
    m, c = hs.shape
    k = wp.shape[-1]
    out = torch.empty_like(hs)
    pe = torch.empty_like(hs) if want_embs else None
    ee = torch.empty_like(hs) if want_embs else None
    check(_lib.load().fcl_variance_embed_add_fwd(_p(hs), _p(p), _p(e), _p(wp), _p(bp), _p(we), _p(be), _p(seg_lo, torch.int32),
                                                 _p(seg_hi, torch.int32), _p(out), _p(pe), _p(ee), m, c, k, _stream()))
    return out, pe, ee


def position_table(dur_i32, lmax):
    n = dur_i32.numel()
    pos = torch.empty(n, lmax, device=dur_i32.device, dtype=torch.float32)
    check(_lib.load().fcl_position_table_fwd(_p(dur_i32, torch.int32), _p(pos), n, lmax, _stream()))
    return pos


def gather_rows(src, idx_i32, want_f32=True, want_planes=False):
    """Returns the gathered fp32 rows, or (fp32 or None, P32 planes) when want_planes."""
    n, c = idx_i32.numel(), src.shape[1]
    dst = torch.empty(n, c, device=src.device, dtype=torch.float32) if want_f32 else None
    dstp = planes_empty(n, c, src.device) if want_planes else None
    check(_lib.load().fcl_gather_rows_fwd(_p(src), _p(idx_i32, torch.int32), _p(dst), _p(dstp, torch.int16), n, c, _stream()))
    return (dst, dstp) if want_planes else dst


_STATUS = {}


def status_word(device):
    """The per-device status word (int32 tensor used as a uint32; include/fcl_hip.h FCL_STATUS_*): kernels that can only detect a failure
    while they run (the cooperating-workgroup BiLSTM's bounded spin) OR a bit into it; check_status() raises on it."""
    key = str(torch.device(device))
    w = _STATUS.get(key)
    if w is None:
        w = _STATUS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def status_message(bits):
    msgs = []
    if bits & _lib.STATUS_GROUP_TIMEOUT:
        msgs.append("a cooperating-workgroup BiLSTM kernel timed out waiting for a group member (its outputs are partial); "
                    "it is single-stream only — run one at a time or set FCL_BILSTM_GROUP=0")
    if bits & _lib.STATUS_ZERO_DURATION:
        msgs.append("zero duration: ds_nonzeros.shape[0] != hs.shape[0] (the reference asserts on a non-padded phoneme of duration 0, "
                    "decoder_sa_kd.py:739); nothing was decoded")
    if bits & _lib.STATUS_LMAX_CAP:
        msgs.append("a duration exceeds the number of decoder steps that were launched (lmax_cap); nothing was decoded -- rerun with a larger cap")
    if bits & _lib.STATUS_FRAMES_CAP:
        msgs.append("the batch has more frames than the frame buffers hold (frames_cap); nothing was decoded -- rerun with a larger cap")
    if bits & _lib.STATUS_ROWS_CAP:
        msgs.append("a decoder step had more live rows than the host's bound for it (rows beyond the bound were not computed)")
    known = (_lib.STATUS_GROUP_TIMEOUT | _lib.STATUS_ZERO_DURATION | _lib.STATUS_LMAX_CAP | _lib.STATUS_FRAMES_CAP | _lib.STATUS_ROWS_CAP)
    if bits & ~known:
        msgs.append("unknown status bits 0x%x" % (bits & ~known))
    return "; ".join(msgs)


def check_status(device, reset=True):
    """Synchronising read of the device status word; raises FclError when a kernel reported a failure."""
    w = status_word(device)
    bits = int(w.item()) & 0xFFFFFFFF
    if bits:
        if reset:
            w.zero_()
        raise _lib.FclError("fcl-taco2_amd: device status 0x%x: %s" % (bits, status_message(bits)))


def bilstm(x, lens_i32, w_ih_f, w_hh_f, b_f, w_ih_r, w_hh_r, b_r, b, t, algo=0, status=None, x_p=None, w_ih_p=None, want_planes=False, row_maps=None):
    """status: device status word for algo 3 (default: the per-device word of status_word(); callers check it at their next sync point).
    x_p / w_ih_p = (forward, reverse): pre-split P32 operands of the input projection (x may then be None); want_planes: also return the
    output as P32 planes.  row_maps: a RowMapsRequest (row_maps_request) whose maps this call also builds (fcl_hip.h fcl_bilstm_fwd)."""
    c, dev = w_ih_f.shape[1], w_ih_f.device
    h = w_hh_f.shape[1]
    lib = _lib.load()
    nbytes = lib.fcl_bilstm_workspace_bytes(b, t, h)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    out = torch.empty(b * t, 2 * h, device=dev, dtype=torch.float32)
    outp = planes_empty(b * t, 2 * h, dev) if want_planes else None
    wf_p, wr_p = w_ih_p if (x_p is not None and w_ih_p is not None) else (None, None)
    check(lib.fcl_bilstm_fwd(_p(x), _p(lens_i32, torch.int32), _p(w_ih_f), _p(w_hh_f), _p(b_f), _p(w_ih_r), _p(w_hh_r), _p(b_r), _p(out),
                             _p(outp, torch.int16), _p(x_p if wf_p is not None else None, torch.int16), _p(wf_p, torch.int16), _p(wr_p, torch.int16),
                             b, t, c, h, algo, ws.data_ptr(), nbytes, _p(status if status is not None else status_word(dev), torch.int32),
                             C.byref(row_maps.struct) if row_maps is not None else None, _stream()))
    return (out, outp) if want_planes else out


def masked_l1_mse(a, b, row_valid, out_f64, b_log_offset=None):
    """Accumulate sum|a-b'|, sum(a-b')^2, count into out_f64[0:3] (device float64, caller-zeroed)."""
    m, c = a.shape
    assert b.shape == a.shape and out_f64.dtype == torch.float64 and out_f64.numel() >= 3
    check(_lib.load().fcl_masked_l1_mse_fwd(_p(a), c, _p(b), c, _p(row_valid, torch.uint8), m, c, int(b_log_offset is not None),
                                            float(b_log_offset or 0.0), out_f64.data_ptr(), _stream()))


def u32_add(word_i32, v=1):
    """*word += v on the current stream (word: 1-element int32 device tensor used as a uint32)."""
    check(_lib.load().fcl_u32_add(_p(word_i32, torch.int32), v, _stream()))


def host_device_ptr(pinned):
    """Device view (an address) of a pinned host tensor, for kernels that read / write host memory directly (feed_copy)."""
    p = _lib.load().fcl_host_device_ptr(pinned.data_ptr())
    if not p:
        msg = _lib.load().fcl_last_error()
        raise _lib.FclError("fcl-taco2_amd: %s" % (msg.decode("utf-8", "replace") if msg else "host_device_ptr failed"))
    return p


def feed_copy(dst_u8, src_devptr, nbytes, seq_dev_i32, seq_host_devptr, bump_i32=None):
    """One kernel on the current stream: dst <- the pinned host block, ++*seq_dev -> *seq_host, ++*bump (include/fcl_hip.h fcl_feed_copy)."""
    check(_lib.load().fcl_feed_copy(_p(dst_u8, torch.uint8), src_devptr, nbytes, _p(seq_dev_i32, torch.int32), seq_host_devptr, _p(bump_i32, torch.int32),
                                    _stream()))


class RowMapsRequest(object):
    """Output tensors (`maps`: the dict row_maps_build returns) + the fcl_row_maps_t that describes them, for a later launch to fill."""
    __slots__ = ("maps", "struct", "_keep")


def row_maps_build(n, b, lmax_cap, frames_cap, dur_i64=None, dur_i32=None, row_src=None, utt_row0=None, t_max=0, pad=None, want_order=False,
                   status=None):
    """Device-built maps, built now: see row_maps_request."""
    req = row_maps_request(n, b, lmax_cap, frames_cap, dur_i64, dur_i32, row_src, utt_row0, t_max, pad, want_order, status)
    check(_lib.load().fcl_row_maps_build(C.byref(req.struct), _stream()))
    return req.maps


def row_maps_request(n, b, lmax_cap, frames_cap, dur_i64=None, dur_i32=None, row_src=None, utt_row0=None, t_max=0, pad=None, want_order=False,
                     status=None):
    """Device-built row / frame maps (include/fcl_hip.h fcl_row_maps_build; bit-identical to engine.build_row_maps on the real rows).
    Row universe: compact rows (row_src [N] / utt_row0 [B + 1] int32 device tensors that depend on the phoneme counts alone) or, with
    row_src=None, the padded [B, t_max] layout (n = B * t_max; `pad` marks the padding rows).  Durations: int64 (fcl_duration_round_fwd's
    output, read through row_src) or int32 per row.  Returns a dict of int32 device tensors: src_rows, dur, frame_off [n]; live_rows
    [lmax_cap + 1]; utt_frame0 [B + 1]; frame_lo / frame_hi [frames_cap]; totals [4] = (frames, max duration, zero-duration rows, 0); order [n] on
    request.  Violations (a zero duration, a duration > lmax_cap, more than frames_cap frames) set FCL_STATUS_* bits in the device status word
    and zero live_rows.  Returns a RowMapsRequest: row_maps_build runs it at once, bilstm(row_maps=...) as part of the encoder's recurrence."""
    ref = dur_i64 if dur_i64 is not None else dur_i32
    dev = ref.device
    # one allocation for all the maps (a graph-private pool then holds one block; slices are 16-byte aligned)
    sizes = dict(src_rows=n, dur=n, frame_off=n, live_rows=lmax_cap + 1, utt_frame0=b + 1, frame_lo=frames_cap, frame_hi=frames_cap, totals=4,
                 scratch=2 * n)
    if want_order:
        sizes["order"] = n
    offs, tot = {}, 0
    for k, v in sizes.items():
        offs[k] = tot
        tot += (v + 3) // 4 * 4
    blk = torch.empty(tot, dtype=torch.int32, device=dev)
    out = {k: blk[offs[k] : offs[k] + v] for k, v in sizes.items()}
    st = status if status is not None else status_word(dev)
    a = _lib.RowMaps(b=b, n=n, lmax_cap=lmax_cap, frames_cap=frames_cap, t_max=t_max, row_src=_p(row_src, torch.int32), utt_row0=_p(utt_row0, torch.int32),
                     pad=_p(pad, torch.uint8), dur_i64=_p(dur_i64, torch.int64), dur_i32=_p(dur_i32, torch.int32), src_rows=_p(out["src_rows"], torch.int32),
                     dur_sorted=_p(out["dur"], torch.int32), frame_off=_p(out["frame_off"], torch.int32),
                     order=_p(out.get("order"), torch.int32), live_rows=_p(out["live_rows"], torch.int32), utt_frame0=_p(out["utt_frame0"], torch.int32),
                     frame_lo=_p(out["frame_lo"], torch.int32), frame_hi=_p(out["frame_hi"], torch.int32), totals=_p(out["totals"], torch.int32),
                     status=st.data_ptr(), scratch=_p(out["scratch"], torch.int32))
    req = RowMapsRequest()
    req.maps, req.struct, req._keep = out, a, (blk, st, row_src, utt_row0, pad, dur_i64, dur_i32)
    return req


def decoder_loop(dw, att_c, dur_i32, live_rows, frame_off_i32, n_frames, teacher_ys=None, dropout_mode=DROP_NONE,
                 prenet_keep=None, seed=0, want_taps=False, seed_dev=None, zero_init=False, att_c_p=None, want_before_p=False, live_rows_dev=None,
                 status=None, tail_from=0):
    """dw: plan.DecoderPack (holds the ctypes DecoderWeights + the tensors it points to).
    live_rows: host numpy int32 [Lmax].  Returns before [F, odim] (+ taps).  att_c_p: P32 planes of att_c (att_c may then be None);
    want_before_p: also return `before` as P32 planes (appended to the result).
    live_rows_dev: device int32 [Lmax + 1] (row_maps_build): the loop is then driven by the DEVICE counts, `live_rows` are per-step upper bounds
    (grid sizes, kernel selection) and n_frames is the capacity of the frame buffers (rows past the real total are never written; the
    postnet's segment bounds of such rows are empty, so nothing reads them either).
    tail_from: fcl_decoder_io_t.tail_from (steps from there on: one launch of the persistent row-tile kernel for the rows still live)."""
    lib = _lib.load()
    dev = att_c.device if att_c is not None else att_c_p.device
    n = att_c.shape[0] if att_c is not None else att_c_p.shape[0]
    lmax = int(live_rows.shape[0])
    nbytes = lib.fcl_decoder_loop_workspace_bytes(C.byref(dw.struct), n)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    alloc = torch.zeros if zero_init else torch.empty  # zero_init: frames no (row, t) maps to stay 0 (padded [B, Lmax] layout)
    before = alloc(n_frames, dw.struct.odim, device=dev, dtype=torch.float32)
    before_p = planes_empty(n_frames, dw.struct.odim, dev) if want_before_p else None
    if live_rows_dev is not None and status is None:
        status = status_word(dev)
    taps = None
    if want_taps:
        taps = (alloc(n_frames, dw.struct.p, device=dev, dtype=torch.float32),
                alloc(n_frames, dw.struct.u, device=dev, dtype=torch.float32),
                alloc(n_frames, dw.struct.u, device=dev, dtype=torch.float32))
    io = _lib.DecoderIO(
        n=n, lmax=lmax, att_c=_p(att_c), dur=_p(dur_i32, torch.int32), live_rows_host=live_rows.ctypes.data,
        frame_off=_p(frame_off_i32, torch.int32), teacher_ys=_p(teacher_ys), dropout_mode=dropout_mode,
        prenet_keep=_p(prenet_keep, torch.uint8), seed=seed & 0xFFFFFFFF, seed_dev=_p(seed_dev, torch.int32), before=_p(before),
        tap_prenet=_p(taps[0]) if taps else None, tap_lstm0=_p(taps[1]) if taps else None, tap_lstm1=_p(taps[2]) if taps else None,
        workspace=ws.data_ptr(), workspace_bytes=nbytes, att_c_p=_p(att_c_p, torch.int16), before_p=_p(before_p, torch.int16),
        live_rows=_p(live_rows_dev, torch.int32), status=None if live_rows_dev is None else status.data_ptr(), tail_from=int(tail_from))
    check(lib.fcl_decoder_loop_fwd(C.byref(dw.struct), C.byref(io), _stream()))
    res = (before, taps) if want_taps else before
    if want_before_p:
        return (res + (before_p,)) if want_taps else (res, before_p)
    return res


# ---- gradient primitives (include/fcl_hip.h "H13") -------------------------------------------------------------------
def gemm_tn(a, b, out, shift=0, seg_lo=None, seg_hi=None):
    """out[n, k] += sum_m a[m, n] * b[m + shift, k]."""
    m, n = a.shape
    k = b.shape[1]
    assert b.shape[0] == m and out.shape == (n, k) and out.stride(1) == 1 and out.dtype == torch.float32 and out.is_cuda
    check(_lib.load().fcl_gemm_tn_fwd(_p(a), n, _p(b), k, out.data_ptr(), out.stride(0), m, n, k, shift, _p(seg_lo, torch.int32), _p(seg_hi, torch.int32), _stream()))
    return out


def pack_planes_t(x, ntaps=1, shift0=0, seg_lo=None, seg_hi=None):
    """Transposed P32 planes (include/fcl_hip.h fcl_pack_planes_t) of x [rows, cols]: int16 [ntaps * cols, ceil(rows/32) * 64]."""
    rows, cols = x.shape
    assert x.stride(1) == 1 and rows > 0
    out = torch.empty(ntaps * cols, (rows + 31) // 32 * 64, device=x.device, dtype=torch.int16)
    assert out.data_ptr() % 128 == 0
    check(_lib.load().fcl_pack_planes_t(x.data_ptr(), x.stride(0), rows, cols, ntaps, shift0, _p(seg_lo, torch.int32), _p(seg_hi, torch.int32),
                                        _p(out, torch.int16), _stream()))
    return out


def gemm_tn_planes(ap_t, bp_t, out, m):
    """out[n, k] += sum_m a[m, n] * b[m, k] from transposed planes; out [n, k] (row stride free) or tap-major [ntaps, n, k / ntaps] (contiguous)."""
    n = ap_t.shape[0]
    k = bp_t.shape[0]
    if out.dim() == 3:
        ntaps, n2, kk = out.shape
        assert n2 == n and ntaps * kk == k and out.is_contiguous()
        check(_lib.load().fcl_gemm_tn_planes(_p(ap_t, torch.int16), _p(bp_t, torch.int16), out.data_ptr(), kk, m, n, k, kk, n * kk, _stream()))
    else:
        assert out.shape == (n, k) and out.stride(1) == 1
        check(_lib.load().fcl_gemm_tn_planes(_p(ap_t, torch.int16), _p(bp_t, torch.int16), out.data_ptr(), out.stride(0), m, n, k, 0, 0, _stream()))
    return out


def gemm_tn_taps(a, b, out, shift0, seg_lo=None, seg_hi=None):
    """out[j, n, k] += sum_m a[m, n] * b[m + shift0 + j, k] for every tap j of out [ntaps, N, K] (contiguous)."""
    m, n = a.shape
    ntaps, n2, k = out.shape
    assert b.shape == (m, k) and n2 == n and out.is_contiguous()
    check(_lib.load().fcl_gemm_tn_taps_fwd(_p(a), n, _p(b), k, _p(out), k, m, n, k, shift0, ntaps, n * k, _p(seg_lo, torch.int32), _p(seg_hi, torch.int32),
                                           _stream()))
    return out


def l1_mse_loss_grad(a, b, row_valid, count, w_l1, w_mse, sums_f64, da=None, b_log_offset=None, want_planes=False):
    """Loss sums (into sums_f64[0:3]) and gradient in one pass; falls back to the two separate kernels when C % 4 != 0.
    want_planes (C % 32 == 0): returns (da, P32 planes of da)."""
    if a.dim() == 1:
        a, b = a.reshape(-1, 1), b.reshape(-1, 1)
    m, c = a.shape
    if c % 4:
        assert not want_planes
        masked_l1_mse(a, b, row_valid, sums_f64, b_log_offset)
        return l1_mse_grad(a, b, row_valid, count, w_l1, w_mse, da=da, b_log_offset=b_log_offset)
    acc = da is not None
    if da is None:
        da = torch.empty_like(a)
    dap = planes_empty(m, c, a.device) if want_planes else None
    check(_lib.load().fcl_l1_mse_loss_grad(_p(a), _p(b), _p(row_valid, torch.uint8), m, c, int(b_log_offset is not None), float(b_log_offset or 0.0),
                                           w_l1, w_mse, float(count), _p(da), int(acc), sums_f64.data_ptr(), _p(dap, torch.int16), _stream()))
    return (da, dap) if want_planes else da


def colsum(x, out, y=None, gamma=None, beta=None, mode=0, out_x=None):
    """out_x: also the plain column sums of x, from the same pass."""
    m, c = x.shape
    check(_lib.load().fcl_colsum2_fwd(_p(x), _p(y), _p(gamma), _p(beta), _p(out), _p(out_x), m, c, mode, _stream()))
    return out


def conv1d_in1_dw(dy, x, dw, db=None, seg_lo=None, seg_hi=None):
    """dw [C, 1, k] += sum_m dy[m, c] x[m + j - pad] (inside each row's utterance), db [C] += colsum(dy): Conv1d(1 -> C, k) weight / bias gradients."""
    m, c = dy.shape
    k = dw.shape[-1]
    assert dy.stride(1) == 1 and x.numel() == m and x.is_contiguous() and dw.is_contiguous() and dw.numel() == c * k
    check(_lib.load().fcl_conv1d_in1_dw(_p(dy), dy.stride(0), _p(x), _p(seg_lo, torch.int32), _p(seg_hi, torch.int32), _p(dw), _p(db), m, c, k, _stream()))
    return dw


def act_bwd(dy, y, act, keep=None, keep_scale=1.0, want_planes=False):
    """dz = dy * act'(y) [* keep * keep_scale]; want_planes (2-D, width % 32 == 0): returns (dz, P32 planes of dz)."""
    dz = torch.empty_like(dy)
    dzp = planes_empty(dy.shape[0], dy.shape[1], dy.device) if want_planes else None
    check(_lib.load().fcl_act_bwd(_p(dy), _p(y), _p(keep, torch.uint8), keep_scale, _p(dz), _p(dzp, torch.int16), dy.shape[-1] if want_planes else 0,
                                  dy.numel(), act, _stream()))
    return (dz, dzp) if want_planes else dz


def l1_mse_grad(a, b, row_valid, count, w_l1, w_mse, da=None, b_log_offset=None):
    if a.dim() == 1:
        a, b = a.reshape(-1, 1), b.reshape(-1, 1)
    m, c = a.shape
    acc = da is not None
    if da is None:
        da = torch.empty_like(a)
    check(_lib.load().fcl_l1_mse_grad(_p(a), _p(b), _p(row_valid, torch.uint8), m, c, int(b_log_offset is not None), float(b_log_offset or 0.0),
                                      w_l1, w_mse, float(count), _p(da), int(acc), _stream()))
    return da


def layernorm_bwd(x, gamma, beta, eps, dgamma, dbeta, dy=None, lin_w=None, ds=None, pad_mask=None, dlin_w=None, dlin_b=None, keep=None,
                  keep_scale=1.0):
    m, c = x.shape
    dx = torch.empty_like(x)
    check(_lib.load().fcl_layernorm_bwd(_p(x), _p(gamma), _p(beta), eps, _p(dy), _p(lin_w), _p(ds), _p(pad_mask, torch.uint8), _p(keep, torch.uint8),
                                        keep_scale, _p(dx), _p(dgamma), _p(dbeta), _p(dlin_w), _p(dlin_b), m, c, _stream()))
    return dx


_BN_WS = {}  # (device, stream) -> zero workspace of fcl_bn_stats_ws_fwd (left zero by every call; calls on one stream are ordered)


def bn_stats(z, eps, momentum=0.1, running_mean=None, running_var=None):
    """Train-mode BatchNorm statistics over the rows of z: returns (mean, invstd); updates the running buffers in place."""
    m, c = z.shape
    mean, invstd = torch.empty(c, device=z.device), torch.empty(c, device=z.device)
    key = (z.device, _stream())
    ws = _BN_WS.get(key)
    need = 2 * c + (c + 63) // 64
    if ws is None or ws.numel() < need:
        ws = _BN_WS[key] = torch.zeros(max(need, 4096), device=z.device, dtype=torch.float64)
    check(_lib.load().fcl_bn_stats_ws_fwd(_p(z), m, c, eps, momentum, _p(mean), _p(invstd), _p(running_mean), _p(running_var), ws.data_ptr(), _stream()))
    return mean, invstd


def conv1d_planes_bn(xp, wpp, seg_lo, seg_hi, cin, cout, k, eps, momentum=0.1, running_mean=None, running_var=None):
    """fcl_conv1d_planes_bn_fwd: Conv1d (no bias) on pre-split operands + the train-mode BatchNorm statistics of its output from the GEMM's epilogue.  wpp = planes of
    the tap-major packed weight [k, cout, cin].  Returns (z, mean, invstd); updates the running buffers in place."""
    m = xp.shape[0]
    z = torch.empty(m, cout, device=xp.device, dtype=torch.float32)
    mean, invstd = torch.empty(cout, device=xp.device), torch.empty(cout, device=xp.device)
    key = (xp.device, _stream())
    ws = _BN_WS.get(key)
    need = 2 * cout + (cout + 63) // 64
    if ws is None or ws.numel() < need:
        ws = _BN_WS[key] = torch.zeros(max(need, 4096), device=xp.device, dtype=torch.float64)
    check(_lib.load().fcl_conv1d_planes_bn_fwd(_p(xp, torch.int16), xp.shape[1] // 64, _p(wpp, torch.int16), _p(seg_lo, torch.int32), _p(seg_hi, torch.int32), _p(z), m,
                                               cin, cout, k, eps, momentum, _p(mean), _p(invstd), _p(running_mean), _p(running_var), ws.data_ptr(), _stream()))
    return z, mean, invstd


def bn_act(z, mean, invstd, gamma, beta, act, keep=None, keep_scale=1.0, want_planes=False):
    """Returns (y_act, y_drop); y_drop is y_act when there is no keep mask.  want_planes: (+ P32 planes of y_drop)."""
    m, c = z.shape
    y_act = torch.empty_like(z)
    y_drop = torch.empty_like(z) if keep is not None else None
    yp = planes_empty(m, c, z.device) if want_planes else None
    check(_lib.load().fcl_bn_act_fwd(_p(z), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(keep, torch.uint8), keep_scale, _p(y_act), _p(y_drop),
                                     _p(yp, torch.int16), m, c, act, _stream()))
    res = (y_act, (y_drop if keep is not None else y_act))
    return res + (yp,) if want_planes else res


def bn_bwd(dy, z, mean, invstd, gamma, dbeta, dgamma, want_planes=False, acc=None):
    """acc = (g_beta, g_gamma): the parameters' gradient accumulators, += this batch's dbeta / dgamma in the same launch."""
    m, c = z.shape
    dz = torch.empty_like(z)
    dzp = planes_empty(m, c, z.device) if want_planes else None
    check(_lib.load().fcl_bn_bwd(_p(dy), _p(z), _p(mean), _p(invstd), _p(gamma), _p(dbeta), _p(dgamma), _p(dz), _p(dzp, torch.int16), m, c,
                                  _p(acc[0]) if acc else None, _p(acc[1]) if acc else None, _stream()))
    return (dz, dzp) if want_planes else dz


def scale_(x, alpha):
    check(_lib.load().fcl_scale(_p(x), x.numel(), float(alpha), _stream()))
    return x


def linear_planes_mse(xp, wpp, target, row_valid, count, sums, m, n, k, want_f32=True, want_planes=True):
    """fcl_linear_planes_mse_fwd: the projection of one KD term, its masked MSE against `target` [m, n] and the gradient 2 (x W^T - target) / count in one
    launch; sums (float64 [3], device) += sum |d|, sum d^2, element count.  Returns (grad fp32 or None, grad planes or None)."""
    g = torch.empty(m, n, device=target.device, dtype=torch.float32) if want_f32 else None
    gp = planes_empty(m, n, target.device) if want_planes else None
    check(_lib.load().fcl_linear_planes_mse_fwd(_p(xp, torch.int16), (k + 31) // 32, _p(wpp, torch.int16), _p(target), target.stride(0), _p(row_valid, torch.uint8),
                                                float(count), _p(g), n, _p(gp, torch.int16), sums.data_ptr(), m, n, k, _stream()))
    return g, gp


def bernoulli_u8(shape, p_one, seed, device, seed_dev=None):
    out = torch.empty(shape, device=device, dtype=torch.uint8)
    check(_lib.load().fcl_bernoulli_u8(_p(out, torch.uint8), out.numel(), float(p_one), seed & 0xFFFFFFFF, _p(seed_dev, torch.int32), _stream()))
    return out


def bernoulli_batch(sites, device):
    """sites: [(shape, p_one, seed)] -> list of uint8 masks, byte for byte what bernoulli_u8 would draw for each, in as few launches as the
    per-launch site limit allows (one, for the groups a training forward asks for) and one allocation."""
    if not sites:
        return []
    sizes = []
    for shape, _, _ in sites:
        n = 1
        for v in shape:
            n *= int(v)
        sizes.append(n)
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 255) // 256 * 256
    buf = torch.empty(tot, device=device, dtype=torch.uint8)
    outs = [buf[o : o + n].view(tuple(shape)) for o, n, (shape, _, _) in zip(offs, sizes, sites)]
    lib = _lib.load()
    for k0 in range(0, len(sites), _lib.BERNOULLI_MAX_SITES):
        grp = sites[k0 : k0 + _lib.BERNOULLI_MAX_SITES]
        arr = (_lib.BernoulliSite * len(grp))()
        for j, (shape, p_one, seed) in enumerate(grp):
            arr[j].out, arr[j].n, arr[j].p_one, arr[j].seed = outs[k0 + j].data_ptr(), sizes[k0 + j], float(p_one), seed & 0xFFFFFFFF
        check(lib.fcl_bernoulli_batch(arr, len(grp), _stream()))
    return outs


def lstm_cell_bwd(gates, c_old, c_new, dh_out, dc_out, zoneout, zone_keep_h=None, zone_keep_c=None, row_len=None, step=0, out=None, dh_out2=None):
    m, u = c_old.shape
    if out is not None:
        dgates, dh_old, dc_old = out
    else:
        dgates = torch.empty(m, 4 * u, device=gates.device, dtype=torch.float32)
        dh_old, dc_old = torch.empty_like(c_old), torch.empty_like(c_old)
    check(_lib.load().fcl_lstm_cell_bwd(_p(gates), _p(c_old), _p(c_new), _p(dh_out), None if dh_out2 is None else dh_out2.data_ptr(),
                                        0 if dh_out2 is None else dh_out2.stride(0), _p(dc_out), zoneout, _p(zone_keep_h, torch.uint8),
                                        _p(zone_keep_c, torch.uint8), _p(row_len, torch.int32), step, _p(dgates), _p(dh_old), _p(dc_old), None, m, u, _stream()))
    return dgates, dh_old, dc_old


def scatter_add_rows(src, idx_i64, dst, skip=-1):
    m, c = src.shape
    check(_lib.load().fcl_scatter_add_rows(_p(src), _p(idx_i64, torch.int64), _p(dst), m, c, skip, _stream()))
    return dst


def add2d(dst, src, alpha=1.0, row_valid=None):
    """dst += alpha * src on valid rows.  dst / src may be column-block VIEWS of row-major matrices (stride(1) == 1)."""
    rows, cols = src.shape
    assert dst.shape == src.shape and dst.stride(1) == 1 and src.stride(1) == 1 and dst.dtype == src.dtype == torch.float32 and dst.is_cuda
    check(_lib.load().fcl_add2d(dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0), rows, cols, alpha, _p(row_valid, torch.uint8), _stream()))
    return dst


def transpose2d(src):
    rows, cols = src.shape
    dst = torch.empty(cols, rows, device=src.device, dtype=torch.float32)
    check(_lib.load().fcl_transpose2d(_p(src), _p(dst), rows, cols, _stream()))
    return dst


class DerivedForms(object):
    """Operand forms that are functions of the PARAMETERS alone (packed Conv1d taps, transposes for the input-gradient GEMMs, column blocks of
    the decoder's LSTM weights, bias sums; fp32 and / or P32 planes), all refreshed by ONE fcl_derive_batch launch per optimizer update instead
    of one small launch each (a KD update used to spend ~150 launches on them).  A form is registered the first time it is asked for (and
    computed on the spot by a one-entry table); from the next refresh() on it is part of the batched table.  Outputs keep their addresses for
    the life of the object, so the table is uploaded only when a new form appears."""

    def __init__(self, device):
        self.device = device
        self.forms = {}  # key -> dict(sig, desc fields, out, outp, keep-alive sources)
        self.order = []
        self.table = None  # (device uint8 tensor, n, total_blocks) or None = rebuild
        self.stamp = None

    def _table(self, forms):
        arr = (_lib.Derive * len(forms))()
        first = 0
        for i, f in enumerate(forms):
            d = arr[i]
            d.src, d.src2, d.dst, d.dst_p = f["src_ptr"], f["src2_ptr"], f["dst_ptr"], f["dstp_ptr"]
            d.a, d.b, d.c, d.sa, d.sb, d.sc = f["geom"]
            d.first_block = first
            first += f["blocks"]
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        return host.to(self.device), len(forms), first

    def _run(self, table):
        dev, n, blocks = table
        check(_lib.load().fcl_derive_batch(dev.data_ptr(), n, blocks, _stream()))

    def refresh(self, stamp):
        """Recompute every registered form (call when the parameters changed: once per update)."""
        if self.order:
            if self.table is None:
                self.table = self._table([self.forms[k] for k in self.order])
            self._run(self.table)
        self.stamp = stamp

    def get(self, stamp, key, src, geom, base=0, src2=None, f32=True, planes=False):
        """geom = (a, b, c, sa, sb, sc) over `src` (a contiguous fp32 parameter tensor) starting at element `base`; see fcl_derive_batch.
        Returns (fp32 [a*b, c] or None, P32 planes or None)."""
        if stamp != self.stamp:
            self.refresh(stamp)
        a, b, c = geom[:3]
        sig = (src.data_ptr() + 4 * base, None if src2 is None else src2.data_ptr() + 4 * base, tuple(geom), bool(f32), bool(planes))
        f = self.forms.get(key)
        if f is not None and f["sig"] == sig:
            return f["out"], f["outp"]
        out = torch.empty(a * b, c, device=self.device, dtype=torch.float32) if f32 else None
        outp = planes_empty(a * b, c, self.device) if planes else None
        blocks = _lib.load().fcl_derive_blocks(a, b, c)
        if blocks <= 0:
            raise _lib.FclError("fcl-taco2_amd: bad derived-form geometry %r" % (geom,))
        f = dict(sig=sig, src_ptr=_p(src) + 4 * base, src2_ptr=None if src2 is None else _p(src2) + 4 * base, dst_ptr=_p(out), dstp_ptr=_p(outp, torch.int16),
                 geom=tuple(int(v) for v in geom), blocks=blocks, out=out, outp=outp, keep=(src, src2))
        if key not in self.forms:
            self.order.append(key)
        self.forms[key] = f
        self.table = None  # rebuilt (and uploaded) at the next refresh
        one = self._table([f])
        self._run(one)
        f["first_table"] = one[0]  # the one-entry table must outlive its launch
        return out, outp

    def clear(self):
        self.forms.clear()
        del self.order[:]
        self.table, self.stamp = None, None


def sumsq_accum(x, out_f64):
    check(_lib.load().fcl_sumsq_accum(_p(x), x.numel(), out_f64.data_ptr(), _stream()))


def adam_step(p, g, m, v, gradnorm_sq_f64, max_norm, lr, beta1, beta2, eps, step_i32, status=None, weight_decay=0.0):
    """step_i32: device int32 count of APPLIED updates (advanced on the device only when this update is applied); status: device status word;
    weight_decay: torch.optim.Adam's L2 term (tts.py:397-399), added to the clipped gradient inside the step."""
    check(_lib.load().fcl_adam_step_wd(_p(p), _p(g), _p(m), _p(v), p.numel(), gradnorm_sq_f64.data_ptr(), max_norm, lr, beta1, beta2, eps,
                                       float(weight_decay), _p(step_i32, torch.int32), _p(status, torch.int32), _stream()))


def act_fwd(x, act, keep=None, keep_scale=1.0, want_planes=False):
    """y = act(x) [* keep * keep_scale]; want_planes (2-D x, width % 32 == 0): returns (y, P32 planes of y)."""
    y = torch.empty_like(x)
    yp = planes_empty(x.shape[0], x.shape[1], x.device) if want_planes else None
    check(_lib.load().fcl_act_fwd(_p(x), _p(keep, torch.uint8), keep_scale, _p(y), _p(yp, torch.int16), x.shape[-1] if want_planes else 0, x.numel(), act,
                                  _stream()))
    return (y, yp) if want_planes else y


def linear_planes(xp, wpp, n, k, bias=None, act=ACT_NONE, want_f32=True, want_planes=False):
    """Linear on pre-split operands: xp = P32 planes of x [m, k], wpp = P32 planes of the weight [n, k].  Returns (y or None, planes or None)."""
    m, ldxp = xp.shape[0], xp.shape[1] // 64
    y = torch.empty(m, n, device=xp.device, dtype=torch.float32) if want_f32 else None
    yp = planes_empty(m, n, xp.device) if want_planes else None
    check(_lib.load().fcl_linear_planes_fwd(_p(xp, torch.int16), ldxp, _p(wpp, torch.int16), _p(bias), _p(y), n, _p(yp, torch.int16), m, n, k, act, _stream()))
    return y, yp


def unpack_conv1d_grad(dwp, dw, scale=None):
    k, cout, cin = dwp.shape
    check(_lib.load().fcl_unpack_conv1d_grad(_p(dwp), _p(scale), _p(dw), cout, cin, k, _stream()))
    return dw


def lstm_step(terms, M, U, h_in, h_out, c, G=None, g_row_mul=1, g_row_add=0, bias=None, rank1_w=None, dur=None, step=0, zoneout=0.0,
              zone_keep_h=None, zone_keep_c=None, row_len=None, out2=None, out2_row_mul=0, out2_row_add=0, ld2=0, out2_col_off=0, save=None):
    """One LSTMCell(+zoneout) step through fcl_lstm_step_fwd.  terms: [(A, W, K)] with A [M, >=K] and W [4U, K] contiguous rows.
    save: optional (gates [M,4U], c_new, c_old, h_old) tensors for the backward pass."""
    a = _lib.LstmStep()
    a.nterms = len(terms)
    for i, (A, W, K) in enumerate(terms):
        a.term[i] = _lib.GemmTerm(A.data_ptr(), W.data_ptr(), A.stride(0), W.stride(0), K, 0, None, None, None, None, 0, 0)
    a.M, a.U = M, U
    a.G, a.g_row_mul, a.g_row_add = _p(G), g_row_mul, g_row_add
    a.bias, a.rank1_w, a.dur, a.step = _p(bias), _p(rank1_w), _p(dur, torch.int32), step
    a.h_in, a.h_out, a.c, a.zoneout = h_in.data_ptr(), h_out.data_ptr(), c.data_ptr(), zoneout
    a.zone_keep_h, a.zone_keep_c, a.row_len = _p(zone_keep_h, torch.uint8), _p(zone_keep_c, torch.uint8), _p(row_len, torch.int32)
    if out2 is not None:
        a.out2, a.out2_row_mul, a.out2_row_add, a.ld2, a.out2_col_off = out2.data_ptr(), out2_row_mul, out2_row_add, ld2, out2_col_off
    if save is not None:
        a.save_gates, a.save_c_new, a.save_c_old, a.save_h_old = [t.data_ptr() for t in save]
    check(_lib.load().fcl_lstm_step_fwd(C.byref(a), _stream()))


# ---- the training step's time loops: one call enqueues a whole recurrence (csrc/train_loops.hip) ---------------------------------------------
def _ptrs(ctype_array, tensors):
    for i, t in enumerate(tensors):
        ctype_array[i] = _p(t)


def decoder_train_fwd(live_rows, p1d, g0, w0_pre, w0_hh, w0_pos, dur_i32, w1_ih, w1_hh, b1, zoneout, zk, s0, s1, h0_all, h1_all, planes=None):
    """live_rows: host int32 [lmax]; zk: None or [[h0, c0], [h1, c1]] uint8 [F, U]; s0/s1: (gates, c_new, c_old, h_old) outputs.
    planes: optional (p1d_p, w0_pre_p, w0_hh_p, w1_ih_p, w1_hh_p) P32 planes -> the big steps run on the LDS-DMA kernels."""
    lib = _lib.load()
    n, u = g0.shape[0], h0_all.shape[1]
    a = _lib.DecoderTrain(n=n, lmax=int(live_rows.shape[0]), u=u, p=p1d.shape[1], live_rows_host=live_rows.ctypes.data, p1d=_p(p1d), g0=_p(g0),
                          w0_pre=_p(w0_pre), w0_hh=_p(w0_hh), w0_pos=_p(w0_pos), dur=_p(dur_i32, torch.int32), w1_ih=_p(w1_ih), w1_hh=_p(w1_hh), b1=_p(b1),
                          zoneout=zoneout, h0_all=_p(h0_all), h1_all=_p(h1_all))
    if zk is not None:
        a.zk_h0, a.zk_c0, a.zk_h1, a.zk_c1 = [_p(t, torch.uint8) for t in (zk[0][0], zk[0][1], zk[1][0], zk[1][1])]
    _ptrs(a.s0, s0)
    _ptrs(a.s1, s1)
    if planes is not None:
        a.p1d_p, a.w0_pre_p, a.w0_hh_p, a.w1_ih_p, a.w1_hh_p = [_p(t, torch.int16) for t in planes]
    nbytes = lib.fcl_decoder_train_workspace_bytes(n, u)
    ws = torch.empty(nbytes, device=g0.device, dtype=torch.uint8)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(lib.fcl_decoder_train_fwd(C.byref(a), _stream()))


def decoder_bptt(live_rows, n, s0, s1, zoneout, zk, dh1_all, dh0_all, w1_ih_t, w1_hh_t, w0_hh_t, dg0_all, dg1_all, planes=None, w1_cat=None):
    """planes: optional (w1_ih_t_p, w1_hh_t_p, w0_hh_t_p, dg0_all_p, dg1_all_p): P32 planes of the transposed weights (in) and of the gate
    gradients of every cell (out) -> the recurrence's GEMMs of the steps with enough live rows run on the pre-split-operand kernels.
    w1_cat: optional ([W1_hh^T ; W1_ih^T] as one [2U, 4U] tensor, its planes or None): four launches per step instead of five."""
    lib = _lib.load()
    u = dh1_all.shape[1]
    a = _lib.DecoderBptt(n=n, lmax=int(live_rows.shape[0]), u=u, live_rows_host=live_rows.ctypes.data, zoneout=zoneout, dh1_all=_p(dh1_all),
                         dh0_all=_p(dh0_all), w1_ih_t=_p(w1_ih_t), w1_hh_t=_p(w1_hh_t), w0_hh_t=_p(w0_hh_t), dg0_all=_p(dg0_all), dg1_all=_p(dg1_all))
    if zk is not None:
        a.zk_h0, a.zk_c0, a.zk_h1, a.zk_c1 = [_p(t, torch.uint8) for t in (zk[0][0], zk[0][1], zk[1][0], zk[1][1])]
    _ptrs(a.s0, s0[:3])
    _ptrs(a.s1, s1[:3])
    if planes is not None:
        a.w1_ih_t_p, a.w1_hh_t_p, a.w0_hh_t_p, a.dg0_all_p, a.dg1_all_p = [_p(t, torch.int16) for t in planes]
    if w1_cat is not None and (planes is None or w1_cat[1] is not None):
        a.w1_cat_t, a.w1_cat_t_p = _p(w1_cat[0]), _p(w1_cat[1], torch.int16)
    nbytes = lib.fcl_decoder_train_workspace_bytes(n, u)
    ws = torch.empty(nbytes, device=dh1_all.device, dtype=torch.uint8)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(lib.fcl_decoder_bptt(C.byref(a), _stream()))


def bilstm_train_fwd(gx, w_hh, lens_i32, b, t, out, s, status=None):
    """Both directions: gx / w_hh: (forward, reverse) pairs; s: per direction (gates, c_new, c_old, h_old), t-major.
    status: device status word (None: the cooperating-workgroup kernel for H = 256 is not used)."""
    lib = _lib.load()
    h = w_hh[0].shape[1]
    a = _lib.BilstmTrain(b=b, t=t, h=h, lens=_p(lens_i32, torch.int32), out=_p(out), status=_p(status, torch.int32))
    for d in range(2):
        a.gx[d], a.w_hh[d] = _p(gx[d]), _p(w_hh[d])
        _ptrs(a.s[d], s[d])
    nbytes = lib.fcl_bilstm_train_workspace_bytes(b, h)
    ws = torch.empty(nbytes, device=out.device, dtype=torch.uint8)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(lib.fcl_bilstm_train_fwd(C.byref(a), _stream()))


def bilstm_bptt(s, lens_i32, b, t, d_out, w_hh_t, dg, status=None):
    lib = _lib.load()
    h = w_hh_t[0].shape[0]
    a = _lib.BilstmBptt(b=b, t=t, h=h, lens=_p(lens_i32, torch.int32), d_out=_p(d_out), ld_dout=d_out.shape[1], status=_p(status, torch.int32))
    for d in range(2):
        a.w_hh_t[d], a.dg[d] = _p(w_hh_t[d]), _p(dg[d])
        _ptrs(a.s[d], s[d][:3])
    nbytes = lib.fcl_bilstm_train_workspace_bytes(b, h)
    ws = torch.empty(nbytes, device=d_out.device, dtype=torch.uint8)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(lib.fcl_bilstm_bptt(C.byref(a), _stream()))


# ---- round 6: the fused small launches of the training update (csrc/fused_small.hip) ---------------------------------------------------------------
def loss_terms_batch(terms, sums_f64):
    """ONE launch for a list of element-wise loss terms.  Each term is a dict: a, b, slot (row of sums_f64 [n, 3]), count, w_l1, w_mse and optionally
    valid, b_log_offset, want_planes, and the second target b2 / valid2 / slot2 / count2 / w_l1_2 / w_mse_2.  Returns [(da, da_planes or None), ...]."""
    assert 1 <= len(terms) <= _lib.LOSS_MAX_TERMS
    arr = (_lib.LossTerm * len(terms))()
    outs = []
    for t, d in zip(arr, terms):
        a, b = d["a"], d["b"]
        if a.dim() == 1:
            a, b = a.reshape(-1, 1), b.reshape(-1, 1)
        m, c = a.shape
        da = torch.empty_like(a)
        dap = planes_empty(m, c, a.device) if d.get("want_planes") else None
        t.a, t.b, t.valid, t.da, t.da_planes = _p(a), _p(b), _p(d.get("valid"), torch.uint8), _p(da), _p(dap, torch.int16)
        t.sums = sums_f64.data_ptr() + 24 * d["slot"]
        t.m, t.c = m, c
        t.b_log, t.b_log_offset = int(d.get("b_log_offset") is not None), float(d.get("b_log_offset") or 0.0)
        t.w_l1, t.w_mse, t.count = d["w_l1"], d["w_mse"], float(d["count"])
        if d.get("b2") is not None:
            b2 = d["b2"].reshape(m, c)
            t.b2, t.valid2 = _p(b2), _p(d.get("valid2"), torch.uint8)
            t.sums2 = sums_f64.data_ptr() + 24 * d["slot2"]
            t.w_l1_2, t.w_mse_2, t.count2 = d["w_l1_2"], d["w_mse_2"], float(d["count2"])
        outs.append((da, dap))
    check(_lib.load().fcl_loss_terms_batch(arr, len(terms), _stream()))
    return outs


def sum_rows(srcs, row_valid=None, want_f32=True, want_planes=False, out=None):
    """row_valid[r] ? sum of the sources' row r : 0 (dense [rows, cols] fp32 matrices, cols % 4 == 0)."""
    rows, cols = srcs[0].shape
    ptrs = (C.c_void_p * len(srcs))(*[_p(s) for s in srcs])
    dst = (out if out is not None else torch.empty_like(srcs[0])) if want_f32 else None
    dp = planes_empty(rows, cols, srcs[0].device) if want_planes else None
    check(_lib.load().fcl_sum_rows(ptrs, len(srcs), _p(row_valid, torch.uint8), _p(dst), _p(dp, torch.int16), rows, cols, _stream()))
    return (dst, dp) if want_planes else dst


def bn_bwd_sums(dy, z, mean, invstd, dgamma, dbeta, act=ACT_NONE, y_act=None, keep=None, keep_scale=1.0, dy2=None):
    """dz = (dy [+ dy2]) [* keep * keep_scale] * act'(y_act); dgamma += colsum(dz * zhat), dbeta += colsum(dz).  Returns dz."""
    m, c = dy.shape
    dz = torch.empty_like(dy)
    check(_lib.load().fcl_bn_bwd_sums(_p(dy), _p(dy2), _p(y_act), _p(keep, torch.uint8), keep_scale, act, _p(z), _p(mean), _p(invstd), _p(dz), _p(dgamma),
                                      _p(dbeta), m, c, _stream()))
    return dz


def act_bwd_sum(dy, dy2, y, act, keep=None, keep_scale=1.0, want_planes=False):
    dz = torch.empty_like(dy)
    cols = dy.shape[-1]
    dzp = planes_empty(dy.numel() // cols, cols, dy.device) if want_planes else None
    check(_lib.load().fcl_act_bwd_sum(_p(dy), _p(dy2), _p(y), _p(keep, torch.uint8), keep_scale, _p(dz), _p(dzp, torch.int16), cols if want_planes else 0,
                                      dy.numel(), act, _stream()))
    return (dz, dzp) if want_planes else dz


def gather_rows_sum(src, src2, src3, idx_i32, want_planes=False):
    n, c = idx_i32.numel(), src.shape[1]
    dst = torch.empty(n, c, device=src.device, dtype=torch.float32)
    dp = planes_empty(n, c, src.device) if want_planes else None
    check(_lib.load().fcl_gather_rows_sum_fwd(_p(src), _p(src2), _p(src3), _p(idx_i32, torch.int32), _p(dst), _p(dp, torch.int16), n, c, _stream()))
    return (dst, dp) if want_planes else dst


def linear2(x, w, x2=None, w2=None, bias=None, residual=None, act=ACT_NONE):
    """act(x . w^T [+ x2 . w2^T] + bias) [+ residual] on fp32 operands."""
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty(m, n, device=x.device, dtype=torch.float32)
    k2 = x2.shape[1] if x2 is not None else 0
    check(_lib.load().fcl_linear2_fwd(_p(x), k, _p(w), k, k, _p(x2), k2, _p(w2), k2, k2, _p(bias), _p(residual), n, _p(y), n, m, n, act, _stream()))
    return y


# ---- compute pipes (fcl_hip.h "Compute pipes"): which streams contend, and a stream that does not --------------------------------------------------------------
class PlacedStream(torch.cuda.ExternalStream):
    """A raw HIP stream made by stream_apart.  It lives as long as the process unless destroy() is called: torch's caching allocator may hold a stream it was
    shown with record_stream() long after the Python object is gone (an event is recorded on it when the block is freed), so the object must not take the
    queue with it.  stream_apart() therefore hands out ONE stream per set of `others` (a cache), not one per caller."""

    def __new__(cls, handle, device, tried):
        self = super().__new__(cls, handle, device=device)
        self.fcl_handle, self.fcl_candidates_tried, self.fcl_placed = handle, tried, True
        return self

    def destroy(self):
        """Only for a stream that no tensor was ever record_stream()-ed on and that is idle (tests, probes)."""
        h, self.fcl_handle = self.fcl_handle, None
        if h:
            check(_lib.load().fcl_stream_destroy(C.c_void_p(h)))


_apart_cache = {}  # (device index, handles of `others`) -> the stream placed apart from them


def streams_share_pipe(a, b):
    """(shared, ratio): do the queues of two idle torch streams sit on the same compute pipe (measured, ~2 ms)?  ratio = pair time / alone."""
    shared, ratio = C.c_int(0), C.c_double(0.0)
    with torch.cuda.device(a.device):
        check(_lib.load().fcl_streams_share_pipe(C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), C.byref(shared), C.byref(ratio)))
    return bool(shared.value), float(ratio.value)


def stream_apart(others, device=None, strict=False, cache=True):
    """A new stream whose queue shares a compute pipe with none of `others` (torch streams; at most three can always be satisfied on MI355X's four pipes).
    Two chains of dependent launches on one pipe run 1.43x slower than apart: for the KD update 12.7 ms instead of 8.4 when the frozen teacher's stream lands
    on the student's pipe, which depends on how many streams the process created before (profiles/r6_stream_placement_kd.log).  When no candidate measures
    apart (more than three `others`, or a process that holds so many streams that new ones keep landing on one queue) the result is an ordinary stream with
    `fcl_placed = False` -- or a RuntimeError under strict=True.  cache=True (default): the same `others` get the same stream again (see PlacedStream)."""
    others = [s for s in others if s is not None]
    device = torch.device(device) if device is not None else (others[0].device if others else torch.device("cuda", torch.cuda.current_device()))
    key = (device.index if device.index is not None else torch.cuda.current_device(), tuple(int(s.cuda_stream) for s in others))
    if cache and key in _apart_cache:
        return _apart_cache[key]
    arr = (C.c_void_p * max(len(others), 1))(*[s.cuda_stream for s in others])
    out, tried = C.c_void_p(), C.c_int(0)
    with torch.cuda.device(device):
        rc = _lib.load().fcl_stream_create_apart(arr, len(others), C.byref(out), C.byref(tried))
        if rc != 0:
            if strict:
                check(rc)
            s = torch.cuda.Stream(device=device)
            s.fcl_candidates_tried, s.fcl_placed = tried.value, False
            return s
    s = PlacedStream(out.value, device, tried.value)
    if cache:
        _apart_cache[key] = s
    return s
