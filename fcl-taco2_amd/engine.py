"""Batched free-running mel synthesis on the HIP path (the BASELINE.json metric path).

Restates, for a batch, what `Tacotron2_sa.inference` does per utterance (reference
nets/knowledge_distillation/e2e_tts_tacotron2_sa_kd_student.py:804-863 and
nets/modules/decoder_sa_kd.py:707-800).  Batched synthesis is a build extension: decoder
rows (phonemes) are independent, so the batch result must equal B independent `inference()` calls — every
stage therefore treats utterance edges as zero padding (no cross-utterance or padding leak).

Host work here is integer index bookkeeping only (numpy): segment bounds, the exclusive cumsum that
places phoneme p's frames in its utterance (H10), and the duration sort that makes the live decoder rows
a shrinking prefix (SURVEY.md §7 "ragged work").  All arithmetic on activations is in libfcl_hip.so.
"""
import os
import time

import numpy as np
import torch

from . import ops
from .hparams import output_act_code
from .plan import LN_EPS


class RowMaps(object):
    """Integer maps between (utterance, phoneme) rows, sorted decoder rows and output frames."""
    __slots__ = ("src_rows", "order", "dur_sorted", "frame_off_sorted", "live_rows", "utt_frames", "n_frames", "lmax",
                 "frame_lo", "frame_hi")


def build_row_maps(lens, durs, t_max, frames_per_step=1):
    """lens [B]; durs: list of int arrays (len_b each).  Raises AssertionError on a zero duration, like the
    reference's `assert ds_nonzeros.shape[0] == hs.shape[0]` (decoder_sa_kd.py:739).
    frames_per_step (`reduction_factor`, decoder_sa.py:573, 627): a duration unit is one decoder STEP and frames_per_step output frames --
    live rows / lmax / dur_sorted count steps, frame offsets and frame totals count frames."""
    B = len(lens)
    lens_np = np.asarray(lens, dtype=np.int64)
    parts = [np.asarray(d).reshape(-1) for d in durs]
    for b in range(B):
        assert parts[b].shape[0] == lens_np[b], "duration count != phoneme count"
    dur = np.concatenate(parts).astype(np.int64) if B else np.zeros(0, dtype=np.int64)
    assert (dur > 0).all(), "zero duration: ds_nonzeros.shape[0] != hs.shape[0]"
    n = dur.shape[0]
    starts = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(lens_np, out=starts[1:])
    src = np.arange(n, dtype=np.int64) + np.repeat(np.arange(B, dtype=np.int64) * t_max - starts[:-1], lens_np)  # row b * t_max + position
    csum = np.cumsum(dur) * int(frames_per_step)
    foff = csum - dur * int(frames_per_step)  # exclusive cumsum over the whole batch = utterance base + exclusive cumsum inside the utterance (H10)
    fstarts = np.zeros(B + 1, dtype=np.int64)
    fstarts[1:] = csum[starts[1:] - 1]
    utt = fstarts[1:] - fstarts[:-1]
    order = np.argsort(-dur, kind="stable")
    m = RowMaps()
    m.order = order
    m.src_rows = src[order].astype(np.int32)
    m.dur_sorted = dur[order].astype(np.int32)
    m.frame_off_sorted = foff[order].astype(np.int32)
    m.lmax = int(m.dur_sorted[0])
    # live_rows[t] = #rows with dur > t  (rows sorted descending => a prefix)
    m.live_rows = np.ascontiguousarray((n - np.cumsum(np.bincount(dur, minlength=m.lmax + 1))[: m.lmax]).astype(np.int32))
    m.utt_frames = [int(v) for v in utt]
    m.n_frames = int(fstarts[-1])
    m.frame_lo = np.repeat(fstarts[:-1], utt).astype(np.int32)
    m.frame_hi = np.repeat(fstarts[1:], utt).astype(np.int32)
    return m


def _predictor_scalar(pp, hs, seg_lo, seg_hi, pad_mask_u8):
    x = hs
    n = len(pp.convs)
    out = None
    for i, cv in enumerate(pp.convs):
        x = ops.conv1d(x, cv.wp, cv.bias, seg_lo, seg_hi, ops.ACT_RELU)
        last = i == n - 1
        x, out = ops.layernorm(x, pp.ln[i][0], pp.ln[i][1], LN_EPS, want_y=not last,
                               lin_w=pp.lin_w if last else None, lin_b=pp.lin_b if last else None,
                               pad_mask=pad_mask_u8 if last else None)
    return out


def _predictor_scalar_planes(pp, hs_p, seg_lo, seg_hi, pad_mask_u8):
    """The same predictor on pre-split operands: the conv reads P32 planes and writes fp32 (the LayerNorm's input), the LayerNorm writes the
    next conv's planes."""
    xp = hs_p
    n = len(pp.convs)
    out = None
    for i, cv in enumerate(pp.convs):
        y, _ = ops.conv1d_planes(xp, cv, seg_lo, seg_hi, ops.ACT_RELU, want_f32=True, want_planes=False)
        last = i == n - 1
        if last:
            _, out = ops.layernorm(y, pp.ln[i][0], pp.ln[i][1], LN_EPS, want_y=False, lin_w=pp.lin_w, lin_b=pp.lin_b, pad_mask=pad_mask_u8)
        else:
            _, _, xp = ops.layernorm(y, pp.ln[i][0], pp.ln[i][1], LN_EPS, want_y=False, want_planes=True)
    return out


def _predictors_grouped(grp, hs_p, seg_lo, seg_hi, pad_u8, m, masked):
    """All predictors of `grp` (plan.PredictorGroup) in one launch per layer: conv-0 with the stacked output channels, then grouped LayerNorm /
    Conv1d / LayerNorm + head.  Returns the [m] outputs in grp.names order; `masked[i]`: apply the pad mask to output i (the duration predictor's
    inference path masks later, in the rounding kernel)."""
    G, C = grp.G, grp.chans
    y, _ = ops.conv1d_planes(hs_p, grp.w0, seg_lo, seg_hi, ops.ACT_RELU, want_f32=True, want_planes=False)  # [m, G * C]
    ldx, gstride = G * C, C
    out = None
    for i in range(grp.layers):
        last = i == grp.layers - 1
        if last:
            same_mask = all(masked) or not any(masked)
            assert same_mask, "grouped predictors share one pad-mask policy"
            _, out, _ = ops.layernorm_group(y, ldx, gstride, grp.gamma[i], grp.beta[i], LN_EPS, m, C, G, lin_w=grp.lin_w, lin_b=grp.lin_b,
                                            pad_mask=pad_u8 if all(masked) else None)
        else:
            _, _, xp = ops.layernorm_group(y, ldx, gstride, grp.gamma[i], grp.beta[i], LN_EPS, m, C, G, want_planes=True)
            y, _ = ops.conv1d_planes_group(xp, m * (C // 32) * 64, grp.wpp[i], grp.bias[i], seg_lo, seg_hi, m, C, C, grp.k, G, act=ops.ACT_RELU)
            ldx, gstride = C, m * C
    return [out[g * m : (g + 1) * m] for g in range(G)]


def use_planes(plan):
    """Pre-split (P32) operands end to end: on by default (FCL_PRECISION=0 / FCL_PLANES=0 turn it off), needs whole 32-column lines."""
    hp = plan.hp
    if getattr(plan, "generic_decoder", False) or hp.elayers != 1:  # structure options beyond the shipped recipes: the fp32-operand path end to end
        return False
    return (ops.planes_enabled() and plan.enc_convs[0].wpp is not None and plan.decoder.struct.w0_att_p is not None
            and all(x % 32 == 0 for x in (hp.embed_dim, hp.econv_chans, hp.eunits, hp.adim, hp.postnet_chans, hp.duration_predictor_chans,
                                          hp.variance_predictor_chans)))


class PreparedBatch(object):
    """Everything `run` needs, resident in HBM: padded ids, segment bounds, and (forced durations) row maps -- built on the host (`maps`), or
    left to the device (`dur_pad`: the forced durations in the padded [B, T] layout; run() then needs `caps`)."""
    __slots__ = ("B", "T", "lens", "ids", "seg_lo", "seg_hi", "pad", "lens_dev", "f0e", "maps", "src_rows", "dur", "frame_off",
                 "frame_lo", "frame_hi", "dur_pad", "spk")


class Caps(object):
    """Capacities of a pass whose row maps are built on the DEVICE (ops.row_maps_build): everything the host would otherwise have to know about the
    durations.  lmax: decoder steps to launch; frames: rows of the frame-major buffers; bounds: int32 [lmax] upper bounds of the live rows per
    step (grid sizes and kernel selection only).  None of them is trusted: a batch that exceeds one raises FCL_STATUS_* in the device status word
    (read it with DeviceFrames.resolve() / ops.check_status) and decodes nothing."""
    __slots__ = ("lmax", "frames", "bounds", "tail_from")

    def __init__(self, lmax, frames, bounds, tail_from=0):
        """tail_from (0 = off): from that decoder step on, the rows still live continue in ONE launch of the persistent row-tile kernel
        (fcl_decoder_io_t.tail_from): set it to where the slack of `lmax` begins -- steps nobody reaches then cost one launch instead of three each."""
        self.lmax, self.frames = int(lmax), int(frames)
        self.bounds = np.ascontiguousarray(np.asarray(bounds, dtype=np.int32))
        self.tail_from = int(tail_from) if 0 < int(tail_from) < self.lmax else 0
        assert self.bounds.shape == (self.lmax,) and self.lmax > 0 and self.frames > 0

    @staticmethod
    def from_maps(maps):
        """Exact capacities of a batch whose host maps are known (calibration of a graph that replays THAT batch)."""
        return Caps(maps.lmax, maps.n_frames, maps.live_rows)

    @staticmethod
    def for_batches(host_maps, slack_steps=0, frame_round=256):
        """Exact capacities of a KNOWN set of batches (their host-built maps): steps = the longest duration (+ slack_steps), frames = the largest
        total rounded up to `frame_round`, per-step row bounds = the maximum over the batches.  This is the best case -- what bench.py's headline
        line uses for the four batches it feeds; a driver that cannot know its batches in advance calibrates with slack (decode._grown_caps)."""
        lmax = max(m.lmax for m in host_maps) + int(slack_steps)
        bounds = np.ones(lmax, dtype=np.int32)
        for m in host_maps:
            bounds[: m.lmax] = np.maximum(bounds[: m.lmax], m.live_rows)
        return Caps(lmax, (max(m.n_frames for m in host_maps) + frame_round - 1) // frame_round * frame_round, bounds)

    @staticmethod
    def generous(n_rows, lmax, frames):
        """Capacities that hold for ANY durations up to lmax per phoneme and `frames` in total: every step may keep every row."""
        return Caps(lmax, frames, np.full(lmax, n_rows, dtype=np.int32))


class DeviceFrames(object):
    """Frame bookkeeping of a device-driven pass: utterance frame starts [B + 1] and totals live in HBM until somebody needs them on the host."""

    def __init__(self, maps, caps, device, status=None):
        self.utt_frame0, self.totals, self.live_rows, self.caps, self.device = maps["utt_frame0"], maps["totals"], maps["live_rows"], caps, device
        self.status = status  # the pass's own status word (None: the per-device one)
        self._host = None

    def resolve(self):
        """Synchronising read: raises FclError when the pass violated a capacity or met a zero duration; returns the per-utterance frame counts."""
        if self._host is None:
            if self.status is None:
                ops.check_status(self.device)
            else:
                bits = int(self.status.item()) & 0xFFFFFFFF
                if bits:
                    self.status.zero_()
                    raise ops._lib.FclError("fcl-taco2_amd: device status 0x%x: %s" % (bits, ops.status_message(bits)))
            f0 = self.utt_frame0.cpu().numpy()
            self._host = [int(v) for v in (f0[1:] - f0[:-1])]
        return self._host


_RING = None
import os as _os

_GROUP_PREDICTORS = _os.environ.get("FCL_PRED_GROUP", "1") not in ("", "0")  # 0: one launch per predictor and layer (rounds 1-2)


def _ring():
    global _RING
    if _RING is None:
        from .hostio import PinnedRing

        _RING = PinnedRing(depth=8)
    return _RING


def _upload_maps(holder, maps, dev):
    """Row maps of a pass whose durations were predicted on the device: one packed int32 block through the pinned ring."""
    holder.maps = maps
    blocks = dict(src_rows=maps.src_rows, dur=maps.dur_sorted, frame_off=maps.frame_off_sorted, frame_lo=maps.frame_lo, frame_hi=maps.frame_hi)
    layout, off = {}, 0
    for k, v in blocks.items():
        layout[k] = (off, v.size)
        off += (v.size + 3) // 4 * 4
    i32 = np.zeros(off, dtype=np.int32)
    for k, v in blocks.items():
        i32[layout[k][0] : layout[k][0] + v.size] = v
    with torch.cuda.device(dev):
        up = _ring().upload({"i32m": torch.from_numpy(i32)}, dev)["i32m"]
    for k, (o, n) in layout.items():
        setattr(holder, k, up[o : o + n])


def prepare(plan, xs, durs=None, f0=None, energy=None, device_maps=False, spembs=None):
    """Input hand-over: pad + upload phoneme ids, build the integer segment bounds, and — when durations are
    forced — the row maps.  This is the host batch layout step (the reference's loader/converter side).  Every integer array of the batch
    travels in ONE packed int32 block (plus the ids and the pad mask) through fixed pinned staging buffers, non-blocking: three copies per
    batch instead of ten pageable ones.  device_maps: forced durations are only uploaded (padded [B, T] int32); the maps are built by
    ops.row_maps_build inside run(), which then needs `caps`."""
    dev = plan.device
    p = PreparedBatch()
    p.B = len(xs)
    p.lens = [int(len(x)) for x in xs]
    p.T = T = max(p.lens)
    ids = np.zeros((p.B, T), dtype=np.int64)
    for b, x in enumerate(xs):
        ids[b, : p.lens[b]] = x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    lens_np = np.asarray(p.lens, dtype=np.int64)
    rows = np.arange(p.B * T, dtype=np.int64)
    b_of = rows // T
    blocks = {"seg_lo": (b_of * T).astype(np.int32), "seg_hi": (b_of * T + lens_np[b_of]).astype(np.int32), "lens_dev": lens_np.astype(np.int32)}
    p.maps, p.dur_pad = None, None
    if durs is not None and device_maps:
        dpad = np.zeros((p.B, T), dtype=np.int32)
        for b in range(p.B):
            d = np.asarray(durs[b]).reshape(-1)
            assert d.shape[0] == p.lens[b], "duration count != phoneme count"
            dpad[b, : p.lens[b]] = d
        blocks["dur_pad"] = dpad.reshape(-1)
    elif durs is not None:
        p.maps = m = build_row_maps(p.lens, durs, T, plan.hp.reduction_factor)
        blocks.update(src_rows=m.src_rows, dur=m.dur_sorted, frame_off=m.frame_off_sorted, frame_lo=m.frame_lo, frame_hi=m.frame_hi)
    layout, off = {}, 0
    for k, v in blocks.items():
        layout[k] = (off, v.size)
        off += (v.size + 3) // 4 * 4  # 16-byte aligned slices
    i32 = np.zeros(off, dtype=np.int32)
    for k, v in blocks.items():
        i32[layout[k][0] : layout[k][0] + v.size] = v
    items = {"ids": torch.from_numpy(ids.reshape(-1)), "i32": torch.from_numpy(i32),
             "pad": torch.from_numpy(((rows % T) >= lens_np[b_of]).astype(np.uint8))}
    if f0 is not None:
        pe = np.zeros((2, p.B, T), dtype=np.float32)
        for b in range(p.B):
            pe[0, b, : p.lens[b]] = np.asarray(f0[b]).reshape(-1)
            pe[1, b, : p.lens[b]] = np.asarray(energy[b]).reshape(-1)
        items["f0e"] = torch.from_numpy(pe.reshape(2, -1))
    if plan.hp.spk_embed_dim is not None:  # one speaker-embedding vector per utterance (tts.py:285-287, ..._sa.py:636-638)
        if spembs is None:
            raise ValueError("fcl-taco2_amd: the model was built with spk_embed_dim=%d: every utterance needs its speaker embedding" % plan.hp.spk_embed_dim)
        sp = np.stack([(s.detach().cpu().numpy() if torch.is_tensor(s) else np.asarray(s)).reshape(-1) for s in spembs]).astype(np.float32)
        assert sp.shape == (p.B, plan.hp.spk_embed_dim), "speaker embeddings must be [B, spk_embed_dim]"
        items["spk"] = torch.from_numpy(np.ascontiguousarray(sp))
    elif spembs is not None:
        raise ValueError("fcl-taco2_amd: speaker embeddings given to a model without spk_embed_dim")
    with torch.cuda.device(dev):
        up = _ring().upload(items, dev)
    p.ids, p.pad, p.f0e, p.spk = up["ids"], up["pad"], up.get("f0e"), up.get("spk")
    for k, (o, n) in layout.items():
        setattr(p, k, up["i32"][o : o + n])
    return p


def encode(plan, prep, bilstm_algo=0, planes=False, row_maps=None):
    """H1-H3: embedding -> 3 x conv/BN/ReLU -> packed BiLSTM.  Returns hs [B*T, C] (planes: (hs, P32 planes of hs)).
    row_maps (ops.RowMapsRequest): maps of FORCED durations, which depend on nothing the encoder computes: built inside the BiLSTM's launch."""
    bl = plan.blstm
    res = plan.hp.use_residual  # convs[i](xs) + xs (encoder_sa_kd.py:213-214): the residual rides in the conv's epilogue (after ReLU), fp32 kept for it
    if planes:  # every GEMM operand travels pre-split: embedding -> planes, conv -> planes, ..., BiLSTM -> fp32 + planes
        x, xp = ops.embedding(prep.ids, plan.embed, want_f32=res, want_planes=True)
        for cv in plan.enc_convs:
            x, xp = ops.conv1d_planes(xp, cv, prep.seg_lo, prep.seg_hi, ops.ACT_RELU, residual=x if res else None, want_f32=res)
        return ops.bilstm(None, prep.lens_dev, bl["w_ih_f"], bl["w_hh_f"], bl["b_f"], bl["w_ih_r"], bl["w_hh_r"], bl["b_r"], prep.B, prep.T, bilstm_algo,
                          x_p=xp, w_ih_p=(bl["w_ih_p_f"], bl["w_ih_p_r"]), want_planes=True, row_maps=row_maps)
    x = ops.embedding(prep.ids, plan.embed)
    for cv in plan.enc_convs:
        x = ops.conv1d(x, cv.wp, cv.bias, prep.seg_lo, prep.seg_hi, ops.ACT_RELU, residual=x if res else None)
    for i, bl in enumerate(plan.blstm_layers):  # (`elayers` > 1: the stacked layers run one after the other; use_planes() is off for them)
        x = ops.bilstm(x, prep.lens_dev, bl["w_ih_f"], bl["w_hh_f"], bl["b_f"], bl["w_ih_r"], bl["w_hh_r"], bl["b_r"], prep.B, prep.T, bilstm_algo,
                       row_maps=row_maps if i == 0 else None)
    return x


class _DevMaps(object):
    """The device-built maps of a pass behind the attribute names run() uses for host-built ones."""

    def __init__(self, dm, caps):
        self.src_rows, self.dur, self.frame_off, self.frame_lo, self.frame_hi = dm["src_rows"], dm["dur"], dm["frame_off"], dm["frame_lo"], dm["frame_hi"]
        self.live_dev, self.live_rows, self.n_frames, self.lmax = dm["live_rows"], caps.bounds, caps.frames, caps.lmax
        self.tail_from = caps.tail_from
        self.order = None


def run(plan, prep, dropout_mode=ops.DROP_RNG, prenet_keep=None, seed=0, bilstm_algo=0, return_intermediates=False, seed_dev=None, caps=None,
        status=None):
    """One pass of the hot path over a prepared batch.  Returns the packed mel [F, odim] (after postnet) and
    the per-utterance frame counts; with forced durations nothing here touches the host.
    caps (engine.Caps): build the row maps on the DEVICE from the durations in HBM — predicted by this pass, or forced and uploaded by
    prepare(device_maps=True) — so that no step waits for the host: the mel buffer then has caps.frames rows (the first total are valid) and
    the frame counts come back as a DeviceFrames to resolve() when the caller synchronises anyway.  Without caps a pass with predicted durations
    makes one host round trip (durations down, maps up)."""
    hp, dev = plan.hp, plan.device
    with torch.cuda.device(dev):
        planes = use_planes(plan)
        hs_p = None
        maps_req = None
        if prep.maps is None and prep.dur_pad is not None and caps is not None:  # forced durations, device-built maps: nothing to wait for
            maps_req = ops.row_maps_request(prep.B * prep.T, prep.B, caps.lmax, caps.frames, dur_i32=prep.dur_pad, t_max=prep.T, pad=prep.pad,
                                            status=status)
        if planes:
            hs, hs_p = encode(plan, prep, bilstm_algo, planes=True, row_maps=maps_req)
            predictor = lambda pp, pad: _predictor_scalar_planes(pp, hs_p, prep.seg_lo, prep.seg_hi, pad)
        else:
            hs = encode(plan, prep, bilstm_algo, row_maps=maps_req)
            predictor = lambda pp, pad: _predictor_scalar(pp, hs, prep.seg_lo, prep.seg_hi, pad)
        inter = {"hs": hs, "T": prep.T} if return_intermediates else None
        if hp.spk_embed_dim is not None:  # hs <- cat[hs, F.normalize(spemb)]: predictors, embeddings and decoder see eunits + spk_embed_dim channels
            if getattr(prep, "spk", None) is None:
                raise ValueError("fcl-taco2_amd: this pass needs speaker embeddings (prepare(..., spembs=...))")
            hs, hs_p = ops.concat_spk(hs, prep.spk, prep.T, want_f32=True, want_planes=planes)
        rm = prep  # holder of the row maps
        frames_info = None
        # the predictors share one geometry in the shipped recipes: one launch per layer for all of them (plan.PredictorGroup) instead of one each
        grouped = planes and prep.f0e is None and _GROUP_PREDICTORS
        m_rows = prep.B * prep.T
        p = e = None
        if prep.maps is None:  # maps depend on durations that live in HBM: predicted by this pass, or forced and uploaded
            d_int = None
            if prep.dur_pad is None:
                if grouped and plan.group_dpe is not None:
                    d_log, p, e = _predictors_grouped(plan.group_dpe, hs_p, prep.seg_lo, prep.seg_hi, prep.pad, m_rows, [True, True, True])
                else:
                    d_log = predictor(plan.duration, None)
                d_int = ops.duration_round(d_log, False, 1.0, prep.pad)
                if inter is not None:
                    inter["d_log"], inter["d_int"] = d_log, d_int
            if caps is not None:  # device-built maps over the padded [B, T] row universe: no host round trip
                dm = maps_req.maps if maps_req is not None else ops.row_maps_build(prep.B * prep.T, prep.B, caps.lmax, caps.frames, dur_i64=d_int,
                                                                                   t_max=prep.T, pad=prep.pad, status=status)
                rm = PreparedBatch()
                rm.maps = _DevMaps(dm, caps)
                rm.src_rows, rm.dur, rm.frame_off, rm.frame_lo, rm.frame_hi = dm["src_rows"], dm["dur"], dm["frame_off"], dm["frame_lo"], dm["frame_hi"]
                frames_info = DeviceFrames(dm, caps, dev, status)
            else:
                if d_int is None:
                    raise ValueError("prepare(device_maps=True) leaves the row maps to the device: run() needs caps")
                d_host = d_int.cpu().numpy().reshape(prep.B, prep.T)  # the one host sync of the predicted-duration path
                rm = PreparedBatch()
                _upload_maps(rm, build_row_maps(prep.lens, [d_host[b, : prep.lens[b]] for b in range(prep.B)], prep.T, hp.reduction_factor), dev)
        if prep.f0e is not None:
            p, e = prep.f0e[0], prep.f0e[1]
        elif p is None and grouped and plan.group_pe is not None:
            p, e = _predictors_grouped(plan.group_pe, hs_p, prep.seg_lo, prep.seg_hi, prep.pad, m_rows, [True, True])
        elif p is None:
            p = predictor(plan.pitch, prep.pad)
            e = predictor(plan.energy, prep.pad)
        att, p_emb, e_emb = ops.variance_embed_add(hs, p, e, plan.pitch_embed_w, plan.pitch_embed_b, plan.energy_embed_w,
                                                   plan.energy_embed_b, prep.seg_lo, prep.seg_hi, want_embs=return_intermediates)
        maps = rm.maps
        att_c, att_c_p = (ops.gather_rows(att, rm.src_rows, want_f32=False, want_planes=True) if planes else (ops.gather_rows(att, rm.src_rows), None))
        keep_dev = None
        if hp.dropout_rate <= 0.0:
            dropout_mode = ops.DROP_NONE
        if dropout_mode == ops.DROP_MASK and maps.order is None:
            raise ValueError("injected prenet masks are given in (utterance, phoneme) order: they need host-built row maps")
        if dropout_mode == ops.DROP_MASK:
            keep = np.ascontiguousarray(np.asarray(prenet_keep)[: maps.lmax][:, :, maps.order, :])  # to sorted row order
            keep_dev = torch.from_numpy(keep).to(dev)
        before = ops.decoder_loop(plan.decoder, att_c, rm.dur, maps.live_rows, rm.frame_off, maps.n_frames,
                                  dropout_mode=dropout_mode, prenet_keep=keep_dev, seed=seed, seed_dev=seed_dev, att_c_p=att_c_p, want_before_p=planes,
                                  live_rows_dev=getattr(maps, "live_dev", None), status=status, tail_from=getattr(maps, "tail_from", 0))
        n_post = len(plan.postnet)
        if planes:
            before, xp = before
            # device-built maps: the frame buffers are capacities (decode driver: x 1.3 slack) -- tiles beyond the batch's real total are skipped
            f_dev = frames_info.totals[:1] if frames_info is not None else None
            for i, cv in enumerate(plan.postnet):
                last = i == n_post - 1
                x, xp = ops.conv1d_planes(xp, cv, rm.frame_lo, rm.frame_hi, ops.ACT_NONE if last else ops.ACT_TANH, residual=before if last else None,
                                          want_f32=last, want_planes=not last, m_dev=f_dev)
        else:
            x = before
            for i, cv in enumerate(plan.postnet):
                last = i == n_post - 1
                x = ops.conv1d(x, cv.wp, cv.bias, rm.frame_lo, rm.frame_hi, ops.ACT_NONE if last else ops.ACT_TANH,
                               residual=before if last else None)
        if hp.output_activation is not None:  # outs = output_activation_fn(before + postnet(before)) (decoder_sa.py:635-636); `before` stays raw
            x = ops.act_fwd(x, output_act_code(hp))
        utt_frames = frames_info if frames_info is not None else maps.utt_frames
        if inter is not None:
            inter.update(p_outs=p, e_outs=e, p_embs=p_emb, e_embs=e_emb, before=before, after=x, maps=maps)
            return x, utt_frames, inter
        return x, utt_frames


def synthesize(plan, xs, durs=None, f0=None, energy=None, dropout_mode=ops.DROP_RNG, prenet_keep=None, seed=0,
               bilstm_algo=0, return_intermediates=False, caps=None, spembs=None):
    """xs: list of 1-D int64 id arrays/tensors; durs: optional list of forced durations (else predicted).
    spembs: list of speaker-embedding vectors [spk_embed_dim], one per utterance (models built with spk_embed_dim).
    f0/energy: optional lists of [T] arrays replacing the predictors (inference(f0=..., energy=...)).
    prenet_keep: optional uint8 [Lmax, 2, N, P] in (utterance, phoneme) row order (FCL_DROP_MASK).
    Returns a list of mel tensors [L_b, odim] (views into one packed device buffer)."""
    prep = prepare(plan, xs, durs, f0, energy, device_maps=caps is not None, spembs=spembs)
    out = run(plan, prep, dropout_mode, prenet_keep, seed, bilstm_algo, return_intermediates, caps=caps)
    import os

    if bilstm_algo == 3 or os.environ.get("FCL_BILSTM_GROUP_INFER", "0") not in ("", "0"):
        ops.check_status(plan.device)  # the cooperating-workgroup BiLSTM (opt-in, single-stream only) reports a timeout here, never silently
    after, utt_frames = out[0], out[1]
    if isinstance(utt_frames, DeviceFrames):
        utt_frames = utt_frames.resolve()
    mels, s = [], 0
    for n in utt_frames:
        mels.append(after[s : s + n])
        s += n
    return (mels, out[2]) if return_intermediates else mels


_SHARED_STREAMS = {}


import contextlib as _contextlib


@_contextlib.contextmanager
def capture(graph, stream, pool=None):
    """hipGraph capture of what runs inside, on `stream` -- torch.cuda.graph() without its device-wide synchronize + empty_cache() in front of EVERY capture
    (~1 ms each: a first decode() call captures 16 graphs back to back; round 6).  The caller has synchronised `stream`; allocations made inside go to the
    graph's private pool as with torch.cuda.graph()."""
    with torch.cuda.stream(stream):
        if pool is not None:
            graph.capture_begin(pool, capture_error_mode="global")
        else:
            graph.capture_begin(capture_error_mode="global")
        try:
            yield
        finally:
            graph.capture_end()


_PLACE_STREAMS = int(os.environ.get("FCL_PLACE_STREAMS", "1")) != 0  # 0: streams as torch's pool hands them out (round 5 behaviour)


def shared_streams(device, n):
    """The first `n` of this process's synthesis streams on `device` (created on first use, then reused by every runner set and by the decode
    driver).  This device runs FOUR concurrently active HIP queues well and falls off a cliff at the fifth, and the runtime
    maps streams onto its hardware queues round-robin: a second set of four streams created beside an idle first set can land two ACTIVE streams
    on one queue (the decode driver inside bench.py measured 28.6 M frames/s on its own fresh streams against 38 M in a process that had created
    no others).  One pool per device keeps the pass streams of a process the same four queues whoever drives them."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    pool = _SHARED_STREAMS.setdefault(key, [])
    if not pool:  # developer aid: FCL_STREAM_SKIP=k leaves the first k streams of torch's per-device pool unused (which hardware queue a stream lands on)
        _SHARED_STREAMS[key + ("skipped",)] = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("FCL_STREAM_SKIP", "0")))]
    while len(pool) < n:
        # round 6: the first four pass streams are MEASURED onto four different compute pipes (ops.stream_apart; two passes on one pipe run as on one queue:
        # the "cliff at the fifth stream" is two streams on one pipe, and which streams share one depends on every stream the process created before)
        if len(pool) < 4 and _PLACE_STREAMS:
            pool.append(ops.stream_apart(pool, device=dev))
        else:
            pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


def packed_words(frames_cap, odim, batch):
    """float32 words of a runner's packed output block / a landing slot: mel | frame starts [batch + 1] | status word, rounded to 256 bytes."""
    return (int(frames_cap) * int(odim) + int(batch) + 2 + 63) // 64 * 64


class GraphRunner(object):
    """One pass of `run` over a prepared batch captured as a hipGraph (guide: capture launch-bound inner loops in graphs).  Every buffer of the
    pass lives in the graph's private pool, so a replay is one host call; the prenet-dropout seed is a device word the graph itself advances, so
    each replay draws fresh masks.  Several runners on different streams keep several batches in flight.
    Forced durations: the host-built maps of `prep` are baked in.  PREDICTED durations (prep built without durs): the graph contains the duration
    predictor, the rounding and the device map builder, i.e. every replay recomputes the durations and the maps in HBM with no host round trip; the
    capacities are calibrated once, eagerly, on this batch (the predictors are deterministic, so they are exact for every replay) and verified on
    the device (a violated capacity raises in check())."""

    def __init__(self, plan, prep, stream=None, dropout_mode=ops.DROP_RNG, seed=0, caps=None):
        self.plan, self.prep = plan, prep
        self.stream = stream if stream is not None else torch.cuda.Stream(device=plan.device)
        with torch.cuda.device(plan.device):
            self.seed_word = torch.zeros(1, dtype=torch.int32, device=plan.device)
            self.caps = caps
            with torch.cuda.stream(self.stream):  # warm-up outside capture (lazy one-time setup in the library)
                if prep.maps is None and caps is None:  # calibration: one pass with the host round trip tells this batch's capacities
                    _, _, inter = run(plan, prep, dropout_mode, seed=seed, seed_dev=self.seed_word, return_intermediates=True)
                    self.caps = Caps.from_maps(inter["maps"])
                    self.utt_frames = list(inter["maps"].utt_frames)
                out = run(plan, prep, dropout_mode, seed=seed, seed_dev=self.seed_word, caps=self.caps)
                if isinstance(out[1], DeviceFrames):
                    self.utt_frames = out[1].resolve()
            self.stream.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                ops.u32_add(self.seed_word, 1)
                self.mel, frames = run(plan, prep, dropout_mode, seed=seed, seed_dev=self.seed_word, caps=self.caps)
            if isinstance(frames, DeviceFrames):
                self.frames = frames  # device-resident frame starts of the LAST replay
            else:
                self.frames, self.utt_frames = None, frames

    def replay(self):
        """Enqueue one pass on this runner's stream; returns the static output tensor [F, odim]."""
        with torch.cuda.stream(self.stream):
            self.graph.replay()
        return self.mel

    def check(self):
        """Device-driven graphs: synchronising check of the status word (capacity violations, zero durations)."""
        ops.check_status(self.plan.device)


class BatchRunner(object):
    """A captured pass whose every size is a CAPACITY: one hipGraph replays for any batch of at most `batch` utterances, `t_cap` phonemes each and
    `caps` (decoder steps, frames, live rows per step).  Per batch the host does three things: pack ids / lengths / pad mask (/ forced durations)
    into ONE pinned block, launch ONE graph, whose first node pulls the block into HBM (ops.feed_copy).  Everything the reference's inference() does per utterance on the
    host -- the `ds_nonzeros` filter, the position table, the per-phoneme trim and concat loops (..._kd_student.py:821-851,
    decoder_sa_kd.py:736-791) -- and this build's own numpy row maps run on the device inside the graph (ops.row_maps_build over the padded
    [batch, t_cap] row universe), with predicted OR forced durations.  Nothing about the durations is known to the host until it reads the mels
    back (`frames()`): capacities are verified on the device (FCL_STATUS_*), a batch that does not fit is reported, never silently truncated.
    forced=True: the graph takes uploaded durations (load(xs, durs)); forced=False: it contains the duration predictor + rounding."""

    def __init__(self, plan, batch, t_cap, caps, forced=True, stream=None, dropout_mode=ops.DROP_RNG, seed=0, depth=3, pack_outputs=False, mempool=None):
        """pack_outputs (round 6, the decode driver): the graph also gathers its three results -- mel [frames cap, odim], frame starts [B + 1], status word --
        into ONE float32 buffer `self.out` (two tiny copies and a ~5 us device copy inside the graph), so that a batch costs the host one device-to-host copy call
        instead of three (the driver's loop is bound by its enqueue time: 3 x ~70 us of copy calls out of ~470 us per batch).
        mempool: a torch graph-pool handle shared by the graphs that replay on THIS stream only (stream order keeps two of them from running at once): later
        captures reuse the intermediates' memory of earlier ones instead of allocating theirs (a first decode() call: ~500 allocations -> ~150)."""
        dev = plan.device
        self.plan, self.B, self.T, self.caps, self.forced = plan, int(batch), int(t_cap), caps, bool(forced)
        self.S = int(plan.hp.spk_embed_dim or 0)  # speaker-embedding width: one vector per utterance rides in the same block
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        n = self.B * self.T
        # one byte block: ids int64 [n] | seg_lo, seg_hi int32 [n] each | dur int32 [n] | lens int32 [B, padded to 4] | pad uint8 [n]
        self._off = {}
        off = 0
        for name, nbytes in (("ids", 8 * n), ("seg_lo", 4 * n), ("seg_hi", 4 * n), ("dur", 4 * n), ("lens", 4 * ((self.B + 3) // 4 * 4)), ("pad", n),
                             ("spk", 4 * self.B * self.S)):
            self._off[name] = (off, nbytes)
            off += (nbytes + 15) // 16 * 16
        self._nbytes = off
        # the feed: by default the graph's FIRST NODE pulls the packed block out of pinned host memory (ops.feed_copy: no copy call per pass, the
        # host waits for the sequence number of its last launch before it repacks the block); FCL_FEED_INGRAPH=0 = one hipMemcpyAsync per pass
        # out of `depth` rotating staging buffers in front of the launch (the form of the first half of round 3)
        self._ingraph = os.environ.get("FCL_FEED_INGRAPH", "1") not in ("", "0")
        self._host = [torch.zeros(off, dtype=torch.uint8).pin_memory() for _ in range(1 if self._ingraph else depth)]
        self._host_ev = [None] * len(self._host)
        self._slot = 0
        self._launched = 0
        self._seq_host = torch.zeros(4, dtype=torch.int32).pin_memory()
        self._seq_np = self._seq_host.numpy()
        rows = np.arange(n, dtype=np.int32)
        self._rows = (rows // self.T, rows - (rows // self.T) * self.T, ((rows // self.T) * self.T).astype(np.int32))
        with torch.cuda.device(dev):
            self._dev = torch.zeros(off, dtype=torch.uint8, device=dev)
            view = lambda name, dt: self._dev[self._off[name][0] : self._off[name][0] + self._off[name][1]].view(dt)
            p = PreparedBatch()
            p.B, p.T, p.lens = self.B, self.T, None
            p.ids, p.seg_lo, p.seg_hi = view("ids", torch.int64), view("seg_lo", torch.int32), view("seg_hi", torch.int32)
            p.lens_dev, p.pad = view("lens", torch.int32)[: self.B], view("pad", torch.uint8)
            p.dur_pad = view("dur", torch.int32) if self.forced else None
            p.f0e, p.maps = None, None
            p.spk = view("spk", torch.float32).view(self.B, self.S) if self.S else None
            self.prep = p
            self.seed_word = torch.zeros(1, dtype=torch.int32, device=dev)
            self.status = torch.zeros(1, dtype=torch.int32, device=dev)  # this runner's own status word (violations are attributed to ITS batches)
            self._seq_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            if self._ingraph:
                self._src_dev, self._seq_host_dev = ops.host_device_ptr(self._host[0]), ops.host_device_ptr(self._seq_host)
            self.stream.wait_stream(torch.cuda.current_stream(dev))  # the zero fills above ran on the caller's stream
            # warm-up on a minimal valid batch (one phoneme of duration 1 per utterance), then capture
            self.load([np.ones(1, dtype=np.int64)] * self.B, [np.ones(1, dtype=np.int64)] * self.B if self.forced else None,
                      [np.ones(self.S, dtype=np.float32)] * self.B if self.S else None)
            # the eager warm-up pass does the library's lazy one-time setup (per-kernel dynamic-LDS opt-ins, which a capture must not contain) for
            # the kernel forms THIS geometry selects: once per (plan, geometry) -- the other runners of a bucket (the decode driver keeps one per
            # stream) capture straight away (round 5: 16 warm-ups of a first decode() call -> 4)
            warm_key = (self.B, self.T, caps.lmax, caps.frames, caps.bounds.tobytes(), caps.tail_from, self.forced, int(dropout_mode), self.S)
            warmed = plan.__dict__.setdefault("_runner_warm", set())
            if warm_key not in warmed:
                with torch.cuda.stream(self.stream):
                    if self._ingraph:
                        self._feed(False)
                    run(plan, p, dropout_mode, seed=seed, seed_dev=self.seed_word, caps=caps, status=self.status)
                self.stream.synchronize()
                if self.forced and int(self.status.item()):
                    raise ops._lib.FclError("fcl-taco2_amd: BatchRunner warm-up failed: %s" % ops.status_message(int(self.status.item()) & 0xFFFFFFFF))
                warmed.add(warm_key)
            else:
                self.stream.synchronize()
            self.status.zero_()  # (predicted durations: the warm-up ids need not predict valid ones)
            self.graph = torch.cuda.CUDAGraph()
            with capture(self.graph, self.stream, mempool):
                if self._ingraph:
                    ops.feed_copy(self._dev, self._src_dev, self._nbytes, self._seq_dev, self._seq_host_dev, self.seed_word)  # + the seed bump
                else:
                    ops.u32_add(self.seed_word, 1)
                self.mel, self._frames = run(plan, p, dropout_mode, seed=seed, seed_dev=self.seed_word, caps=caps, status=self.status)
                self.out = None
                if pack_outputs:
                    n, b1 = self.mel.numel(), self.B + 1
                    self.out = torch.empty(packed_words(self.mel.shape[0], self.mel.shape[1], self.B), dtype=torch.float32, device=dev)
                    self.out[:n].view_as(self.mel).copy_(self.mel)
                    self.out[n : n + b1].view(torch.int32).copy_(self._frames.utt_frame0[:b1])
                    self.out[n + b1 : n + b1 + 1].view(torch.int32).copy_(self.status)
        self.n_loaded = 0

    def _feed(self, bump):
        """The feed node outside the graph (warm-up): same kernel, counted like a launch."""
        ops.feed_copy(self._dev, self._src_dev, self._nbytes, self._seq_dev, self._seq_host_dev, self.seed_word if bump else None)
        self._launched += 1

    def _wait_consumed(self):
        """In-graph feed: the block may be repacked once the feed node of the last launch has read it (normally long ago: it is that pass's first node)."""
        if int(self._seq_np[0]) == self._launched:
            return
        t0 = time.perf_counter()
        while int(self._seq_np[0]) != self._launched:
            time.sleep(0)  # (yields the GIL: the decode driver's writer thread runs beside this one)
            if time.perf_counter() - t0 > 60.0:
                raise ops._lib.FclError("fcl-taco2_amd: BatchRunner: the feed node of launch %d never ran (sequence word %d; device hung?)"
                                        % (self._launched, int(self._seq_np[0])))

    def load(self, xs, durs=None, spembs=None):
        """Hand one batch to the graph's input block: host packing into the pinned block the graph's first node reads (FCL_FEED_INGRAPH=0: + ONE
        non-blocking copy on this runner's stream, ordered before the next replay).
        spembs: one speaker-embedding vector per utterance (models built with spk_embed_dim)."""
        nb = len(xs)
        if (spembs is not None) != bool(self.S):
            raise ValueError("BatchRunner: speaker embeddings %s" % ("missing (the model has spk_embed_dim=%d)" % self.S if self.S else "given to a model without spk_embed_dim"))
        if nb > self.B or nb == 0:
            raise ValueError("BatchRunner: %d utterances, capacity %d" % (nb, self.B))
        if (durs is not None) != self.forced:
            raise ValueError("BatchRunner(forced=%s): durations %s" % (self.forced, "missing" if self.forced else "not taken (the graph predicts them)"))
        B, T, n = self.B, self.T, self.B * self.T
        # every check first: the pinned block is only repacked once the whole batch is known to fit (a load() that raises leaves the previous
        # batch in place, so a replay() after a caught error still runs a consistent block)
        as_np = lambda v: v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        xs = [as_np(x).reshape(-1) for x in xs]
        ln = np.fromiter((x.shape[0] for x in xs), np.int32, nb)
        if int(ln.max()) > T or int(ln.min()) < 1:
            raise ValueError("BatchRunner: utterances of %d..%d phonemes, capacity %d" % (int(ln.min()), int(ln.max()), T))
        dcat = None
        if durs is not None:
            if len(durs) != nb or any(np.asarray(d).size != k for d, k in zip(durs, ln)):
                raise ValueError("duration count != phoneme count")
            dcat = np.concatenate([np.asarray(d).reshape(-1) for d in durs])
        spk_rows = None
        if self.S:
            spk_rows = [as_np(v).reshape(-1) for v in spembs]
            if len(spk_rows) != nb or any(v.shape[0] != self.S for v in spk_rows):
                raise ValueError("BatchRunner: speaker embeddings must be %d vectors of %d values" % (nb, self.S))
        j = self._slot % len(self._host)
        self._slot += 1
        if self._ingraph:
            self._wait_consumed()
        elif self._host_ev[j] is not None:
            self._host_ev[j].synchronize()  # the copy that last read this staging buffer (depth loads ago)
        hb = self._host[j].numpy()
        seg = lambda name, dt: hb[self._off[name][0] : self._off[name][0] + self._off[name][1]].view(dt)
        ids, dur, lens = seg("ids", np.int64), seg("dur", np.int32), seg("lens", np.int32)
        lens[:nb] = ln
        lens[nb:] = 0
        b_of, t_of, base = self._rows
        lfull = lens[:B][b_of]
        valid = t_of < lfull  # the non-padded rows, row-major: exactly the order of the concatenated utterances
        ids[:] = 0
        ids[valid] = np.concatenate(xs)
        if dcat is not None:
            dur[:] = 0
            dur[valid] = dcat
        if self.S:
            sp = seg("spk", np.float32).reshape(B, self.S)
            sp[nb:] = 0.0
            for i_, v in enumerate(spk_rows):
                sp[i_] = v
        seg("seg_lo", np.int32)[:] = base
        np.add(base, lfull, out=seg("seg_hi", np.int32))
        np.logical_not(valid, out=seg("pad", np.uint8).view(np.bool_))
        if not self._ingraph:
            with torch.cuda.stream(self.stream):
                self._dev.copy_(self._host[j], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self._host_ev[j] = ev
        self.n_loaded = nb

    def replay(self):
        """Enqueue one pass over the loaded batch; returns the static mel buffer [caps.frames, odim] (rows past the batch's total are not valid)."""
        with torch.cuda.stream(self.stream):
            self.graph.replay()
        self._launched += 1
        return self.mel

    def frames(self):
        """Synchronising: per-utterance frame counts of the last replay (raises FclError on a violated capacity / zero duration)."""
        self.stream.synchronize()
        self._frames._host = None
        return self._frames.resolve()[: self.n_loaded]
