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
import numpy as np
import torch

from . import ops
from .plan import LN_EPS


class RowMaps(object):
    """Integer maps between (utterance, phoneme) rows, sorted decoder rows and output frames."""
    __slots__ = ("src_rows", "order", "dur_sorted", "frame_off_sorted", "live_rows", "utt_frames", "n_frames", "lmax",
                 "frame_lo", "frame_hi")


def build_row_maps(lens, durs, t_max):
    """lens [B]; durs: list of int arrays (len_b each).  Raises AssertionError on a zero duration, like the
    reference's `assert ds_nonzeros.shape[0] == hs.shape[0]` (decoder_sa_kd.py:739)."""
    B = len(lens)
    src, dur, foff, utt_frames = [], [], [], []
    base = 0
    for b in range(B):
        d = np.asarray(durs[b]).reshape(-1).astype(np.int64)
        assert d.shape[0] == lens[b], "duration count != phoneme count"
        assert (d > 0).all(), "zero duration: ds_nonzeros.shape[0] != hs.shape[0]"
        src.append(b * t_max + np.arange(lens[b], dtype=np.int64))
        dur.append(d)
        foff.append(base + np.concatenate([[0], np.cumsum(d)[:-1]]))  # exclusive cumsum (H10)
        utt_frames.append(int(d.sum()))
        base += utt_frames[-1]
    src, dur, foff = np.concatenate(src), np.concatenate(dur), np.concatenate(foff)
    order = np.argsort(-dur, kind="stable")
    m = RowMaps()
    m.order = order
    m.src_rows = src[order].astype(np.int32)
    m.dur_sorted = dur[order].astype(np.int32)
    m.frame_off_sorted = foff[order].astype(np.int32)
    m.lmax = int(m.dur_sorted[0])
    # live_rows[t] = #rows with dur > t  (rows sorted descending => a prefix)
    m.live_rows = np.ascontiguousarray((m.dur_sorted[None, :] > np.arange(m.lmax)[:, None]).sum(axis=1).astype(np.int32))
    m.utt_frames = utt_frames
    m.n_frames = base
    starts = np.concatenate([[0], np.cumsum(utt_frames)])
    m.frame_lo = np.repeat(starts[:-1], utt_frames).astype(np.int32)
    m.frame_hi = np.repeat(starts[1:], utt_frames).astype(np.int32)
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


def encode(plan, ids_padded, lens, bilstm_algo=0):
    """H1-H3 for a padded id matrix [B, T] (device int64): returns hs [B*T, C], seg_lo/seg_hi, pad mask."""
    B, T = ids_padded.shape
    dev = plan.device
    lens_np = np.asarray(lens, dtype=np.int64)
    rows = np.arange(B * T, dtype=np.int64)
    b_of = rows // T
    seg_lo = torch.from_numpy((b_of * T).astype(np.int32)).to(dev)
    seg_hi = torch.from_numpy((b_of * T + lens_np[b_of]).astype(np.int32)).to(dev)
    pad = torch.from_numpy(((rows % T) >= lens_np[b_of]).astype(np.uint8)).to(dev)
    x = ops.embedding(ids_padded.reshape(-1), plan.embed)
    for cv in plan.enc_convs:
        x = ops.conv1d(x, cv.wp, cv.bias, seg_lo, seg_hi, ops.ACT_RELU)
    lens_dev = torch.from_numpy(lens_np.astype(np.int32)).to(dev)
    bl = plan.blstm
    hs = ops.bilstm(x, lens_dev, bl["w_ih_f"], bl["w_hh_f"], bl["b_f"], bl["w_ih_r"], bl["w_hh_r"], bl["b_r"], B, T, bilstm_algo)
    return hs, seg_lo, seg_hi, pad


def synthesize(plan, xs, durs=None, f0=None, energy=None, dropout_mode=ops.DROP_RNG, prenet_keep=None, seed=0,
               bilstm_algo=0, return_intermediates=False):
    """xs: list of 1-D int64 id arrays/tensors; durs: optional list of forced durations (else predicted).
    f0/energy: optional lists of [T] arrays replacing the predictors (inference(f0=..., energy=...)).
    prenet_keep: optional uint8 [Lmax, 2, N, P] in (utterance, phoneme) row order (FCL_DROP_MASK).
    Returns a list of mel tensors [L_b, odim] (views into one packed device buffer)."""
    hp, dev = plan.hp, plan.device
    B = len(xs)
    lens = [int(len(x)) for x in xs]
    T = max(lens)
    ids = np.zeros((B, T), dtype=np.int64)
    for b, x in enumerate(xs):
        ids[b, : lens[b]] = x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    with torch.cuda.device(dev):
        ids_dev = torch.from_numpy(ids).to(dev)
        hs, seg_lo, seg_hi, pad = encode(plan, ids_dev, lens, bilstm_algo)
        inter = {"hs": hs} if return_intermediates else None
        if durs is None:
            d_log = _predictor_scalar(plan.duration, hs, seg_lo, seg_hi, None)
            d_int = ops.duration_round(d_log, False, 1.0, pad)
            d_host = d_int.cpu().numpy().reshape(B, T)  # the one host sync of the predicted-duration path
            durs = [d_host[b, : lens[b]] for b in range(B)]
            if inter is not None:
                inter["d_log"], inter["d_int"] = d_log, d_int
        if f0 is None:
            p = _predictor_scalar(plan.pitch, hs, seg_lo, seg_hi, pad)
            e = _predictor_scalar(plan.energy, hs, seg_lo, seg_hi, pad)
        else:
            pe = np.zeros((2, B, T), dtype=np.float32)
            for b in range(B):
                pe[0, b, : lens[b]] = np.asarray(f0[b]).reshape(-1)
                pe[1, b, : lens[b]] = np.asarray(energy[b]).reshape(-1)
            pe = torch.from_numpy(pe).to(dev)
            p, e = pe[0].reshape(-1).contiguous(), pe[1].reshape(-1).contiguous()
        att, p_emb, e_emb = ops.variance_embed_add(hs, p, e, plan.pitch_embed_w, plan.pitch_embed_b, plan.energy_embed_w,
                                                   plan.energy_embed_b, seg_lo, seg_hi, want_embs=return_intermediates)
        if inter is not None:
            inter.update(p_outs=p, e_outs=e, p_embs=p_emb, e_embs=e_emb)
        maps = build_row_maps(lens, durs, T)
        att_c = ops.gather_rows(att, torch.from_numpy(maps.src_rows).to(dev))
        dur_dev = torch.from_numpy(maps.dur_sorted).to(dev)
        foff_dev = torch.from_numpy(maps.frame_off_sorted).to(dev)
        keep_dev = None
        if hp.dropout_rate <= 0.0:
            dropout_mode = ops.DROP_NONE
        if dropout_mode == ops.DROP_MASK:
            keep = np.ascontiguousarray(np.asarray(prenet_keep)[: maps.lmax][:, :, maps.order, :])  # to sorted row order
            keep_dev = torch.from_numpy(keep).to(dev)
        before = ops.decoder_loop(plan.decoder, att_c, dur_dev, maps.live_rows, foff_dev, maps.n_frames,
                                  dropout_mode=dropout_mode, prenet_keep=keep_dev, seed=seed)
        f_lo, f_hi = torch.from_numpy(maps.frame_lo).to(dev), torch.from_numpy(maps.frame_hi).to(dev)
        x = before
        n_post = len(plan.postnet)
        for i, cv in enumerate(plan.postnet):
            last = i == n_post - 1
            x = ops.conv1d(x, cv.wp, cv.bias, f_lo, f_hi, ops.ACT_NONE if last else ops.ACT_TANH, residual=before if last else None)
        after = x
        mels, s = [], 0
        for n in maps.utt_frames:
            mels.append(after[s : s + n])
            s += n
        if inter is not None:
            inter.update(before=before, after=after, maps=maps, T=T)
            return mels, inter
        return mels
