"""Teacher-forced `forward()` on the HIP path, evaluation mode.

Restates `Tacotron2_sa.forward` of the three reference classes for `model.eval()` + `torch.no_grad()` — the mode the
reference's `CustomEvaluator` runs every epoch (tts.py:76-108): BatchNorm uses running statistics, every nn.Dropout is
off, zoneout takes its expectation form (decoder_sa.py:96), the prenet's dropout stays on (decoder_sa.py:156-158).
  teacher     nets/teacher_training/e2e_tts_tacotron2_sa.py:520-622             -> named losses
  kd_teacher  nets/knowledge_distillation/e2e_tts_tacotron2_sa_kd_teacher.py:521-603 -> the 5-tuple of distillation items
  student     nets/knowledge_distillation/e2e_tts_tacotron2_sa_kd_student.py:673-802 -> named losses incl. the KD terms
Batched `forward()` keeps the reference's padding semantics: convolutions run over the zero-padded [B, Tmax] /
[B, Lmax] rows WITHOUT masking between layers (the padding leak of SURVEY.md §7), unlike the synthesis path.
The train-mode forward (batch-statistics BN, dropout everywhere, sampled zoneout) and the backward pass live in training.py (TrainEngine).
"""
import numpy as np
import torch

from . import ops
from .hparams import output_act_code
from .engine import _predictor_scalar


class ForwardResult(object):
    pass


def _dev(t, device, dtype=None):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device).contiguous()


def forward_pass(plan, batch, dropout_mode=ops.DROP_RNG, prenet_keep=None, seed=0):
    """Runs H1-H11 teacher-forced.  batch: the converter's dict (reference tts.py:277-305).  Returns a ForwardResult
    with every tensor the losses / distillation items need, in padded row layouts [B*T, .] and [B*L, .]."""
    if plan.hp.reduction_factor != 1:
        raise NotImplementedError("fcl-taco2_amd: the no-gradient teacher-forced forward (model.eval(); model(**batch)) covers reduction_factor 1; with r > 1 "
                                  "the plug-in classes run TrainEngine.forward_backward(batch, mode='eval') (nets/base.py forward)")
    hp, dev = plan.hp, plan.device
    ilens = [int(v) for v in batch["ilens"]]
    olens = [int(v) for v in batch["olens"]]
    B, T, L = len(ilens), max(ilens), max(olens)
    r = ForwardResult()
    r.B, r.T, r.L, r.ilens, r.olens = B, T, L, ilens, olens
    with torch.cuda.device(dev):
        xs = _dev(batch["xs"][:, :T], dev, torch.int64).reshape(-1)
        rows = np.arange(B * T)
        b_of = rows // T
        lens_np = np.asarray(ilens)
        seg_lo = torch.from_numpy((b_of * T).astype(np.int32)).to(dev)
        seg_hi = torch.from_numpy((b_of * T + T).astype(np.int32)).to(dev)  # whole padded row: the reference does not mask
        pad_np = (rows % T) >= lens_np[b_of]
        r.enc_pad = torch.from_numpy(pad_np.astype(np.uint8)).to(dev)
        r.enc_valid = torch.from_numpy((~pad_np).astype(np.uint8)).to(dev)
        # H1-H3
        emb = ops.embedding(xs, plan.embed)
        r.enc_taps = [emb]
        x = emb
        for cv in plan.enc_convs:
            x = ops.conv1d(x, cv.wp, cv.bias, seg_lo, seg_hi, ops.ACT_RELU, residual=x if hp.use_residual else None)  # convs[i](xs) + xs
            r.enc_taps.append(x)
        bl = plan.blstm
        lens_dev = torch.from_numpy(lens_np.astype(np.int32)).to(dev)
        hs = x
        for bl in plan.blstm_layers:
            hs = ops.bilstm(hs, lens_dev, bl["w_ih_f"], bl["w_hh_f"], bl["b_f"], bl["w_ih_r"], bl["w_hh_r"], bl["b_r"], B, T)
        r.hs = hs  # (the encoder-KD tap: BEFORE the speaker embedding is appended, encoder_sa_kd.py:178-188)
        if hp.spk_embed_dim is not None:  # hs <- cat[hs, F.normalize(spembs)] (..._sa.py:555-557)
            if batch.get("spembs") is None:
                raise ValueError("fcl-taco2_amd: the model was built with spk_embed_dim=%d: forward() needs spembs" % hp.spk_embed_dim)
            hs = ops.concat_spk(hs, _dev(batch["spembs"], dev, torch.float32), T)[0]
        # H4/H5 (log-domain duration output, pitch, energy; masked_fill on padded positions)
        r.d_outs = _predictor_scalar(plan.duration, hs, seg_lo, seg_hi, r.enc_pad)
        r.p_outs = _predictor_scalar(plan.pitch, hs, seg_lo, seg_hi, r.enc_pad)
        r.e_outs = _predictor_scalar(plan.energy, hs, seg_lo, seg_hi, r.enc_pad)
        # embeds of the GROUND-TRUTH f0 / energy (..._sa.py:582-583)
        f0 = _dev(batch["f0"][:, :T], dev, torch.float32).reshape(-1)
        en = _dev(batch["energy"][:, :T], dev, torch.float32).reshape(-1)
        r.f0, r.energy = f0, en
        att, r.p_embs, r.e_embs = ops.variance_embed_add(hs, f0, en, plan.pitch_embed_w, plan.pitch_embed_b, plan.energy_embed_w,
                                                         plan.energy_embed_b, seg_lo, seg_hi, want_embs=True)
        # H9: row compaction (row-major over (b, t)), duration sort, frame offsets in the padded [B, L] layout
        nzm = np.asarray(batch["non_zero_lens_mask"])[:, :T] != 0
        dsn = np.asarray(batch["ds_nonzeros"]).astype(np.int64)
        src = np.flatnonzero(nzm.reshape(-1))
        assert src.shape[0] == dsn.shape[0], "hs.shape[0] != len(ds_nonzeros)"  # decoder_sa.py:468
        b_row = src // T
        starts = np.zeros(B + 1, dtype=np.int64)
        np.add.at(starts, b_row + 1, dsn)
        assert np.array_equal(starts[1:], np.asarray(olens)), "sum of durations != olens"
        excl = np.cumsum(dsn) - dsn
        first = np.concatenate([[0], np.cumsum(np.bincount(b_row, minlength=B))[:-1]])
        in_utt = excl - excl[first[b_row]]  # exclusive cumsum restarted per utterance
        foff = b_row * L + in_utt
        order = np.argsort(-dsn, kind="stable")
        dur_s = dsn[order].astype(np.int32)
        lmax = int(dur_s[0])
        live = np.ascontiguousarray((dur_s[None, :] > np.arange(lmax)[:, None]).sum(1).astype(np.int32))
        att_c = ops.gather_rows(att, torch.from_numpy(src[order].astype(np.int32)).to(dev))
        new_ys = _dev(batch["new_ys"], dev, torch.float32)
        assert new_ys.shape[1] == lmax
        tys = ops.gather_rows(new_ys.reshape(new_ys.shape[0], -1), torch.from_numpy(order.astype(np.int32)).to(dev))
        keep_dev = None
        if hp.dropout_rate <= 0.0:
            dropout_mode = ops.DROP_NONE
        if dropout_mode == ops.DROP_MASK:
            keep_dev = torch.from_numpy(np.ascontiguousarray(np.asarray(prenet_keep)[:lmax][:, :, order, :])).to(dev)
        before, taps = ops.decoder_loop(plan.decoder, att_c, torch.from_numpy(dur_s).to(dev), live,
                                        torch.from_numpy(foff[order].astype(np.int32)).to(dev), B * L, teacher_ys=tys,
                                        dropout_mode=dropout_mode, prenet_keep=keep_dev, seed=seed, want_taps=True, zero_init=True)
        # H11 over the zero-padded [B, L] rows, all five layer outputs kept (decoder_sa_kd.py:344-352)
        frows = np.arange(B * L)
        f_lo = torch.from_numpy(((frows // L) * L).astype(np.int32)).to(dev)
        f_hi = torch.from_numpy(((frows // L) * L + L).astype(np.int32)).to(dev)
        r.frame_valid = torch.from_numpy(((frows % L) < np.asarray(olens)[frows // L]).astype(np.uint8)).to(dev)
        post = []
        x = before
        n_post = len(plan.postnet)
        for i, cv in enumerate(plan.postnet):
            x = ops.conv1d(x, cv.wp, cv.bias, f_lo, f_hi, ops.ACT_NONE if i == n_post - 1 else ops.ACT_TANH)
            post.append(x)
        # after = before + conv4 (one more pass of the residual epilogue would need conv4 twice; a plain add kernel suffices)
        r.after = ops.conv1d(post[-2], plan.postnet[-1].wp, plan.postnet[-1].bias, f_lo, f_hi, ops.ACT_NONE, residual=before) if n_post >= 2 else None
        r.before = before
        if hp.output_activation is not None:  # decoder_sa.py:538-540: both outputs activated AFTER the postnet has read the raw `before`
            r.after = ops.act_fwd(r.after, output_act_code(hp))
            r.before = ops.act_fwd(before, output_act_code(hp))
        r.dec_taps = list(taps) + post
        r.ys = _dev(batch["ys"][:, :L], dev, torch.float32).reshape(B * L, -1)
        r.ds = _dev(batch["extras"][:, :T], dev, torch.float32).reshape(-1)
    return r


class LossAccumulator(object):
    """All loss sums land in one device buffer; a single D2H copy turns them into the reference's named scalars."""

    def __init__(self, device, n=32):
        self.buf = torch.zeros(n, 3, dtype=torch.float64, device=device)
        self.names = []

    def add(self, name, a, b, valid, b_log_offset=None):
        i = len(self.names)
        self.names.append(name)
        if a.dim() == 1:
            a, b = a.reshape(-1, 1), b.reshape(-1, 1)
        ops.masked_l1_mse(a.contiguous(), b.contiguous(), valid, self.buf[i], b_log_offset)

    def means(self):
        host = self.buf[: len(self.names)].cpu().numpy()
        return {n: (host[i, 0] / host[i, 2], host[i, 1] / host[i, 2]) for i, n in enumerate(self.names)}


def base_losses(acc, r, use_masking=True):
    """Tacotron2Loss + duration + pitch + energy (..._sa.py:601-613).  use_masking False: the mel and prosody terms run over the padded tensors
    (row_valid None = every row); the duration loss is masked regardless (..._sa.py:561-565)."""
    fv, ev = (r.frame_valid, r.enc_valid) if use_masking else (None, None)
    acc.add("after", r.after, r.ys, fv)
    acc.add("before", r.before, r.ys, fv)
    acc.add("dur", r.d_outs, r.ds, r.enc_valid, b_log_offset=1.0)  # DurationPredictorLoss: MSE vs log(d + 1)
    acc.add("pitch", r.p_outs, r.f0, ev)
    acc.add("energy", r.e_outs, r.energy, ev)


def finish_base(m):
    rep = dict(l1_loss=m["after"][0] + m["before"][0], mse_loss=m["after"][1] + m["before"][1], dur_loss=m["dur"][1],
               pitch_loss=m["pitch"][1], energy_loss=m["energy"][1])
    rep["loss"] = rep["l1_loss"] + rep["mse_loss"] + rep["dur_loss"] + rep["pitch_loss"] + rep["energy_loss"]
    return rep


def teacher_forward(plan, batch, **kw):
    plan.hp.check_loss_supported()  # (use_weighted_masking raises instead of silently computing another objective)
    r = forward_pass(plan, batch, **kw)
    acc = LossAccumulator(plan.device)
    base_losses(acc, r, plan.hp.use_masking)
    return finish_base(acc.means()), r


def knowledge_tuple(r):
    """The KD teacher's 5-tuple (..._kd_teacher.py:597-603), tensors shaped as the reference's."""
    B, T, L = r.B, r.T, r.L
    e = lambda x: x.reshape(B, T, -1)
    f = lambda x: x.reshape(B, L, -1)
    return (f(r.after), f(r.before), [e(t) for t in r.enc_taps] + [e(r.hs)], [f(t) for t in r.dec_taps],
            [e(r.d_outs), e(r.p_outs), e(r.e_outs), e(r.p_embs), e(r.e_embs)])


def student_forward(plan, batch, teacher_knowledge, share_proj=True, distill=(True, True, True, True), **kw):
    """distill = (output, encoder, decoder, prosody) flags (..._kd_student.py:778-797)."""
    plan.hp.check_loss_supported()
    r = forward_pass(plan, batch, **kw)
    dev = plan.device
    acc = LossAccumulator(dev, 48)
    base_losses(acc, r, plan.hp.use_masking)
    t_after, t_before, t_enc, t_dec, t_pro = teacher_knowledge
    flat = lambda t: _dev(t, dev, torch.float32).reshape(-1, t.shape[-1])
    P = plan.proj
    lin = lambda x, key: ops.linear(x, P[key])
    if share_proj:
        cp, lp, pp = ["enc.convs_proj.0"] * 3, ["dec.lstm_proj"] * 2, ["dec.post_proj"] * 4
    else:
        cp = ["enc.convs_proj.%d" % i for i in range(3)]
        lp, pp = ["dec.lstm0_proj", "dec.lstm1_proj"], ["dec.post%d_proj" % i for i in range(4)]
    if distill[0]:
        fv = r.frame_valid if plan.hp.use_masking else None  # Tacotron2Loss_KD follows use_masking too (..._kd_student.py:120-125)
        acc.add("o_after", r.after, flat(t_after), fv)
        acc.add("o_before", r.before, flat(t_before), fv)
    if distill[1]:
        s_enc = [lin(r.enc_taps[0], "enc.embed_proj")] + [lin(r.enc_taps[1 + i], cp[i]) for i in range(3)] + [lin(r.hs, "enc.blstm_proj")]
        for i, (s, t) in enumerate(zip(s_enc, t_enc)):
            acc.add("enc%d" % i, s, flat(t), r.enc_valid)
    if distill[2]:
        s_dec = [lin(r.dec_taps[0], "dec.prenet_proj"), lin(r.dec_taps[1], lp[0]), lin(r.dec_taps[2], lp[1])] + \
                [lin(r.dec_taps[3 + i], pp[i]) for i in range(4)] + [r.dec_taps[7]]
        for i, (s, t) in enumerate(zip(s_dec, t_dec)):
            acc.add("dec%d" % i, s, flat(t), r.frame_valid)
    if distill[3]:
        s_pro = [r.d_outs, r.p_outs, r.e_outs, lin(r.p_embs, "pemb_proj"), lin(r.e_embs, "eemb_proj")]
        for i, (s, t) in enumerate(zip(s_pro, t_pro)):
            acc.add("pro%d" % i, s.reshape(-1, 1) if s.dim() == 1 else s, flat(t), r.enc_valid)
    m = acc.means()
    rep = finish_base(m)
    if distill[0]:
        rep["output_l1_loss"] = m["o_after"][0] + m["o_before"][0]
        rep["output_mse_loss"] = m["o_after"][1] + m["o_before"][1]
        rep["loss"] += rep["output_l1_loss"] + rep["output_mse_loss"]
    for flag, key, pre, n in ((distill[1], "encoder_loss", "enc", 5), (distill[2], "decoder_loss", "dec", 8), (distill[3], "prosody_loss", "pro", 5)):
        if flag:
            rep[key] = sum(m["%s%d" % (pre, i)][1] for i in range(n))
            rep["loss"] += rep[key]
    return rep, r
