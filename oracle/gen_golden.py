#!/usr/bin/env python3
"""gen_golden.py — run the REAL reference (/root/reference) on CPU and dump golden vectors.

TEST TOOLING.  Runs only in the container that has /root/reference (never on the GPU box).  It imports
the reference's `nets/**`, `variance_predictor.py` and `tts.py:CustomConverter` UNMODIFIED through
`oracle/espnet_shim` (+ attribute-auto-stub modules for chainer/apex/kaldiio/... that tts.py imports at
module level but the converter never touches), loads closed-form weights
(`fcl_taco2_amd.synthetic.closed_form_state_dict`) with `load_state_dict`, and writes small `.npz` /
`.json` fixtures to tests/golden/.  Nothing of the reference's source text is written anywhere.

    python -B oracle/gen_golden.py            # regenerates every set (≈1 min)

Sets: manifest, G1 (tiny dims, every stage, inference+forward, teacher+student,
share_proj on/off), G2 (full S dims, 1 utt, prenet dropout 0), G2T (full T dims), G2B (3 utterances for
the batched extension), G3 (injected prenet dropout masks), G4 (integer: duration rounding, converter
layout), G5 (training losses + a few gradients, eval-dropout-off), G6 (padding leak + zero-duration
failure record).
"""
import argparse
import dataclasses
import contextlib
import io
import json
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "espnet_shim"))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import fcl_taco2_amd  # noqa: E402,F401
from fcl_taco2_amd import hparams as HP  # noqa: E402
from fcl_taco2_amd import synthetic as SYN  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


class _AutoStub(types.ModuleType):
    """Module whose every attribute is a fresh empty class (usable as base class or callable)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _install_stubs():
    for m in ["chainer", "chainer.training", "chainer.training.extensions", "kaldiio", "matplotlib",
              "tensorboardX", "apex", "apex.amp", "espnet.asr", "espnet.asr.asr_utils",
              "espnet.asr.pytorch_backend", "espnet.asr.pytorch_backend.asr_init", "espnet.utils.dataset",
              "espnet.utils.dynamic_import", "espnet.utils.training", "espnet.utils.training.evaluator",
              "espnet.utils.deterministic_utils", "espnet.utils.training.train_utils",
              "espnet.utils.training.iterators", "espnet.utils.training.tensorboard_logger"]:
        if m not in sys.modules:
            sys.modules[m] = _AutoStub(m)
    sys.modules["chainer"].training = sys.modules["chainer.training"]
    sys.modules["chainer.training"].extensions = sys.modules["chainer.training.extensions"]
    sys.modules["apex"].amp = sys.modules["apex.amp"]
    sys.modules["matplotlib"].use = lambda *a, **k: None


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ns(hp, **extra):
    d = {k: getattr(hp, k) for k in (
        "embed_dim elayers eunits econv_layers econv_chans econv_filts dlayers dunits prenet_layers "
        "prenet_units postnet_layers postnet_chans postnet_filts use_batch_norm use_concate use_residual "
        "reduction_factor dropout_rate zoneout_rate use_masking duration_predictor_layers "
        "duration_predictor_chans duration_predictor_kernel_size duration_predictor_dropout_rate output_activation spk_embed_dim").split()}
    d["encoder_resume"] = None
    d.update(extra)
    return argparse.Namespace(**d)


COM = dict(use_fe_condition=True, append_position=True, distill_output_knowledge=True,
           distill_encoder_knowledge=True, distill_decoder_knowledge=True, distill_prosody_knowledge=True,
           is_train=True)


def build(role, hp, thp=None, share_proj=True):
    from nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_student import Tacotron2_sa as Student
    from nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_teacher import Tacotron2_sa as KDTeacher
    from nets.teacher_training.e2e_tts_tacotron2_sa import Tacotron2_sa as Teacher

    com = argparse.Namespace(share_proj=share_proj, **dict(COM, append_position=bool(getattr(hp, "append_position", True))))
    if role == "student":
        m = _quiet(Student, hp.idim, hp.odim, ns(hp), com, ns(thp))
        spec = HP.param_spec(hp, thp, share_proj)
    elif role == "kd_teacher":
        m = _quiet(KDTeacher, hp.idim, hp.odim, ns(hp), com)
        spec = HP.param_spec(hp)
    else:
        m = _quiet(Teacher, hp.idim, hp.odim, ns(hp), com)
        spec = HP.param_spec(hp)
    ref_spec = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert ref_spec == {k: tuple(v) for k, v in spec.items()}, "param_spec != reference state_dict manifest"
    sd = SYN.closed_form_state_dict(spec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m.eval()
    return m, spec


@contextlib.contextmanager
def injected_dropout(keep_masks):
    """Patch torch.nn.functional.dropout so every training=True call consumes the next closed-form keep
    mask.  keep_masks: iterator of uint8 arrays."""
    import torch.nn.functional as F

    orig = F.dropout
    it = iter(keep_masks)

    def fake(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        keep = torch.from_numpy(next(it)).to(x.dtype)
        assert keep.shape == x.shape
        return x * keep * (1.0 / (1.0 - p))

    F.dropout = fake
    try:
        yield
    finally:
        F.dropout = orig


def t2n(x):
    return x.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB  %s" % (name + ".npz", os.path.getsize(path) / 1024.0, sorted(arrs)[:6]))


TINY_S = HP.student_hparams(idim=12, odim=8, embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20,
                            postnet_chans=12, duration_predictor_chans=20, dropout_rate=0.0)
TINY_T = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28,
                            postnet_chans=20, duration_predictor_chans=20, dropout_rate=0.0)


def gen_manifest():
    man = {}
    S, T = HP.student_hparams(), HP.teacher_hparams()
    for tag, role, hp, thp, share in [("student_share", "student", S, T, True), ("student_noshare", "student", S, T, False),
                                      ("teacher", "teacher", T, None, True), ("kd_teacher", "kd_teacher", T, None, True)]:
        m, spec = build(role, hp, thp, share)
        man[tag] = {k: list(v) for k, v in spec.items()}
        man[tag + "_nparams"] = int(sum(p.numel() for p in m.parameters()))
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("manifest.json", {k: v for k, v in man.items() if k.endswith("_nparams")})


def stage_dump(m, x, dur):
    """Every tensor after every H-row of inference(), by calling the reference sub-modules in the order
    ..._kd_student.py:804-863 does."""
    from espnet.nets.pytorch_backend.nets_utils import make_pad_mask

    with torch.no_grad():
        emb = m.enc.embed(x.unsqueeze(0)).transpose(1, 2)
        c = emb
        convs = []
        for i in range(len(m.enc.convs)):
            c = m.enc.convs[i](c)
            convs.append(c)
        h = m.enc.inference(x)
        ilens = torch.LongTensor([h.shape[0]])
        pad = make_pad_mask(ilens)
        d_log = m.duration_predictor(h.unsqueeze(0), pad)
        d_int = m.duration_predictor.inference(h.unsqueeze(0), pad)
        p = m.pitch_predictor(h.unsqueeze(0), pad.unsqueeze(-1))
        e = m.energy_predictor(h.unsqueeze(0), pad.unsqueeze(-1))
        pe = m.pitch_embed(p.transpose(1, 2)).transpose(1, 2)
        ee = m.energy_embed(e.transpose(1, 2)).transpose(1, 2)
        after = m.inference(x, None, dur=dur)
    out = dict(x=t2n(x), dur=t2n(dur), embed=t2n(emb[0].t()), h=t2n(h), d_log=t2n(d_log[0]), d_int=t2n(d_int[0]),
               p_outs=t2n(p[0]), e_outs=t2n(e[0]), p_embs=t2n(pe[0]), e_embs=t2n(ee[0]), after=t2n(after))
    for i, cv in enumerate(convs):
        out["conv%d" % i] = t2n(cv[0].t())
    return out


def before_from(m, x, dur):
    """`before` (pre-postnet) mel: re-run dec.inference with the postnet output captured by a hook."""
    cap = {}
    hk = m.dec.postnet.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("before", inp[0].detach().clone()))
    with torch.no_grad():
        m.inference(x, None, dur=dur)
    hk.remove()
    return t2n(cap["before"][0].t())


def make_converter_batch(hp, seed, batch=4):
    from tts import CustomConverter

    xs, ys, ds, f0, en = SYN.training_batch(hp.odim, hp.idim, batch=batch, seed=seed)
    conv = CustomConverter(reduction_factor=1, use_fe_condition=True, append_position=True)
    b = conv([(xs, ys, None, ds, f0, en)])
    return (xs, ys, ds, f0, en), b


def gen_g1():
    rng = np.random.RandomState(11)
    x = torch.from_numpy(rng.randint(1, TINY_S.idim, size=7).astype(np.int64))
    dur = torch.tensor([1, 3, 2, 5, 1, 4, 2])
    for tag, role, hp, thp, share in [("student_share", "student", TINY_S, TINY_T, True),
                                      ("student_noshare", "student", TINY_S, TINY_T, False),
                                      ("teacher", "teacher", TINY_T, None, True)]:
        m, _ = build(role, hp, thp, share)
        d = stage_dump(m, x, dur)
        d["before"] = before_from(m, x, dur)
        save("g1_infer_" + tag, **d)
    # teacher-forced forward: kd_teacher 5-tuple -> student losses; and the plain teacher's losses
    raw, b = make_converter_batch(TINY_S, seed=7)
    kt, _ = build("kd_teacher", TINY_T)
    with torch.no_grad():
        know = kt(**b)
    flat = dict(t_after=t2n(know[0]), t_before=t2n(know[1]))
    for grp, items in (("t_enc", know[2]), ("t_dec", know[3]), ("t_pro", know[4])):
        for i, it in enumerate(items):
            flat["%s%d" % (grp, i)] = t2n(it)
    for share in (True, False):
        st, _ = build("student", TINY_S, TINY_T, share)
        with torch.no_grad():
            loss = st(teacher_knowledge=know, **b)
        rep = {list(d.keys())[0]: list(d.values())[0] for d in st.reporter.last}
        flat["student_%s_loss" % ("share" if share else "noshare")] = np.float32(loss.item())
        for k, v in rep.items():
            flat["student_%s_%s" % ("share" if share else "noshare", k)] = np.float32(v)
    te, _ = build("teacher", TINY_T)
    with torch.no_grad():
        loss = te(**b)
    for d in te.reporter.last:
        for k, v in d.items():
            flat["teacher_" + k] = np.float32(v)
    save("g1_forward", **flat)


def c1_inputs(hp, n=80, seed=137):
    x, d = SYN.utterance_c1(hp.idim, n, seed)
    return torch.from_numpy(x), torch.from_numpy(d)


def gen_g2_g3():
    S0 = HP.student_hparams(dropout_rate=0.0)
    T = HP.teacher_hparams()
    x, dur = c1_inputs(S0)
    m, _ = build("student", S0, T, True)
    with torch.no_grad():
        after = m.inference(x, None, dur=dur)
        h = m.enc.inference(x)
    save("g2_student_c1", x=t2n(x), dur=t2n(dur), after=t2n(after), before=before_from(m, x, dur), h=t2n(h))
    # G2B: three utterances of different lengths -> the batched extension must equal per-utterance inference
    xs, ds = SYN.batch_c2(S0.idim, batch=3, t_lo=20, t_hi=40, seed=99)
    d = {}
    for i, (xi, di) in enumerate(zip(xs, ds)):
        with torch.no_grad():
            d["after%d" % i] = t2n(m.inference(torch.from_numpy(xi), None, dur=torch.from_numpy(di)))
        d["x%d" % i], d["dur%d" % i] = xi, di
    save("g2b_student_batch3", **d)
    # G3: dropout 0.5 with injected masks
    S = HP.student_hparams()
    m, _ = build("student", S, T, True)
    n_steps, N = int(dur.max()), int((dur > 0).sum())
    keep = SYN.closed_form_keep_mask((n_steps, 2, N, S.prenet_units), seed=2024)
    with injected_dropout(keep[t, l] for t in range(n_steps) for l in range(2)):
        with torch.no_grad():
            after = m.inference(x, None, dur=dur)
    save("g3_student_c1_masked", x=t2n(x), dur=t2n(dur), after=t2n(after), keep_seed=np.int64(2024))
    # run-to-run stochasticity record (D8): two un-patched calls differ
    with torch.no_grad():
        a, b = m.inference(x, None, dur=dur), m.inference(x, None, dur=dur)
    rec = {"d8_two_calls_max_abs_diff": float((a - b).abs().max())}
    # G2T: full T dims, config[0] of BASELINE.json (batch 1, CPU)
    T0 = HP.teacher_hparams(dropout_rate=0.0)
    mt, _ = build("teacher", T0)
    with torch.no_grad():
        after = mt.inference(x, None, dur=dur)
    save("g2t_teacher_c1", x=t2n(x), dur=t2n(dur), after=t2n(after))
    return rec


def gen_g4():
    # duration rounding: ties in the linear domain (round-half-even), negatives, clamp
    lin = np.array([-3.0, -0.5, -0.49, 0.0, 0.49, 0.5, 0.51, 1.5, 2.5, 3.5, 4.5, 7.49, 7.5, 49.5, 50.5, 1e4], np.float32)
    lin_round = t2n(torch.clamp(torch.round(torch.from_numpy(lin)), min=0).long())
    logits = np.array([-20.0, -1.0, -0.2, 0.0, 0.3, 0.6931472, 1.0, 1.5, 2.0, 2.3978953, 3.0, 3.9, 5.0], np.float32)
    from espnet.nets.pytorch_backend.fastspeech.duration_predictor import DurationPredictor  # the shim's

    dp = DurationPredictor(4)
    dp.eval()
    ref = t2n(torch.clamp(torch.round(torch.from_numpy(logits).exp() - dp.offset), min=0).long())
    d = dict(lin=lin, lin_round=lin_round, logits=logits, logits_round=ref)
    # converter layout incl. zero-duration phonemes (B=4) — the real tts.py:CustomConverter
    raw, b = make_converter_batch(TINY_S, seed=7)
    xs, ys, ds, f0, en = raw
    for i in range(len(xs)):
        d["in_xs%d" % i], d["in_ys%d" % i], d["in_ds%d" % i], d["in_f0%d" % i], d["in_en%d" % i] = xs[i], ys[i], ds[i], f0[i], en[i]
    for k, v in b.items():
        d["out_" + k] = t2n(v)
    # a second, larger one at mel dim 80 for the index maps only
    raw2, b2 = make_converter_batch(HP.student_hparams(), seed=21, batch=6)
    for i in range(6):
        d["in2_ds%d" % i] = raw2[2][i]
    for k in ("non_zero_lens_mask", "ds_nonzeros", "output_masks", "position", "ilens", "olens"):
        d["out2_" + k] = t2n(b2[k])
    save("g4_integer", **d)


def gen_g5():
    """Training: total loss, every named loss and gradients of a few parameters for one B=4 batch,
    model.eval() (running-stat BN, nn.Dropout off, eval zoneout), prenet dropout 0."""
    raw, b = make_converter_batch(TINY_S, seed=7)
    te, _ = build("teacher", TINY_T)
    for p in te.parameters():
        p.grad = None
    loss = te(**b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()))
    for k in ["dec.feat_out.weight", "enc.embed.weight", "duration_predictor.linear.weight",
              "dec.lstm.0.cell.weight_hh", "dec.prenet.prenet.0.0.bias", "pitch_embed.0.weight",
              "dec.postnet.postnet.4.0.weight", "enc.blstm.weight_hh_l0_reverse"]:
        d["grad:" + k] = t2n(dict(te.named_parameters())[k].grad).copy()  # copy: clip_grad_norm_ below scales in place
    gn = torch.nn.utils.clip_grad_norm_(te.parameters(), 1.0)
    d["grad_norm"] = np.float32(float(gn))
    save("g5_teacher_train", **d)


def gen_g6(rec):
    m, _ = build("teacher", TINY_T)
    rng = np.random.RandomState(5)
    x0 = rng.randint(1, TINY_T.idim, size=9).astype(np.int64)
    x1 = rng.randint(1, TINY_T.idim, size=5).astype(np.int64)
    xs = torch.zeros(2, 9, dtype=torch.long)
    xs[0, :9], xs[1, :5] = torch.from_numpy(x0), torch.from_numpy(x1)
    with torch.no_grad():
        hs, hlens = m.enc(xs, [9, 5])
        h1 = m.enc.inference(torch.from_numpy(x1))
    save("g6_padding_leak", xs=t2n(xs), ilens=np.array([9, 5]), hs_batched=t2n(hs), h1_single=t2n(h1))
    rec["g6_leak_max_abs_on_shorter"] = float((hs[1, :5] - h1).abs().max())
    # zero predicted duration crashes reference inference
    try:
        with torch.no_grad():
            m.inference(torch.from_numpy(x1), None, dur=torch.tensor([2, 0, 1, 3, 1]))
        rec["zero_duration"] = "no error"
    except AssertionError:
        rec["zero_duration"] = "AssertionError"
    with open(os.path.join(OUT, "records.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("records.json", rec)


@contextlib.contextmanager
def injected_randomness(log):
    """Train-mode goldens: every Bernoulli draw of a training forward -- F.dropout (nn.Dropout and the prenet) and ZoneOutCell's
    `h.new(...).bernoulli_(p)` (decoder_sa.py:93) -- is replaced by closed_form_keep_mask(shape, seed=<draw counter>) and appended to `log`,
    so the reference, the oracle and the HIP path consume identical masks."""
    import torch.nn.functional as F

    orig_drop, orig_bern = F.dropout, torch.Tensor.bernoulli_

    def fake_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        keep = SYN.closed_form_keep_mask(tuple(x.shape), 1000 + len(log))
        log.append(keep)
        return x * torch.from_numpy(keep).to(x.dtype) * (1.0 / (1.0 - p))

    def fake_bernoulli_(self, p=0.5, generator=None):
        keep = SYN.closed_form_keep_mask(tuple(self.shape), 1000 + len(log))
        log.append(keep)
        self.copy_(torch.from_numpy(keep).to(self.dtype))
        return self

    F.dropout, torch.Tensor.bernoulli_ = fake_dropout, fake_bernoulli_
    try:
        yield
    finally:
        F.dropout, torch.Tensor.bernoulli_ = orig_drop, orig_bern


GRAD_KEYS = ["dec.feat_out.weight", "enc.embed.weight", "duration_predictor.linear.weight", "dec.lstm.0.cell.weight_hh",
             "dec.prenet.prenet.0.0.bias", "pitch_embed.0.weight", "dec.postnet.postnet.4.0.weight", "enc.blstm.weight_hh_l0_reverse",
             "enc.convs.1.1.weight", "dec.postnet.postnet.0.1.bias", "energy_predictor.conv.1.2.weight", "dec.lstm.1.cell.bias_ih"]
KD_KEYS = ["enc.embed_proj.weight", "enc.convs_proj.0.weight", "enc.blstm_proj.weight", "dec.prenet_proj.weight", "dec.lstm_proj.weight",
           "dec.post_proj.weight", "pemb_proj.weight", "eemb_proj.weight"]


def _grads(model, keys, d):
    named = dict(model.named_parameters())
    for k in keys:
        d["grad:" + k] = t2n(named[k].grad).copy()
    gn = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 1.0)
    d["grad_norm"] = np.float32(float(gn))


def _named_losses(model, d, prefix=""):
    for item in model.reporter.last:
        for k, v in item.items():
            d[prefix + k] = np.float32(v)


def gen_g7_g8_g9():
    """G7: teacher step in TRAIN mode (batch-stat BatchNorm, every dropout and zoneout draw injected) -- loss, gradients, running statistics.
    G8: student KD step, eval form (teacher knowledge from the eval-mode KD teacher): loss and gradients incl. the projections.
    G9: the reference's KD update (tts_distill.py:159-161) in train mode: frozen train-mode teacher -> student forward/backward."""
    T7 = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20,
                            duration_predictor_chans=20, dropout_rate=0.5)
    S7 = HP.student_hparams(idim=12, odim=8, embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12,
                            duration_predictor_chans=20, dropout_rate=0.5)
    raw, b = make_converter_batch(TINY_S, seed=7)
    # ---- G7
    te, _ = build("teacher", T7)
    te.train()
    log = []
    with injected_randomness(log):
        loss = te(**b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()), n_masks=np.int64(len(log)))
    _named_losses(te, d)
    _grads(te, GRAD_KEYS, d)
    sd = te.state_dict()
    for k in ("enc.convs.0.1.running_mean", "enc.convs.0.1.running_var", "dec.postnet.postnet.4.1.running_mean", "dec.postnet.postnet.4.1.running_var",
              "enc.convs.0.1.num_batches_tracked"):
        if k in sd:
            d["buf:" + k] = t2n(sd[k]).copy()
    for i, mk in enumerate(log):
        d["mask%03d" % i] = mk
    save("g7_teacher_train_mode", **d)
    # ---- G8
    kt, _ = build("kd_teacher", TINY_T)
    with torch.no_grad():
        know = kt(**b)
    st, _ = build("student", TINY_S, TINY_T, True)
    loss = st(teacher_knowledge=know, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()))
    _named_losses(st, d)
    _grads(st, [k for k in GRAD_KEYS] + KD_KEYS, d)
    save("g8_student_kd_eval", **d)
    # ---- G9
    kt, _ = build("kd_teacher", T7)
    for p in kt.parameters():
        p.requires_grad = False  # tts_distill.py:398
    kt.train()
    st, _ = build("student", S7, T7, True)
    st.train()
    tlog, slog = [], []
    with injected_randomness(tlog):
        know = kt(**b)
    with injected_randomness(slog):
        loss = st(teacher_knowledge=know, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()), n_tmasks=np.int64(len(tlog)), n_smasks=np.int64(len(slog)), t_after=t2n(know[0]),
             t_dec1=t2n(know[3][1]), t_pro3=t2n(know[4][3]))
    _named_losses(st, d)
    _grads(st, [k for k in GRAD_KEYS] + KD_KEYS, d)
    d["buf:teacher.enc.convs.0.1.running_mean"] = t2n(kt.state_dict()["enc.convs.0.1.running_mean"]).copy()
    for i, mk in enumerate(tlog):
        d["tmask%03d" % i] = mk
    for i, mk in enumerate(slog):
        d["smask%03d" % i] = mk
    save("g9_kd_step_train_mode", **d)


def gen_g10():
    """G10: the loss variants of `--use-masking False` (the reference's argparse default, ..._sa.py:251-262; the shipped yaml sets True): mel L1 / MSE
    and the prosody MSEs over the PADDED tensors, the output-KD term likewise; the duration loss and the encoder / decoder / prosody KD terms stay
    masked (..._kd_student.py:719, 134-179).  Teacher step and student KD step, eval form: named losses + gradients.  Also records what the
    reference does with `--use-weighted-masking True` (unreduced duration / prosody losses of different shapes are added: it fails)."""
    kw = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, use_masking=False)
    TU = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, **kw)
    SU = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, **kw)
    raw, b = make_converter_batch(TINY_S, seed=7)
    te, _ = build("teacher", TU)
    assert te.taco2_loss.use_masking is False
    loss = te(**b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()))
    _named_losses(te, d)
    _grads(te, GRAD_KEYS, d)
    save("g10_teacher_unmasked", **d)
    kt, _ = build("kd_teacher", TINY_T)
    with torch.no_grad():
        know = kt(**b)
    st, _ = build("student", SU, TU, True)
    loss = st(teacher_knowledge=know, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()))
    _named_losses(st, d)
    _grads(st, [k for k in GRAD_KEYS] + KD_KEYS, d)
    save("g10_student_kd_unmasked", **d)
    rec_path = os.path.join(OUT, "records.json")
    rec = json.load(open(rec_path)) if os.path.exists(rec_path) else {}
    try:
        from nets.teacher_training.e2e_tts_tacotron2_sa import Tacotron2_sa as Teacher

        tw = _quiet(Teacher, TU.idim, TU.odim, ns(TU, use_weighted_masking=True), argparse.Namespace(share_proj=True, **COM))
        tw.eval()
        float(tw(**b).mean())
        rec["use_weighted_masking"] = "runs"
    except Exception as e:  # noqa: BLE001 - the record is the exception itself
        rec["use_weighted_masking"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:160])
    with open(rec_path, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("records.json use_weighted_masking ->", rec["use_weighted_masking"])


def gen_g11():
    """G11: `--use-residual True` (the reference's argparse default; the shipped yaml sets false): `convs[i](xs) + xs` in the encoder
    (encoder_sa_kd.py:158-171, 213-214).  Teacher: inference mel + training step (eval form); student: KD step against a use_residual teacher."""
    kw = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, use_residual=True)
    TR = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, **kw)
    SR = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, **kw)
    rng = np.random.RandomState(11)
    x = torch.from_numpy(rng.randint(1, TR.idim, size=7).astype(np.int64))
    dur = torch.tensor([1, 3, 2, 5, 1, 4, 2])
    te, _ = build("teacher", TR)
    assert te.enc.use_residual is True
    with torch.no_grad():
        after = te.inference(x, None, dur=dur)
        h = te.enc.inference(x)
    d = dict(x=t2n(x), dur=t2n(dur), after=t2n(after), h=t2n(h))
    save("g11_teacher_residual", **d)
    raw, b = make_converter_batch(TINY_S, seed=7)
    # the plain teacher class cannot TRAIN with it: encoder_sa.py:137-138 adds in place (`xs += self.convs[i](xs)`) and autograd refuses
    rec_path = os.path.join(OUT, "records.json")
    rec = json.load(open(rec_path)) if os.path.exists(rec_path) else {}
    try:
        te(**b).backward()
        rec["use_residual_teacher_training"] = "runs"
    except RuntimeError as e:
        rec["use_residual_teacher_training"] = "RuntimeError: " + str(e).splitlines()[0][:120]
    with open(rec_path, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("records.json use_residual_teacher_training ->", rec["use_residual_teacher_training"])
    kt, _ = build("kd_teacher", TR)
    with torch.no_grad():
        know = kt(**b)
    st, _ = build("student", SR, TR, True)
    loss = st(teacher_knowledge=know, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()), t_enc1=t2n(know[2][1]), t_enc4=t2n(know[2][4]))
    _named_losses(st, d)
    _grads(st, [k for k in GRAD_KEYS] + KD_KEYS, d)
    save("g11_student_kd_residual", **d)


def gen_g12():
    """G12: `--output-activation sigmoid` (decoder_sa.py:353-360 resolves torch.nn.functional.<name>): the free-running loop feeds the ACTIVATED
    frame back (:614-617) and activates the final output (:635-636); forward() activates before_outs and after_outs behind the postnet (:538-540,
    decoder_sa_kd.py:698-700).  Teacher: inference mel + training step (eval form); student: KD step against a teacher with the same activation."""
    kw = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, output_activation="sigmoid")
    TA = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, **kw)
    SA = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, **kw)
    rng = np.random.RandomState(12)
    x = torch.from_numpy(rng.randint(1, TA.idim, size=7).astype(np.int64))
    dur = torch.tensor([2, 1, 4, 3, 1, 5, 2])
    te, _ = build("teacher", TA)
    assert te.dec.output_activation_fn is torch.nn.functional.sigmoid
    with torch.no_grad():
        after = te.inference(x, None, dur=dur)
    save("g12_teacher_sigmoid_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    raw, b = make_converter_batch(TINY_S, seed=7)
    loss = te(**b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()))
    _named_losses(te, d)
    _grads(te, GRAD_KEYS, d)
    save("g12_teacher_sigmoid", **d)
    kt, _ = build("kd_teacher", TA)
    with torch.no_grad():
        know = kt(**b)
    st, _ = build("student", SA, TA, True)
    loss = st(teacher_knowledge=know, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()), t_after=t2n(know[0]), t_before=t2n(know[1]))
    _named_losses(st, d)
    _grads(st, [k for k in GRAD_KEYS] + KD_KEYS, d)
    save("g12_student_kd_sigmoid", **d)


def gen_g13():
    """G13: speaker embeddings (`spk_embed_dim`, ..._sa.py:380-384, 555-557, 636-638): F.normalize(spemb) appended to every encoder state; the
    predictors, the pitch / energy embeddings and the decoder run on eunits + spk_embed_dim channels.  Teacher: inference mel + training step (eval
    form); KD teacher: the 5-tuple (the encoder tap stays eunits wide).  The KD student cannot run with them in the reference (records.json)."""
    kw = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, spk_embed_dim=8)
    TK = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, **kw)
    rng = np.random.RandomState(13)
    x = torch.from_numpy(rng.randint(1, TK.idim, size=7).astype(np.int64))
    dur = torch.tensor([3, 1, 2, 4, 1, 2, 3])
    raw, b = make_converter_batch(TINY_S, seed=7)
    spk = torch.from_numpy(rng.randn(b["xs"].shape[0], 8).astype(np.float32))
    te, _ = build("teacher", TK)
    with torch.no_grad():
        after = te.inference(x, None, spemb=spk[1], dur=dur)
    save("g13_teacher_spk_inference", x=t2n(x), dur=t2n(dur), spemb=t2n(spk[1]), after=t2n(after))
    loss = te(spembs=spk, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()), spembs=t2n(spk))
    _named_losses(te, d)
    _grads(te, GRAD_KEYS + ["pitch_embed.0.weight", "duration_predictor.conv.0.0.weight", "enc.blstm.weight_hh_l0"], d)
    save("g13_teacher_spk", **d)
    kt, _ = build("kd_teacher", TK)
    with torch.no_grad():
        know = kt(spembs=spk, **b)
    save("g13_kd_teacher_spk", after=t2n(know[0]), enc4=t2n(know[2][4]), dec1=t2n(know[3][1]), p_embs=t2n(know[4][3]), d_outs=t2n(know[4][0]))


def gen_g14():
    """G14: decoder options outside the shipped recipes, all three at once -- zoneout_rate 0 (plain LSTMCell: the parameters lose the `.cell` level,
    decoder_sa.py:366-369), use_concate False (feat_out reads the LSTM state alone, :397, :505-511), append_position False (no position column in
    LSTM 0's input, :361-365, :494-498).  Teacher: inference mel + training step (eval form); student: KD step against a teacher with the same options."""
    kw = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, zoneout_rate=0.0, use_concate=False, append_position=False)
    TA = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, **kw)
    SA = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, **kw)
    rng = np.random.RandomState(14)
    x = torch.from_numpy(rng.randint(1, TA.idim, size=7).astype(np.int64))
    dur = torch.tensor([2, 4, 1, 3, 1, 2, 5])
    te, spec = build("teacher", TA)
    assert "dec.lstm.0.weight_ih" in spec and spec["dec.lstm.0.weight_ih"][1] == TA.adim + TA.prenet_units and spec["dec.feat_out.weight"][1] == TA.dunits
    with torch.no_grad():
        after = te.inference(x, None, dur=dur)
    save("g14_teacher_options_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    raw, b = make_converter_batch(TINY_S, seed=7)
    loss = te(**b)
    loss.backward()
    keys = [k.replace(".cell.", ".") for k in GRAD_KEYS]
    d = dict(loss=np.float32(loss.item()))
    _named_losses(te, d)
    _grads(te, keys, d)
    save("g14_teacher_options", **d)
    # the KD pair: the reference's KD decoder cannot run forward() with use_concate False (decoder_sa_kd.py:617-622 hands feat_out the LIST z_list[-1]:
    # records.json), so the KD step carries the other two options; the student's inference() works with all three (:765-770)
    st3, _ = build("student", SA, TA, True)
    with torch.no_grad():
        after = st3.inference(x, None, dur=dur)
    save("g14_student_options_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    TK, SK = dataclasses.replace(TA, use_concate=True), dataclasses.replace(SA, use_concate=True)
    kt, _ = build("kd_teacher", TK)
    with torch.no_grad():
        know = kt(**b)
    st, _ = build("student", SK, TK, True)
    loss = st(teacher_knowledge=know, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()), t_after=t2n(know[0]), t_before=t2n(know[1]))
    _named_losses(st, d)
    _grads(st, keys + KD_KEYS, d)
    save("g14_student_kd_options", **d)


def gen_g15():
    """G15: `--use-batch-norm false` (encoder_sa.py:63-90, decoder_sa.py:203-263; decoder_sa_kd.py likewise): the encoder blocks are Conv1d -> ReLU ->
    Dropout and the postnet blocks Conv1d -> Tanh -> Dropout with no normalisation layer (and no `.1.*` parameters).  Teacher: inference mel + training
    step (eval form); student: inference mel + KD step against a teacher with the same option."""
    kw = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, use_batch_norm=False)
    TA = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, **kw)
    SA = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, **kw)
    rng = np.random.RandomState(15)
    x = torch.from_numpy(rng.randint(1, TA.idim, size=7).astype(np.int64))
    dur = torch.tensor([1, 3, 2, 5, 1, 2, 4])
    te, spec = build("teacher", TA)
    assert "enc.convs.0.1.weight" not in spec and "dec.postnet.postnet.0.1.weight" not in spec
    with torch.no_grad():
        after = te.inference(x, None, dur=dur)
    save("g15_teacher_nobn_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    raw, b = make_converter_batch(TINY_S, seed=7)
    loss = te(**b)
    loss.backward()
    keys = [k for k in GRAD_KEYS if ".1.weight" not in k and ".1.bias" not in k] + ["enc.convs.1.0.weight", "dec.postnet.postnet.0.0.weight"]
    d = dict(loss=np.float32(loss.item()))
    _named_losses(te, d)
    _grads(te, keys, d)
    save("g15_teacher_nobn", **d)
    kt, _ = build("kd_teacher", TA)
    with torch.no_grad():
        know = kt(**b)
    st, _ = build("student", SA, TA, True)
    with torch.no_grad():
        after = st.inference(x, None, dur=dur)
    save("g15_student_nobn_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    loss = st(teacher_knowledge=know, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()), t_after=t2n(know[0]), t_before=t2n(know[1]), t_enc1=t2n(know[2][1]))
    _named_losses(st, d)
    _grads(st, keys + KD_KEYS, d)
    save("g15_student_kd_nobn", **d)


def gen_g16():
    """G16: encoder widths that all differ (embed_dim != econv_chans != eunits; every shipped recipe sets them equal): the first encoder block maps
    embed_dim -> econv_chans, the BiLSTM econv_chans -> eunits, and everything behind the encoder runs on eunits.  Teacher: inference mel + training
    step (eval form); student (its three widths differ too, and from the teacher's): inference mel + KD step."""
    kw = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0)
    TA = HP.teacher_hparams(embed_dim=24, econv_chans=32, eunits=40, dunits=40, prenet_units=28, postnet_chans=20, **kw)
    SA = HP.student_hparams(embed_dim=12, econv_chans=16, eunits=24, dunits=24, prenet_units=20, postnet_chans=12, **kw)
    rng = np.random.RandomState(16)
    x = torch.from_numpy(rng.randint(1, TA.idim, size=7).astype(np.int64))
    dur = torch.tensor([3, 1, 2, 2, 4, 1, 3])
    te, _ = build("teacher", TA)
    with torch.no_grad():
        after = te.inference(x, None, dur=dur)
    save("g16_teacher_widths_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    raw, b = make_converter_batch(TINY_S, seed=7)
    loss = te(**b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()))
    _named_losses(te, d)
    _grads(te, GRAD_KEYS + ["enc.convs.0.0.weight", "enc.blstm.weight_ih_l0"], d)
    save("g16_teacher_widths", **d)
    kt, _ = build("kd_teacher", TA)
    with torch.no_grad():
        know = kt(**b)
    st, _ = build("student", SA, TA, True)
    with torch.no_grad():
        after = st.inference(x, None, dur=dur)
    save("g16_student_widths_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    loss = st(teacher_knowledge=know, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()), t_after=t2n(know[0]), t_before=t2n(know[1]), t_enc0=t2n(know[2][0]), t_enc4=t2n(know[2][4]))
    _named_losses(st, d)
    _grads(st, GRAD_KEYS + KD_KEYS + ["enc.convs.0.0.weight", "enc.blstm.weight_ih_l0"], d)
    save("g16_student_kd_widths", **d)


def gen_g17():
    """G17: layer counts outside the shipped recipes that the plain teacher class runs (the KD classes index fixed tap lists and raise on them:
    records.json): econv_layers 2, postnet_layers 3.  Teacher: inference mel + training step (eval form)."""
    TA = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20,
                            duration_predictor_chans=20, dropout_rate=0.0, econv_layers=2, postnet_layers=3)
    rng = np.random.RandomState(17)
    x = torch.from_numpy(rng.randint(1, TA.idim, size=7).astype(np.int64))
    dur = torch.tensor([2, 2, 5, 1, 3, 1, 2])
    te, spec = build("teacher", TA)
    assert "enc.convs.2.0.weight" not in spec and "dec.postnet.postnet.3.0.weight" not in spec and "dec.postnet.postnet.2.0.weight" in spec
    with torch.no_grad():
        after = te.inference(x, None, dur=dur)
    save("g17_teacher_layers_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    raw, b = make_converter_batch(TINY_S, seed=7)
    loss = te(**b)
    loss.backward()
    keys = [k for k in GRAD_KEYS if "postnet.4" not in k] + ["dec.postnet.postnet.2.0.weight", "enc.convs.0.0.weight"]
    d = dict(loss=np.float32(loss.item()))
    _named_losses(te, d)
    _grads(te, keys, d)
    save("g17_teacher_layers", **d)


def _teacher_variant(name, hp, seed, extra_keys):
    """One golden pair for a structure option the reference's plain teacher class runs: inference mel + training step (eval form)."""
    rng = np.random.RandomState(seed)
    x = torch.from_numpy(rng.randint(1, hp.idim, size=7).astype(np.int64))
    dur = torch.tensor([2, 1, 4, 1, 3, 2, 2])
    te, spec = build("teacher", hp)
    with torch.no_grad():
        after = te.inference(x, None, dur=dur)
    save(name + "_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    raw, b = make_converter_batch(TINY_S, seed=7)
    loss = te(**b)
    loss.backward()
    keys = [k for k in GRAD_KEYS if k in spec] + [k for k in extra_keys if k in spec]
    d = dict(loss=np.float32(loss.item()))
    _named_losses(te, d)
    _grads(te, keys, d)
    save(name, **d)
    return spec


def gen_g18_g19_g20():
    """G18 - G20 (round 5): the structure options of the reference's teacher class that the HIP path used to refuse (hparams.py:70-86).
    G18 `dlayers` 1 and 3 (decoder_sa.py:357-369, 500-504: a stack of ZoneOut LSTMCells, cell l > 0 on cell l - 1's state, feat_out on the last);
    G19 `prenet_layers` 1 and 3 (decoder_sa.py:119-158); G20 `elayers` 2 (encoder_sa.py:96-100: a two-layer bidirectional nn.LSTM).
    Each: inference mel and training step (losses + gradients, eval form) of the real class.  The KD classes index fixed tap lists and fail on
    most of these (records.json): teacher class only."""
    kw = dict(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, duration_predictor_chans=20, dropout_rate=0.0)
    for dl in (1, 3):
        spec = _teacher_variant("g18_teacher_dlayers%d" % dl, HP.teacher_hparams(dlayers=dl, **kw), 180 + dl,
                                ["dec.lstm.%d.cell.weight_ih" % (dl - 1), "dec.lstm.%d.cell.weight_hh" % (dl - 1), "dec.lstm.%d.cell.bias_ih" % (dl - 1)])
        assert ("dec.lstm.2.cell.weight_ih" in spec) == (dl == 3) and ("dec.lstm.1.cell.weight_ih" in spec) == (dl == 3)
    for pl in (1, 3):
        spec = _teacher_variant("g19_teacher_prenet%d" % pl, HP.teacher_hparams(prenet_layers=pl, **kw), 190 + pl,
                                ["dec.prenet.prenet.%d.0.weight" % (pl - 1), "dec.prenet.prenet.%d.0.bias" % (pl - 1)])
        assert ("dec.prenet.prenet.2.0.weight" in spec) == (pl == 3)
    spec = _teacher_variant("g20_teacher_elayers2", HP.teacher_hparams(elayers=2, **kw), 200,
                            ["enc.blstm.weight_ih_l0", "enc.blstm.weight_hh_l0_reverse", "enc.blstm.weight_ih_l1", "enc.blstm.weight_hh_l1", "enc.blstm.bias_ih_l1_reverse",
                             "enc.blstm.weight_ih_l1_reverse", "enc.convs.0.0.weight"])
    assert "enc.blstm.weight_ih_l1_reverse" in spec


def gen_g21():
    """G21 (round 5): `reduction_factor` 2 on the teacher class (decoder_sa.py:397-398 feat_out emits r frames per step as [odim, r]; :456-457 /
    :488-489 every r-th target frame is the next step's input; :512-516 / :627 the r frames of a step are consecutive output frames; converter
    tts.py:250-258: segments, ds_nonzeros and the position table in FRAMES = r x the annotated durations; inference position t / d in STEPS,
    ..._sa.py:665-669).  Targets hold exactly r * sum(d) frames per utterance (the class splits the concatenated frames by `olens`,
    decoder_sa.py:519-522: any other length mis-assigns frames across utterances).  The real CustomConverter's outputs are stored too (bit-exact pin
    of the vectorised converter at r = 2), then the inference mel and the training step."""
    from tts import CustomConverter

    r = 2
    hp = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, duration_predictor_chans=20,
                            dropout_rate=0.0, reduction_factor=r)
    rng = np.random.RandomState(21)
    x = torch.from_numpy(rng.randint(1, hp.idim, size=7).astype(np.int64))
    dur = torch.tensor([2, 1, 3, 1, 2, 2, 1])
    te, spec = build("teacher", hp)
    assert tuple(spec["dec.feat_out.weight"]) == (hp.odim * r, hp.dunits + hp.eunits)
    with torch.no_grad():
        after = te.inference(x, None, dur=dur)
    assert after.shape[0] == r * int(dur.sum())
    save("g21_teacher_r2_inference", x=t2n(x), dur=t2n(dur), after=t2n(after))
    xs, ys, ds, f0, en = SYN.training_batch(hp.odim, hp.idim, batch=4, seed=7)
    ys = [rng.standard_normal((r * int(np.asarray(d).sum()), hp.odim)).astype(np.float32) for d in ds]  # r frames per annotated duration unit
    conv = CustomConverter(reduction_factor=r, use_fe_condition=True, append_position=True)
    b = conv([(xs, ys, None, ds, f0, en)])
    d = {}
    for i in range(4):
        d["in_xs%d" % i], d["in_ys%d" % i], d["in_ds%d" % i], d["in_f0%d" % i], d["in_en%d" % i] = xs[i], ys[i], np.asarray(ds[i]), f0[i], en[i]
    for k in ("xs", "ilens", "ys", "olens", "extras", "new_ys", "non_zero_lens_mask", "ds_nonzeros", "output_masks", "position", "f0", "energy"):
        d["out_" + k] = t2n(b[k])
    loss = te(**b)
    loss.backward()
    d["loss"] = np.float32(loss.item())
    _named_losses(te, d)
    _grads(te, GRAD_KEYS, d)
    save("g21_teacher_r2", **d)


def gen_g22():
    """G22 (round 5): the KD classes with the structure options their tap lists allow (records.json: `prenet_layers` 1 / 3 and `elayers` 2 run in
    ..._kd_teacher / ..._kd_student; `dlayers` 1 raises IndexError, 3 taps the middle cell): a KD teacher with THREE prenet blocks and TWO BiLSTM
    layers, a student with ONE prenet block and two BiLSTM layers.  The prenet tap is the last block's output, the encoder taps are unchanged."""
    kw = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, elayers=2)
    TA = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, prenet_layers=3, **kw)
    SA = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, prenet_layers=1, **kw)
    raw, b = make_converter_batch(TINY_S, seed=7)
    kt, _ = build("kd_teacher", TA)
    with torch.no_grad():
        know = kt(**b)
    st, spec = build("student", SA, TA, True)
    assert "dec.prenet.prenet.1.0.weight" not in spec and "enc.blstm.weight_ih_l1_reverse" in spec
    loss = st(teacher_knowledge=know, **b)
    loss.backward()
    d = dict(loss=np.float32(loss.item()), t_after=t2n(know[0]), t_before=t2n(know[1]), t_enc4=t2n(know[2][4]), t_dec0=t2n(know[3][0]), t_dec2=t2n(know[3][2]))
    _named_losses(st, d)
    _grads(st, [k for k in GRAD_KEYS if k in spec] + KD_KEYS + ["enc.blstm.weight_ih_l1", "enc.blstm.weight_hh_l1_reverse", "dec.prenet.prenet.0.0.weight"], d)
    save("g22_student_kd_structure", **d)


def gen_option_records():
    """records.json: what the reference itself does with the options the HIP path refuses (nets/base.py): speaker embeddings and reduction_factor > 1.
    Neither is in a shipped recipe (conf/*.yaml; LJSpeech is single-speaker).  The KD student cannot run with speaker embeddings in the reference:
    pemb_proj / eemb_proj are built for `eunits` inputs (..._kd_student.py:602-603) but receive eunits + spk_embed_dim channels (:709-711, :749-750)."""
    from nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_student import Tacotron2_sa as Student
    from nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_teacher import Tacotron2_sa as KDTeacher
    from nets.teacher_training.e2e_tts_tacotron2_sa import Tacotron2_sa as Teacher

    com = argparse.Namespace(share_proj=True, **COM)
    raw, b = make_converter_batch(TINY_S, seed=7)
    spk = torch.from_numpy(np.random.RandomState(3).randn(b["xs"].shape[0], 6).astype(np.float32))
    rec_path = os.path.join(OUT, "records.json")
    rec = json.load(open(rec_path)) if os.path.exists(rec_path) else {}

    def attempt(name, fn):
        try:
            fn()
            rec[name] = "runs"
        except Exception as e:  # noqa: BLE001 - the record is the exception itself
            rec[name] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:160])
        print("records.json %s -> %s" % (name, rec[name]))

    def teacher_spk():
        m = _quiet(Teacher, TINY_T.idim, TINY_T.odim, ns(TINY_T, spk_embed_dim=6), com)
        m.eval()
        m(spembs=spk, **b).backward()
        m.inference(torch.tensor([1, 2, 3, 4, 5]), None, spemb=spk[0], dur=torch.tensor([1, 2, 3, 1, 2]))

    def student_spk():
        kt = _quiet(KDTeacher, TINY_T.idim, TINY_T.odim, ns(TINY_T, spk_embed_dim=6), com)
        kt.eval()
        with torch.no_grad():
            know = kt(spembs=spk, **b)
        st = _quiet(Student, TINY_S.idim, TINY_S.odim, ns(TINY_S, spk_embed_dim=6), com, ns(TINY_T, spk_embed_dim=6))
        st.eval()
        st(teacher_knowledge=know, spembs=spk, **b).backward()

    def kd_no_concat():
        kt = _quiet(KDTeacher, TINY_T.idim, TINY_T.odim, ns(TINY_T, use_concate=False), com)
        kt.eval()
        with torch.no_grad():
            kt(**b)

    attempt("use_concate_false_kd_forward", kd_no_concat)

    # structural options the HIP path refuses (hparams.check_supported): what the reference's own classes do with them on the tiny shapes --
    # training step + inference of the plain teacher class, and the KD pair (KD teacher forward, student step, student inference)
    def teacher_with(com_kw=None, **kw):
        def f():
            c = argparse.Namespace(**dict(vars(com), **(com_kw or {})))
            m = _quiet(Teacher, TINY_T.idim, TINY_T.odim, ns(TINY_T, **kw), c)
            m.eval()
            m(**b).backward()
            m.inference(torch.tensor([1, 2, 3, 4, 5]), None, dur=torch.tensor([1, 2, 3, 1, 2]))
        return f

    def kd_with(com_kw=None, **kw):
        def f():
            c = argparse.Namespace(**dict(vars(com), **(com_kw or {})))
            kt = _quiet(KDTeacher, TINY_T.idim, TINY_T.odim, ns(TINY_T, **kw), c)
            kt.eval()
            with torch.no_grad():
                know = kt(**b)
            st = _quiet(Student, TINY_S.idim, TINY_S.odim, ns(TINY_S, **kw), c, ns(TINY_T, **kw))
            st.eval()
            st(teacher_knowledge=know, **b).backward()
            st.inference(torch.tensor([1, 2, 3, 4, 5]), None, dur=torch.tensor([1, 2, 3, 1, 2]))
        return f

    attempt("use_fe_condition_false_teacher", teacher_with(dict(use_fe_condition=False)))
    attempt("use_fe_condition_false_kd", kd_with(dict(use_fe_condition=False)))
    for key, vals in (("elayers", (2,)), ("dlayers", (1, 3)), ("prenet_layers", (0, 1, 3)), ("postnet_layers", (0, 1, 3)), ("econv_layers", (0, 2))):
        for v in vals:
            attempt("%s_%d_teacher" % (key, v), teacher_with(**{key: v}))
            attempt("%s_%d_kd" % (key, v), kd_with(**{key: v}))
    attempt("spk_embed_teacher_training_and_inference", teacher_spk)
    attempt("spk_embed_student_kd_training", student_spk)
    with open(rec_path, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)


def main():
    assert os.path.isdir(REF), "gen_golden.py needs /root/reference (survey container only)"
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    only = set(sys.argv[1:])  # e.g. `gen_golden.py g10`: that set alone (every set is a pure function of the reference + closed-form inputs)
    if only:
        assert only <= {"g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g21", "g22", "records"}, only
        if "g22" in only:
            gen_g22()
        if "g21" in only:
            gen_g21()
        if "g18" in only:
            gen_g18_g19_g20()
        if "g17" in only:
            gen_g17()
        if "g16" in only:
            gen_g16()
        if "g15" in only:
            gen_g15()
        if "g14" in only:
            gen_g14()
        if "g13" in only:
            gen_g13()
        if "records" in only:
            gen_option_records()
        if "g12" in only:
            gen_g12()
        if "g10" in only:
            gen_g10()
        if "g11" in only:
            gen_g11()
        return
    gen_manifest()
    gen_g1()
    rec = gen_g2_g3()
    gen_g4()
    gen_g5()
    gen_g6(rec)
    gen_g7_g8_g9()
    gen_g10()
    gen_g11()
    gen_g12()
    gen_g13()
    gen_g14()
    gen_g15()
    gen_g16()
    gen_g17()
    gen_g18_g19_g20()
    gen_g21()
    gen_g22()
    gen_option_records()


if __name__ == "__main__":
    main()
