"""Shared implementation of the three `Tacotron2_sa` plug-in classes (teacher, KD teacher, KD student).

The reference resolves `--model-module pkg.mod:Class` and instantiates `Class(idim, odim, args, com_args
[, teacher_args])` (tts.py:356-357, tts_distill.py:364-375).  These classes keep that surface — argument
names, flag names, `state_dict()` key names/shapes, `inference()` / `forward()` signatures, `reporter`,
`base_plot_keys` — while every FLOP of `inference()` runs in libfcl_hip.so.  The torch modules built here
are parameter CONTAINERS (names, shapes, reference initialisation); they are never called.
"""
import argparse

import torch

from .. import engine, ops
from ..hparams import HParams, param_spec
from ..plan import SynthesisPlan
from ..tts_interface import TTSInterface


def strtobool(x):
    s = str(x).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return True
    if s in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError("invalid truth value %r" % (x,))


# (flag, default, type) — the reference's model flags (..._kd_student.py:241-390 == ..._sa.py:138-287)
MODEL_FLAGS = [
    ("embed-dim", 512, int), ("elayers", 1, int), ("eunits", 512, int), ("econv-layers", 3, int), ("econv-chans", 512, int),
    ("econv-filts", 5, int), ("dlayers", 2, int), ("dunits", 1024, int), ("prenet-layers", 2, int), ("prenet-units", 256, int),
    ("postnet-layers", 5, int), ("postnet-chans", 512, int), ("postnet-filts", 5, int), ("output-activation", None, str),
    ("use-batch-norm", True, strtobool), ("use-concate", True, strtobool), ("use-residual", True, strtobool),
    ("dropout-rate", 0.5, float), ("zoneout-rate", 0.1, float), ("reduction-factor", 1, int), ("spk-embed-dim", None, int),
    ("spc-dim", None, int), ("pretrained-model", None, str), ("use-masking", False, strtobool),
    ("use-weighted-masking", False, strtobool), ("duration-predictor-layers", 2, int), ("duration-predictor-chans", 384, int),
    ("duration-predictor-kernel-size", 3, int), ("duration-predictor-dropout-rate", 0.1, float),
]


def add_model_arguments(parser):
    group = parser.add_argument_group("tacotron 2 model setting")
    for flag, default, typ in MODEL_FLAGS:
        kw = dict(default=default, type=typ)
        if flag == "output-activation":
            kw["nargs"] = "?"
        names = ["--" + flag] + (["-u"] if flag == "eunits" else [])
        group.add_argument(*names, **kw)
    return parser


def fill_missing_args(args, add_arguments):
    defaults, _ = add_arguments(argparse.ArgumentParser()).parse_known_args([])
    d = {} if args is None else dict(vars(args))
    for k, v in vars(defaults).items():
        d.setdefault(k, v)
    return argparse.Namespace(**d)


def hparams_from_args(idim, odim, args):
    return HParams(
        idim=idim, odim=odim, embed_dim=args.embed_dim, elayers=args.elayers, eunits=args.eunits, econv_layers=args.econv_layers,
        econv_chans=args.econv_chans, econv_filts=args.econv_filts, dlayers=args.dlayers, dunits=args.dunits,
        prenet_layers=args.prenet_layers, prenet_units=args.prenet_units, postnet_layers=args.postnet_layers,
        postnet_chans=args.postnet_chans, postnet_filts=args.postnet_filts, use_batch_norm=args.use_batch_norm,
        use_concate=args.use_concate, use_residual=args.use_residual, reduction_factor=args.reduction_factor,
        dropout_rate=args.dropout_rate, zoneout_rate=args.zoneout_rate,
        duration_predictor_layers=args.duration_predictor_layers, duration_predictor_chans=args.duration_predictor_chans,
        duration_predictor_kernel_size=args.duration_predictor_kernel_size,
        duration_predictor_dropout_rate=args.duration_predictor_dropout_rate,
        use_fe_condition=args.use_fe_condition, append_position=args.append_position, use_masking=bool(args.use_masking),
        use_weighted_masking=bool(getattr(args, "use_weighted_masking", False)), output_activation=args.output_activation,
        spk_embed_dim=args.spk_embed_dim,
    ).check_supported()


# ---- parameter containers with the reference's names and initialisation ---------------------------
def _seq(*mods):
    return torch.nn.Sequential(*mods)


def _conv_bn(cin, cout, k, act, p, bn=True):
    layers = [torch.nn.Conv1d(cin, cout, k, stride=1, padding=(k - 1) // 2, bias=False)] + ([torch.nn.BatchNorm1d(cout)] if bn else [])
    if act is not None:
        layers.append(act)
    layers.append(torch.nn.Dropout(p))
    return _seq(*layers)


class _Cell(torch.nn.Module):  # ZoneOutCell wrapper: parameters live under `.cell` (decoder_sa.py:26-61)
    def __init__(self, cin, units):
        super().__init__()
        self.cell = torch.nn.LSTMCell(cin, units)


class _Prenet(torch.nn.Module):
    def __init__(self, idim, units, layers):
        super().__init__()
        self.prenet = torch.nn.ModuleList(_seq(torch.nn.Linear(idim if i == 0 else units, units), torch.nn.ReLU()) for i in range(layers))


class _Postnet(torch.nn.Module):
    def __init__(self, hp):
        super().__init__()
        n, cp = hp.postnet_layers, hp.postnet_chans
        self.postnet = torch.nn.ModuleList(
            _conv_bn(hp.odim if i == 0 else cp, hp.odim if i == n - 1 else cp, hp.postnet_filts,
                     None if i == n - 1 else torch.nn.Tanh(), hp.dropout_rate, hp.use_batch_norm) for i in range(n))


class _Predictor(torch.nn.Module):
    def __init__(self, idim, layers, chans, k, p):
        super().__init__()
        self.conv = torch.nn.ModuleList(
            _seq(torch.nn.Conv1d(idim if i == 0 else chans, chans, k, stride=1, padding=(k - 1) // 2), torch.nn.ReLU(),
                 torch.nn.LayerNorm(chans, eps=1e-12), torch.nn.Dropout(p)) for i in range(layers))
        self.linear = torch.nn.Linear(chans, 1)


class _Encoder(torch.nn.Module):
    def __init__(self, hp, thp, share_proj):
        super().__init__()
        self.embed = torch.nn.Embedding(hp.idim, hp.embed_dim, padding_idx=0)
        self.convs = torch.nn.ModuleList(
            _conv_bn(hp.embed_dim if i == 0 else hp.econv_chans, hp.econv_chans, hp.econv_filts, torch.nn.ReLU(), hp.dropout_rate, hp.use_batch_norm)
            for i in range(hp.econv_layers))
        self.blstm = torch.nn.LSTM(hp.econv_chans, hp.eunits // 2, hp.elayers, batch_first=True, bidirectional=True)
        if thp is not None:
            self.embed_proj = torch.nn.Linear(hp.embed_dim, thp.embed_dim, bias=False)
            self.convs_proj = torch.nn.ModuleList(torch.nn.Linear(hp.econv_chans, thp.econv_chans, bias=False)
                                                  for _ in range(1 if share_proj else hp.econv_layers))
            self.blstm_proj = torch.nn.Linear(hp.eunits, thp.eunits, bias=False)
        for m in self.modules():  # encoder_init (encoder_sa.py:15-18)
            if isinstance(m, torch.nn.Conv1d):
                torch.nn.init.xavier_uniform_(m.weight, torch.nn.init.calculate_gain("relu"))


class _Decoder(torch.nn.Module):
    def __init__(self, hp, thp, share_proj):
        super().__init__()
        d, u, p = hp.adim, hp.dunits, hp.prenet_units
        i0 = d + p + (1 if hp.append_position else 0)  # decoder_sa.py:361-365
        mk = _Cell if hp.zoneout_rate > 0.0 else torch.nn.LSTMCell  # ZoneOutCell wraps the cell only for a positive rate (:366-369)
        self.lstm = torch.nn.ModuleList(mk(i0 if l == 0 else u, u) for l in range(hp.dlayers))
        self.prenet = _Prenet(hp.odim, p, hp.prenet_layers)
        self.postnet = _Postnet(hp)
        self.feat_out = torch.nn.Linear(u + d if hp.use_concate else u, hp.odim * hp.reduction_factor, bias=False)  # decoder_sa.py:397-398
        if thp is not None:
            self.prenet_proj = torch.nn.Linear(p, thp.prenet_units, bias=False)
            if share_proj:
                self.lstm_proj = torch.nn.Linear(u, thp.dunits, bias=False)
                self.post_proj = torch.nn.Linear(hp.postnet_chans, thp.postnet_chans, bias=False)
            else:
                self.lstm0_proj = torch.nn.Linear(u, thp.dunits, bias=False)
                self.lstm1_proj = torch.nn.Linear(u, thp.dunits, bias=False)
                for i in range(4):
                    setattr(self, "post%d_proj" % i, torch.nn.Linear(hp.postnet_chans, thp.postnet_chans, bias=False))
        for m in self.modules():  # decoder_init (decoder_sa.py:20-23)
            if isinstance(m, torch.nn.Conv1d):
                torch.nn.init.xavier_uniform_(m.weight, torch.nn.init.calculate_gain("tanh"))


class _HipLoss(torch.autograd.Function):
    """loss value computed by the HIP engine; backward returns the engine's gradients (scaled by the incoming grad)."""

    @staticmethod
    def forward(ctx, value, grads, *params):
        ctx.grads = grads
        return value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        return (None, None) + tuple(g * grad_out for g in ctx.grads)


class Tacotron2Base(TTSInterface, torch.nn.Module):
    """role: "teacher" | "kd_teacher" | "student"."""

    role = "teacher"

    @staticmethod
    def add_arguments(parser):
        return add_model_arguments(parser)

    def _setup(self, idim, odim, args, com_args, teacher_args):
        TTSInterface.__init__(self)
        torch.nn.Module.__init__(self)
        args = fill_missing_args(args, self.add_arguments)
        d = vars(args)
        extra = ["use_fe_condition", "append_position"]
        if self.role == "student":
            extra += ["distill_output_knowledge", "distill_encoder_knowledge", "distill_decoder_knowledge",
                      "distill_prosody_knowledge", "is_train", "share_proj"]
        for k in extra:  # driver flags come from `args` if present, else from com_args (..._kd_student.py:438-456)
            if k not in d:
                d[k] = getattr(com_args, k)
        d.setdefault("encoder_resume", None)
        args = argparse.Namespace(**d)
        self.idim, self.odim = idim, odim
        self.hp = hparams_from_args(idim, odim, args)
        self.embed_dim, self.spk_embed_dim, self.reduction_factor = args.embed_dim, args.spk_embed_dim, args.reduction_factor
        self.use_fe_condition, self.append_position = args.use_fe_condition, args.append_position
        thp = None
        self.share_proj = False
        if self.role == "student":
            for k in extra[2:]:
                setattr(self, k, d[k])
            targs = fill_missing_args(teacher_args, self.add_arguments)
            vars(targs).setdefault("use_fe_condition", True)
            vars(targs).setdefault("append_position", True)
            self.teacher_hp = hparams_from_args(idim, odim, targs)
            thp = self.teacher_hp if self.is_train else None  # is_student = is_train (..._kd_student.py:473-476)
        hp = self.hp
        self.enc = _Encoder(hp, thp, self.share_proj)
        self.dec = _Decoder(hp, thp, self.share_proj)
        self.duration_predictor = _Predictor(hp.adim, hp.duration_predictor_layers, hp.duration_predictor_chans,
                                             hp.duration_predictor_kernel_size, hp.duration_predictor_dropout_rate)
        for nm in ("pitch", "energy"):
            setattr(self, nm + "_predictor", _Predictor(hp.adim, hp.variance_predictor_layers, hp.variance_predictor_chans,
                                                        hp.variance_predictor_kernel_size, hp.variance_predictor_dropout_rate))
            k = hp.variance_embed_kernel_size
            setattr(self, nm + "_embed", _seq(torch.nn.Conv1d(1, hp.adim, k, padding=(k - 1) // 2),
                                              torch.nn.Dropout(hp.variance_embed_dropout_rate)))
        if self.role == "student":  # created whenever use_fe_condition, regardless of is_train (..._kd_student.py:602-603)
            self.pemb_proj = torch.nn.Linear(hp.eunits, self.teacher_hp.eunits, bias=False)
            self.eemb_proj = torch.nn.Linear(hp.eunits, self.teacher_hp.eunits, bias=False)
        spec = param_spec(hp, thp, self.share_proj)
        if self.role == "student" and thp is None:
            spec["pemb_proj.weight"] = (self.teacher_hp.eunits, hp.eunits)
            spec["eemb_proj.weight"] = (self.teacher_hp.eunits, hp.eunits)
        mine = {k: tuple(v.shape) for k, v in self.state_dict().items()}
        assert mine == {k: tuple(v) for k, v in spec.items()}, "parameter tree does not match hparams.param_spec"
        self._plan = None
        self._plan_key = None
        # `--encoder-resume PATH` (tts_train.py:319; teacher and student constructors pass it to their Encoder, ..._sa.py:378,
        # ..._kd_student.py:504): the encoder's OWN state_dict, loaded strictly in place of encoder_init (encoder_sa.py:117-120,
        # encoder_sa_kd.py:137-140 -- for a student under KD training that includes its embed_proj / convs_proj / blstm_proj)
        if args.encoder_resume is not None:
            self.enc.load_state_dict(torch.load(args.encoder_resume, map_location="cpu", weights_only=False))
        # `--pretrained-model PATH`: the whole model, after construction (..._sa.py:480-481, ..._kd_student.py:622-623, ..._kd_teacher.py:481-482)
        if getattr(args, "pretrained_model", None) is not None:
            self.load_pretrained_model(args.pretrained_model)

    def load_pretrained_model(self, model_path):
        """ESPnet's TTSInterface.load_pretrained_model = espnet.asr.pytorch_backend.asr_init / asr_utils.torch_load(model_path, self) (ESPnet
        0.8, restated: parity unpinned at this boundary): a trainer snapshot (file name contains "snapshot") holds the weights under "model",
        any other file IS the state_dict; keys of a DataParallel-wrapped model ("module." prefix) are accepted."""
        import os

        obj = torch.load(model_path, map_location="cpu", weights_only=False)
        if "snapshot" in os.path.basename(model_path) or (isinstance(obj, dict) and isinstance(obj.get("model"), dict)):
            obj = obj["model"]
        self.load_state_dict({k[len("module."):] if k.startswith("module.") else k: v for k, v in obj.items()})

    # ---- plan management --------------------------------------------------------------------------
    def _load_from_state_dict(self, *a, **k):  # any (re)load invalidates the packed device weights
        self._plan = None
        if getattr(self, "_engine", None) is not None:
            self._engine.invalidate_planes()
        return super()._load_from_state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        self._plan = None
        r = super().load_state_dict(*a, **k)
        if getattr(self, "_engine", None) is not None:
            self._engine.invalidate_planes()
        return r

    def plan(self, device=None):
        """Packed device weights for the HIP path (rebuilt after load_state_dict / refresh_plan())."""
        if device is None:
            p = next(self.parameters())
            device = p.device if p.is_cuda else torch.device("cuda:0")
        key = str(device)
        if self._plan is None or self._plan_key != key:
            self._plan = SynthesisPlan({k: v.detach() for k, v in self.state_dict().items()}, self.hp, device)
            self._plan_key = key
        return self._plan

    def refresh_plan(self):
        self._plan = None

    # ---- the reference's inference(): one utterance -----------------------------------------------
    @torch.no_grad()
    def inference(self, x, inference_args=None, spemb=None, dur=None, f0=None, energy=None, utt_id=None, y=None, *args, **kwargs):
        """x: LongTensor (T,) -> Tensor (L, odim), as ..._kd_student.py:804-863 / ..._sa.py:624-683.
        Prenet dropout stays ON (decoder_sa.py:156-158); masks come from the on-device generator, seeded from
        torch's default generator so `torch.manual_seed` makes a run repeatable."""
        mels = self.inference_batch([x], None if dur is None else [dur], None if f0 is None else [f0],
                                    None if energy is None else [energy], spembs=None if spemb is None else [spemb])
        return mels[0]

    @torch.no_grad()
    def inference_batch(self, xs, durs=None, f0s=None, energies=None, dropout_mode=ops.DROP_RNG, prenet_keep=None, seed=None, spembs=None):
        """Build extension: equals len(xs) independent inference() calls, in one pass."""
        plan = self.plan(xs[0].device if torch.is_tensor(xs[0]) and xs[0].is_cuda else None)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        cpu = lambda a: a.detach().cpu().numpy() if torch.is_tensor(a) else a
        return engine.synthesize(plan, [cpu(x) for x in xs], None if durs is None else [cpu(d).reshape(-1) for d in durs],
                                 None if f0s is None else [cpu(f) for f in f0s], None if energies is None else [cpu(e) for e in energies],
                                 dropout_mode=dropout_mode, prenet_keep=prenet_keep, seed=seed, spembs=spembs)

    def forward(self, xs, ilens, ys, olens, spembs=None, extras=None, new_ys=None, non_zero_lens_mask=None, ds_nonzeros=None,
                output_masks=None, position=None, f0=None, energy=None, teacher_knowledge=None, *args, **kwargs):
        """Teacher-forced forward on the HIP path: returns the scalar loss (kd_teacher: the 5-tuple) and reports the named losses.  Same
        keyword batch as the reference (tts.py:277-305); `output_masks` / `position` are accepted and recomputed on device
        (make_non_pad_mask(ds_nonzeros); t/d in-kernel).
          .eval()   what the reference's CustomEvaluator runs (tts.py:76-108): no gradients;
          .train()  what CustomUpdater runs (tts.py:156-161, tts_distill.py:159-161): forward AND backward run fused in libfcl_hip.so
                    (fcl_taco2_amd.training.TrainEngine, train-form BatchNorm / dropout / zoneout, masks drawn on the device); the returned
                    loss is attached to the parameters through a torch.autograd.Function whose backward hands the HIP-computed gradients
                    to autograd, so the reference's `loss.backward(); clip_grad_norm_(model.parameters()); optimizer.step()` works as is."""
        if self.role == "student" and self.hp.spk_embed_dim is not None:
            # the reference's own KD student cannot run forward() with speaker embeddings: pemb_proj / eemb_proj are Linear(eunits, ...) but receive
            # eunits + spk_embed_dim channels (..._kd_student.py:602-603 vs :709-711, 749-750; the RuntimeError is recorded in tests/golden/records.json)
            raise NotImplementedError("fcl-taco2_amd: KD training with speaker embeddings is undefined in the reference (its student's pemb_proj / "
                                      "eemb_proj do not fit eunits + spk_embed_dim inputs); synthesis with a speaker-embedding student is supported")
        batch = dict(xs=xs, ilens=ilens, ys=ys, olens=olens, extras=extras, new_ys=new_ys, non_zero_lens_mask=non_zero_lens_mask,
                     ds_nonzeros=ds_nonzeros, f0=f0, energy=energy, spembs=spembs)
        if self.training:
            return self._forward_train(batch, teacher_knowledge, kwargs.get("masks"))
        order = ["l1_loss", "mse_loss", "dur_loss", "pitch_loss", "energy_loss", "output_l1_loss", "output_mse_loss", "encoder_loss",
                 "decoder_loss", "prosody_loss", "loss"]
        if self.hp.reduction_factor != 1:
            # round 5: the no-gradient forward of teacher_forced.py covers reduction_factor 1; with r > 1 (teacher class only: the reference's KD
            # classes fail on it) the evaluator's forward runs the training engine's eval form -- running-stat BatchNorm, expectation zoneout, the
            # prenet's dropout on as in the reference (decoder_sa.py:156-158) -- and leaves the engine's gradient buffers as it found them
            eng = self.train_engine()
            eng.invalidate_planes()
            kept = eng.gflat.clone()
            try:
                with torch.no_grad():
                    rep = eng.forward_backward(batch, None, mode="eval", masks=kwargs.get("masks"), reduce=False)
            finally:
                eng.gflat.copy_(kept)
            self.reporter.report([{k: float(rep[k])} for k in order if k in rep])
            return torch.tensor(float(rep["loss"]), dtype=torch.float32, device=eng.dev)
        from .. import teacher_forced as TF

        plan = self.plan(xs.device if xs.is_cuda else None)
        kw = dict(seed=int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        kw.update({k: kwargs[k] for k in ("dropout_mode", "prenet_keep", "seed") if k in kwargs})
        with torch.no_grad():
            if self.role == "kd_teacher":
                return TF.knowledge_tuple(TF.forward_pass(plan, batch, **kw))
            if self.role == "student":
                rep, _ = TF.student_forward(plan, batch, teacher_knowledge, self.share_proj,
                                            (self.distill_output_knowledge, self.distill_encoder_knowledge,
                                             self.distill_decoder_knowledge, self.distill_prosody_knowledge), **kw)
            else:
                rep, _ = TF.teacher_forward(plan, batch, **kw)
        self.reporter.report([{k: float(rep[k])} for k in order if k in rep])
        return torch.tensor(float(rep["loss"]), dtype=torch.float32, device=plan.device)

    def train_engine(self, **kw):
        """The fused forward/backward engine bound to this module's parameters (created on first use; parameters are re-pointed into its
        flat buffer, so torch optimizers and state_dict() keep working on the same storage)."""
        if getattr(self, "_engine", None) is None:
            from ..training import TrainEngine

            self._engine = TrainEngine(self, **kw)
        return self._engine

    def _forward_train(self, batch, teacher_knowledge, masks=None):
        eng = self.train_engine()
        self._plan = None  # weights are about to move
        # on this path the optimizer is the CALLER's (torch.optim.*.step(), p.data.copy_(), ...): writes through the re-pointed parameter views do
        # not bump the flat buffer's version counter, so the engine's stamp cannot see them -- every cached operand form of the parameters (packed
        # conv taps, transposes, LSTM column blocks, P32 planes) is rebuilt per forward here (round-2 ADVICE; the native TrainEngine.train_step path
        # knows its own updates and keeps the once-per-update cache)
        eng.invalidate_planes()
        if self.role == "kd_teacher":
            with torch.no_grad():
                return eng.knowledge(batch, mode="train", masks=masks)
        eng.zero_grad()  # accumulation across micro-batches is autograd's job on this path (p.grad += ...)
        accum, eng.accum_grad = eng.accum_grad, 1
        try:
            # reduce=False: on this path the gradients go to autograd (p.grad), so averaging them across ranks is the caller's optimizer-side
            # job, exactly as with the reference's own updater; the engine's buckets stay idle
            rep = eng.forward_backward(batch, teacher_knowledge, mode="train", masks=masks, reduce=False)
        finally:
            eng.accum_grad = accum
        order = ["l1_loss", "mse_loss", "dur_loss", "pitch_loss", "energy_loss", "output_l1_loss", "output_mse_loss", "encoder_loss",
                 "decoder_loss", "prosody_loss", "loss"]
        self.reporter.report([{k: float(rep[k])} for k in order if k in rep])
        names = [k for k, p in self.named_parameters() if p.requires_grad]
        params = [p for _, p in self.named_parameters() if p.requires_grad]
        value = torch.tensor(float(rep["loss"]), dtype=torch.float32, device=eng.dev)
        snap = eng.gflat.clone()  # this forward's gradients: a later forward() (before this loss's backward()) must not overwrite them
        offs = eng.param_offsets()
        return _HipLoss.apply(value, [snap[offs[k][0] : offs[k][0] + offs[k][1]].view(offs[k][2]) for k in names], *params)

    @property
    def base_plot_keys(self):
        keys = ["loss", "l1_loss", "mse_loss", "dur_loss"]
        if self.use_fe_condition:
            keys += ["pitch_loss", "energy_loss"]
        if self.role == "student":
            if self.distill_output_knowledge:
                keys += ["output_l1_loss", "output_mse_loss"]
            if self.distill_encoder_knowledge:
                keys += ["encoder_loss"]
            if self.distill_decoder_knowledge:
                keys += ["decoder_loss"]
            if self.use_fe_condition and self.distill_prosody_knowledge:
                keys += ["prosody_loss"]
        return keys
