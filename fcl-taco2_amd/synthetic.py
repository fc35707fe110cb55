"""Closed-form weights, dropout masks and synthetic LJSpeech-shape workloads.

There is no network for checkpoints or datasets, so weights come from a closed-form generator keyed on
the state_dict name (no dependence on torch's RNG stream, no weight files in the repo) and inputs from
`numpy.random.RandomState(seed)`.  Used by the golden generator, the tests and bench.py alike, so the
HIP path, the oracle and the real reference all see byte-identical parameters.
"""
import zlib
from collections import OrderedDict

import numpy as np


def _key(name):
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def closed_form_tensor(name, shape):
    """float32 ndarray for state_dict entry `name` (int64 scalar 0 for num_batches_tracked)."""
    if name.endswith("num_batches_tracked"):
        return np.zeros((), dtype=np.int64)
    k = _key(name)
    n = int(np.prod(shape)) if len(shape) else 1
    b = 1.0 + (k % 9973) / 9973.0  # radians per element, in [1, 2)
    phase = (k >> 8) % 6283 / 1000.0
    wave = np.sin(np.arange(n, dtype=np.float64) * b + phase)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        if name.endswith("embed.weight") and len(shape) == 2 and "proj" not in name:
            v = wave  # embedding table: O(1) entries
        else:
            v = wave * (1.4 / np.sqrt(fan_in))
        v = v.reshape(shape)
        if name == "enc.embed.weight":
            v[0] = 0.0  # padding_idx row (reference encoder_sa.py:58)
    elif leaf == "running_var":
        v = (1.0 + 0.5 * wave * wave).reshape(shape)
    elif leaf == "weight":  # BatchNorm / LayerNorm scale
        v = (1.0 + 0.1 * wave).reshape(shape)
    else:  # biases, running_mean
        v = (0.1 * wave).reshape(shape)
    return np.ascontiguousarray(v, dtype=np.float32)


def closed_form_state_dict(spec):
    """{name: ndarray} for an ordered {name: shape} manifest (hparams.param_spec)."""
    return OrderedDict((k, closed_form_tensor(k, tuple(s))) for k, s in spec.items())


def closed_form_keep_mask(shape, seed):
    """uint8 {0,1} keep-mask with P(keep)=0.5 from an integer hash of the flat index (no RNG stream).

    Used to inject the reference's always-on prenet dropout (decoder_sa.py:156-158) identically into
    the reference, the oracle and the HIP kernels.
    """
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    h = (idx * np.uint64(2654435761) + np.uint64(seed & 0xFFFFFFFF) * np.uint64(40503)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    return ((h >> np.uint64(7)) & np.uint64(1)).astype(np.uint8).reshape(shape)


def durations(rng, n, lam=10.0, lo=1, hi=50):
    """clip(Poisson(lam), lo, hi) — LJSpeech-like phoneme durations (preprocess.py:203 caps at 50)."""
    return np.clip(rng.poisson(lam, size=n), lo, hi).astype(np.int64)


def utterance_c1(vocab=80, n_phonemes=80, seed=137):
    """BASELINE config 1 input: one utterance, forced durations."""
    rng = np.random.RandomState(seed)
    x = rng.randint(1, vocab, size=n_phonemes).astype(np.int64)
    d = durations(rng, n_phonemes)
    return x, d


def batch_c2(vocab=80, batch=32, t_lo=60, t_hi=100, seed=1234, zero_frac=0.0):
    """BASELINE config 2 input: `batch` utterances, phoneme counts U{t_lo..t_hi} sorted descending,
    per-phoneme forced durations.  Returns lists of id / duration arrays."""
    rng = np.random.RandomState(seed)
    lens = np.sort(rng.randint(t_lo, t_hi + 1, size=batch))[::-1]
    xs, ds = [], []
    for n in lens:
        xs.append(rng.randint(1, vocab, size=int(n)).astype(np.int64))
        d = durations(rng, int(n))
        if zero_frac > 0.0:
            d[rng.rand(int(n)) < zero_frac] = 0
        ds.append(d)
    return xs, ds


def training_batch(hp_odim=80, vocab=80, batch=4, t_lo=5, t_hi=9, seed=7, zero_frac=0.1, lam=3.0, hi=8):
    """Small teacher-forced batch in the layout the reference's loader hands CustomConverter
    (tts.py:228): lists of xs [T], ys [L, odim], durations [T], f0 [T,1], energy [T,1]; sorted by
    descending phoneme count (io_utils_fcl.py:316-318)."""
    rng = np.random.RandomState(seed)
    lens = np.sort(rng.randint(t_lo, t_hi + 1, size=batch))[::-1]
    xs, ys, ds, f0, en = [], [], [], [], []
    for n in lens:
        n = int(n)
        xs.append(rng.randint(1, vocab, size=n).astype(np.int64))
        d = np.clip(rng.poisson(lam, size=n), 1, hi).astype(np.int64)
        z = rng.rand(n) < zero_frac
        z[0] = False  # keep every utterance non-empty
        d[z] = 0
        ds.append(d.astype(np.float32).reshape(n, 1))  # 'extras' carries durations as floats
        ys.append(rng.randn(int(d.sum()), hp_odim).astype(np.float32))
        p = rng.randn(n, 1).astype(np.float32)
        p[rng.rand(n) < 0.3] = 0.0  # unvoiced
        f0.append(p)
        en.append(rng.randn(n, 1).astype(np.float32))
    return xs, ys, ds, f0, en


def build_model(role, hp, thp=None, device="cuda:0", share_proj=True, weights="closed_form", seed=0):
    """A plug-in model ("teacher" | "kd_teacher" | "student"), the way bench / tools / smoke build one.  weights: "closed_form" (the goldens'
    generator: a stiff, badly conditioned net -- fine for timing and forward parity) or "init" (the reference's own initialisation, i.e. what
    training starts from: torch defaults + xavier on the convolutions, drawn from torch.manual_seed(seed))."""
    import argparse

    import torch

    from . import hparams as HP
    from .nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_student import Tacotron2_sa as Student
    from .nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_teacher import Tacotron2_sa as KDTeacher
    from .nets.teacher_training.e2e_tts_tacotron2_sa import Tacotron2_sa as Teacher

    def ns(h):
        return argparse.Namespace(embed_dim=h.embed_dim, eunits=h.eunits, econv_chans=h.econv_chans, dunits=h.dunits, prenet_units=h.prenet_units,
                                  postnet_chans=h.postnet_chans, use_residual=False, use_masking=True, dropout_rate=h.dropout_rate,
                                  duration_predictor_chans=h.duration_predictor_chans)

    com = argparse.Namespace(use_fe_condition=True, append_position=True, distill_output_knowledge=True, distill_encoder_knowledge=True,
                             distill_decoder_knowledge=True, distill_prosody_knowledge=True, is_train=True, share_proj=share_proj)
    if weights == "init":
        torch.manual_seed(seed)
    elif weights != "closed_form":
        raise ValueError("weights must be 'closed_form' or 'init'")
    if role == "student":
        m = Student(hp.idim, hp.odim, ns(hp), com, ns(thp))
        spec = HP.param_spec(hp, thp, share_proj)
    else:
        m = (Teacher if role == "teacher" else KDTeacher)(hp.idim, hp.odim, ns(hp), com)
        spec = HP.param_spec(hp)
    if weights == "init":
        return m.to(device)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in closed_form_state_dict(spec).items()})
    return m.to(device)


def positive_duration_head(sd, mean_log=2.3, spread=0.35):
    """A copy of a closed-form state dict whose duration predictor predicts usable durations: the closed-form head gives log-durations ~ N(0, 1),
    i.e. mostly 0 frames ("random weights predict ~ 0 frames", and a predicted 0 is an error, D9).  Scaling the head's weight by
    `spread` and setting its bias to `mean_log` gives log-durations ~ N(mean_log, spread): 4 .. 25 frames per phoneme, LJSpeech-like, never 0 --
    so the predicted-duration path (predictor -> rounding -> device-built row maps -> decoder) can be exercised and timed with synthetic weights."""
    out = OrderedDict((k, np.array(v, copy=True)) for k, v in sd.items())
    out["duration_predictor.linear.weight"] = (out["duration_predictor.linear.weight"] * spread).astype(np.float32)
    out["duration_predictor.linear.bias"] = np.full_like(out["duration_predictor.linear.bias"], mean_log)
    return out
