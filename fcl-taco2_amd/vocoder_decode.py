"""mel -> waveform driver on MI355X: the step the reference delegates to the external `parallel-wavegan-decode --checkpoint vocoder/PWG/PWG.pkl
--feats-scp <decode>/feats.scp --outdir <wav dir>` (inference_student.sh:20-23, inference_teacher.sh:20-23, README.md:46).  Same flag names and
the same outputs (`<outdir>/<utt_id>_gen.wav`, 16-bit PCM at the generator's sampling rate); the generator is fcl_taco2_amd/vocoder.py (published
architecture, parity unpinned: DESIGN.md §6b).

    python -m fcl_taco2_amd.vocoder_decode --checkpoint vocoder/PWG/PWG.pkl --feats-scp exp/student/test/feats.scp --outdir exp/student/test/wav

Utterances are sorted by length and packed into batches of about `--batch-frames` mel frames; batch i + 1 runs on the GPU while batch i's waveform
travels to the host (one packed non-blocking copy on a side stream) and a writer thread encodes the wav files.  `--nj / --job` shard the scp by
utterance (one process per GPU, no collective), like decode.py.
"""
import argparse
import logging
import os
import queue
import threading
import time
import wave

import numpy as np
import torch

from . import kaldi_io, ops
from .sharding import shard_utterances
from .vocoder import CONFIG, PWGPlan, ParallelWaveGANGenerator


def generator_config(checkpoint, config=None):
    """Generator geometry from the `config.yml` parallel_wavegan keeps beside its checkpoints (`generator_params`); the v1 defaults without one.
    Returns (cfg overrides for PWGPlan, sampling rate)."""
    path = config or os.path.join(os.path.dirname(os.path.abspath(checkpoint)), "config.yml")
    if not os.path.exists(path):
        if config:
            raise FileNotFoundError(config)
        return {}, 22050
    import yaml

    with open(path) as f:
        y = yaml.safe_load(f) or {}
    gp = y.get("generator_params", {})
    unsupported = {k: gp[k] for k, dflt in (("in_channels", 1), ("out_channels", 1), ("use_causal_conv", False), ("upsample_net", "ConvInUpsampleNetwork"))
                   if gp.get(k, dflt) != dflt}
    if unsupported:  # (dropout is an inference no-op, bias / weight norm are read from the state dict itself)
        raise NotImplementedError("fcl-taco2_amd: generator_params %r are not supported on the HIP path" % unsupported)
    cfg = {k: gp[k] for k in ("layers", "stacks", "residual_channels", "gate_channels", "skip_channels", "aux_channels", "aux_context_window", "kernel_size")
           if k in gp}
    up = gp.get("upsample_params", {})
    bad_up = {k: up[k] for k, dflt in (("nonlinear_activation", None), ("interpolate_mode", "nearest"), ("use_causal_conv", False),
                                       ("freq_axis_kernel_size", 1)) if up.get(k, dflt) != dflt}
    if bad_up:  # they change the upsampling network's arithmetic; silently ignoring them would produce wrong audio
        raise NotImplementedError("fcl-taco2_amd: upsample_params %r are not supported on the HIP path" % bad_up)
    if "upsample_scales" in up:
        cfg["upsample_scales"] = tuple(int(s) for s in up["upsample_scales"])
    return cfg, int(y.get("sampling_rate", 22050))


def load_checkpoint(path, allow_pickle=False):
    """torch.load of a parallel_wavegan checkpoint ({"model": {"generator": sd}}, or a bare state dict) -> the same nesting with numpy arrays.
    Loaded with weights_only=True (tensors and plain containers only); a checkpoint that needs the full unpickler — which executes whatever code
    the file carries — is read only with allow_pickle=True (`--unsafe-pickle`), i.e. when the caller vouches for the file."""
    def conv(o):
        if torch.is_tensor(o):
            return o.detach().cpu().float().numpy()
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        return o

    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        if not allow_pickle:
            raise RuntimeError("fcl-taco2_amd: %s does not load with weights_only=True (%s); pass --unsafe-pickle / allow_pickle=True only for a "
                               "checkpoint you trust" % (path, str(e).splitlines()[0] if str(e) else type(e).__name__))
        obj = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict) and "generator" in obj["model"]:
        return {"model": {"generator": conv(obj["model"]["generator"])}}  # (the discriminator / optimizer states are not needed)
    return conv(obj)


def write_wav(path, samples, rate):
    """float waveform in [-1, 1] -> 16-bit PCM mono (as `soundfile.write(..., "PCM_16")`: scaled by 0x7FFF, rounded to nearest; clipped here)."""
    pcm = np.clip(np.rint(np.asarray(samples, dtype=np.float64) * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(pcm.tobytes())


def make_batches(lengths, batch_frames):
    """Indices sorted by length (longest first), cut into batches of at most `batch_frames` frames (at least one utterance each)."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    out, cur, tot = [], [], 0
    for i in order:
        if cur and tot + lengths[i] > batch_frames:
            out.append(cur)
            cur, tot = [], 0
        cur.append(i)
        tot += lengths[i]
    if cur:
        out.append(cur)
    return out


def decode(gen, feats, outdir, rate, batch_frames=51200, seed=0, depth=2):
    """feats: list of (utt_id, [T', aux] float32 array).  Writes <outdir>/<utt_id>_gen.wav; returns (samples, seconds)."""
    os.makedirs(outdir, exist_ok=True)
    dev = gen.plan.device
    hop = gen.plan.hop
    batches = make_batches([m.shape[0] for _, m in feats], batch_frames)
    wq, werr = queue.Queue(maxsize=depth + 1), []

    def writer():
        try:
            while True:
                item = wq.get()
                if item is None:
                    return
                for uid, wav in item:
                    write_wav(os.path.join(outdir, uid + "_gen.wav"), wav, rate)
        except Exception as e:  # surfaced by the main thread
            werr.append(e)
            while wq.get() is not None:
                pass

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    with torch.cuda.device(dev):  # the copies' queue on a compute pipe apart from the generator's stream (ops.stream_apart: fcl_hip.h "Compute pipes")
        copy_stream = ops.stream_apart([torch.cuda.current_stream(dev)], device=dev) if os.environ.get("FCL_PLACE_STREAMS", "1") != "0" else torch.cuda.Stream(device=dev)
    slots = [None] * (depth + 1)  # pinned staging, one per batch in flight
    pending, total = [], 0
    t0 = time.perf_counter()

    def harvest(p):
        ids, lens, host, ev = p
        ev.synchronize()
        arr = host.numpy()
        items, s = [], 0
        for uid, n in zip(ids, lens):
            items.append((uid, arr[s : s + n * hop].copy()))
            s += n * hop
        wq.put(items)
        return s

    with torch.cuda.device(dev):
        for bi, idx in enumerate(batches):
            mels = [feats[i][1] for i in idx]
            lens = [int(m.shape[0]) for m in mels]
            packed = torch.from_numpy(np.ascontiguousarray(np.concatenate(mels), dtype=np.float32)).to(dev, non_blocking=True)
            wavs = gen.synthesize_packed(packed, lens, seed=seed + bi)
            n = sum(lens) * hop
            flat = wavs[0]._base if wavs[0]._base is not None else wavs[0]  # the batch's waveforms are views into one buffer
            j = bi % (depth + 1)
            if slots[j] is None or slots[j].numel() < n:
                slots[j] = torch.empty(max(n, 1 << 20), dtype=torch.float32, pin_memory=True)
            done = torch.cuda.Event()
            done.record()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done)
                host = slots[j][:n]
                host.copy_(flat[:n], non_blocking=True)
                flat.record_stream(copy_stream)
                ev = torch.cuda.Event()
                ev.record()
            pending.append(([feats[i][0] for i in idx], lens, host, ev))
            while len(pending) > depth:
                total += harvest(pending.pop(0))
        while pending:
            total += harvest(pending.pop(0))
    wq.put(None)
    th.join()
    torch.cuda.synchronize()
    if werr:
        raise werr[0]
    return total, time.perf_counter() - t0


def main(argv=None):
    ap = argparse.ArgumentParser(description="Parallel WaveGAN decoding on MI355X (drop-in for `parallel-wavegan-decode`)")
    ap.add_argument("--checkpoint", required=True, help="generator checkpoint ({'model': {'generator': state_dict}} or a bare state_dict)")
    ap.add_argument("--feats-scp", "--scp", dest="feats_scp", required=True, help="Kaldi scp of [T', aux] float matrices (decode.py's <out>.scp)")
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--config", default=None, help="parallel_wavegan config.yml (default: next to the checkpoint; v1 geometry without one)")
    ap.add_argument("--batch-frames", type=int, default=51200, help="mel frames per GPU batch")
    ap.add_argument("--nj", type=int, default=1, help="number of utterance shards (one process per GPU)")
    ap.add_argument("--job", type=int, default=0, help="this process's shard (0-based)")
    ap.add_argument("--seed", type=int, default=0, help="seed of the device noise")
    ap.add_argument("--verbose", type=int, default=1)
    ap.add_argument("--unsafe-pickle", action="store_true", help="allow the full unpickler for checkpoints that weights_only=True rejects (runs code "
                    "embedded in the file: trusted checkpoints only)")
    args = ap.parse_args(argv)
    torch.set_num_threads(4)
    logging.basicConfig(level=logging.INFO if args.verbose else logging.WARN, format="%(asctime)s %(levelname)s: %(message)s")
    dev = "cuda:%d" % (args.job % max(torch.cuda.device_count(), 1))
    cfg, rate = generator_config(args.checkpoint, args.config)
    gen = ParallelWaveGANGenerator(PWGPlan(load_checkpoint(args.checkpoint, args.unsafe_pickle), dev, cfg))
    feats = sorted(kaldi_io.read_scp(args.feats_scp).items())
    aux = dict(CONFIG, **cfg)["aux_channels"]
    for uid, m in feats:
        if m.ndim != 2 or m.shape[1] != aux or m.shape[0] < 1:
            raise ValueError("%s: expected a [T', %d] feature matrix, got %r" % (uid, aux, m.shape))
    mine = shard_utterances([m.shape[0] for _, m in feats], args.nj)[args.job]
    samples, secs = decode(gen, [feats[i] for i in mine], args.outdir, rate, args.batch_frames, args.seed)
    audio = samples / float(rate)
    logging.info("generated %d utterances, %.1f s of audio in %.2f s (RTF = %.5f)", len(mine), audio, secs, secs / max(audio, 1e-9))
    return samples, secs


if __name__ == "__main__":
    main()
