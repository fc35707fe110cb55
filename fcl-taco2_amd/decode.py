"""Decode driver on the HIP path — the role of the reference's `tts_decode.py` + `tts.decode` /
`tts_distill.decode` (tts.py:605-686, tts_distill.py:626-718) with the same inputs and outputs:

  * `--model-conf model.json`  = `[idim, odim, vars(train_args)]` as written by the reference (tts.py:341-348);
    `model_module` values of the reference (`nets....:Tacotron2_sa`) are mapped onto this package's classes;
  * `--model` = `snapshot.ep.N` / `model.loss.best` (ESPnet snapshot: dict with a "model" entry, or a bare state_dict)
    or `amp_checkpoint_*.pt` (`{"model", "optimizer", "amp"}`, tts.py:193-198);
  * `--json` = the data manifest (`{"utts": {id: {"output": [{"tokenid": "1 2 3"}, ...]}}}`, preprocess.py make_json);
  * `--out PREFIX` -> PREFIX.ark / PREFIX.scp (Kaldi float matrices, one mel per utterance) and the mean
    frames / second (stream-synchronised, unlike the reference's unsynchronised clock, tts.py:665-667).
Differences: utterances are synthesised `--batch-size` at a time (sorted by length), and `--nj/--job` shard the manifest
by utterance (what `splitjson.py` + one process per split did), one process per GPU, no collective.
"""
import argparse
import importlib
import json
import logging
import os
import queue
import threading
import time

import numpy as np
import torch

from .kaldi_io import ArkScpWriter
from .sharding import shard_utterances

MODULE_MAP = {
    "nets.teacher_training.e2e_tts_tacotron2_sa": "fcl_taco2_amd.nets.teacher_training.e2e_tts_tacotron2_sa",
    "nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_student": "fcl_taco2_amd.nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_student",
    "nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_teacher": "fcl_taco2_amd.nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_teacher",
}


def get_model_conf(conf_path):
    """model.json -> (idim, odim, Namespace)   (espnet.asr.asr_utils.get_model_conf)."""
    with open(conf_path, "rb") as f:
        idim, odim, args = json.load(f)
    return idim, odim, argparse.Namespace(**args)


def dynamic_import(path):
    mod, cls = path.split(":")
    return getattr(importlib.import_module(MODULE_MAP.get(mod, mod)), cls)


def load_state_dict(path):
    obj = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        obj = obj["model"]  # ESPnet snapshot or apex-AMP checkpoint
    return {k[len("module."):] if k.startswith("module.") else k: v for k, v in obj.items()}


def build_model(model_path, conf_path, teacher_conf=None, device="cuda:0", share_proj=None):
    idim, odim, train_args = get_model_conf(conf_path)
    cls = dynamic_import(train_args.model_module)
    com = argparse.Namespace(use_fe_condition=True, append_position=True, distill_output_knowledge=True, distill_encoder_knowledge=True,
                             distill_decoder_knowledge=True, distill_prosody_knowledge=True, is_train=True,
                             share_proj=bool(getattr(train_args, "share_proj", False)) if share_proj is None else share_proj)
    train_args.encoder_resume = None
    if cls.role == "student":
        targs = get_model_conf(teacher_conf)[2] if teacher_conf else argparse.Namespace(use_residual=False)
        model = cls(idim, odim, train_args, com, targs)
    else:
        model = cls(idim, odim, train_args, com)
    model.load_state_dict(load_state_dict(model_path))
    return model.eval().to(device)


def read_manifest(json_path):
    with open(json_path, "rb") as f:
        js = json.load(f)["utts"]
    utts = []
    for k, v in js.items():
        ids = np.array(list(map(int, v["output"][0]["tokenid"].split())), dtype=np.int64)
        if len(v.get("input", [])) > 1 and "feat" in v["input"][1]:  # speaker embedding: `--use-speaker-embedding` data (tts.py:327-332, 285-287)
            from .kaldi_io import read_vec

            path, off = v["input"][1]["feat"].rsplit(":", 1)
            utts.append((k, ids, read_vec(path, int(off))))
        else:
            utts.append((k, ids))
    return utts


class _Slot(object):
    """One pinned landing area of a runner's outputs (mel buffer, frame starts, status word); `free` is set by the writer thread once the ark
    holds its utterances, and waited on before the next device-to-host copy may overwrite it."""

    def __init__(self, frames_cap, odim, batch, slab=None):
        """slab: a pinned float32 tensor of at least words(frames_cap, odim, batch) elements to carve the three buffers from (round 6: one pinned allocation
        per bucket instead of three per slot -- 96 hipHostMalloc calls of a first decode() call -> 4)."""
        if slab is None:
            slab = torch.empty(self.words(frames_cap, odim, batch), dtype=torch.float32, pin_memory=True)
        self.whole = slab[: self.words(frames_cap, odim, batch)]  # the layout of BatchRunner.out: one device-to-host copy fills all three
        n = frames_cap * odim
        self.mel = slab[:n].view(frames_cap, odim)
        self.f0 = slab[n : n + batch + 1].view(torch.int32)
        self.st = slab[n + batch + 1 : n + batch + 2].view(torch.int32)
        self.free = threading.Event()
        self.free.set()

    @staticmethod
    def words(frames_cap, odim, batch):
        from .engine import packed_words

        return packed_words(frames_cap, odim, batch)


class _Pool(object):
    """The capacity graphs of one phoneme-length bucket: BatchRunners (predicted durations, created on first use) sharing the bucket's
    capacities; kept on the plan, so later decode() calls on the same model replay the graphs captured by earlier ones."""

    def __init__(self, plan, batch, t_cap, caps, streams, seed):
        self.plan, self.batch, self.caps, self.t_cap, self.streams, self.seed = plan, batch, caps, t_cap, streams, seed
        self.runners, self.slots, self.next, self.grow = [None] * len(streams), [None] * len(streams), 0, None
        self._slab = None

    def runner(self, j):
        from . import engine

        if self.runners[j] is None:
            pools_ = self.plan.__dict__.setdefault("_decode_cache", {}).setdefault("graph_mempools", {})
            mp = pools_.setdefault(self.streams[j].cuda_stream, torch.cuda.graph_pool_handle()) if SHARE_GRAPH_POOLS else None  # one per pass stream (stream-ordered graphs)
            r = self.runners[j] = engine.BatchRunner(self.plan, self.batch, self.t_cap, self.caps, forced=False, stream=self.streams[j], seed=self.seed + 7919 * j,
                                                     pack_outputs=True, mempool=mp)
            odim = int(r.mel.shape[1])
            w = _Slot.words(self.caps.frames, odim, self.batch)
            if self._slab is None:  # the landing areas of every runner of the bucket: one pinned allocation
                self._slab = torch.empty(2 * len(self.streams) * w, dtype=torch.float32, pin_memory=True)
            self.slots[j] = [_Slot(self.caps.frames, odim, self.batch, self._slab[(2 * j + q) * w : (2 * j + q + 1) * w]) for q in range(2)]
        return self.runners[j]


def _grown_caps(engine, maps, n_rows, scale=1.3):
    """Capacities for the batches that follow the one whose exact maps are `maps` (they are no longer than it): some slack on every count."""
    lmax = max(16, int(maps.lmax * 1.5) + 4)
    bounds = np.full(lmax, 1, dtype=np.int32)
    live = np.minimum(n_rows, (maps.live_rows.astype(np.float64) * scale).astype(np.int64) + 32)
    bounds[: maps.lmax] = live
    bounds[maps.lmax :] = live[-1]
    bounds = np.maximum.accumulate(bounds[::-1])[::-1].astype(np.int32)  # non-increasing, as the loop requires
    # the steps past this batch's own longest duration (+ 2) are slack: the rows that do reach them continue in one launch of the row-tile kernel
    return engine.Caps(lmax, (int(maps.n_frames * scale) + 255) // 256 * 256, bounds, tail_from=maps.lmax + 2)


class _ScaledMaps(object):
    """The exact maps of a calibration batch rescaled to another batch's phoneme count (round 6): durations are a per-phoneme quantity, so a
    later bucket's frame total and live-row profile are the calibrated ones x (its phonemes / the calibrated batch's phonemes); _grown_caps adds the slack
    and a batch that still overflows is reported by the device and redone eagerly (pool.grow), as before."""

    def __init__(self, maps, n_ph_cal, n_ph):
        ratio = float(n_ph) / float(max(n_ph_cal, 1))
        self.lmax = int(maps.lmax)
        self.live_rows = np.ceil(np.asarray(maps.live_rows, dtype=np.float64) * ratio).astype(np.int64)
        self.n_frames = int(np.ceil(maps.n_frames * ratio))


SHARE_GRAPH_POOLS = os.environ.get("FCL_DECODE_SHARE_POOLS", "1") not in ("", "0")  # the graphs of one pass stream share a memory pool (first call: fewer allocations)
ESTIMATE_CAPS = os.environ.get("FCL_DECODE_ESTIMATE_CAPS", "1") not in ("", "0")  # capacities of later buckets from phoneme counts (0: one eager batch per bucket)
MAX_BUCKETS = 8  # captured-graph pools kept per (batch size, depth): least recently used buckets are released beyond this


def release_graphs(model_or_plan):
    """Drop every captured decode graph (and its pinned landing buffers) kept on the model's plan by earlier decode() calls."""
    plan = model_or_plan.plan() if hasattr(model_or_plan, "plan") and callable(model_or_plan.plan) else model_or_plan
    plan.__dict__.pop("_decode_cache", None)


@torch.no_grad()
def decode(model, utts, out_prefix, batch_size=32, seed=137, depth=4, stats=None, keep_graphs=True, max_buckets=None):
    """utts: [(utt_id, ids)] or, for a model with spk_embed_dim, [(utt_id, ids, spemb)].  Writes PREFIX.ark/.scp (out_prefix None: nothing is written); returns (frames, seconds).
    Every batch runs as ONE captured graph with predicted durations (engine.BatchRunner): the host packs the phoneme ids, enqueues one
    graph launch (its first node pulls the packed block into HBM) and one D2H copy of the mel buffer + the per-utterance frame starts, and only synchronises on a batch when it
    harvests it `depth` batches later -- no read-back of the predicted durations in the middle of a pass (rounds 1-2 did one per batch).  The
    mels go from the pinned landing buffer straight into the ark on a writer thread (no intermediate copy on the submitting thread).
    Utterances are sorted by length and bucketed by padded phoneme count (multiples of 16); the first batch of a bucket runs eagerly with the
    host round trip, which both produces its mels and calibrates the bucket's capacities (decoder steps, frames, live rows per step, with
    slack); a later batch that exceeds them is reported by the device (FCL_STATUS_*), re-run eagerly, and the bucket's capacities grow.  The
    captured graphs stay with the model's plan: a second decode() on the same model replays them -- at most `max_buckets` (MAX_BUCKETS) length
    buckets per (batch size, depth), least recently used first out (each holds up to `depth` graphs with private memory pools and two pinned
    landing buffers per graph); keep_graphs=False releases all of them when the call returns (release_graphs() does it later).
    An error inside the loop (a zero-duration phoneme raising like the reference, a failing writer) still drains the device, stops the writer
    thread and closes the ark before it propagates.
    stats (dict, optional): receives `device_seconds` — first submit -> last batch complete on the GPU, excluding the ark writing."""
    from . import engine, ops

    torch.manual_seed(seed)
    order = sorted(range(len(utts)), key=lambda i: -len(utts[i][1]))
    dev = next(model.parameters()).device
    plan = model.plan(dev)
    has_spk = plan.hp.spk_embed_dim is not None
    if has_spk and any(len(u) < 3 for u in utts):
        raise ValueError("fcl-taco2_amd: the model has spk_embed_dim=%d: every utterance needs a speaker embedding" % plan.hp.spk_embed_dim)
    spk_of = (lambda chunk: [u[2] for u in chunk]) if has_spk else (lambda chunk: None)
    frames = 0
    depth = max(1, int(depth))
    cache = plan.__dict__.setdefault("_decode_cache", {})
    streams = engine.shared_streams(dev, depth)  # the process's pass streams (one pool per device: see engine.shared_streams)
    import collections

    pools = cache.setdefault(("pools", batch_size, depth), collections.OrderedDict())
    max_buckets = max(1, int(MAX_BUCKETS if max_buckets is None else max_buckets))
    pending = []
    n_eager = n_graph = n_redo = n_evict = n_est = 0
    # dlayers / prenet_layers / elayers other than 2 / 2 / 1 run on the fp32-operand loop with host row counts (fcl_decoder_weights_t.dlayers ...)
    eager_only = bool(getattr(plan, "generic_decoder", False)) or plan.hp.elayers != 1

    # the ark / scp file is written by a worker thread (file writes release the GIL): storage keeps up with the GPU instead of stalling the loop
    wq = queue.Queue(maxsize=4 * (depth + 1))
    werr = []

    def writer(w):
        while True:
            item = wq.get()
            if item is None:
                return
            chunk, arr, counts, slot = item
            try:
                if not werr:
                    w.write_batch([u[0] for u in chunk], arr[: int(sum(counts))], counts)
            except Exception as e:  # surfaced by the main thread after the join
                werr.append(e)
            finally:
                if slot is not None:
                    slot.free.set()

    def eager(chunk):
        """Host-round-trip pass (calibration / fallback): exact maps of THIS batch."""
        prep = engine.prepare(plan, [u[1] for u in chunk], spembs=spk_of(chunk))
        mel, utt_frames, inter = engine.run(plan, prep, ops.DROP_RNG, seed=int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), return_intermediates=True)
        arr = mel.cpu().numpy()
        wq.put((chunk, arr, list(utt_frames), None))
        return int(arr.shape[0]), inter["maps"]

    def harvest(item):
        nonlocal n_redo
        pool, j, chunk, slot, ev = item
        ev.synchronize()
        if int(slot.st[0]) != 0:  # a capacity of the bucket did not hold for this batch (or a phoneme got duration 0: eager() then raises like the reference)
            pool.runners[j].status.zero_()
            slot.free.set()
            n_redo += 1
            got, pool.grow = eager(chunk)
            return got
        f0 = slot.f0.numpy()
        total = int(f0[len(chunk)])
        wq.put((chunk, slot.mel.numpy()[:total], [int(v) for v in np.diff(f0[: len(chunk) + 1])], slot))
        return total

    class _Discard(dict):  # out_prefix None: synthesis + device-to-host hand-over only (benchmarks)
        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def __setitem__(self, k, v):
            pass

        def write_batch(self, keys, mats, counts):
            pass

    with (ArkScpWriter(out_prefix) if out_prefix is not None else _Discard()) as w, torch.cuda.device(dev):
        th = threading.Thread(target=writer, args=(w,), daemon=True)
        th.start()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            for bi, s in enumerate(range(0, len(order), batch_size)):
                chunk = [utts[i] for i in order[s : s + batch_size]]
                if eager_only:  # structure options beyond the shipped recipes: the launch-by-launch pass (no capacity graphs for them)
                    got, _ = eager(chunk)
                    frames += got
                    n_eager += 1
                    continue
                t_cap = (max(len(u[1]) for u in chunk) + 15) // 16 * 16
                pool = pools.get(t_cap)
                if pool is not None:
                    pools.move_to_end(t_cap)
                if pool is not None and pool.grow is not None:  # a batch overflowed this bucket: drain it, widen the capacities, capture anew
                    for it in [p_ for p_ in pending if p_[0] is pool]:
                        pending.remove(it)
                        frames += harvest(it)
                    g = _grown_caps(engine, pool.grow, batch_size * t_cap, scale=1.6)
                    lmax = max(g.lmax, pool.caps.lmax)
                    caps = engine.Caps(lmax, max(g.frames, pool.caps.frames), np.full(lmax, batch_size * t_cap, np.int32), tail_from=g.tail_from)
                    pool = pools[t_cap] = _Pool(plan, batch_size, t_cap, caps, streams, seed + 31 * bi)
                cal = cache.get(("calibration", batch_size))  # (exact maps, phoneme count) of the first eagerly calibrated batch of this model
                if pool is None and cal is not None and ESTIMATE_CAPS:
                    # a later bucket: capacities ESTIMATED from the phoneme count (no eager batch, no host round trip); the batch itself goes through the graph below
                    est = _ScaledMaps(cal[0], cal[1], sum(len(u[1]) for u in chunk))
                    pool = pools[t_cap] = _Pool(plan, batch_size, t_cap, _grown_caps(engine, est, batch_size * t_cap), streams, seed + 31 * bi)
                    n_est += 1
                    while len(pools) > max_buckets:
                        old_cap, old_pool = next(iter(pools.items()))
                        for it in [p_ for p_ in pending if p_[0] is old_pool]:
                            pending.remove(it)
                            frames += harvest(it)
                        for sl in [x for pair in old_pool.slots if pair for x in pair]:
                            sl.free.wait()
                        del pools[old_cap]
                        n_evict += 1
                if pool is None:  # first batch of the bucket: eager pass = its result + the bucket's calibration
                    got, maps = eager(chunk)
                    frames += got
                    n_eager += 1
                    cache.setdefault(("calibration", batch_size), (maps, sum(len(u[1]) for u in chunk)))
                    pools[t_cap] = _Pool(plan, batch_size, t_cap, _grown_caps(engine, maps, batch_size * t_cap), streams, seed + 31 * bi)
                    while len(pools) > max_buckets:  # least recently used bucket out: its batches in flight are harvested first
                        old_cap, old_pool = next(iter(pools.items()))
                        for it in [p_ for p_ in pending if p_[0] is old_pool]:
                            pending.remove(it)
                            frames += harvest(it)
                        for sl in [x for pair in old_pool.slots if pair for x in pair]:
                            sl.free.wait()  # the writer thread still reads the pinned landing buffers it was handed
                        del pools[old_cap]
                        n_evict += 1
                    continue
                j = pool.next % len(streams)
                pool.next += 1
                for it in [p_ for p_ in pending if p_[1] == j]:  # stream j's previous batch (of any bucket) must have left the runner's static buffers
                    pending.remove(it)
                    frames += harvest(it)
                r = pool.runner(j)
                slot = pool.slots[j][(pool.next // len(streams)) % 2]
                slot.free.wait()  # the writer thread is done with what this landing buffer held
                slot.free.clear()
                try:
                    r.load([u[1] for u in chunk], spembs=spk_of(chunk))
                except Exception:
                    slot.free.set()
                    raise
                r.replay()
                with torch.cuda.stream(r.stream):
                    if r.out is not None and r.out.numel() == slot.whole.numel():
                        slot.whole.copy_(r.out, non_blocking=True)  # mel | frame starts | status in one call (round 6)
                    else:
                        slot.mel.copy_(r.mel, non_blocking=True)
                        slot.f0.copy_(r._frames.utt_frame0, non_blocking=True)
                        slot.st.copy_(r.status, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(r.stream)
                pending.append((pool, j, chunk, slot, ev))
                n_graph += 1
            while pending:
                frames += harvest(pending.pop(0))
            torch.cuda.synchronize()
            dev_secs = time.perf_counter() - t0
        except BaseException:
            # graphs already launched keep copying into the pinned landing buffers: wait for the device, hand the slots back, and let the
            # writer finish what it was given -- then the ark is closed by the `with` and the error propagates
            torch.cuda.synchronize()
            for it in pending:
                it[0].runners[it[1]].status.zero_()
                it[3].free.set()
            del pending[:]
            raise
        finally:
            wq.put(None)
            th.join()
            if not keep_graphs:
                release_graphs(plan)
        secs = time.perf_counter() - t0
        if werr:
            raise werr[0]
    if stats is not None:
        stats.update(device_seconds=dev_secs, eager_batches=n_eager, graph_batches=n_graph, redone_batches=n_redo, buckets=len(pools), evicted_buckets=n_evict,
                     estimated_buckets=n_est)
    return frames, secs


def main(argv=None):
    ap = argparse.ArgumentParser(description="FCL-taco2 mel synthesis on MI355X (reference-compatible decode driver)")
    ap.add_argument("--model", required=True)
    ap.add_argument("--model-conf", required=True)
    ap.add_argument("--teacher-config", default=None, help="model.json of the teacher (student checkpoints trained with KD projections)")
    ap.add_argument("--json", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--nj", type=int, default=1, help="number of utterance shards (one process per GPU)")
    ap.add_argument("--job", type=int, default=0, help="this process's shard (0-based)")
    ap.add_argument("--seed", type=int, default=137)
    ap.add_argument("--verbose", type=int, default=1)
    args = ap.parse_args(argv)
    torch.set_num_threads(4)  # kernels are launched from this thread; a one-thread-per-core intra-op pool spinning beside it slows them
    logging.basicConfig(level=logging.INFO if args.verbose else logging.WARN, format="%(asctime)s %(levelname)s: %(message)s")
    dev = "cuda:%d" % (args.job % max(torch.cuda.device_count(), 1))
    model = build_model(args.model, args.model_conf, args.teacher_config, dev)
    utts = read_manifest(args.json)
    mine = shard_utterances([len(u[1]) for u in utts], args.nj)[args.job]
    out = args.out if args.nj == 1 else "%s.%d" % (args.out, args.job + 1)
    frames, secs = decode(model, [utts[i] for i in mine], out, args.batch_size, args.seed)
    logging.info("average inference speed = %.1f frames / sec. (%d utterances, %d frames)", frames / max(secs, 1e-9), len(mine), frames)
    return frames, secs


if __name__ == "__main__":
    main()
