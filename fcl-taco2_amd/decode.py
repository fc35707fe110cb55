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
    return [(k, np.array(list(map(int, v["output"][0]["tokenid"].split())), dtype=np.int64)) for k, v in js.items()]


@torch.no_grad()
def decode(model, utts, out_prefix, batch_size=32, seed=137, depth=2):
    """utts: [(utt_id, ids)].  Writes PREFIX.ark/.scp; returns (frames, seconds).
    Pipelined: a batch's packed mel [F, odim] leaves the device in ONE non-blocking copy into pinned memory on a copy stream, and up to `depth`
    batches are in flight, so the host prepares and enqueues batch i+1 (and writes batch i-1 to the ark) while the GPU runs batch i.  The clock
    covers first submit -> last mel on the host (the only synchronisation points are the predicted-duration read-back inside a pass and the
    copy-complete events)."""
    torch.manual_seed(seed)
    order = sorted(range(len(utts)), key=lambda i: -len(utts[i][1]))
    dev = next(model.parameters()).device
    frames = 0
    copy_stream = torch.cuda.Stream(device=dev)
    pinned = [None] * (depth + 1)
    pending = []

    # the ark / scp file is written by a worker thread (file writes release the GIL): storage keeps up with the GPU instead of stalling the loop
    wq = queue.Queue(maxsize=2 * (depth + 1))
    werr = []

    def writer(w):
        while True:
            item = wq.get()
            if item is None:
                return
            try:
                if not werr:
                    for uid, arr in item:
                        w[uid] = arr
            except Exception as e:  # surfaced by the main thread after the join
                werr.append(e)

    def harvest(item):
        chunk, host, counts, ev = item
        ev.synchronize()
        arr, s0, out = host.numpy().copy(), 0, []  # one copy out of the pinned slot, which is reused `depth + 1` batches later
        for (uid, _), c in zip(chunk, counts):
            out.append((uid, arr[s0 : s0 + c]))
            s0 += c
        wq.put(out)
        return s0

    with ArkScpWriter(out_prefix) as w, torch.cuda.device(dev):
        th = threading.Thread(target=writer, args=(w,), daemon=True)
        th.start()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for bi, s in enumerate(range(0, len(order), batch_size)):
            chunk = [utts[i] for i in order[s : s + batch_size]]
            mels = model.inference_batch([x for _, x in chunk])  # views into one packed device buffer, utterance-major
            counts = [int(m.shape[0]) for m in mels]
            total = sum(counts)
            packed = mels[0]._base if mels[0]._base is not None else torch.cat(mels)
            packed = packed[:total]
            slot = bi % (depth + 1)
            if pinned[slot] is None or pinned[slot].shape[0] < total:
                pinned[slot] = torch.empty(max(total * 5 // 4, 1), packed.shape[1], dtype=torch.float32, pin_memory=True)
            host = pinned[slot][:total]
            copy_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(copy_stream):
                host.copy_(packed, non_blocking=True)
                packed.record_stream(copy_stream)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            pending.append((chunk, host, counts, ev))
            while len(pending) > depth:  # the slot written `depth + 1` batches ago is free again only after its harvest
                frames += harvest(pending.pop(0))
        while pending:
            frames += harvest(pending.pop(0))
        wq.put(None)
        th.join()
        torch.cuda.synchronize()
        secs = time.perf_counter() - t0
        if werr:
            raise werr[0]
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
    mine = shard_utterances([len(x) for _, x in utts], args.nj)[args.job]
    out = args.out if args.nj == 1 else "%s.%d" % (args.out, args.job + 1)
    frames, secs = decode(model, [utts[i] for i in mine], out, args.batch_size, args.seed)
    logging.info("average inference speed = %.1f frames / sec. (%d utterances, %d frames)", frames / max(secs, 1e-9), len(mine), frames)
    return frames, secs


if __name__ == "__main__":
    main()
