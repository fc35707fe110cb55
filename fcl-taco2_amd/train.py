"""Training driver on the HIP path — the role of the reference's `tts_train.py` + `tts.train` / `tts_distill.train`
(tts.py:309-603, tts_distill.py:312-624) reduced to what the hot path needs, with the reference's flag names, inputs and artefacts:

  inputs   `--train-json/--valid-json` data manifests (preprocess.py:199-241: per utterance input1 mel [L,80], input2 durations [T,1],
           input3 f0 [T,1], input4 energy [T,1] as .npy paths, output tokenid string); for KD `--teacher-conf model.json --teacher-model
           amp_checkpoint_*.pt | snapshot` (tts_distill.py:367-398: the teacher is frozen and LEFT IN TRAIN MODE);
  batching `--batch-size`, `--batch-sort-key shuffle|input|output` (shuffle = the shipped recipes: fixed-size random batches), every batch
           sorted by descending phoneme count (io_utils_fcl.py:316-318), laid out by the bit-exact vectorised CustomConverter (tts.py:215-306);
  step     fcl_taco2_amd.training.TrainEngine: forward/backward on the HIP path, `--accum-grad`, bucketed gradient all-reduce when launched
           under torch.distributed.run (one process per GPU, batches sharded round-robin), clip `--grad-clip`, NaN guard, Adam `--lr --eps`;
  outputs  `outdir/model.json` (`[idim, odim, vars(args)]`, tts.py:341-348), `outdir/snapshot.ep.N` ({"model", "optimizer", "epoch",
           "iteration"}), `outdir/amp_checkpoint_ep{N}.pt` ({"model", "optimizer" (torch.optim.Adam layout), "amp": apex's
           `amp.state_dict()` layout {"loss_scaler0": {"loss_scale", "unskipped"}}}, tts.py:193-198 — the file the reference's KD recipe
           loads its teacher from and its `--amp-checkpoint` resume reads with `amp.load_state_dict`, tts.py:418-423), `outdir/model.loss.best` (bare state_dict of the best validation
           loss, ESPnet torch_save), `outdir/log` (JSON list of per-epoch means of every reported loss, main/ and validation/main/ prefixed).
Not here: chainer extensions (plots, tensorboard), sortagrad, apex AMP, early stopping.
"""
import argparse
import json
import logging
import os
import time

import numpy as np
import torch

from .converter import CustomConverter
from .decode import build_model, dynamic_import, load_state_dict


# ---- data ------------------------------------------------------------------------------------------------------------------------------
def read_train_manifest(path):
    with open(path, "rb") as f:
        utts = json.load(f)["utts"]
    out = []
    for uid, info in utts.items():
        inputs = {i["name"]: i for i in info["input"]}
        out.append(dict(id=uid, tokenid=info["output"][0]["tokenid"], mel=inputs["input1"]["feat"], dur=inputs["input2"]["feat"],
                        f0=inputs["input3"]["feat"], energy=inputs["input4"]["feat"], ilen=int(info["output"][0]["shape"][0]),
                        olen=int(inputs["input1"]["shape"][0]), num_phns=int(info["output"][0]["shape"][1])))
    return out


def make_batchset(utts, batch_size, sort_key="shuffle", seed=1, min_batch_size=1):
    """batchfy_fcl.make_batchset reduced to count="seq": "shuffle" = random fixed-size batches (the shipped recipes), "input"/"output" =
    sort by that length (descending) and cut; batches smaller than min_batch_size (= ranks) are dropped like the reference does for ngpu > 1."""
    idx = list(range(len(utts)))
    if sort_key == "shuffle":
        np.random.RandomState(seed).shuffle(idx)
    elif sort_key in ("input", "output"):
        idx.sort(key=lambda i: -(utts[i]["ilen"] if sort_key == "input" else utts[i]["olen"]))
    else:
        raise ValueError("batch-sort-key must be shuffle, input or output")
    batches = [[utts[i] for i in idx[s : s + batch_size]] for s in range(0, len(idx), batch_size)]
    return [b for b in batches if len(b) >= min_batch_size]


def load_batch(batch, cache=None):
    """LoadInputsAndTargets(mode="tts", use_second_target=True) for one batch: lists sorted by descending phoneme count."""
    def get(p):
        if cache is not None and p in cache:
            return cache[p]
        a = np.load(p)
        if cache is not None:
            cache[p] = a
        return a

    rows = []
    for u in batch:
        x = np.fromiter(map(int, u["tokenid"].split()), dtype=np.int64)
        if len(x) == 0:
            continue
        rows.append((x, get(u["mel"]).astype(np.float32), np.asarray(get(u["dur"]), dtype=np.float32).reshape(-1, 1),
                     np.asarray(get(u["f0"]), dtype=np.float32).reshape(-1, 1), np.asarray(get(u["energy"]), dtype=np.float32).reshape(-1, 1)))
    rows.sort(key=lambda r: -len(r[0]))
    xs, ys, ds, f0, en = (list(c) for c in zip(*rows))
    return xs, ys, None, ds, f0, en


PINNED = ("xs", "ys", "extras", "f0", "energy")
_ring = None


def to_device(batch, dev):
    """Hand the float / id tensors of a converted batch to the GPU through fixed pinned staging buffers without blocking the host (the integer
    layout tensors stay on the host: the engine builds its index maps from them).  Inputs are resident when the step's kernels reach them."""
    global _ring
    if _ring is None:
        from .training import PinnedRing

        _ring = PinnedRing()
    batch.update(_ring.upload({k: batch[k] for k in PINNED}, dev))
    return batch


class _BatchDataset(torch.utils.data.Dataset):
    """One item = one converted batch incl. the host half of the engine's index maps: everything numpy-heavy happens in the loader processes
    (the role of the reference's ChainerDataLoader workers, tts.py:509-528; --num-iter-processes)."""

    def __init__(self, batches, conv, cache):
        self.batches, self.conv, self.cache = batches, conv, cache

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, i):
        from .training import build_maps_host

        b = self.conv([load_batch(self.batches[i], self.cache)])
        b["_fcl_maps_host"] = build_maps_host(b)
        return b


def batch_feed(batches, conv, cache, workers):
    """Iterator of converted batches in order; workers > 0 = forked loader processes (they never touch the GPU), 0 = inline."""
    ds = _BatchDataset(batches, conv, cache)
    if workers <= 0 or len(batches) == 0:
        return (ds[i] for i in range(len(ds)))

    def feed():
        # The loader processes are forked from a process that holds a GPU context and its runtime threads: a worker can die at birth (seen once in a few
        # hundred epochs: SIGSEGV before its first item).  Items are a pure function of their index, so a dead loader costs the epoch its prefetching, not the
        # run: the remaining batches are converted in this process, loudly.
        done = 0
        try:
            for b in torch.utils.data.DataLoader(ds, batch_size=None, shuffle=False, num_workers=workers, prefetch_factor=2, persistent_workers=False):
                done += 1
                yield b
        except RuntimeError as e:
            if "DataLoader worker" not in str(e):
                raise
            logging.warning("fcl-taco2_amd: %s -- converting the remaining %d batches of this epoch in the training process", e, len(ds) - done)
            for i in range(done, len(ds)):
                yield ds[i]

    return feed()


# ---- checkpoints -----------------------------------------------------------------------------------------------------------------------
def adam_state_dict(engine):
    """The engine's flat Adam moments in torch.optim.Adam.state_dict() layout (parameters in model.parameters() order)."""
    names = [k for k, _ in engine.model.named_parameters()]
    offs = engine.param_offsets()
    state = {}
    for i, k in enumerate(names):
        o, n, shape = offs[k]
        state[i] = {"step": torch.tensor(float(engine.step_count)), "exp_avg": engine.mflat[o : o + n].view(shape).detach().cpu().clone(),
                    "exp_avg_sq": engine.vflat[o : o + n].view(shape).detach().cpu().clone()}
    group = {"lr": engine.lr, "betas": tuple(engine.betas), "eps": engine.eps, "weight_decay": getattr(engine, "weight_decay", 0.0), "amsgrad": False, "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(engine, sd):
    names = [k for k, _ in engine.model.named_parameters()]
    offs = engine.param_offsets()
    for i, k in enumerate(names):
        st = sd["state"].get(i)
        if st is None:
            continue
        o, n, shape = offs[k]
        engine.mflat[o : o + n].view(shape).copy_(st["exp_avg"])
        engine.vflat[o : o + n].view(shape).copy_(st["exp_avg_sq"])
        engine.step_count = int(float(st["step"]))
    g = sd["param_groups"][0]
    engine.lr, engine.eps, engine.betas = g["lr"], g["eps"], tuple(g["betas"])
    engine.weight_decay = float(g.get("weight_decay", 0.0))  # (torch.optim.Adam.load_state_dict restores the group's options too)


def amp_state_dict(engine):
    """The `amp` entry of an amp_checkpoint in the layout apex's `amp.state_dict()` writes and `amp.load_state_dict()` reads (tts.py:193-198, 418-423):
    one entry per loss scaler, {"loss_scale": float, "unskipped": int}.  Arithmetic here is bf16 without loss scaling, so the scaler is
    carried, not used: a state loaded from a reference checkpoint is written back unchanged, a fresh run writes apex's initial dynamic scaler
    (2^16, 0 unskipped steps) so that the reference's O1 recipe can resume from the file."""
    from collections import OrderedDict

    st = getattr(engine, "amp_state", None)
    if st:
        return OrderedDict((k, dict(v)) for k, v in st.items())
    return OrderedDict([("loss_scaler0", {"loss_scale": 65536.0, "unskipped": 0})])


def load_amp_state_dict(engine, sd):
    """Keep a checkpoint's `amp` entry (apex layout; None / missing in files written before round 6) on the engine for the next save."""
    engine.amp_state = None
    if isinstance(sd, dict):
        keep = {k: {"loss_scale": float(v["loss_scale"]), "unskipped": int(v["unskipped"])} for k, v in sd.items()
                if "loss_scaler" in k and isinstance(v, dict) and "loss_scale" in v and "unskipped" in v}
        engine.amp_state = keep or None
    return engine.amp_state


def save_checkpoints(outdir, engine, epoch, iteration, valid_loss, best):
    model_sd = {k: v.detach().cpu().clone() for k, v in engine.model.state_dict().items()}
    opt = adam_state_dict(engine)
    torch.save({"model": model_sd, "optimizer": opt, "epoch": epoch, "iteration": iteration, "amp": amp_state_dict(engine)}, os.path.join(outdir, "snapshot.ep.%d" % epoch))
    torch.save({"model": model_sd, "optimizer": opt, "amp": amp_state_dict(engine)}, os.path.join(outdir, "amp_checkpoint_ep%d.pt" % epoch))
    if valid_loss is not None and valid_loss < best:
        torch.save(model_sd, os.path.join(outdir, "model.loss.best"))
        return valid_loss
    return best


# ---- the loop ----------------------------------------------------------------------------------------------------------------------------
def get_parser():
    p = argparse.ArgumentParser(description="Train FCL-taco2 (teacher or KD student) on the MI355X HIP path")
    p.add_argument("--outdir", required=True)
    p.add_argument("--train-json", required=True)
    p.add_argument("--valid-json", default=None)
    p.add_argument("--model-module", default="fcl_taco2_amd.nets.teacher_training.e2e_tts_tacotron2_sa:Tacotron2_sa")
    p.add_argument("--teacher-conf", default=None, help="KD: the teacher's model.json")
    p.add_argument("--teacher-model", default=None, help="KD: the teacher's amp_checkpoint_*.pt / snapshot")
    p.add_argument("--resume", "-r", default=None, help="snapshot.ep.N to continue from")
    p.add_argument("--amp-checkpoint", default=None, help="amp_checkpoint_*.pt to initialise model, optimizer and the carried amp scaler state from "
                                                           "(teacher_parser.py:312-315, tts.py:418-423)")
    p.add_argument("--encoder-resume", default=None, type=str, help="state_dict file of the ENCODER alone, loaded in place of its initialisation "
                   "(tts_train.py:319-323; encoder_sa.py:117-120).  --pretrained-model (a model flag) loads the whole model")
    p.add_argument("--batch-size", "-b", default=32, type=int)
    p.add_argument("--batch-sort-key", default="shuffle", choices=["shuffle", "input", "output"])
    p.add_argument("--epochs", "-e", default=30, type=int)
    p.add_argument("--lr", default=1e-3, type=float)
    p.add_argument("--eps", default=1e-6, type=float)
    p.add_argument("--weight-decay", default=0.0, type=float)
    p.add_argument("--grad-clip", default=1.0, type=float)
    p.add_argument("--accum-grad", default=1, type=int)
    p.add_argument("--seed", default=1, type=int)
    p.add_argument("--save-interval-epochs", default=1, type=int)
    p.add_argument("--eval-interval-epochs", default=1, type=int)
    p.add_argument("--report-interval-iters", default=100, type=int)
    p.add_argument("--keep-all-data-on-mem", action="store_true")
    p.add_argument("--num-iter-processes", default=0, type=int, help="loader processes (npy reads, converter, host index maps); 0 = inline, "
                   "which keeps the loop GPU-bound at the shipped batch sizes (converter + maps ~4 ms per batch)")
    p.add_argument("--host-threads", default=4, type=int, help="torch intra-op CPU threads of the training process (see train())")
    p.add_argument("--use-amp", default=False, type=lambda s: str(s).lower() in ("1", "true", "yes"),
                   help="mixed precision as in the shipped recipes (teacher_parser.py:306, student_model_training.sh:29: apex O1): here bf16-rounded GEMM "
                        "operands with fp32 accumulation, fp32 master weights / norms / losses / Adam, no loss scaling (TrainEngine(amp='bf16'))")
    p.add_argument("--use-fe-condition", default=True, type=lambda s: str(s).lower() in ("1", "true", "yes"))
    p.add_argument("--append-position", default=True, type=lambda s: str(s).lower() in ("1", "true", "yes"))
    for k in ("output", "encoder", "decoder", "prosody"):
        p.add_argument("--distill-%s-knowledge" % k, default=True, type=lambda s: str(s).lower() in ("1", "true", "yes"))
    p.add_argument("--share-proj", default=False, type=lambda s: str(s).lower() in ("1", "true", "yes"))
    p.add_argument("--is-train", default=True, type=lambda s: str(s).lower() in ("1", "true", "yes"))
    return p


def _mean_reports(reps):
    keys = sorted({k for r in reps for k in r})
    return {k: float(np.mean([r[k] for r in reps if k in r])) for k in keys}


def train(argv=None):
    parser = get_parser()
    args, _ = parser.parse_known_args(argv)
    cls = dynamic_import(args.model_module)
    cls.add_arguments(parser)
    args = parser.parse_args(argv)
    # The host side of a step is ~1000 kernel launches issued from this thread.  torch's default intra-op pool (one thread per core: 256 on an
    # MI355X host) spins after every small CPU op of the converter and slows those launches 4x (46 vs 16 ms per KD update, measured); cap it.
    torch.set_num_threads(max(1, args.host_threads))
    torch.manual_seed(args.seed)  # before any module is built: initial weights depend on --seed only (set_deterministic_pytorch, tts.py:321)
    np.random.seed(args.seed)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = "cuda:%d" % local_rank
    if not torch.cuda.is_available():
        raise RuntimeError("fcl-taco2_amd: training needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(dev))
    from .training import KDPipeline, TrainEngine

    train_utts = read_train_manifest(args.train_json)
    valid_utts = read_train_manifest(args.valid_json) if args.valid_json else []
    idim, odim = train_utts[0]["num_phns"], int(np.load(train_utts[0]["mel"]).shape[1])
    os.makedirs(args.outdir, exist_ok=True)
    if rank == 0:
        with open(os.path.join(args.outdir, "model.json"), "wb") as f:
            f.write(json.dumps((idim, odim, vars(args)), indent=4, ensure_ascii=False, sort_keys=True).encode("utf_8"))
    kd = cls.role == "student"
    if kd:
        if not (args.teacher_conf and args.teacher_model):
            raise ValueError("KD training needs --teacher-conf and --teacher-model (tts_distill.py:370-371)")
        from .decode import get_model_conf

        teacher = build_model(args.teacher_model, args.teacher_conf, None, dev)
        if teacher.role != "kd_teacher":  # a teacher trained with the plain class: same parameter tree, KD class returns the knowledge tuple
            from .nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_teacher import Tacotron2_sa as KDTeacher

            t_idim, t_odim, targs = get_model_conf(args.teacher_conf)
            kt = KDTeacher(t_idim, t_odim, targs, argparse.Namespace(use_fe_condition=True, append_position=args.append_position))
            kt.load_state_dict(teacher.state_dict())
            teacher = kt.to(dev)
        for p_ in teacher.parameters():
            p_.requires_grad = False  # tts_distill.py:397-398
        teacher.train()  # the reference never puts the teacher in eval mode
        teng = TrainEngine(teacher, amp="bf16" if args.use_amp else None)
        model = cls(idim, odim, args, args, get_model_conf(args.teacher_conf)[2])
    else:
        teng = None
        model = cls(idim, odim, args, args)
    model = model.to(dev)
    eng = TrainEngine(model, lr=args.lr, eps=args.eps, grad_clip=args.grad_clip, accum_grad=args.accum_grad, seed=args.seed * 1000 + rank, weight_decay=args.weight_decay,
                      amp="bf16" if args.use_amp else None)
    epoch0, iteration = 0, 0
    if args.resume:
        snap = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.load_state_dict(load_state_dict(args.resume))
        if "optimizer" in snap:
            load_adam_state_dict(eng, snap["optimizer"])
        load_amp_state_dict(eng, snap.get("amp"))
        epoch0, iteration = int(snap.get("epoch", 0)), int(snap.get("iteration", 0))
    if getattr(args, "amp_checkpoint", None):  # tts.py:418-423: model, optimizer and amp state from the reference's own checkpoint layout
        ck = torch.load(args.amp_checkpoint, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["model"])
        load_adam_state_dict(eng, ck["optimizer"])
        load_amp_state_dict(eng, ck.get("amp"))
    conv = CustomConverter(getattr(args, "reduction_factor", 1), args.use_fe_condition, args.append_position)  # tts.py:361-364
    cache = {} if args.keep_all_data_on_mem else None
    log, best = [], float("inf")
    for epoch in range(epoch0 + 1, args.epochs + 1):
        batches = make_batchset(train_utts, args.batch_size, args.batch_sort_key, seed=args.seed + epoch, min_batch_size=1)
        n_iter = len(batches) // world  # every rank takes the same number of steps: collectives stay matched
        reps, t0, frames = [], time.time(), 0
        micro = 0
        pipe = KDPipeline(teng, eng) if (kd and eng.accum_grad == 1) else None  # frozen teacher one batch ahead on a second stream
        feed = batch_feed([batches[it * world + rank] for it in range(n_iter)], conv, cache, args.num_iter_processes)
        nxt = next(feed, None)
        nxt = to_device(nxt, dev) if nxt is not None else None
        for it in range(n_iter):
            batch = nxt
            nxt = next(feed, None)
            if nxt is not None:
                nxt = to_device(nxt, dev)
            if pipe is not None:
                rep = pipe.step(batch, nxt)
                iteration += 1
            else:
                know = teng.knowledge(batch, mode="train") if kd else None
                if micro == 0:
                    eng.zero_grad()
                rep = eng.forward_backward(batch, know, mode="train", reduce=micro == eng.accum_grad - 1)  # all-reduce on the last micro-batch only
                micro += 1
                if micro == eng.accum_grad:  # tts.py:166-171
                    eng.optimizer_step()
                    micro = 0
                    iteration += 1
            reps.append(rep)
            frames += int(sum(int(v) for v in batch["olens"]))
            if rank == 0 and args.report_interval_iters and (it + 1) % args.report_interval_iters == 0:
                m = _mean_reports(reps[-args.report_interval_iters:])
                logging.info("epoch %d iter %d/%d loss %.4f (%.0f frames/s/GPU)", epoch, it + 1, n_iter, m["loss"], frames / (time.time() - t0))
        torch.cuda.synchronize()
        entry = {"epoch": epoch, "iteration": iteration, "elapsed_time": time.time() - t0}
        entry.update({"main/" + k: v for k, v in _mean_reports(reps).items()})
        valid_loss = None
        if valid_utts and epoch % args.eval_interval_epochs == 0 and rank == 0:
            model.eval()  # CustomEvaluator: student in eval mode, teacher untouched (tts_distill.py:91-111)
            vreps = []
            with torch.no_grad():
                for vb in make_batchset(valid_utts, args.batch_size, "input"):
                    batch = conv([load_batch(vb, cache)])
                    kw = dict(teacher_knowledge=teng.knowledge(batch, mode="train")) if kd else {}
                    model(**{k: v for k, v in batch.items() if not k.startswith("_")}, **kw)
                    vreps.append(dict(model.reporter.last))
            model.train()
            vm = _mean_reports(vreps)
            entry.update({"validation/main/" + k: v for k, v in vm.items()})
            valid_loss = vm.get("loss")
        if rank == 0:
            log.append(entry)
            with open(os.path.join(args.outdir, "log"), "w") as f:
                json.dump(log, f, indent=4)
            if epoch % args.save_interval_epochs == 0 or epoch == args.epochs:
                best = save_checkpoints(args.outdir, eng, epoch, iteration, valid_loss, best)
            logging.info("epoch %d done: %s", epoch, json.dumps(entry))
        if world > 1:
            dist.barrier()
    if world > 1:
        dist.destroy_process_group()
    return log


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    train(argv)


if __name__ == "__main__":
    main()
