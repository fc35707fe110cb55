"""The training driver: manifest -> batches -> converter on CPU; a tiny end-to-end teacher run, a KD run from the teacher's
own amp_checkpoint, resume, and the decode driver reading the result on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

import fcl_taco2_amd  # noqa: F401
from fcl_taco2_amd import synthetic as SYN
from fcl_taco2_amd import train as TR


def write_dataset(root, n=6, seed=3, odim=8, vocab=12):
    """The layout preprocess.py writes: .npy per utterance and feature + {train,valid}_data.json manifests."""
    xs, ys, ds, f0, en = SYN.training_batch(odim, vocab, batch=n, t_lo=4, t_hi=7, seed=seed)
    utts = {}
    for i in range(n):
        uid = "LJ%03d" % i
        paths = {}
        for name, arr in (("mels", ys[i]), ("durations_MFA", ds[i].astype(np.int64)), ("f0", f0[i]), ("energy", en[i])):
            os.makedirs(os.path.join(root, name), exist_ok=True)
            paths[name] = os.path.join(root, name, uid + ".npy")
            np.save(paths[name], arr)
        utts[uid] = {"input": [{"feat": paths["mels"], "filetype": "npy", "name": "input1", "shape": list(ys[i].shape)},
                               {"feat": paths["durations_MFA"], "filetype": "npy", "name": "input2", "shape": [len(xs[i]), 1]},
                               {"feat": paths["f0"], "filetype": "npy", "name": "input3", "shape": [len(xs[i]), 1]},
                               {"feat": paths["energy"], "filetype": "npy", "name": "input4", "shape": [len(xs[i]), 1]}],
                     "output": [{"name": "target1", "shape": [len(xs[i]), vocab], "text": "", "token": "", "tokenid": " ".join(str(int(v)) for v in xs[i])}],
                     "utt2spk": "LJ"}
    path = os.path.join(root, "train_data.json")
    with open(path, "w") as f:
        json.dump({"utts": utts}, f)
    return path, (xs, ys, ds, f0, en)


def test_manifest_batching_and_loader_cpu(tmp_path):
    path, (xs, ys, ds, f0, en) = write_dataset(str(tmp_path))
    utts = TR.read_train_manifest(path)
    assert len(utts) == 6 and utts[0]["num_phns"] == 12
    b1 = TR.make_batchset(utts, 4, "shuffle", seed=5)
    b2 = TR.make_batchset(utts, 4, "shuffle", seed=5)
    assert [[u["id"] for u in b] for b in b1] == [[u["id"] for u in b] for b in b2] and [len(b) for b in b1] == [4, 2]
    assert sorted(u["id"] for b in b1 for u in b) == sorted(u["id"] for u in utts)
    assert TR.make_batchset(utts, 4, "shuffle", seed=5, min_batch_size=3) == b1[:1]  # short tail batch dropped for multi-rank runs
    srt = TR.make_batchset(utts, 6, "output")[0]
    assert [u["olen"] for u in srt] == sorted((u["olen"] for u in utts), reverse=True)
    raw = TR.load_batch(b1[0])
    lens = [len(x) for x in raw[0]]
    assert lens == sorted(lens, reverse=True) and raw[2] is None
    from fcl_taco2_amd.converter import CustomConverter

    batch = CustomConverter(1, True, True)([raw])
    assert batch["ys"].shape[0] == 4 and int(batch["olens"].sum()) == int(batch["ds_nonzeros"].sum())
    i = [u["id"] for u in utts].index(b1[0][0]["id"])  # durations survive the int64 .npy round trip as floats
    j = [len(x) for x in raw[0]].index(len(xs[i]))
    assert raw[3][0].dtype == np.float32 and raw[3][j].shape[1] == 1


TEACHER_FLAGS = ["--embed-dim", "32", "--eunits", "32", "--econv-chans", "32", "--dunits", "40", "--prenet-units", "28", "--postnet-chans", "20",
                 "--duration-predictor-chans", "20", "--use-residual", "false", "--use-masking", "true"]
STUDENT_FLAGS = ["--embed-dim", "16", "--eunits", "16", "--econv-chans", "16", "--dunits", "24", "--prenet-units", "20", "--postnet-chans", "12",
                 "--duration-predictor-chans", "20", "--use-residual", "false", "--use-masking", "true"]


class _FakeEngine(object):
    """What train.save_checkpoints / load_* touch of a TrainEngine, on CPU tensors (the checkpoint LAYOUT is host logic)."""

    def __init__(self):
        self.model = torch.nn.Linear(3, 2)
        n = sum(p.numel() for p in self.model.parameters())
        self.mflat, self.vflat = torch.arange(n, dtype=torch.float32), torch.arange(n, dtype=torch.float32) * 2
        self.step_count, self.lr, self.eps, self.betas, self.weight_decay = 4, 1e-3, 1e-6, (0.9, 0.999), 0.0

    def param_offsets(self):
        out, o = {}, 0
        for k, p in self.model.named_parameters():
            out[k] = (o, p.numel(), tuple(p.shape))
            o += p.numel()
        return out


def test_amp_checkpoint_layout_round_trips_with_the_reference_recipe(tmp_path):
    """the reference resumes with `model.load_state_dict(ck['model']); optimizer.load_state_dict(ck['optimizer']);
    amp.load_state_dict(ck['amp'])` (tts.py:418-423) and writes `amp.state_dict()` (tts.py:193-198) = apex's {"loss_scalerN": {"loss_scale",
    "unskipped"}}.  A checkpoint written here carries that layout; one read from it is preserved through save."""
    eng = _FakeEngine()
    TR.save_checkpoints(str(tmp_path), eng, 1, 7, None, float("inf"))
    ck = torch.load(str(tmp_path / "amp_checkpoint_ep1.pt"), weights_only=False)
    assert set(ck) == {"model", "optimizer", "amp"}
    # exactly what apex's amp.load_state_dict iterates over: keys containing 'loss_scaler', each with 'loss_scale' and 'unskipped'
    assert list(ck["amp"]) == ["loss_scaler0"] and set(ck["amp"]["loss_scaler0"]) == {"loss_scale", "unskipped"}
    assert ck["amp"]["loss_scaler0"] == {"loss_scale": 65536.0, "unskipped": 0}  # apex's initial dynamic scaler
    torch.optim.Adam(eng.model.parameters()).load_state_dict(ck["optimizer"])  # torch.optim.Adam accepts the optimizer entry
    snap = torch.load(str(tmp_path / "snapshot.ep.1"), weights_only=False)
    assert snap["iteration"] == 7 and snap["amp"] == ck["amp"]
    # a reference-written scaler state (after some overflow-free steps) is kept through load -> save
    ref_amp = {"loss_scaler0": {"loss_scale": 32768.0, "unskipped": 1234}}
    torch.save({"model": ck["model"], "optimizer": ck["optimizer"], "amp": ref_amp}, str(tmp_path / "ref_amp.pt"))
    eng2 = _FakeEngine()
    got = torch.load(str(tmp_path / "ref_amp.pt"), weights_only=False)
    TR.load_adam_state_dict(eng2, got["optimizer"])
    assert TR.load_amp_state_dict(eng2, got["amp"]) == ref_amp
    TR.save_checkpoints(str(tmp_path), eng2, 2, 9, None, float("inf"))
    assert torch.load(str(tmp_path / "amp_checkpoint_ep2.pt"), weights_only=False)["amp"] == ref_amp
    # files written before round 6 ("amp": None) and junk entries load as "no state"
    assert TR.load_amp_state_dict(eng2, None) is None and TR.load_amp_state_dict(eng2, {"x": 1}) is None
    assert TR.amp_state_dict(eng2)["loss_scaler0"]["loss_scale"] == 65536.0


@pytest.mark.gpu
def test_train_teacher_then_kd_student_then_decode(tmp_path):
    from fcl_taco2_amd import decode as D

    path, _ = write_dataset(str(tmp_path))
    tdir, sdir = str(tmp_path / "exp_teacher"), str(tmp_path / "exp_student")
    common = ["--train-json", path, "--valid-json", path, "--batch-size", "3", "--report-interval-iters", "0", "--seed", "2"]
    log = TR.train(["--outdir", tdir, "--epochs", "3", "--model-module", "fcl_taco2_amd.nets.teacher_training.e2e_tts_tacotron2_sa:Tacotron2_sa"]
                   + common + TEACHER_FLAGS)
    assert len(log) == 3 and log[-1]["iteration"] == 6 and all(np.isfinite(e["main/loss"]) and np.isfinite(e["validation/main/loss"]) for e in log)
    assert log[-1]["main/loss"] < log[0]["main/loss"]  # it learns
    for f in ("model.json", "snapshot.ep.3", "amp_checkpoint_ep3.pt", "model.loss.best", "log"):
        assert os.path.exists(os.path.join(tdir, f)), f
    idim, odim, targs = D.get_model_conf(os.path.join(tdir, "model.json"))
    assert (idim, odim) == (12, 8) and targs.dunits == 40
    ck = torch.load(os.path.join(tdir, "amp_checkpoint_ep3.pt"), weights_only=False)
    assert set(ck) == {"model", "optimizer", "amp"} and ck["optimizer"]["param_groups"][0]["eps"] == 1e-6
    opt_n = len(ck["optimizer"]["state"])
    assert opt_n == len([k for k in ck["model"] if "running" not in k and "num_batches" not in k])
    assert float(ck["optimizer"]["state"][0]["step"]) == 6.0 and float(ck["optimizer"]["state"][0]["exp_avg_sq"].abs().sum()) > 0
    # resume: two more epochs continue the iteration count and the Adam moments
    log2 = TR.train(["--outdir", tdir, "--epochs", "4", "--resume", os.path.join(tdir, "snapshot.ep.3"),
                     "--model-module", "fcl_taco2_amd.nets.teacher_training.e2e_tts_tacotron2_sa:Tacotron2_sa"] + common + TEACHER_FLAGS)
    assert len(log2) == 1 and log2[0]["epoch"] == 4 and log2[0]["iteration"] == 8
    # KD: the student is distilled from the teacher's amp checkpoint (tts_distill.py:370-375), with accum_grad 2
    slog = TR.train(["--outdir", sdir, "--epochs", "2", "--accum-grad", "2", "--share-proj", "true",
                     "--model-module", "fcl_taco2_amd.nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_student:Tacotron2_sa",
                     "--teacher-conf", os.path.join(tdir, "model.json"), "--teacher-model", os.path.join(tdir, "amp_checkpoint_ep3.pt")]
                    + common + STUDENT_FLAGS)
    assert len(slog) == 2 and slog[-1]["iteration"] == 2 and {"main/encoder_loss", "main/decoder_loss", "main/prosody_loss", "main/output_l1_loss"} <= set(slog[0])
    assert np.isfinite(slog[-1]["validation/main/loss"])
    # accum_grad 1: the driver runs the frozen teacher one batch ahead on a second stream (KDPipeline); same artefacts, finite losses
    plog = TR.train(["--outdir", sdir + "_pipe", "--epochs", "2", "--share-proj", "true",
                     "--model-module", "fcl_taco2_amd.nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_student:Tacotron2_sa",
                     "--teacher-conf", os.path.join(tdir, "model.json"), "--teacher-model", os.path.join(tdir, "amp_checkpoint_ep3.pt")]
                    + common + STUDENT_FLAGS)
    assert len(plog) == 2 and plog[-1]["iteration"] == 4 and all(np.isfinite(e["main/loss"]) and np.isfinite(e["validation/main/loss"]) for e in plog)
    assert plog[-1]["main/decoder_loss"] < plog[0]["main/decoder_loss"]  # the distillation terms go down from the first epoch on
    # loader processes (--num-iter-processes): converter + host index maps in a forked worker, same artefacts
    wlog = TR.train(["--outdir", tdir + "_workers", "--epochs", "1", "--num-iter-processes", "1",
                     "--model-module", "fcl_taco2_amd.nets.teacher_training.e2e_tts_tacotron2_sa:Tacotron2_sa"] + common + TEACHER_FLAGS)
    assert len(wlog) == 1 and wlog[0]["iteration"] == 2 and np.isfinite(wlog[0]["main/loss"])
    assert abs(wlog[0]["main/loss"] - log[0]["main/loss"]) < 1e-3 * abs(log[0]["main/loss"])  # same seed, same batches: the same first epoch
    # the decode driver reads what the train driver wrote
    model = D.build_model(os.path.join(sdir, "model.loss.best"), os.path.join(sdir, "model.json"), os.path.join(tdir, "model.json"))
    mel = model.inference(torch.tensor([3, 5, 2, 7]), None, dur=torch.tensor([2, 1, 3, 2]))
    assert tuple(mel.shape) == (8, 8) and bool(torch.isfinite(mel).all())


def _conv_that_dies_in_a_worker(items):
    import signal

    i = items[0]
    if torch.utils.data.get_worker_info() is not None and i >= 2:
        os.kill(os.getpid(), signal.SIGSEGV)  # a loader process dying mid-epoch
    return {"index": i}


def test_a_dead_loader_process_costs_the_epoch_its_prefetching_not_the_run(monkeypatch, caplog):
    """batch_feed: items are a pure function of their index, so when a forked loader dies (seen once on the GPU box: SIGSEGV at birth, forked from a process
    holding a GPU context) the remaining batches are converted in the training process, with a warning."""
    import fcl_taco2_amd.training as TRN

    monkeypatch.setattr(TR, "load_batch", lambda b, cache: b)
    monkeypatch.setattr(TRN, "build_maps_host", lambda b: None)
    with caplog.at_level("WARNING"):
        got = [b["index"] for b in TR.batch_feed(list(range(6)), _conv_that_dies_in_a_worker, None, workers=1)]
    assert got == list(range(6))
    assert any("DataLoader worker" in r.getMessage() for r in caplog.records)
    assert [b["index"] for b in TR.batch_feed(list(range(3)), _conv_that_dies_in_a_worker, None, workers=0)] == [0, 1, 2]  # inline: no worker, nothing dies
