"""-m gpu: the persistent row-tile form of the free-running decoder loop (csrc/decoder_tile.hip: FCL_DEC_TILE=1, opt-in) against the per-step
launches and the oracle.  The tunable is read once per process, so the tile path runs in a child process (started with subprocess; this process
keeps its own GPU context) that writes its mels to a file; this process computes the default path's mels on the same seeded inputs.
What must hold: same frames, <= 5e-6 between the two paths with dropout off AND in RNG mode (both draw the same counter-hash bits from (seed, step,
row, column)), <= 1e-3 against the oracle's per-utterance inference() (north_star), device-built row maps (capacity graphs) included, and the kernel
must actually have run in the child (its name in the library's launch profile)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import max_abs, np_state_dict, torch_state_dict
from fcl_taco2_amd import hparams as HP, synthetic as SYN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import np_state_dict
from fcl_taco2_amd import _lib, engine, hparams as HP, ops, synthetic as SYN
from fcl_taco2_amd.plan import SynthesisPlan
res = {}
for tag, drop in (("nodrop", 0.0), ("rng", 0.5)):
    hp = HP.student_hparams(dropout_rate=drop)
    plan = SynthesisPlan(np_state_dict(hp), hp, "cuda:0")
    assert plan.decoder.struct.stream, "the plan did not pack the weight stream"
    for b, (lo, hi) in ((32, (60, 100)), (5, (3, 40))):
        xs, ds = SYN.batch_c2(hp.idim, batch=b, t_lo=lo, t_hi=hi, seed=1234)
        _lib.prof_enable(True)
        mels = engine.synthesize(plan, xs, ds, seed=7)
        torch.cuda.synchronize()
        prof = _lib.prof_collect(); _lib.prof_enable(False)
        assert any(k.startswith("decoder_tile_kernel") for k in prof) and not any(k.startswith("plstm") or k.startswith("lstm_small") for k in prof), sorted(prof)
        res["%%s_b%%d" %% (tag, b)] = torch.cat(mels).cpu().numpy()
    if drop == 0.0:  # capacity graph, device-built maps, two different batches through one captured graph
        B, T_CAP = 32, 100
        batches = [SYN.batch_c2(hp.idim, batch=B, t_hi=T_CAP, seed=1234 + 1000 * j) for j in range(2)]
        maps = [engine.build_row_maps([len(x) for x in b[0]], b[1], T_CAP) for b in batches]
        r = engine.BatchRunner(plan, B, T_CAP, engine.Caps.for_batches(maps), forced=True, seed=77)
        for j in range(2):
            r.load(*batches[j]); mel = r.replay(); fr = r.frames()
            res["runner_%%d" %% j] = mel[: sum(fr)].cpu().numpy()
np.savez(sys.argv[1], **res)
''' % (ROOT, os.path.join(ROOT, "tests"))


def test_tile_decoder_equals_the_per_step_path_and_the_oracle(tmp_path):
    from fcl_taco2_amd import engine
    from fcl_taco2_amd.plan import SynthesisPlan
    from oracle import fcl_oracle as O

    if os.environ.get("FCL_PRECISION", "1") == "0":
        pytest.skip("the tile kernel runs on pre-split operands")
    out = str(tmp_path / "tile.npz")
    env = dict(os.environ, FCL_DEC_TILE="1", FCL_DEC_TILE_MIN_ROWS="1")
    r = subprocess.run([sys.executable, "-c", CHILD, out], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    tile = np.load(out)
    for tag, drop in (("nodrop", 0.0), ("rng", 0.5)):
        hp = HP.student_hparams(dropout_rate=drop)
        plan = SynthesisPlan(np_state_dict(hp), hp, DEV)
        for b, (lo, hi) in ((32, (60, 100)), (5, (3, 40))):
            xs, ds = SYN.batch_c2(hp.idim, batch=b, t_lo=lo, t_hi=hi, seed=1234)
            mels = engine.synthesize(plan, xs, ds, seed=7)
            ref = torch.cat(mels).cpu().numpy()
            got = tile["%s_b%d" % (tag, b)]
            assert got.shape == ref.shape and float(np.abs(got - ref).max()) < 5e-6, (tag, b, float(np.abs(got - ref).max()))
            if drop == 0.0:
                sd = torch_state_dict(hp)
                starts = np.concatenate([[0], np.cumsum([m.shape[0] for m in mels])])
                for u in (0, b // 2, b - 1):
                    with torch.no_grad():
                        want = O.inference(sd, hp, torch.from_numpy(xs[u]), dur=torch.from_numpy(ds[u]))["after"].numpy()
                    assert float(np.abs(got[starts[u] : starts[u + 1]] - want).max()) < 1e-3, (b, u)
        if drop == 0.0:
            for j in range(2):
                bx = SYN.batch_c2(hp.idim, batch=32, t_hi=100, seed=1234 + 1000 * j)
                ref = torch.cat(engine.synthesize(plan, *bx)).cpu().numpy()
                got = tile["runner_%d" % j]
                assert got.shape == ref.shape and float(np.abs(got - ref).max()) < 5e-6, j


def test_tail_hand_over_to_the_tile_kernel_in_a_capacity_graph():
    """fcl_decoder_io_t.tail_from / engine.Caps(tail_from=...): the per-step loop runs steps 0 .. tail_from - 1, then the rows still live continue in ONE
    launch of the row-tile kernel from the loop's own fp32 states.  Captured capacity graphs (device-built maps, slack steps beyond the longest duration)
    with the hand-over (a) deep inside the real steps, (b) at the last real step, (c) inside the slack (nothing left to do: the launch exits), each
    replayed over two different batches, against the eager per-step pass: same frames, <= 5e-6; dropout off and RNG mode (same counter-hash bits)."""
    from fcl_taco2_amd import _lib, engine, ops
    from fcl_taco2_amd.plan import SynthesisPlan

    if not ops.planes_enabled():
        pytest.skip("the tile kernel runs on pre-split operands")
    B, T_CAP = 32, 100
    for drop in (0.0, 0.5):
        hp = HP.student_hparams(dropout_rate=drop)
        plan = SynthesisPlan(np_state_dict(hp), hp, DEV)
        batches = [SYN.batch_c2(hp.idim, batch=B, t_hi=T_CAP, seed=4321 + 1000 * j) for j in range(2)]
        maps = [engine.build_row_maps([len(x) for x in b[0]], b[1], T_CAP) for b in batches]
        lmax = max(m.lmax for m in maps)
        base = engine.Caps.for_batches(maps, slack_steps=6)
        for tail in (5, min(m.lmax for m in maps) - 1, lmax + 2):
            caps = engine.Caps(base.lmax, base.frames, base.bounds, tail_from=tail)
            assert caps.tail_from == tail
            r = engine.BatchRunner(plan, B, T_CAP, caps, forced=True, seed=77)
            for j in range(2):
                r.load(*batches[j])
                mel = r.replay()
                fr = r.frames()
                # the runner's seed stream: replay k of a runner draws with (seed, k): the eager pass below uses the same pair through seed_dev
                got = mel[: sum(fr)].clone()
                ref = engine.BatchRunner(plan, B, T_CAP, engine.Caps(base.lmax, base.frames, base.bounds), forced=True, seed=77)
                for jj in range(j + 1):  # same number of replays -> same seed word
                    ref.load(*batches[jj])
                    mel_ref = ref.replay()
                    fr_ref = ref.frames()
                assert fr == fr_ref and max_abs(got, mel_ref[: sum(fr_ref)]) < 5e-6, (drop, tail, j)
    # the kernel really ran: the same pass launched eagerly (device-built maps from uploaded forced durations), profiled by the library's hooks
    hp = HP.student_hparams(dropout_rate=0.0)
    plan = SynthesisPlan(np_state_dict(hp), hp, DEV)
    xs, ds = SYN.batch_c2(hp.idim, batch=B, t_hi=T_CAP, seed=99)
    want = torch.cat(engine.synthesize(plan, xs, ds))
    m = engine.build_row_maps([len(x) for x in xs], ds, T_CAP)
    caps = engine.Caps(m.lmax + 4, (m.n_frames + 255) // 256 * 256, np.concatenate([m.live_rows, np.full(4, m.live_rows[-1], np.int32)]), tail_from=7)
    prep = engine.prepare(plan, xs, ds, device_maps=True)
    _lib.prof_enable(True)
    mel, frames = engine.run(plan, prep, ops.DROP_NONE, caps=caps)
    torch.cuda.synchronize()
    prof = _lib.prof_collect()
    _lib.prof_enable(False)
    assert any(k.startswith("decoder_tile_kernel") for k in prof), sorted(prof)
    total = sum(frames.resolve())
    assert total == want.shape[0] and max_abs(mel[:total], want) < 5e-6
