"""-m gpu: parity of exactly what bench.py times, and of the BASELINE configs the earlier rounds only covered in pieces (VERDICT r3 "next" #1).

(a) the default bench line's configuration (bench.py `--feed fresh`): four engine.BatchRunner graphs on four HIP streams, batch 32, t_cap 100,
    capacities = engine.Caps.for_batches over four SYN.batch_c2 batches, >= 12 round-robin passes back to back -- every utterance against the
    eager engine.synthesize of its batch (<= 2e-5) and sampled utterances against the oracle's per-utterance inference() (<= 1e-3, north_star);
(b) BASELINE configs[4]: FCL-taco2-S synthesis at batch 64 (forced durations, predicted pitch / energy: the sizes at which the two-stage 64 KB
    ring and the wide-tile thresholds switch) against the oracle, then mel -> ParallelWaveGANGenerator on that batch equal to per-utterance runs;
(c) the exact-fp32 mode behind bench.py's `value_fp32_exact` (FCL_PRECISION=0 is read once per process): the reference-golden tests G2 / G2T / G3
    and the decoder-loop-vs-oracle cases re-run in a child process under that mode.
Dropout is off (dropout_rate 0) wherever values are compared: the production RNG mode is covered statistically elsewhere."""
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


def _plan(hp):
    from fcl_taco2_amd.plan import SynthesisPlan

    return SynthesisPlan(np_state_dict(hp), hp, DEV)


def test_the_bench_timed_configuration_equals_eager_and_the_oracle():
    """bench.py:597-639 with dropout off: same batches (seeds 1234 + 1000 j), same capacities, same runner seeds, four streams, no host
    synchronisation between the passes (the host runs ahead exactly as in the timed region; each pass's mel buffer is cloned on its own stream)."""
    from fcl_taco2_amd import engine, ops
    from oracle import fcl_oracle as O

    hp = HP.student_hparams(dropout_rate=0.0)
    plan = _plan(hp)
    B, T_CAP, STREAMS, PASSES = 32, 100, 4, 14
    batches = [SYN.batch_c2(hp.idim, batch=B, t_hi=T_CAP, seed=1234 + 1000 * j) for j in range(4)]
    host_maps = [engine.build_row_maps([len(x) for x in b[0]], b[1], T_CAP) for b in batches]
    caps = engine.Caps.for_batches(host_maps)
    assert caps.lmax == max(m.lmax for m in host_maps) and caps.frames % 256 == 0 and caps.frames >= max(m.n_frames for m in host_maps)
    runners = [engine.BatchRunner(plan, B, T_CAP, caps, forced=True, seed=77 + 1000 * j) for j in range(STREAMS)]
    got = []
    for i in range(PASSES):  # 14 passes over 4 runners x 4 batches: every (runner, batch) residue pair of i % 4 twice, and batches change under each runner
        r, j = runners[i % STREAMS], (i + i // 4) % len(batches)
        r.load(*batches[j])
        mel = r.replay()
        with torch.cuda.stream(r.stream):
            got.append((j, mel[: host_maps[j].n_frames].clone()))
    for r in runners:
        fr = r.frames()  # synchronises; raises on a violated capacity
        assert sum(fr) in [m.n_frames for m in host_maps]
    torch.cuda.synchronize()
    eager = [torch.cat(engine.synthesize(plan, *b)) for b in batches]
    for j, mel in got:
        assert mel.shape == eager[j].shape and max_abs(mel, eager[j]) < 2e-5, j
    # the oracle, per utterance as the reference's inference() runs: 3 utterances of every batch (first / longest, a middle one, last / shortest)
    sd = torch_state_dict(hp)
    last = {j: mel for j, mel in got}
    for j, (xs, ds) in enumerate(batches):
        starts = np.concatenate([[0], np.cumsum(host_maps[j].utt_frames)])
        for u in (0, 13, B - 1):
            with torch.no_grad():
                ref = O.inference(sd, hp, torch.from_numpy(xs[u]), dur=torch.from_numpy(ds[u]))["after"]
            mine = last[j][starts[u] : starts[u + 1]]
            assert mine.shape == ref.shape and max_abs(mine.cpu(), ref) < 1e-3, (j, u)
    assert ops.planes_enabled() == (os.environ.get("FCL_PRECISION", "1") != "0" and os.environ.get("FCL_PLANES", "1") != "0")


def test_configs4_student_batch64_synthesis_and_the_vocoder_chain():
    """BASELINE configs[4] at its size: 64 utterances of 60-100 phonemes (~51 k frames, ~5 k decoder rows) through FCL-taco2-S with forced
    durations and PREDICTED pitch / energy, eager and as a captured graph; sampled utterances against the oracle; the kernel forms this size
    selects must be on the tested path; then the batch's mels through the Parallel WaveGAN generator: every sampled utterance equals the same
    utterance vocoded alone with the same noise (the vocoder itself is `parity unpinned`: oracle/pwg_oracle.py is its only reference)."""
    from fcl_taco2_amd import _lib, engine, ops, vocoder as V
    from oracle import fcl_oracle as O

    hp = HP.student_hparams(dropout_rate=0.0)
    plan = _plan(hp)
    B = 64
    xs, ds = SYN.batch_c2(hp.idim, batch=B, seed=1234)
    _lib.prof_enable(True)
    mels = engine.synthesize(plan, xs, ds)
    torch.cuda.synchronize()
    prof = _lib.prof_collect()
    _lib.prof_enable(False)
    assert [m.shape[0] for m in mels] == [int(d.sum()) for d in ds] and all(bool(torch.isfinite(m).all()) for m in mels)
    if ops.planes_enabled():  # the pre-split-operand LSTM step carried the big steps of this batch (5 k rows: the >= 300-tile two-stage form switches on here)
        assert any(k.startswith("plstm_kernel") for k in prof), sorted(prof)
        assert any(k.startswith("pconv_kernel") or k.startswith("pgemm_kernel") for k in prof), sorted(prof)
    sd = torch_state_dict(hp)
    for u in (0, 21, 40, B - 1):
        with torch.no_grad():
            ref = O.inference(sd, hp, torch.from_numpy(xs[u]), dur=torch.from_numpy(ds[u]))["after"]
        assert mels[u].shape == ref.shape and max_abs(mels[u].cpu(), ref) < 1e-3, u
    # the same batch as a captured graph (what bench.py --workload tts_e2e replays)
    runner = engine.GraphRunner(plan, engine.prepare(plan, xs, ds))
    packed = runner.replay()
    torch.cuda.synchronize()
    assert max_abs(packed, torch.cat(mels)) < 2e-5 and list(runner.utt_frames) == [m.shape[0] for m in mels]
    if not ops.planes_enabled():
        return  # FCL_PRECISION=0: the vocoder exists on the pre-split operand path only
    # mel -> waveform on the whole batch (13 M samples) vs the same utterances alone
    vsd = {k: SYN.closed_form_tensor("pwg." + k, tuple(s)) for k, s in V.param_spec().items()}
    gen = V.ParallelWaveGANGenerator(V.PWGPlan(vsd, DEV))
    rng = np.random.RandomState(3)
    noise = [rng.standard_normal(m.shape[0] * gen.plan.hop).astype(np.float32) for m in mels]
    mel_np = [m.cpu().numpy() for m in mels]
    full = gen.synthesize(mel_np, noise=noise)
    torch.cuda.synchronize()
    assert sum(w.numel() for w in full) == sum(m.shape[0] for m in mels) * gen.plan.hop
    for u in (0, 33, B - 1):
        alone = gen.synthesize([mel_np[u]], noise=[noise[u]])[0]
        peak = float(alone.abs().max())
        assert peak > 0 and bool(torch.isfinite(full[u]).all()) and max_abs(full[u].cpu(), alone.cpu()) < 2e-4 * peak, u


@pytest.mark.skipif(os.environ.get("FCL_PRECISION", "1") == "0", reason="already the exact-fp32 process")
def test_exact_fp32_mode_in_a_child_process_meets_the_reference_goldens():
    """FCL_PRECISION=0 (every contraction on v_mfma_f32_16x16x4_f32: the mode of bench.py's value_fp32_exact) is chosen once per process, so the
    driver's single `pytest -m gpu` run never sees it: this test starts a child pytest under that mode on the reference-golden mel tests (G2, G2T,
    G3), the decoder loop against the oracle and this file's bench-configuration test.  The child is a separate process started with
    subprocess (this process keeps running and holds its own GPU context)."""
    env = dict(os.environ, FCL_PRECISION="0")
    sel = ("test_g2_student_c1_mel_vs_reference or test_g2t_teacher_c1_mel_vs_reference or test_g3_injected_dropout_vs_reference "
           "or test_decoder_loop_vs_oracle or test_the_bench_timed_configuration_equals_eager_and_the_oracle "
           # (round 5) the training-step tests whose default-arithmetic form needs a loose gradient tolerance (ill-conditioned closed-form nets):
           # here every tensor is held at 5e-4 against the reference and the oracle
           "or test_structure_options_vs_reference_g18_g19_g20 or test_no_batch_norm_vs_reference_g15 or test_kd_classes_with_structure_options_vs_reference_g22")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_bench_config.py"),
           os.path.join(ROOT, "tests", "test_gpu_training.py"), "-m", "gpu", "-q", "-x", "-k", sel, "-p", "no:cacheprovider"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = "\n".join((r.stdout + "\n" + r.stderr).strip().splitlines()[-25:])
    assert r.returncode == 0, "the FCL_PRECISION=0 child failed:\n" + tail
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail
    n_passed = int(r.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    assert n_passed >= 17, tail  # G2 + G2T + G3 + 8 decoder-loop cases + the bench configuration + 5 structure variants (+ G15)
