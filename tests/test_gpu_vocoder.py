"""-m gpu: the Parallel WaveGAN generator on the HIP path (fcl_taco2_amd/vocoder.py, csrc/pwg.hip) against the CPU restatement of the published
architecture (oracle/pwg_oracle.py; no vocoder source or vectors exist in the reference: parity unpinned).  Closed-form weights, explicit noise;
full v1 configuration and a small irregular one (aux channels not a multiple of 32, scales 2 x 3, two stacks), batches of ragged utterances incl.
a one-frame utterance (every dilation then reaches past both edges), per-layer taps, weight-normed checkpoints."""
import numpy as np
import pytest
import torch

from helpers import max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def voc():
    assert torch.cuda.is_available()
    import fcl_taco2_amd  # noqa: F401
    from fcl_taco2_amd import _lib, ops, vocoder

    _lib.load()
    if not ops.planes_enabled():
        pytest.skip("FCL_PRECISION=0 / FCL_PLANES=0: the vocoder needs the pre-split operand path")
    return vocoder


def weights(voc, cfg=None):
    from fcl_taco2_amd import synthetic as SYN

    return {k: SYN.closed_form_tensor("pwg." + k, tuple(s)) for k, s in voc.param_spec(cfg).items()}


def run_both(voc, cfg, lens, seed):
    from oracle import pwg_oracle as O

    rng = np.random.RandomState(seed)
    sd = weights(voc, cfg)
    full = dict(voc.CONFIG, **(cfg or {}))
    hop = int(np.prod(full["upsample_scales"]))
    mels = [rng.standard_normal((n, full["aux_channels"])).astype(np.float32) for n in lens]
    noise = [rng.standard_normal(n * hop).astype(np.float32) for n in lens]
    gen = voc.ParallelWaveGANGenerator(voc.PWGPlan(sd, DEV, cfg))
    got, aux = gen.synthesize(mels, noise=noise, return_intermediates=True)
    torch.cuda.synchronize()
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    want = [O.inference(tsd, m, z, cfg) for m, z in zip(mels, noise)]
    return got, want, aux, (tsd, mels, noise, hop)


def test_full_v1_generator_matches_the_oracle(voc):
    got, want, aux, _ = run_both(voc, None, [5, 1, 3], 0)
    for g, w in zip(got, want):
        assert g.shape == w.shape
        err = max_abs(g.cpu(), w) / float(w.abs().max())
        print('rel err', err)
        assert err < 1e-3  # relative to the waveform's peak (the closed-form generator's output is small)


def test_small_irregular_generator_and_per_layer_taps(voc):
    import torch.nn.functional as F

    from oracle import pwg_oracle as O

    cfg = dict(layers=4, stacks=2, residual_channels=32, gate_channels=64, skip_channels=32, aux_channels=20, upsample_scales=(2, 3))
    got, want, aux, (tsd, mels, noise, hop) = run_both(voc, cfg, [7, 2, 1, 9], 1)
    for g, w in zip(got, want):
        assert max_abs(g.cpu(), w) < 1e-3 * float(w.abs().max())
    # the residual stream after every layer, utterance 0
    full = dict(O.CONFIG, **cfg)
    c = torch.from_numpy(mels[0]).t().unsqueeze(0)
    c = F.pad(c, (full["aux_context_window"],) * 2, mode="replicate")
    with torch.no_grad():
        _, taps, skips = O.generator_forward(tsd, torch.from_numpy(noise[0]).reshape(1, 1, -1), O.upsample(tsd, c, cfg), cfg, return_taps=True)
    n0 = mels[0].shape[0] * hop
    for l, t in enumerate(taps):
        assert max_abs(aux["taps"][l][:n0].cpu(), t[0].t()) < 5e-5, l


def test_weight_normed_checkpoint_and_published_call(voc):
    """A parallel_wavegan checkpoint ({"model": {"generator": ...}} with weight_g / weight_v) gives the same waveform; inference(c, x) -> [T, 1];
    device noise is reproducible per seed and differs between seeds."""
    rng = np.random.RandomState(2)
    cfg = dict(layers=2, stacks=1, residual_channels=32, gate_channels=64, skip_channels=32, aux_channels=16, upsample_scales=(4,))
    sd = weights(voc, cfg)
    wn = {}
    for k, v in sd.items():
        if k.endswith("weight"):
            g = rng.uniform(0.5, 2.0, size=[v.shape[0]] + [1] * (v.ndim - 1)).astype(np.float32)
            norm = np.sqrt((v.reshape(v.shape[0], -1).astype(np.float64) ** 2).sum(1)).reshape(g.shape)
            wn[k + "_g"], wn[k + "_v"] = (norm * 1.0).astype(np.float32), (v * g).astype(np.float32)  # g * v / |v| with |g v| = g |v|: folds back to v
        else:
            wn[k] = v
    mel, z = rng.standard_normal((4, 16)).astype(np.float32), rng.standard_normal(16).astype(np.float32)
    a = voc.ParallelWaveGANGenerator(voc.PWGPlan(sd, DEV, cfg)).inference(mel, z)
    b = voc.ParallelWaveGANGenerator(voc.PWGPlan({"model": {"generator": wn}}, DEV, cfg)).inference(mel, z)
    assert a.shape == (16, 1) and max_abs(a.cpu(), b.cpu()) < 1e-5
    gen = voc.ParallelWaveGANGenerator(voc.PWGPlan(sd, DEV, cfg))
    w0, w0b, w1 = gen.synthesize([mel], seed=7)[0], gen.synthesize([mel], seed=7)[0], gen.synthesize([mel], seed=8)[0]
    assert torch.equal(w0, w0b) and not torch.equal(w0, w1)


def test_fused_and_unfused_blocks_agree(voc, monkeypatch):
    """The one-launch residual block (default for the v1 geometry) against the four-launch form of the same algebra."""
    rng = np.random.RandomState(4)
    sd = weights(voc)
    mels = [rng.standard_normal((n, 80)).astype(np.float32) for n in (6, 2)]
    noise = [rng.standard_normal(n * 256).astype(np.float32) for n in (6, 2)]
    gen = voc.ParallelWaveGANGenerator(voc.PWGPlan(sd, DEV))
    a, ia = gen.synthesize(mels, noise=noise, return_intermediates=True)
    monkeypatch.setenv("FCL_PWG_FUSED", "0")
    b, ib = gen.synthesize(mels, noise=noise, return_intermediates=True)
    torch.cuda.synchronize()
    for l in (0, 9, 29):
        assert max_abs(ia["taps"][l].cpu(), ib["taps"][l].cpu()) < 1e-4, l
    assert max_abs(ia["skips"].cpu(), ib["skips"].cpu()) < 1e-4
    for x, y in zip(a, b):
        assert max_abs(x.cpu(), y.cpu()) < 1e-3 * float(y.abs().max())


def test_frame_rate_auxiliary_term_across_window_boundaries(voc, monkeypatch):
    """The default one-launch block evaluates conv1x1_aux(upsample(c)) at frame rate (coefficient lines x a 32-frame window of W_aux c_in, two window
    alignments).  70 frames in ragged utterances put tiles on every frame position mod 32, incl. the four that switch to the shifted windows and
    utterance edges right at a window boundary; against the oracle and against the upsampled-feature form of the same block."""
    lens = [29, 3, 1, 33, 2, 2]
    got, want, aux, (tsd, mels, noise, hop) = run_both(voc, None, lens, 5)
    for g, w in zip(got, want):
        assert max_abs(g.cpu(), w) < 1e-3 * float(w.abs().max())
    monkeypatch.setenv("FCL_PWG_AUX_FRAME_RATE", "0")
    gen = voc.ParallelWaveGANGenerator(voc.PWGPlan(weights(voc), DEV))
    ref, ir = gen.synthesize(mels, noise=noise, return_intermediates=True)
    torch.cuda.synchronize()
    for l in (0, 1, 15, 29):
        assert max_abs(aux["taps"][l].cpu(), ir["taps"][l].cpu()) < 1e-4, l
    assert max_abs(aux["skips"].cpu(), ir["skips"].cpu()) < 1e-4
    for x, y in zip(got, ref):
        assert max_abs(x.cpu(), y.cpu()) < 1e-3 * float(y.abs().max())


def test_decode_driver_writes_the_generators_waveforms(voc, tmp_path):
    """`python -m fcl_taco2_amd.vocoder_decode --checkpoint --feats-scp --outdir` (the reference's `parallel-wavegan-decode` call,
    inference_student.sh:20-23): feats.scp in, <utt>_gen.wav out; the PCM equals the generator's output for the same batches and seeds."""
    import wave

    from fcl_taco2_amd import vocoder_decode as VD
    from fcl_taco2_amd.kaldi_io import ArkScpWriter

    rng = np.random.RandomState(6)
    sd = weights(voc)
    torch.save({"model": {"generator": {k: torch.from_numpy(v) for k, v in sd.items()}}}, tmp_path / "PWG.pkl")
    feats = {"utt_%d" % i: rng.standard_normal((n, 80)).astype(np.float32) for i, n in enumerate([4, 9, 2, 6, 1])}
    with ArkScpWriter(str(tmp_path / "feats")) as w:
        for k, m in feats.items():
            w[k] = m
    samples, _ = VD.main(["--checkpoint", str(tmp_path / "PWG.pkl"), "--feats-scp", str(tmp_path / "feats.scp"), "--outdir", str(tmp_path / "wav"),
                          "--batch-frames", "10", "--seed", "3", "--verbose", "0"])
    assert samples == 22 * 256
    gen = voc.ParallelWaveGANGenerator(voc.PWGPlan(sd, DEV))
    items = sorted(feats.items())
    peak = 0.0
    for bi, idx in enumerate(VD.make_batches([m.shape[0] for _, m in items], 10)):
        want = gen.synthesize([items[i][1] for i in idx], seed=3 + bi)
        for i, y in zip(idx, want):
            with wave.open(str(tmp_path / "wav" / (items[i][0] + "_gen.wav"))) as f:
                assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 22050, items[i][1].shape[0] * 256)
                pcm = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").astype(np.float64)
            ref = np.clip(np.rint(y.cpu().numpy().astype(np.float64) * 32767.0), -32768, 32767)
            assert np.array_equal(pcm, ref)
            peak = max(peak, float(np.abs(ref).max()))
    assert peak > 0


def test_full_size_batch_equals_single_utterances(voc):
    """BASELINE configs[4] size (64 utterances, ~800 frames each = 13 M samples, frame windows up to index ~1600): utterances are independent, so
    any utterance of the batch equals the same utterance synthesised alone with the same noise -- edges, tile / window bookkeeping at full scale."""
    rng = np.random.RandomState(7)
    lens = [int(n) for n in rng.randint(700, 900, size=64)]
    mels = [rng.standard_normal((n, 80)).astype(np.float32) for n in lens]
    noise = [rng.standard_normal(n * 256).astype(np.float32) for n in lens]
    gen = voc.ParallelWaveGANGenerator(voc.PWGPlan(weights(voc), DEV))
    full = gen.synthesize(mels, noise=noise)
    torch.cuda.synchronize()
    assert sum(w.numel() for w in full) == sum(lens) * 256 and all(bool(torch.isfinite(w).all()) for w in full)
    for i in (0, 17, 63):
        alone = gen.synthesize([mels[i]], noise=[noise[i]])[0]
        peak = float(alone.abs().max())
        assert peak > 0 and max_abs(full[i].cpu(), alone.cpu()) < 2e-4 * peak, i


def test_device_noise_is_standard_normal(voc):
    from fcl_taco2_amd import _lib, ops

    z = torch.empty(1 << 20, device=DEV)
    _lib.check(_lib.load().fcl_pwg_noise(z.data_ptr(), z.numel(), 123, ops._stream()))
    torch.cuda.synchronize()
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    assert abs(float((z.abs() < 1.0).float().mean()) - 0.6827) < 3e-3 and float(z.abs().max()) < 6.5
