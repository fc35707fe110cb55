"""H15: the vectorised CustomConverter reproduces the REAL reference converter bit for bit (G4)."""
import time

import numpy as np
import torch

import fcl_taco2_amd  # noqa: F401
from fcl_taco2_amd import synthetic as SYN
from fcl_taco2_amd.converter import CustomConverter


def _raw(g, n, pre="in_"):
    return ([g[pre + "xs%d" % i] for i in range(n)], [g[pre + "ys%d" % i] for i in range(n)], None,
            [g[pre + "ds%d" % i] for i in range(n)], [g[pre + "f0%d" % i] for i in range(n)], [g[pre + "en%d" % i] for i in range(n)])


def test_converter_bit_exact_vs_reference(golden):
    g = golden("g4_integer")
    out = CustomConverter(1, use_fe_condition=True, append_position=True)([_raw(g, 4)])
    keys = ("xs", "ilens", "ys", "olens", "extras", "new_ys", "non_zero_lens_mask", "ds_nonzeros", "output_masks", "position", "f0", "energy")
    assert sorted(out) == sorted(keys)
    for k in keys:
        ref = g["out_" + k]
        assert out[k].numpy().dtype == ref.dtype and np.array_equal(out[k].numpy(), ref), k
    assert (g["out_non_zero_lens_mask"].sum(1) < g["out_ilens"]).any()  # zero-duration phonemes were present


def test_converter_second_case_index_maps(golden):
    g = golden("g4_integer")
    xs, ys, ds, f0, en = SYN.training_batch(80, 80, batch=6, seed=21)  # the generator call gen_golden.py made
    assert all(np.array_equal(ds[i], g["in2_ds%d" % i]) for i in range(6))
    out = CustomConverter(1, True, True)([(xs, ys, None, ds, f0, en)])
    for k in ("non_zero_lens_mask", "ds_nonzeros", "output_masks", "position", "ilens", "olens"):
        assert np.array_equal(out[k].numpy(), g["out2_" + k]), k


def test_converter_reduction_factor_2_bit_exact(golden):
    """tts.py:250-258 with reduction_factor 2 (segments, ds_nonzeros and the position table in frames = 2 x the durations): the vectorised
    converter against the REAL class's outputs stored in G21."""
    g = golden("g21_teacher_r2")
    raw = ([g["in_xs%d" % i] for i in range(4)], [g["in_ys%d" % i] for i in range(4)], None, [g["in_ds%d" % i] for i in range(4)],
           [g["in_f0%d" % i] for i in range(4)], [g["in_en%d" % i] for i in range(4)])
    out = CustomConverter(2, True, True)([raw])
    for k in ("xs", "ilens", "ys", "olens", "extras", "new_ys", "non_zero_lens_mask", "ds_nonzeros", "output_masks", "position", "f0", "energy"):
        ref = g["out_" + k]
        assert out[k].numpy().dtype == ref.dtype and np.array_equal(out[k].numpy(), ref), k


def test_converter_is_fast_at_bench_scale():
    xs, ds = SYN.batch_c2(batch=32)
    rng = np.random.RandomState(0)
    ys = [rng.randn(int(d.sum()), 80).astype(np.float32) for d in ds]
    f0 = [rng.randn(len(x), 1).astype(np.float32) for x in xs]
    conv = CustomConverter(1, True, True)
    batch = [(xs, ys, None, [d.astype(np.float32).reshape(-1, 1) for d in ds], f0, f0)]
    conv(batch)
    t0 = time.perf_counter()
    out = conv(batch)
    dt = time.perf_counter() - t0
    assert out["new_ys"].shape[0] == sum(len(x) for x in xs)
    assert dt < 0.25  # the reference's loops take ~1.2 s for this batch
