"""N>1 path on CPU: world_size-2 gloo processes exercise the utterance sharding and the bench reductions
(the data path itself has no collective)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fcl_taco2_amd  # noqa: F401
from fcl_taco2_amd import sharding, synthetic as SYN


def test_shard_utterances_is_a_balanced_partition():
    xs, ds = SYN.batch_c2(batch=32)
    frames = [int(d.sum()) for d in ds]
    for world in (1, 2, 4, 8):
        parts = sharding.shard_utterances(frames, world)
        assert sorted(i for p in parts for i in p) == list(range(32))  # exact partition, nothing dropped or duplicated
        loads = [sum(frames[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(frames)  # greedy LPT bound
    assert sharding.shard_utterances([5, 5], 4) == [[0], [1], [], []]  # more ranks than utterances: empty shards are legal


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xs, ds = SYN.batch_c2(batch=8, seed=5)
    frames = [int(d.sum()) for d in ds]
    mine = sharding.shard_utterances(frames, world)[rank]
    local_frames = sum(frames[i] for i in mine)
    t, f = sharding.aggregate_throughput(1.0 + rank, local_frames, dist)
    counts = sharding.gather_frame_counts([frames[i] for i in mine], dist)
    recs = sharding.verify_world(dist, world)  # bench.py --gpus N: every rank proves the communicator holds N ranks
    wrong = False
    try:
        sharding.verify_world(dist, world + 1)  # a job that believes it is larger than its communicator must be loud on every rank
    except RuntimeError:
        wrong = True
    q.put((rank, mine, t, f, counts, recs, wrong))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduction():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xs, ds = SYN.batch_c2(batch=8, seed=5)
    total = sum(int(d.sum()) for d in ds)
    (r0, m0, t0, f0, c0, v0, w0), (r1, m1, t1, f1, c1, v1, w1) = res
    assert w0 and w1 and v0 == v1 and [r["rank"] for r in v0] == [0, 1]
    assert all(r["world_size_seen"] == 2 and r["allreduce_of_ones"] == 2.0 for r in v0)
    assert sharding.verify_world(None, 1)[0]["world_size_seen"] == 1  # single process: no group needed
    assert sorted(m0 + m1) == list(range(8)) and not set(m0) & set(m1)
    assert t0 == t1 == 2.0  # MAX over ranks
    assert f0 == f1 == float(total)  # SUM over ranks
    assert c0 == c1 and sum(sum(c) for c in c0) == total


# ---- training: the gradient exchange step (one all-reduce per iteration, bucketed, overlapped with backward) ----------------
def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fcl_taco2_amd import hparams as HP
    from fcl_taco2_amd.training import GradBuckets, flat_layout

    spec = HP.param_spec(HP.student_hparams(idim=12, odim=8, embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12,
                                            duration_predictor_chans=20),
                         HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20,
                                            duration_predictor_chans=20), True)
    names, offs, bounds = flat_layout([(k, int(np.prod(v))) for k, v in spec.items() if "running" not in k and "num_batches" not in k])
    flat = torch.from_numpy(np.random.RandomState(100 + rank).randn(int(offs[-1])).astype(np.float32))
    mine = flat.clone()
    b = GradBuckets(flat, bounds)
    for i in range(len(bounds) - 1):  # the engine launches bucket i when backward has finished group i
        b.launch(i)
    b.finish()
    # gradient accumulation (accum_grad = 2): micro-batch 1 only accumulates, the buckets are launched by the LAST micro-batch, once each
    rs = np.random.RandomState(200 + rank)
    m1, m2 = (torch.from_numpy(rs.randn(int(offs[-1])).astype(np.float32)) for _ in range(2))
    acc = torch.zeros(int(offs[-1]))
    b2 = GradBuckets(acc, bounds)
    acc += m1  # micro-batch 1: reduce=False, nothing launched
    acc += m2  # micro-batch 2 ...
    for i in range(len(bounds) - 1):
        b2.launch(i)  # ... launches as its backward completes each group
    twice = False
    try:
        b2.launch(0)  # a second launch before finish() is the race ADVICE r1 describes: it must be loud
    except RuntimeError:
        twice = True
    b2.finish()
    b2.finish()  # nothing launched since: must not rescale again
    # round 6: the self-deciding placement -- warm-up, alternating trial, one decision shared by the ranks, correct means throughout
    g3 = torch.zeros(int(offs[-1]))
    b3 = GradBuckets(g3, bounds)
    n_trial = GradBuckets.TRIAL_TOTAL
    means_ok = True
    for u in range(n_trial + 2):
        g3.copy_(torch.from_numpy(np.random.RandomState(1000 * u + rank).randn(int(offs[-1])).astype(np.float32)))
        want = sum(np.random.RandomState(1000 * u + r).randn(int(offs[-1])).astype(np.float32) for r in range(world)) / world
        for i in range(len(bounds) - 1):
            b3.launch(i)
        b3.finish()
        means_ok = means_ok and bool(np.allclose(g3.numpy()[: bounds[-1]], want[: bounds[-1]], atol=1e-6))
    sched = b3.schedule()
    q.put((rank, bounds, mine.numpy(), flat.numpy(), (m1 + m2).numpy(), acc.numpy(), twice, (means_ok, b3.auto, list(b3.forms_used), sched)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_buckets_average():
    from fcl_taco2_amd.training import _GROUPS, flat_layout, group_of

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, bounds, g0, avg0, s0, acc0, tw0, pol0), (_, _, g1, avg1, s1, acc1, tw1, pol1) = res
    assert tw0 and tw1  # double launch raises
    # round 6: the placement policy decides by measurement -- the trial alternated both forms on both ranks, every update still averaged correctly, and the
    # two ranks took the SAME decision (the medians are MAX-reduced before the comparison)
    from fcl_taco2_amd.training import GradBuckets

    for ok, auto, forms, sched in (pol0, pol1):
        assert ok and auto
        w, k = GradBuckets.TRIAL_WARMUP, GradBuckets.TRIAL_UPDATES
        assert forms[: w + 3 * k] == ["inline"] * (w + k) + ["async"] * k + ["inline"] * k  # stream-ordered | backend stream | stream-ordered
        assert sched["policy"] in ("inline", "async") and sched["decided_after_updates"] == w + 3 * k and forms[w + 3 * k :] == [sched["policy"]] * 2
        assert set(sched["trial_median_update_ms"]) == {"inline", "async", "inline_after"} and sched["world"] == 2 and len(sched["bucket_bytes"]) == len(bounds) - 1
    assert pol0[3]["policy"] == pol1[3]["policy"] and pol0[3]["trial_median_update_ms"] == pol1[3]["trial_median_update_ms"]
    assert np.array_equal(acc0, acc1) and np.allclose(acc0, (s0 + s1) / 2, atol=1e-6)  # accum_grad=2: mean over ranks of the accumulated sums, once
    assert bounds[0] == 0 and bounds[-1] == g0.shape[0] and all(a <= b for a, b in zip(bounds, bounds[1:])) and len(bounds) == len(_GROUPS) + 1
    assert np.array_equal(avg0, avg1)  # every rank ends with the same gradients -> same grad-norm -> all skip / step together (tts.py:173-178)
    assert np.allclose(avg0, (g0 + g1) / 2, atol=1e-7)
    # layout: parameters are grouped by when backward finishes them, slots 256-byte aligned
    names, offs, b2 = flat_layout([("enc.embed.weight", 7), ("dec.postnet.postnet.0.0.weight", 5), ("dec.feat_out.weight", 3), ("pitch_embed.0.bias", 2)])
    assert names == ["dec.postnet.postnet.0.0.weight", "dec.feat_out.weight", "pitch_embed.0.bias", "enc.embed.weight"]
    assert list(offs) == [0, 64, 128, 192, 256] and b2 == [0, 64, 128, 192, 256] and [group_of(n) for n in names] == [0, 1, 2, 3]
