"""The training step on the HIP path: forward with saved activations, backward, clip + Adam, data-parallel all-reduce.

Restates the reference's update (teacher: tts.py:137-179; KD: tts_distill.py:143-182):
    [teacher_knowledge = teacher(**x)]  loss = model(**x).mean() / accum_grad ; loss.backward()
    clip_grad_norm_(params, grad_clip) ; skip if NaN ; Adam.step() ; zero_grad()
with every FLOP in libfcl_hip.so (forward GEMMs reuse the synthesis kernels; gradients use backward.hip).  torch supplies device memory,
streams and torch.distributed only; there is no autograd graph.

Two forms of the stochastic layers, both pinned against the real reference (tests/golden/g5, g7, g8, g9):
  mode="eval"   the graph the reference builds under model.eval(): BatchNorm = affine map of the running statistics, nn.Dropout off, zoneout in
                its expectation form (decoder_sa.py:96); the prenet's dropout stays on whenever dropout_rate > 0 (decoder_sa.py:156-158);
  mode="train"  model.train(): BatchNorm normalises with the statistics of ALL positions of the padded batch and updates the running buffers,
                Dropout after every encoder / postnet conv block, after every predictor LayerNorm and after the pitch / energy embeds, zoneout
                samples keep-old masks for h and c (decoder_sa.py:92-94).
Every Bernoulli draw is either injected (`masks=`, the layouts of oracle.masks_from_sequence: how the goldens pin this path) or produced on the
device by fcl_bernoulli_u8 (production; statistically, not bit-wise, torch's stream).

Layouts: encoder rows (b, t) -> b*T + t, frame rows (b, l) -> b*L + l (zero padded like the reference's batch), decoder cells STEP-MAJOR: phoneme
rows sorted by duration (descending), cell (t, m) at offset[t] + m, so the live rows of a step are one contiguous slice of every saved tensor.
Gradients, Adam moments and (re-pointed) parameters live in three flat buffers ordered by when backward finishes them, so the optimizer is one
launch and the all-reduce runs over contiguous buckets while backward is still producing the later ones.
"""
import contextlib
import ctypes as C
import os
import zlib

import numpy as np
import torch

from . import _lib, ops
from .hparams import lstm_key, output_act_code
from .plan import BN_EPS, LN_EPS

BN_MOMENTUM = 0.1  # torch.nn.BatchNorm1d default

# parameter groups in the order backward completes them (= all-reduce bucket order)
_GROUPS = (("dec.postnet.", "dec.post_proj", "dec.post0_proj", "dec.post1_proj", "dec.post2_proj", "dec.post3_proj"),
           ("dec.",),
           ("pitch_embed.", "energy_embed.", "pemb_proj", "eemb_proj", "duration_predictor.", "pitch_predictor.", "energy_predictor."),
           ("enc.",))


def _i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


def _u8(a, dev):
    if torch.is_tensor(a):
        return a.to(device=dev, dtype=torch.uint8).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(dev)


def group_of(name):
    for g, prefixes in enumerate(_GROUPS):
        if any(name.startswith(p) for p in prefixes):
            return g
    return len(_GROUPS) - 1


class GradBuckets(object):
    """Bucketed gradient averaging over torch.distributed (RCCL on GPUs, gloo in the CPU tests).  launch(i) may be called as soon as bucket i
    is final; the collective runs on the backend's own stream while the caller keeps producing the later buckets; finish() waits for all.
    gloo + device tensors (the 2-processes-on-one-GPU test): buckets are staged through host memory in finish(), without overlap."""

    def __init__(self, flat, bounds, group=None, force=None):
        """force: run the collective branch for a ONE-rank group too (default: the FCL_DP_FORCE_COLLECTIVE environment flag).  A one-rank
        `nccl` group then executes exactly what a rank of an N-GPU job executes -- all_reduce(AVG, async_op=True) on slices of the flat device
        buffer, ordered against the compute streams -- with the identity as its result: the single-GPU test / timing of the data-parallel schedule."""
        import torch.distributed as dist

        self.flat, self.bounds, self.group, self.work, self.staged, self.launched = flat, bounds, group, [], [], set()
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ready else 1
        if force is None:
            force = os.environ.get("FCL_DP_FORCE_COLLECTIVE", "0") not in ("", "0")
        self.active = self.world > 1 or (bool(force) and ready)
        backend = dist.get_backend(group) if self.active else None
        self.avg = backend == "nccl"
        # gloo + device tensors: staged through host memory in finish() by default; FCL_DP_GLOO_DIRECT=1 issues the collective on the device slice
        # at launch() time instead (torch's gloo backend orders it behind the issuing stream and copies through pinned memory itself): the same
        # point of the backward at which a RCCL job issues it -- a bucket launched before its gradients are final would then be averaged stale
        # (the 2-process ordering test, VERDICT r4 #7b)
        self.stage_host = backend == "gloo" and flat.is_cuda and os.environ.get("FCL_DP_GLOO_DIRECT", "0") in ("", "0")
        self.collectives = 0  # all_reduce calls issued so far (tests assert the branch really ran)
        # inline (default): the collective is enqueued with async_op=False, i.e. stream-ordered on / behind the ISSUING stream (the engine's
        # weight-gradient stream) with no host wait, instead of async_op=True + wait() at the optimizer: the backend's extra stream is one more
        # ACTIVE hardware queue beside main / weight-gradient / frozen-teacher, and a fifth active queue costs this device a third of its
        # throughput (r4, one-rank RCCL group on one GPU: KD update 14.5 ms with async collectives against 10.7 ms without any).
        # FCL_DP_INLINE=0: the async form; =1: inline always; unset ("auto"): inline for buckets whose wire time is small against the
        # update -- a ring all-reduce of b bytes moves ~2 b per rank over xGMI links of ~100 GB/s achievable: <= 64 MB is <= ~1.3 ms on the
        # weight-gradient stream (4.8 ms busy of a 9.6 ms KD update; the four buckets of FCL-taco2-S are 6.5 MB each, of FCL-taco2-T 29 MB) --,
        # async (overlapped on the backend's stream, at the price of that fifth queue) for anything larger.  Unmeasured on a multi-GPU node.
        # Round 6: "auto" DECIDES BY MEASUREMENT on the job it runs in.  No 8-GPU node has been available to tune the placement on, and on one
        # GPU the wire is free, so the first updates of a job with a real group time both forms -- after `TRIAL_WARMUP` updates the placement runs three blocks
        # of `TRIAL_UPDATES` updates (stream-ordered, backend stream, stream-ordered), each update's wall time (finish() to finish(), device events) is booked to the form it ran under, the medians
        # are MAX-reduced over the ranks (one 2-float collective, once: every rank takes the same decision) and the faster form is kept; the choice,
        # both medians and the per-bucket wire times of the stream-ordered form are in `schedule()` (bench.py prints them as `dp_schedule`).
        env = os.environ.get("FCL_DP_INLINE", "auto")
        self.inline = env not in ("", "0")
        self.inline_max_bytes = (64 << 20) if env == "auto" else (1 << 62)
        # (RCCL, and gloo on HOST tensors -- the CPU tests run the same trial; gloo on device tensors keeps its fixed forms: host-staged, or asynchronous under
        # FCL_DP_GLOO_DIRECT, the ordering test)
        self.auto = env == "auto" and self.active and (self.avg or not flat.is_cuda)
        self.forms_used = []                   # the placement of every update so far (tests read it)
        self.updates = 0                       # finish() calls that reduced
        self.trial = {"inline": [], "async": [], "inline2": []}
        self.decision = None if self.auto else ("inline" if self.inline else "async")
        self.decided_at = None
        self._prev_mark = None                 # device event / host time of the previous finish()
        self._wire = {}                        # bucket -> [(start event, end event)] of its stream-ordered collectives during the trial
        self.wire_ms = {}

    TRIAL_WARMUP, TRIAL_UPDATES, TRIAL_BLOCKS = 3, 3, 3  # warm-up, updates per block, blocks: stream-ordered | backend stream | stream-ordered
    TRIAL_TOTAL = TRIAL_WARMUP + TRIAL_UPDATES * TRIAL_BLOCKS

    def _form_now(self):
        """The placement of the CURRENT update: the decision once taken, else the trial's -- warm-up and a block of TRIAL_UPDATES updates stream-ordered, then a
        block on the backend's stream (blocks, not alternation: a backend stream that has just run a collective stays an active hardware queue for a while and
        would tax the stream-ordered updates next to it; the first update of each block is the transition and is not booked)."""
        if self.decision is not None or not self.auto:
            return self.decision or ("inline" if self.inline else "async")
        blk = (self.updates - self.TRIAL_WARMUP) // self.TRIAL_UPDATES if self.updates >= self.TRIAL_WARMUP else 0
        return "async" if blk == 1 else "inline"  # A B A: the two stream-ordered blocks bracket the backend-stream block, so a drift of the job's own pace cannot pick the winner

    def _mark(self):
        if self.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        import time as _time

        return _time.perf_counter()

    @staticmethod
    def _elapsed_ms(a, b):
        if isinstance(a, float):
            return 1e3 * (b - a)
        b.synchronize()
        return a.elapsed_time(b)

    def _trial_step(self):
        """Called at the end of finish(): book this update's duration, decide when the trial is complete."""
        import torch.distributed as dist

        mark = self._mark()
        u = self.updates
        first_of_block = u >= self.TRIAL_WARMUP and (u - self.TRIAL_WARMUP) % self.TRIAL_UPDATES == 0
        if self._prev_mark is not None and u >= self.TRIAL_WARMUP and not first_of_block:
            blk = (u - self.TRIAL_WARMUP) // self.TRIAL_UPDATES
            self.trial["async" if blk == 1 else ("inline" if blk == 0 else "inline2")].append((self._prev_mark, mark))
        self._prev_mark = mark
        self.updates = u + 1
        if self.updates < self.TRIAL_TOTAL:
            return
        med = {}
        for k, pairs in self.trial.items():
            ms = sorted(self._elapsed_ms(a, b) for a, b in pairs)
            med[k] = ms[len(ms) // 2] if ms else float("inf")
        t = torch.tensor([med["inline"], med["async"], med.get("inline2", float("inf"))], dtype=torch.float32, device=self.flat.device if self.avg else "cpu")
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        m_in, m_as, m_in2 = (float(v) for v in t.tolist())
        # the backend's stream is taken only if it beats BOTH stream-ordered blocks around it (a one-sided comparison picked it on a one-rank group whose first
        # block was still warming up: 10.35 vs 9.65 ms in the trial, 10.2 against 9.0 ms in steady state)
        self.decision = "async" if m_as < min(m_in, m_in2) else "inline"
        self.decided_at = self.updates
        self.trial_ms = {"inline": m_in, "async": m_as, "inline_after": m_in2}
        for i, pairs in self._wire.items():
            ms = sorted(self._elapsed_ms(a, b) for a, b in pairs)
            self.wire_ms[i] = ms[len(ms) // 2]
        self._wire, self.trial = {}, {"inline": [], "async": [], "inline2": []}
        import logging

        logging.info("GradBuckets: collective placement decided after %d updates: %s (median update %.3f ms stream-ordered on the issuing stream, %.3f ms on the "
                     "backend's stream; per-bucket wire ms %s)", self.updates, self.decision, m_in, m_as, {k: round(v, 3) for k, v in sorted(self.wire_ms.items())})

    def schedule(self):
        """What bench.py prints as `dp_schedule.policy`."""
        return {"policy": self.decision or "trial (%d of %d updates)" % (self.updates, self.TRIAL_TOTAL), "auto": bool(self.auto),
                "decided_after_updates": self.decided_at, "trial_median_update_ms": getattr(self, "trial_ms", None),
                "bucket_wire_ms": {str(k): v for k, v in sorted(self.wire_ms.items())}, "bucket_bytes": [int((self.bounds[i + 1] - self.bounds[i]) * self.flat.element_size())
                                                                                                         for i in range(len(self.bounds) - 1)],
                "world": self.world}

    def launch(self, i):
        """Start averaging bucket i.  Once per optimizer step: with gradient accumulation only the LAST micro-batch may launch (an in-flight
        all-reduce must not overlap the next micro-batch's writes into the same gradient memory), so a second launch before finish() raises.
        The collective is ordered behind the CURRENT stream (torch's ProcessGroupNCCL makes its own stream wait for it): the caller issues it
        from the stream that wrote the bucket last."""
        if not self.active:
            return
        import torch.distributed as dist

        if i in self.launched:
            raise RuntimeError("GradBuckets: bucket %d launched twice before finish() — reduce only on the last micro-batch of an accumulation" % i)
        self.launched.add(i)
        a, b = self.bounds[i], self.bounds[i + 1]
        if b <= a:
            return
        if self.stage_host:
            self.staged.append((a, b))
            return
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        if self.auto:
            inline = self._form_now() == "inline"
        else:
            inline = self.inline and self.avg and (b - a) * self.flat.element_size() <= self.inline_max_bytes
        if inline:
            timing = self.auto and self.decision is None and self.flat.is_cuda and self.updates >= self.TRIAL_WARMUP
            e0 = self._mark() if timing else None
            dist.all_reduce(self.flat[a:b], op=op, group=self.group, async_op=False)  # the issuing stream is ordered behind it; the host is not
            if timing:  # (events on the issuing stream around a stream-ordered collective: what the stream stood still for = the bucket's wire time)
                self._wire.setdefault(i, []).append((e0, self._mark()))
        else:
            self.work.append(dist.all_reduce(self.flat[a:b], op=op, group=self.group, async_op=True))
        self.collectives += 1

    def finish(self, scale_fn=None):
        import torch.distributed as dist

        for a, b in self.staged:
            host = self.flat[a:b].cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat[a:b].copy_(host)
        self.staged = []
        for w in self.work:
            w.wait()
        self.work = []
        reduced, self.launched = bool(self.launched), set()
        if not reduced:
            return
        if self.active and not self.avg:
            (scale_fn or (lambda t, s: t.mul_(s)))(self.flat, 1.0 / self.world)
        self.forms_used.append(self._form_now())
        if self.auto and self.decision is None:
            self._trial_step()


def flat_layout(names_sizes):
    """Order parameters by backward-completion group; returns (ordered names, offsets [n+1], bucket bounds [len(_GROUPS)+1])."""
    order = sorted(range(len(names_sizes)), key=lambda i: (group_of(names_sizes[i][0]), i))
    names = [names_sizes[i][0] for i in order]
    sizes = [(names_sizes[i][1] + 63) // 64 * 64 for i in order]  # 256-byte aligned slots (kernels want 16-byte aligned rows); padding stays 0
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    bounds = [0]
    for g in range(len(_GROUPS)):
        last = [j for j, n in enumerate(names) if group_of(n) <= g]
        bounds.append(int(offs[last[-1] + 1]) if last else bounds[-1])
    return names, offs, bounds


from .hostio import PinnedRing  # noqa: E402,F401  (fixed pinned staging buffers; shared with engine.prepare)


def build_maps_host(batch, r=1):
    """The host half of the index maps (SURVEY.md H9/H10 for the teacher-forced layout): pure numpy on the converter's integer tensors, safe to run
    in a loader process.  Returns scalars, two packed blocks (int32, uint8) with their {name: (offset, size)} layouts, and the position column.
    r (`reduction_factor`, decoder_sa.py:487-489, 512-516): the converter's ds_nonzeros count FRAMES (tts.py:256); a decoder cell is one STEP = r
    consecutive frames.  Cells, live rows and lmax count steps; `cell_frame` is the first frame of a cell's group, `prev_frame` the frame before it
    (the teacher-forced input), `frame_cell` / `cell_frame` index GROUPS of r frames when r > 1 (a [B * L, odim] frame buffer viewed as
    [B * L / r, r * odim]); the position column is t / (r d) as the converter's table has it (tts.py:258)."""
    ilens = [int(v) for v in batch["ilens"]]
    olens = [int(v) for v in batch["olens"]]
    B, T, L = len(ilens), max(ilens), max(olens)
    rows = np.arange(B * T)
    b_of = rows // T
    lens_np = np.asarray(ilens)
    pad_np = (rows % T) >= lens_np[b_of]
    frows = np.arange(B * L)
    fvalid_np = (frows % L) < np.asarray(olens)[frows // L]
    nzm = np.asarray(batch["non_zero_lens_mask"])[:, :T] != 0
    dsn_frames = np.asarray(batch["ds_nonzeros"]).astype(np.int64)
    r = int(r)
    if r > 1 and ((dsn_frames % r).any() or any(o % r for o in olens) or L % r):
        raise ValueError("fcl-taco2_amd: reduction_factor %d needs segment and utterance lengths that are multiples of it (the converter's ds_nonzeros "
                         "are r x the durations, tts.py:256; the class splits the concatenated frames by olens, decoder_sa.py:519-522)" % r)
    dsn = dsn_frames // r  # decoder STEPS per phoneme row
    src = np.flatnonzero(nzm.reshape(-1))
    assert src.shape[0] == dsn.shape[0], "hs.shape[0] != len(ds_nonzeros)"  # decoder_sa.py:468
    N = src.shape[0]
    b_row = src // T
    excl = np.cumsum(dsn_frames) - dsn_frames
    first = np.concatenate([[0], np.cumsum(np.bincount(b_row, minlength=B))[:-1]])
    foff = b_row * L + (excl - excl[first[b_row]])  # frame row (b * L + l) of the phoneme's first frame
    order = np.argsort(-dsn, kind="stable")
    dur_s, foff_s = dsn[order], foff[order]
    lmax = int(dur_s[0])
    live = (dur_s[None, :] > np.arange(lmax)[:, None]).sum(1)
    offs = np.concatenate([[0], np.cumsum(live)])  # step-major cell offsets
    cell_row = np.concatenate([np.arange(n) for n in live])  # sorted-row index of every cell
    cell_t = np.repeat(np.arange(lmax), live)
    cell_frame = foff_s[cell_row] + cell_t * r  # frame row (b*L + l) of every cell's (first) frame
    F = int(offs[-1])
    assert F * r == int(fvalid_np.sum()), "sum of durations != olens"
    frame_cell = np.full(B * L // r, -1, dtype=np.int64)  # per GROUP of r frames: the cell that emits it
    frame_cell[cell_frame // r] = np.arange(F)
    inv = np.full(B * T, -1, dtype=np.int64)
    inv[src[order]] = np.arange(N)
    i32 = dict(lens_dev=lens_np, e_lo=b_of * T, e_hi=b_of * T + T, f_lo=(frows // L) * L, f_hi=(frows // L) * L + L, src_sorted=src[order],
               row_of_enc=inv, cell_frame=cell_frame // r, frame_cell=frame_cell, prev_frame=np.where(cell_t > 0, cell_frame - 1, -1),
               cell_row_i32=cell_row, dur_dev=dur_s * r, perm_tb=(rows % T) * B + rows // T)  # dur_dev: the in-kernel position's divisor, in frames
    u8 = dict(enc_pad=pad_np, enc_valid=~pad_np, frame_valid=fvalid_np)
    out = dict(scalars=dict(B=B, T=T, L=L, N=N, F=F, lmax=lmax, live=live, offs=offs, order=order, cell_row=cell_row, cell_t=cell_t, dur_s=dur_s,
                            n_enc=float((~pad_np).sum()), n_frames=float(fvalid_np.sum()), live_i32=np.ascontiguousarray(live, dtype=np.int32)))
    for name, block, dtype in (("i32", i32, np.int32), ("u8", u8, np.uint8)):
        layout, o = {}, 0
        for k, v in block.items():
            layout[k] = (o, int(v.size))
            o += (v.size + 15) // 16 * 16  # 64-byte aligned slots
        host = np.zeros(o, dtype=dtype)
        for k, v in block.items():
            host[layout[k][0] : layout[k][0] + v.size] = v
        out[name] = torch.from_numpy(host)
        out[name + "_layout"] = layout
    pos = np.zeros((F, 4), dtype=np.float32)
    pos[:, 0] = cell_t.astype(np.float32) / (dur_s[cell_row] * r).astype(np.float32)  # the position input t / (r d), padded to 4 columns (TN-GEMM operand)
    out["pos4"] = torch.from_numpy(pos)
    return out


_RING = PinnedRing()  # staging of the index maps (one process = one GPU)


class _Ctx(object):
    pass


class _ConvP(object):
    """What ops.conv1d_planes needs from a convolution: P32 planes of the packed taps, bias, sizes."""
    __slots__ = ("wpp", "bias", "cin", "cout", "k")

    def __init__(self, wpp, bias, cout, cin, k):
        self.wpp, self.bias, self.cout, self.cin, self.k = wpp, bias, cout, cin, k


class LossReport(dict):
    """The named losses of a step.  The fp64 loss sums are copied to pinned host memory asynchronously; the dict fills itself on first access, so a
    training loop that only logs every N steps never stalls the CPU on the GPU (the .cpu() read-back used to cost a full pipeline drain per step)."""

    def __init__(self, sums_dev, names, extra=None, status_dev=None, host=None):
        """host = (pinned [len(names), 3] float64, pinned int32 [1] or None): the copies were already enqueued on the current stream by the caller
        (the native step, fcl_te_forward_backward); slots whose count is 0 (terms the configuration does not compute) are ignored."""
        super().__init__()
        self._names = list(names)
        if host is not None:
            self._host, self._status = host
        else:
            self._host = torch.empty((len(names), 3), dtype=torch.float64, pin_memory=True)
            self._host.copy_(sums_dev[: len(names)], non_blocking=True)
            self._status = None
            if status_dev is not None:  # the device status word rides along: 4 bytes on the same stream
                self._status = torch.empty(1, dtype=torch.int32, pin_memory=True)
                self._status.copy_(status_dev, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        self._extra = dict(extra or {})
        self._done = False

    def resolve(self):
        if self._done:
            return self
        self._event.synchronize()
        if self._status is not None:
            bits = int(self._status.numpy()[0]) & 0xFFFFFFFF
            if bits:
                from ._lib import FclError

                raise FclError("fcl-taco2_amd: device status 0x%x during this step (the parameter update was skipped on the device): %s"
                               % (bits, ops.status_message(bits)))
        host = self._host.numpy()
        m = {n: (host[i, 0] / host[i, 2], host[i, 1] / host[i, 2]) for i, n in enumerate(self._names) if host[i, 2] > 0}
        rep = dict(l1_loss=m["after"][0] + m["before"][0], mse_loss=m["after"][1] + m["before"][1], dur_loss=m["dur"][1], pitch_loss=m["pitch"][1],
                   energy_loss=m["energy"][1])
        rep["loss"] = rep["l1_loss"] + rep["mse_loss"] + rep["dur_loss"] + rep["pitch_loss"] + rep["energy_loss"]
        if "o_after" in m:
            rep["output_l1_loss"] = m["o_after"][0] + m["o_before"][0]
            rep["output_mse_loss"] = m["o_after"][1] + m["o_before"][1]
            rep["loss"] += rep["output_l1_loss"] + rep["output_mse_loss"]
        for key, pre, n in (("encoder_loss", "enc", 5), ("decoder_loss", "dec", 8), ("prosody_loss", "pro", 5)):
            if pre + "0" in m:
                rep[key] = sum(m["%s%d" % (pre, i)][1] for i in range(n))
                rep["loss"] += rep[key]
        for k, v in self._extra.items():
            rep[k] = v() if callable(v) else v
        self._done = True
        super().update(rep)
        return self

    def __getitem__(self, k):
        return dict.__getitem__(self.resolve(), k)

    def __contains__(self, k):
        return dict.__contains__(self.resolve(), k)

    def __iter__(self):
        return dict.__iter__(self.resolve())

    def keys(self):
        return dict.keys(self.resolve())

    def items(self):
        return dict.items(self.resolve())

    def values(self):
        return dict.values(self.resolve())

    def get(self, k, d=None):
        return dict.get(self.resolve(), k, d)

    def __len__(self):
        return dict.__len__(self.resolve())

    def __repr__(self):
        return dict.__repr__(self.resolve())

    def __setitem__(self, k, v):
        if self._done:
            dict.__setitem__(self, k, v)
        else:
            self._extra[k] = v


class ZeroArena(object):
    """Zero-initialised scratch of one training step from ONE buffer cleared by ONE launch: a step used to issue ~90 fill launches for its
    accumulation targets (per-layer dbeta / dgamma / packed dW / scatter targets / loss sums: torch.zeros each).  take() bumps a pointer; begin()
    -- called when a step starts, i.e. after the previous step's side-stream consumers were joined -- clears what the previous step used.  The
    first step (and any step that outgrows the buffer) falls back to torch.zeros and sizes the buffer for the next one."""

    def __init__(self, device):
        self.device, self.buf, self.used, self.want = device, None, 0, 0

    def begin(self):
        need = self.want
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.zeros(max(need * 5 // 4, 1 << 20), dtype=torch.uint8, device=self.device) if need else None
        elif self.used:
            self.buf[: self.used].zero_()
        self.used, self.want = 0, 0

    def take(self, shape, dtype=torch.float32):
        n = 1
        for v in shape:
            n *= int(v)
        nbytes = (n * torch.empty(0, dtype=dtype).element_size() + 255) // 256 * 256
        self.want += nbytes
        if self.buf is None or self.used + nbytes > self.buf.numel():
            return torch.zeros(tuple(shape), dtype=dtype, device=self.device)
        out = self.buf[self.used : self.used + nbytes].view(dtype)[:n].view(tuple(shape))
        self.used += nbytes
        return out


class NativeKnowledge(object):
    """The frozen KD teacher's knowledge as produced by the native step (csrc/train_engine.hip): device pointers into one of the teacher engine's
    two alternating arenas (valid until the second next knowledge() call on that engine), decoder taps CELL-major (teacher and student share the
    batch's index maps, so the frame round trip of the reference's tuple is skipped)."""
    __slots__ = ("struct", "engine", "batch_id")

    def __init__(self, struct, engine, batch_id):
        self.struct, self.engine, self.batch_id = struct, engine, batch_id


class _NativeStep(object):
    """TrainEngine's native form of the step: fcl_te_* of include/fcl_hip.h.  The ~530 launches of a KD update are issued by ONE C++ routine per
    backward stage instead of one ctypes call each (8.7 of 11.0 ms per update were host time)."""

    @staticmethod
    def unsupported(eng):
        """None when the native routine covers this engine's configuration, else the reason (the per-launch Python path is used then)."""
        hp = eng.hp
        if os.environ.get("FCL_TRAIN_NATIVE", "1") in ("", "0"):
            return "FCL_TRAIN_NATIVE=0"
        if not ops.planes_enabled():
            return "pre-split operands are off (FCL_PRECISION=0 / FCL_PLANES=0)"
        if hp.spk_embed_dim is not None or hp.use_residual or hp.output_activation is not None:
            return "speaker embeddings / residual encoder / output activation"
        if not (hp.zoneout_rate > 0.0 and hp.use_concate and hp.append_position and hp.use_batch_norm):
            return "zoneout_rate 0 / use_concate False / append_position False / use_batch_norm False (options outside the shipped recipes)"
        if not eng.overlap_dw:
            return "overlap_dw=False"
        widths = (hp.embed_dim, hp.econv_chans, hp.dunits, hp.prenet_units, hp.postnet_chans, hp.duration_predictor_chans, hp.variance_predictor_chans)
        if any(v % 32 for v in widths) or hp.eunits % 64 or hp.odim % 4 or not (hp.embed_dim == hp.econv_chans == hp.eunits):
            return "channel widths that are not multiples of 32"
        if (hp.elayers, hp.dlayers, hp.prenet_layers, hp.reduction_factor) != (1, 2, 2, 1):
            return "elayers / dlayers / prenet_layers / reduction_factor other than 1 / 2 / 2 / 1 (the native routine issues the shipped structure's launches)"
        if hp.econv_layers > 8 or hp.postnet_layers > 8 or hp.duration_predictor_layers > 4 or hp.variance_predictor_layers > 4:
            return "more layers than the native routine's tables hold (econv / postnet <= 8, predictors <= 4: fcl_te_create)"  # 
        if eng.role == "student" and eng.distill[2] and hp.postnet_layers != 5:
            return "decoder distillation with postnet_layers != 5"
        if eng.role == "student" and eng.distill[1] and hp.econv_layers != 3:
            return "encoder distillation with econv_layers != 3"
        return None

    def __init__(self, eng):
        import ctypes as C
        from . import _lib

        self.eng, self.C, self._lib = eng, C, _lib
        lib = self.lib = _lib.load()
        hp = eng.hp
        role = {"teacher": _lib.TE_TEACHER, "kd_teacher": _lib.TE_KD_TEACHER, "student": _lib.TE_STUDENT}[eng.role]
        cfg = _lib.TeConfig(role=role, idim=hp.idim, odim=hp.odim, embed_dim=hp.embed_dim, econv_layers=hp.econv_layers, econv_chans=hp.econv_chans,
                            econv_filts=hp.econv_filts, eunits=hp.eunits, dunits=hp.dunits, prenet_units=hp.prenet_units, postnet_layers=hp.postnet_layers,
                            postnet_chans=hp.postnet_chans, postnet_filts=hp.postnet_filts, dp_layers=hp.duration_predictor_layers,
                            dp_chans=hp.duration_predictor_chans, dp_kernel=hp.duration_predictor_kernel_size, vp_layers=hp.variance_predictor_layers,
                            vp_chans=hp.variance_predictor_chans, vp_kernel=hp.variance_predictor_kernel_size, ve_kernel=hp.variance_embed_kernel_size,
                            dropout_rate=hp.dropout_rate, zoneout_rate=float(hp.zoneout_rate), dp_dropout=hp.duration_predictor_dropout_rate,
                            vp_dropout=hp.variance_predictor_dropout_rate, ve_dropout=hp.variance_embed_dropout_rate, use_masking=int(bool(hp.use_masking)),
                            accum_grad=eng.accum_grad, seed=eng.seed & 0xFFFFFFFF, dw_planes_min=eng._dw_planes_min,
                            pred_stream=int(eng.pstream is not None), late_losses=int(eng._late_losses))
        if eng.role == "student":
            thp = eng.model.teacher_hp
            cfg.t_embed_dim, cfg.t_econv_chans, cfg.t_eunits = thp.embed_dim, thp.econv_chans, thp.eunits
            cfg.t_prenet_units, cfg.t_dunits, cfg.t_postnet_chans = thp.prenet_units, thp.dunits, thp.postnet_chans
            cfg.share_proj = int(eng.share_proj)
            cfg.distill_output, cfg.distill_encoder, cfg.distill_decoder, cfg.distill_prosody = [int(v) for v in eng.distill]
        # mask seeds: training.py's per-site tags (crc32 of the Python site name), in the order of the library's site table
        for i in range(_lib.TE_MAX_SITES):
            nm = lib.fcl_te_site_name(i)
            if nm is None:
                break
            parts = nm.decode().split("/")
            key = (parts[0],) + tuple(int(v) for v in parts[1:])
            cfg.site_tag[i] = zlib.crc32(repr(key).encode()) & 0x7FFFFFFF
        self.loss_names = []
        for i in range(_lib.TE_MAX_LOSSES):
            nm = lib.fcl_te_loss_name(i)
            if nm is None:
                break
            self.loss_names.append(nm.decode())
        h = C.c_void_p()
        _lib.check(lib.fcl_te_create(C.byref(cfg), C.byref(h)))
        self.h, self.cfg = h, cfg
        for k, v in eng.P.items():
            _lib.check(lib.fcl_te_bind_param(h, k.encode(), v.data_ptr(), eng.G[k].data_ptr(), v.numel()))
        for k, v in eng.B.items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                assert v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()
                _lib.check(lib.fcl_te_bind_buffer(h, k.encode(), v.data_ptr()))
        _lib.check(lib.fcl_te_finalize(h, eng.status.data_ptr()))
        self.side_moved = False
        if int(os.environ.get("FCL_PLACE_STREAMS", "1")):  # the weight-gradient stream on a compute pipe of its own (fcl_hip.h "Compute pipes")
            moved = C.c_int(0)
            with torch.cuda.device(eng.dev):
                _lib.check(lib.fcl_te_place_streams(h, C.c_void_p(torch.cuda.current_stream(eng.dev).cuda_stream), C.byref(moved)))
            self.side_moved = moved.value  # 1 moved, 0 fine as created, -1 contends and could not be placed
        self.side = torch.cuda.ExternalStream(lib.fcl_te_side_stream(h), device=eng.dev)
        self._hosts = []  # pinned landing buffers of the last steps' loss sums (kept until their copies have certainly run)
        self.stage_hook = None  # optional callable(stage) run after backward stage 0 .. 3 has been enqueued (KDPipeline)
        self._keep = None

    def __del__(self):
        try:
            if getattr(self, "h", None) is not None and self.h.value:
                self.lib.fcl_te_destroy(self.h)
                self.h = None
        except Exception:  # pragma: no cover - interpreter shutdown
            pass

    def params_changed(self):
        self._lib.check(self.lib.fcl_te_params_changed(self.h))

    def batch_struct(self, batch):
        """fcl_te_batch_t of a converter batch: the engine's index maps (TrainEngine._maps: built once per batch, shared by teacher and student) plus
        the float inputs, flattened once and cached on the batch."""
        eng, _lib = self.eng, self._lib
        hit = batch.get("_fcl_te_batch") if isinstance(batch, dict) else None
        if hit is not None and hit[0] == str(eng.dev):
            return hit[1]
        c = _Ctx()
        eng._maps(c, batch)
        dev, T, L = eng.dev, c.T, c.L
        xs = batch["xs"][:, :T].to(dev).to(torch.int64).reshape(-1).contiguous()
        ys = batch["ys"][:, :L].to(dev).float().reshape(c.B * L, eng.hp.odim).contiguous()
        f0 = batch["f0"][:, :T].to(dev).float().reshape(-1).contiguous()
        en = batch["energy"][:, :T].to(dev).float().reshape(-1).contiguous()
        ds = batch["extras"][:, :T].to(dev).float().reshape(-1).contiguous()
        live = np.ascontiguousarray(c.live_i32, dtype=np.int32)
        st = _lib.TeBatch(B=c.B, T=T, L=L, N=c.N, F=c.F, lmax=c.lmax, xs=xs.data_ptr(), ys=ys.data_ptr(), f0=f0.data_ptr(), energy=en.data_ptr(), ds=ds.data_ptr(),
                          lens=c.lens_dev.data_ptr(), e_lo=c.e_lo.data_ptr(), e_hi=c.e_hi.data_ptr(), f_lo=c.f_lo.data_ptr(), f_hi=c.f_hi.data_ptr(),
                          src_sorted=c.src_sorted.data_ptr(), row_of_enc=c.row_of_enc.data_ptr(), cell_frame=c.cell_frame.data_ptr(),
                          frame_cell=c.frame_cell.data_ptr(), prev_frame=c.prev_frame.data_ptr(), cell_row=c.cell_row_i32.data_ptr(), dur=c.dur_dev.data_ptr(),
                          perm_tb=c.perm_tb.data_ptr(), cell_row_i64=c.cell_row_i64.data_ptr(), enc_pad=c.enc_pad.data_ptr(), enc_valid=c.enc_valid.data_ptr(),
                          frame_valid=c.frame_valid.data_ptr(), cell_valid=c.cell_valid.data_ptr(), pos4=c.pos4.data_ptr(), live_rows_host=live.ctypes.data,
                          n_enc=float(c.n_enc), n_frames=float(c.n_frames))
        keep = (xs, ys, f0, en, ds, live, c)  # everything the struct points at lives as long as the batch's cache entry
        if isinstance(batch, dict):
            batch["_fcl_te_batch"] = (str(eng.dev), (st, keep))
        return st, keep

    def knowledge(self, batch):
        eng, C = self.eng, self.C
        st, keep = self.batch_struct(batch)
        eng.forward_count += 1
        know = self._lib.TeKnowledge()
        with torch.cuda.device(eng.dev), ops.gemm_mode(eng.amp):
            self._lib.check(self.lib.fcl_te_knowledge(self.h, C.byref(st), eng.forward_count & 0xFFFFFFFF, C.byref(know), ops._stream()))
        eng._native_nbt()
        return NativeKnowledge(know, eng, id(batch))

    def _knowledge_struct(self, know, c_batch):
        """fcl_te_knowledge_t of whatever the caller handed to the student: a NativeKnowledge, or the reference-shaped 5-tuple of tensors."""
        if isinstance(know, NativeKnowledge):
            return know.struct, know
        eng = self.eng
        t_after, t_before, t_enc, t_dec, t_pro = know
        flat = lambda t: t.to(device=eng.dev, dtype=torch.float32).reshape(-1, t.shape[-1]).contiguous()
        tens = [flat(t_after), flat(t_before)] + [flat(t) for t in t_enc] + [flat(t) for t in t_dec] + [flat(t) for t in t_pro]
        ks = self._lib.TeKnowledge(after=tens[0].data_ptr(), before=tens[1].data_ptr(), dec_cell_major=0)
        for i in range(5):
            ks.enc[i] = tens[2 + i].data_ptr()
        for i in range(8):
            ks.dec[i] = tens[7 + i].data_ptr()
        for i in range(5):
            ks.pro[i] = tens[15 + i].data_ptr()
        return ks, tens

    def forward_backward(self, batch, teacher_knowledge, reduce):
        eng, C, lib = self.eng, self.C, self.lib
        st, keep = self.batch_struct(batch)
        kptr = None
        if eng.role == "student":
            ks, kkeep = self._knowledge_struct(teacher_knowledge, st)
            kptr, self._keep = C.byref(ks), (ks, kkeep, keep)
        else:
            self._keep = keep
        eng.forward_count += 1
        host = torch.empty((self._lib.TE_MAX_LOSSES, 3), dtype=torch.float64, pin_memory=True)
        stat = torch.empty(1, dtype=torch.int32, pin_memory=True)
        self._hosts.append([host, stat, None])
        while len(self._hosts) > 16:  #  a landing buffer is released only once the copy that fills it has certainly run
            old = self._hosts.pop(0)
            if old[2] is not None:
                old[2].synchronize()
        do_reduce = reduce and eng.buckets.active
        main = torch.cuda.current_stream(eng.dev)

        def bucket(i):
            if do_reduce:
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    eng.buckets.launch(i)

        with torch.cuda.device(eng.dev), ops.gemm_mode(eng.amp):
            s = ops._stream()
            self._lib.check(lib.fcl_te_forward_backward(self.h, C.byref(st), kptr, eng.forward_count & 0xFFFFFFFF, host.data_ptr(), stat.data_ptr(), s))
            self._hosts[-1][2] = torch.cuda.Event()
            self._hosts[-1][2].record(main)  # behind the two device-to-host copies of this step's loss sums / status word
            rep = LossReport(None, self.loss_names, host=(host, stat))
            eng._native_nbt()
            bucket(0)
            hook = self.stage_hook  # KDPipeline: the frozen teacher's next forward may be enqueued between two backward stages
            if hook is not None:
                hook(0)
            for stage in (1, 2, 3):
                self._lib.check(lib.fcl_te_backward_stage(self.h, stage, s))
                bucket(stage)
                if hook is not None:
                    hook(stage)
            self._lib.check(lib.fcl_te_join(self.h, s))
        return rep

    def launches(self):
        return int(self.lib.fcl_te_last_launches(self.h))


_DW_PLANES_MIN = int(os.environ.get("FCL_DW_PLANES_MIN", "0"))  # output elements from which a weight gradient runs on transposed planes (0: per role, below)
# (measured r3: sending the student's long-contraction / small-output gradients -- 13 GFLOP into [1024, 256] over ~25 k frames, 65 - 110 TFLOP/s on the
# fp32-operand kernel -- to the planes kernel as well is a wash: its two transposing passes cost what the faster GEMM saves; KD update 12.66 vs 12.59 ms)


class TrainEngine(object):
    def __init__(self, model, lr=1e-3, eps=1e-6, betas=(0.9, 0.999), grad_clip=1.0, accum_grad=1, seed=0, group=None, overlap_dw=True, amp=None, native=True,
                 weight_decay=0.0):
        """native: train-form passes without injected masks run as ONE C++ routine per backward stage (fcl_te_*, csrc/train_engine.hip) when the
        configuration is covered (`native_reason` says why not); False = always the per-launch path below, which is also its reference.
        amp: None = fp32-equivalent arithmetic; "bf16" = the mixed-precision form of the reference's `--use-amp True` recipe (apex O1,
        tts.py:414-416) with bf16 in place of fp16: the big-tile forward / input-gradient / weight-gradient GEMMs take bf16-rounded operands and
        accumulate in fp32 (ops.gemm_mode), master weights, norms, losses, gradients and Adam stay fp32, no loss scaling is needed."""
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise RuntimeError("fcl-taco2_amd: TrainEngine needs the model on a GPU (no CPU fallback)")
        self.model, self.hp, self.dev, self.role = model, model.hp, p0.device, model.role
        # weight gradients on transposed planes from 1 M outputs on; from 256 k in the teacher's OWN update (same-box A/B at the end of round 3: teacher
        # update 11.32 -> 11.15 ms; the KD update, whose weight-gradient stream also carries the predictors and the late loss terms beside the frozen
        # teacher's forward, loses with it: 10.78 -> 10.83 ms)
        # (round 6, profiles/r6_dw_route_ab.log: with dw_mfma_kernel and an update whose pace is set by the bytes it moves, the teacher's threshold is 4 M outputs --
        # only the 4 096 x 1 024 LSTM matrices keep the transposed-planes route: teacher update 9.70 -> 9.44 ms same box; 256 k, 1 M and "never" are all slower)
        self._dw_planes_min = _DW_PLANES_MIN or ((1 << 22) if model.role == "teacher" else (1 << 20))
        if self.role == "student" and self.hp.spk_embed_dim is not None:
            raise NotImplementedError("fcl-taco2_amd: KD training with speaker embeddings is undefined in the reference (its student's pemb_proj / eemb_proj "
                                      "are built for eunits inputs but receive eunits + spk_embed_dim channels: tests/golden/records.json)")
        if self.role in ("student", "kd_teacher") and not self.hp.use_concate:
            raise NotImplementedError("fcl-taco2_amd: KD training with use_concate False is undefined in the reference (its KD decoder's forward() hands "
                                      "feat_out the list z_list[-1]: decoder_sa_kd.py:617-622, tests/golden/records.json)")
        if self.role in ("student", "kd_teacher") and (self.hp.econv_layers != 3 or self.hp.postnet_layers != 5):
            raise NotImplementedError("fcl-taco2_amd: KD training needs econv_layers 3 and postnet_layers 5: the reference's KD classes index fixed lists of "
                                      "encoder / postnet taps and raise IndexError otherwise (tests/golden/records.json)")
        if self.role in ("student", "kd_teacher") and self.hp.reduction_factor != 1:
            raise NotImplementedError("fcl-taco2_amd: KD training is built for reduction_factor 1 (the KD taps are frame-level; no shipped recipe sets r > 1); "
                                      "the teacher class trains and synthesises with r > 1")
        if self.role in ("student", "kd_teacher") and self.hp.dlayers != 2:
            # the reference's KD decoder taps cells 0 and 1 by index (decoder_sa_kd.py:626-627): IndexError with one cell (records.json); with three it
            # runs, tapping the MIDDLE cell -- not built here (no shipped recipe, no golden): refused rather than guessed
            raise NotImplementedError("fcl-taco2_amd: KD training needs dlayers 2 (the reference's KD decoder taps lstm cells 0 and 1 by index: "
                                      "decoder_sa_kd.py:626-627); the teacher class trains with dlayers 1 .. 3")
        if self.role != "kd_teacher":  # the frozen KD teacher computes no loss
            self.hp.check_loss_supported()
        self.share_proj = bool(getattr(model, "share_proj", True))
        self.distill = tuple(bool(getattr(model, "distill_%s_knowledge" % k, True)) for k in ("output", "encoder", "decoder", "prosody"))
        pd = dict(model.named_parameters())
        names, offs, bounds = flat_layout([(k, v.numel()) for k, v in pd.items()])
        total = int(offs[-1])
        self.pflat = torch.zeros(total, device=self.dev)  # zeros: the 256-byte alignment padding between parameters must stay inert
        self.gflat, self.mflat, self.vflat = (torch.zeros(total, device=self.dev) for _ in range(3))
        self.P, self.G, self._offsets = {}, {}, {}
        for k, o in zip(names, offs[:-1]):
            self._offsets[k] = (int(o), pd[k].numel(), tuple(pd[k].shape))
            n = pd[k].numel()
            view = self.pflat[o : o + n].view(pd[k].shape)
            view.copy_(pd[k].data)
            pd[k].data = view  # the module's parameters now live in the flat buffer: state_dict() / plan() see every update
            self.P[k] = view
            self.G[k] = self.gflat[o : o + n].view(pd[k].shape)
        self.buckets = GradBuckets(self.gflat, bounds, group)
        self.B = dict(model.named_buffers())
        self.lr, self.eps, self.betas, self.grad_clip, self.accum_grad = lr, eps, betas, grad_clip, int(accum_grad)
        self.weight_decay = float(weight_decay)  # torch.optim.Adam(weight_decay=...) of tts.py:397-399 / tts_distill.py:418-420
        if self.weight_decay < 0.0:
            raise ValueError("Invalid weight_decay value: %g" % self.weight_decay)  # (torch.optim.Adam's own check)
        self.arena = ZeroArena(self.dev)
        self._z = self.arena.take
        self.forward_count, self.seed = 0, int(seed)
        if amp not in (None, "bf16"):
            raise ValueError("amp must be None or 'bf16'")
        self.amp = amp
        self.update_calls = 0  # optimizer_step() calls (host side; no sync)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)  # APPLIED updates = torch.optim.Adam's per-parameter `step`
        self.status = ops.status_word(self.dev)  # shared per device: a failure reported by the frozen teacher's kernels stops this update too
        self.gn_sq = torch.zeros(1, dtype=torch.float64, device=self.dev)
        # weight gradients are off the critical path of backward (only the input-gradient chain is sequential): they are enqueued on a side
        # stream and joined at the end of backward / before a bucket's all-reduce
        self.overlap_dw = overlap_dw
        if overlap_dw and int(os.environ.get("FCL_PLACE_STREAMS", "1")):
            with torch.cuda.device(self.dev):
                self.side = ops.stream_apart([torch.cuda.current_stream(self.dev)], device=self.dev)  # on a compute pipe of its own (ops.stream_apart)
        else:
            self.side = torch.cuda.Stream(device=self.dev) if overlap_dw else None
        # The three predictors feed nothing but their own losses in a teacher-forced step (the variance embeddings take the ground-truth pitch /
        # energy): their forward runs beside the decoder's forward and their backward beside the decoder's backward -- ~90 small dependent
        # launches off the main stream's chain -- on the weight-gradient stream, which is idle through the forward and far from full in the
        # backward.  NOT a stream of their own: with the frozen teacher's stream that made five active queues in a KD update (18.6 instead of
        # 11.7 ms: DESIGN §5, the device runs four queues well).  FCL_PRED_STREAM=0: in line, as before
        self.pstream = self.side if (overlap_dw and os.environ.get("FCL_PRED_STREAM", "1") not in ("", "0")) else None
        self._pred_ev = None
        self._late_ev = None
        self._late_losses = os.environ.get("FCL_KD_LATE_LOSSES", "1") not in ("", "0")
        self._dw_keep = []
        self._plane_cache = {}
        # operand forms of the parameters (packed taps, transposes, column blocks, bias sums, their P32 planes): one batched launch per update
        self._forms = ops.DerivedForms(self.dev)
        self._recipe = {}  # data_ptr of a derived fp32 form -> (key, src, geom, base, src2): lets _wplanes / _wt derive further forms from the source
        self._param_ptr = {v.data_ptr(): k for k, v in self.P.items()}
        nbt = [k for k in self.B if k.endswith("num_batches_tracked")]
        self._nbt_flat = torch.zeros(len(nbt), dtype=torch.int64, device=self.dev) if nbt else None  # one add_ per forward instead of one per layer
        for i, k in enumerate(nbt):
            self._nbt_flat[i] = self.B[k].to(self.dev)
            self.B[k].data = self._nbt_flat[i]
        try:
            model._engine = self  # load_state_dict() on the module reaches invalidate_planes() through this back-reference
        except Exception:  # pragma: no cover - a module that refuses attributes only loses the cache invalidation hook
            pass
        # the native form of the step (csrc/train_engine.hip), created on first use; None + a reason when this configuration is not covered
        self._native, self.native_reason = None, _NativeStep.unsupported(self) if native else "native=False"

    @property
    def native(self):
        """The native step (fcl_te_*), or None: mode="train" passes without injected masks go through it."""
        if self.native_reason is not None:
            return None
        if self._native is None:
            self._native = _NativeStep(self)
        return self._native

    def _native_nbt(self):
        """torch's integer BatchNorm bookkeeping (num_batches_tracked) after a native train-form forward: every layer advanced once."""
        if self._nbt_flat is not None:
            self._nbt_flat.add_(1)

    @property
    def step_count(self):
        """Updates actually applied (skipped NaN / failed steps do not count: tts.py:173-179).  Synchronising read of the device counter."""
        return int(self.step_dev.item())

    @step_count.setter
    def step_count(self, v):
        self.step_dev.fill_(int(v))

    def _wplanes(self, key, w, cache=True):
        """P32 planes of a weight matrix (ops.pack_planes).  cache: `w` is a function of the PARAMETERS only, so its planes stay valid until
        the next optimizer step (the frozen KD teacher packs once, a model in training once per update; load_state_dict clears the cache).
        Matrices that also depend on buffers (eval-mode BatchNorm folds) pass cache=False.  None when the pre-split path is off."""
        if not ops.planes_enabled():
            return None
        rec = self._recipe_of(w) if cache else None
        if rec is not None:  # a parameter or a form derived from one: its planes come from the source in the batched per-update launch
            return self._forms.get(self._stamp(), ("p",) + rec[0], rec[1], rec[2], rec[3], rec[4], f32=False, planes=True)[1]
        # validity stamp: the engine's own optimizer steps (raw-pointer kernels) + torch's version counter of the flat buffer (in-place writes to
        # pflat / the P[...] views: finite-difference tests, checkpoint loads through the engine).  Writes through the MODULE's parameters
        # (p.data = view: torch optimizers, p.data.copy_()) do NOT bump that counter -- Tacotron2Base._forward_train and load_state_dict call
        # invalidate_planes() instead
        stamp = (self.update_calls, self.pflat._version, tuple(w.shape))
        hit = self._plane_cache.get(key) if cache else None
        if hit is not None and hit[0] == stamp:
            return hit[1]
        pl = ops.pack_planes(w.reshape(w.shape[0], -1) if w.dim() != 2 else w)
        if cache:
            self._plane_cache[key] = (stamp, pl)
        return pl

    def invalidate_planes(self):
        """Parameters were overwritten in place (load_state_dict): cached weight planes are stale."""
        self._plane_cache.clear()
        self._forms.stamp = None
        if self._native is not None:
            self._native.params_changed()

    def _stamp(self):
        return (self.update_calls, self.pflat._version)

    def _recipe_of(self, w):
        """(key, src, geom, base, src2) when `w` is a whole parameter seen as a 2-D matrix, or a derived form of one; else None."""
        ptr = w.data_ptr()
        rec = self._recipe.get(ptr)
        if rec is not None:
            a, b, c = rec[2][:3]
            live = self._forms.forms.get(("f",) + rec[0])  # the address must still belong to that form (a freed form's address can be reused)
            if live is None or live["out"] is None or live["out"].data_ptr() != ptr:
                del self._recipe[ptr]
                return None
            return rec if w.numel() == a * b * c and w.shape[-1] == c else None
        name = self._param_ptr.get(ptr)
        if name is not None and w.dim() == 2 and w.is_contiguous() and w.numel() == self.P[name].numel():
            rows, cols = w.shape
            return ((name, "w", rows), self.P[name], (1, rows, cols, 0, cols, 1), 0, None)
        return None

    def _form(self, key, src, geom, base=0, src2=None):
        """fp32 form [a*b, c] of a parameter (ops.DerivedForms / fcl_derive_batch), refreshed with all the others once per update."""
        out = self._forms.get(self._stamp(), ("f",) + key, src, geom, base, src2, f32=True, planes=False)[0]
        self._recipe[out.data_ptr()] = (key, src, geom, base, src2)
        return out

    def _cols(self, w, col0, n):
        """Contiguous copy of the parameter block w[:, col0:col0+n]."""
        rows, ld = w.shape
        return self._form((self._param_ptr[w.data_ptr()], "c", col0, n), w, (1, rows, n, 0, ld, 1), base=col0)

    def _zeros_const(self, name, shape):
        """A zero tensor that lives as long as the engine (the stand-in of a parameter block an option removed)."""
        z = self._zero_consts.get(name) if hasattr(self, "_zero_consts") else None
        if z is None:
            if not hasattr(self, "_zero_consts"):
                self._zero_consts = {}
            z = self._zero_consts[name] = torch.zeros(*shape, device=self.dev, dtype=torch.float32)
        return z

    def _bsum(self, b1, b2):
        """bias_ih + bias_hh."""
        return self._form((self._param_ptr[b1.data_ptr()], "bs"), b1, (1, 1, b1.numel(), 0, 0, 1), src2=b2).reshape(-1)

    def _dw(self, fn):
        """Run a weight-gradient closure on the side stream, after everything enqueued on the main stream so far.  The closure (and through it
        every tensor it reads) is kept alive until the streams are joined."""
        if self.side is None:
            fn()
            return
        self.side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.side):
            fn()
        self._dw_keep.append(fn)

    def _dw_gemm(self, dz, pairs, taps=None):
        """Weight gradients out += dz^T x for every (x, out) in pairs (dz [m, n], x [m, k], out [n, k]); taps = (ksz, seg_lo, seg_hi): one x and a
        tap-major out [ksz, n, k] (Conv1d).  Large outputs run on transposed planes (fcl_pack_planes_t + fcl_gemm_tn_planes: the LDS-DMA GEMM,
        2-3x the rate of the fp32-operand kernel; the transposing pass over dz is shared by the pairs), small ones on fcl_gemm_tn_fwd, whose
        per-launch cost is lower than the two extra passes."""
        m, n = dz.shape
        outs = sum(out.numel() for _, out in pairs)
        if ops.planes_enabled() and outs >= self._dw_planes_min and all(x.shape[1] % 4 == 0 for x, _ in pairs) and m >= 512:
            ap = ops.pack_planes_t(dz)
            for x, out in pairs:
                if taps is None:
                    ops.gemm_tn_planes(ap, ops.pack_planes_t(x), out, m)
                else:
                    ops.gemm_tn_planes(ap, ops.pack_planes_t(x, ntaps=taps[0], shift0=-((taps[0] - 1) // 2), seg_lo=taps[1], seg_hi=taps[2]), out, m)
            return
        for x, out in pairs:
            if taps is None:
                ops.gemm_tn(dz, x, out)
            else:
                ops.gemm_tn_taps(dz, x, out, -((taps[0] - 1) // 2), seg_lo=taps[1], seg_hi=taps[2])  # all taps in one launch

    def _join_dw(self):
        if self.side is not None and self._dw_keep:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            self._dw_keep.clear()  # main-stream reuse of these buffers is ordered after the join

    def _launch_bucket(self, c, i):
        """Bucket i's gradients are final once (a) the main stream has reached this point (BatchNorm / LayerNorm / bias gradients are written
        there) and (b) the weight-gradient stream has run the closures enqueued so far.  The collective is therefore issued FROM the
        weight-gradient stream after it has waited for the main stream's position -- a wait every later _dw() would make anyway -- and RCCL's
        stream orders itself behind that: the main stream, i.e. the backward's critical path, never waits for weight gradients here (until round 4
        it joined the side stream at every bucket, which since round 3 also carries the predictors' and the late KD terms' backward)."""
        if not c.reduce:  # a micro-batch that is not the last of its accumulation (or the autograd path): gradients stay local
            return
        if not self.buckets.active:
            return
        if self.side is None:
            self.buckets.launch(i)
            return
        self.side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.side):
            self.buckets.launch(i)
        self._dw_keep.append(None)  # the end-of-backward join must happen even if no weight-gradient closure followed this bucket

    def param_offsets(self):
        """{parameter name: (offset, numel, shape)} into pflat / gflat / mflat / vflat (checkpoint writers)."""
        return dict(self._offsets)

    # ------------------------------------------------------------------------------------------------ randomness
    def _keep(self, c, name, shape, p_one):
        """uint8 mask [shape] with P(1) = p_one: injected (c.masks) or drawn on the device."""
        if c.masks is not None:
            m = c.masks
            for part in name:
                m = m[part]
            return _u8(m, self.dev).reshape(shape)
        tag = zlib.crc32(repr(name).encode()) & 0x7FFFFFFF
        return ops.bernoulli_u8(shape, p_one, self.seed * 7919 + c.draw * 104729 + tag, self.dev)

    def _keeps(self, c, sites):
        """[(name, shape, p_one)] -> masks: what _keep would return for each site (same seeds, same bytes), drawn in ONE launch per group of sites
        (ops.bernoulli_batch) instead of one each: a train-mode forward has ~22 sites."""
        if c.masks is not None:
            return [self._keep(c, name, shape, p) for name, shape, p in sites]
        seeds = [self.seed * 7919 + c.draw * 104729 + (zlib.crc32(repr(name).encode()) & 0x7FFFFFFF) for name, _, _ in sites]
        return ops.bernoulli_batch([(shape, p, sd) for (_, shape, p), sd in zip(sites, seeds)], self.dev)

    def device_masks(self, batch, draw=None):
        """Every Bernoulli draw of the train-mode forward with ordinal `draw` (default: the last one) as host {0,1} arrays in the dictionary layout
        of injected masks (`masks=` here, oracle.masks_from_sequence): the device draws are a pure function of (engine seed, forward ordinal, site
        tag), so what the native / per-launch step used can be replayed into an independent implementation (tests: the native step vs the oracle)."""
        hp, dev = self.hp, self.dev
        c = _Ctx()
        c.masks, c.draw = None, self.forward_count if draw is None else int(draw)
        self._maps(c, batch)
        B, T, L, N, F, lmax = c.B, c.T, c.L, c.N, c.F, c.lmax
        C = hp.adim
        host = lambda t: t.cpu().numpy()
        bt = lambda sites: [host(m).reshape(B, -1, m.shape[-1]) for m in self._keeps(c, sites)]
        out = {"enc.convs": None, "postnet": None, "prenet": None}
        with torch.cuda.device(dev):
            if hp.dropout_rate > 0:
                out["enc.convs"] = bt([(("enc.convs", i), (B * T, hp.econv_chans), 1.0 - hp.dropout_rate) for i in range(hp.econv_layers)])
                n_post = hp.postnet_layers
                out["postnet"] = bt([(("postnet", i), (B * L, hp.odim if i == n_post - 1 else hp.postnet_chans), 1.0 - hp.dropout_rate) for i in range(n_post)])
            for nm, layers, chans, pd_ in (("duration_predictor", hp.duration_predictor_layers, hp.duration_predictor_chans, hp.duration_predictor_dropout_rate),
                                           ("pitch_predictor", hp.variance_predictor_layers, hp.variance_predictor_chans, hp.variance_predictor_dropout_rate),
                                           ("energy_predictor", hp.variance_predictor_layers, hp.variance_predictor_chans, hp.variance_predictor_dropout_rate)):
                out[nm] = bt([((nm, i), (B * T, chans), 1.0 - pd_) for i in range(layers)]) if pd_ > 0 else [np.ones((B, T, chans), np.uint8)] * layers
            p_emb = hp.variance_embed_dropout_rate
            for nm in ("pitch_embed", "energy_embed"):
                out[nm] = bt([((nm,), (B * T, C), 1.0 - p_emb)])[0] if p_emb > 0 else np.ones((B, T, C), np.uint8)

            def rows(cells):  # step-major cells [F, X] of the duration-sorted rows -> [lmax, N (the converter's row order), X]; cells no row reaches: 1
                cells = host(cells)
                full = np.ones((lmax, N, cells.shape[-1]), np.uint8)
                for t in range(lmax):
                    full[t, c.order[: c.live[t]]] = cells[c.offs[t] : c.offs[t + 1]]
                return full

            if hp.dropout_rate > 0:
                k0, k1 = self._keeps(c, [(("prenet", l), (F, hp.prenet_units), 1.0 - hp.dropout_rate) for l in range(2)])
                out["prenet"] = np.stack([rows(k0), rows(k1)], 1)
            zr = float(hp.zoneout_rate)
            if zr > 0:
                zk = self._keeps(c, [(("zoneout", l, j), (F, hp.dunits), zr) for l in range(2) for j in range(2)])
                out["zoneout"] = np.stack([np.stack([rows(zk[2 * l + j]) for j in range(2)], 1) for l in range(2)], 1)
        return out

    # ------------------------------------------------------------------------------------------------ layers
    def _wt(self, w):
        rec = self._recipe_of(w)
        if rec is not None and rec[2][0] == 1 and rec[4] is None:  # transpose of a parameter (block): swap the roles of b and c
            key, src, (a, b, c, sa, sb, sc), base, _ = rec
            return self._form(key + ("t",), src, (1, c, b, 0, sc, sb), base=base)
        return ops.transpose2d(w.contiguous())

    def _conv_pack(self, w, scale=None, need_t=True):
        name = self._param_ptr.get(w.data_ptr())
        if scale is None and name is not None:  # a function of the parameter alone: part of the batched per-update launch
            cout, cin, k = w.shape
            wp = self._form((name, "cp"), w, (k, cout, cin, 1, cin * k, k)).view(k, cout, cin)
            wt = self._form((name, "ct"), w, (k, cin, cout, -1, k, cin * k), base=k - 1).view(k, cin, cout) if need_t else None
            return wp, wt
        wp = ops.pack_conv1d_weight(w, scale)  # [k, Cout, Cin]
        if not need_t:  # forward only (the frozen teacher): the transposed taps are the backward's operand
            return wp, None
        k = wp.shape[0]
        wt = torch.stack([ops.transpose2d(wp[k - 1 - j]) for j in range(k)]).contiguous()  # [k, Cin, Cout], taps reversed
        return wp, wt

    def _conv_bn_fwd(self, c, x, prefix, lo, hi, act, keep, xp=None, want_planes=False):
        """Conv1d(no bias) -> BatchNorm -> act -> Dropout.  Returns (block output, cache[, P32 planes of the output]).
        xp: P32 planes of x -> the convolution runs on the pre-split-operand kernels (its weight planes are cached per update)."""
        P, B = self.P, self.B
        ks = 1.0 / (1.0 - self.hp.dropout_rate) if keep is not None else 1.0
        cout, cin, k = P[prefix + ".0.weight"].shape
        use_p = xp is not None
        want_planes = want_planes and cout % 32 == 0

        def conv(wp, bias):
            if use_p:
                wpp = self._wplanes(prefix, wp.reshape(k * cout, cin), cache=c.train)  # eval mode folds the running statistics into the taps
                return ops.conv1d_planes(xp, _ConvP(wpp, bias, cout, cin, k), lo, hi, ops.ACT_NONE, want_f32=True, want_planes=False)[0]
            return ops.conv1d(x, wp, bias, lo, hi, ops.ACT_NONE)

        if not self.hp.use_batch_norm:  # `--use-batch-norm false`: Conv1d(no bias) -> act -> Dropout (encoder_sa.py:78-90, decoder_sa.py:219-232)
            wp, wt = self._conv_pack(P[prefix + ".0.weight"], need_t=c.save)
            z = conv(wp, None)
            y_act = ops.act_fwd(z, act) if act != ops.ACT_NONE else z
            yp = None
            if keep is not None or want_planes:
                r = ops.act_fwd(y_act, ops.ACT_NONE, keep, ks, want_planes=want_planes)
                y, yp = r if want_planes else (r, None)
            else:
                y = y_act
            cache = dict(x=x, z=z, y_act=y_act, wt=wt, scale=None, prefix=prefix, act=act, lo=lo, hi=hi, keep=keep, ks=ks, no_bn=True)
            return (y, cache, yp) if want_planes else (y, cache)
        if c.train:
            wp, wt = self._conv_pack(P[prefix + ".0.weight"], need_t=c.save)
            z = conv(wp, None)
            mean, invstd = ops.bn_stats(z, BN_EPS, BN_MOMENTUM, B[prefix + ".1.running_mean"], B[prefix + ".1.running_var"])
            if prefix + ".1.num_batches_tracked" in B:
                c.bn_run.append(prefix + ".1.num_batches_tracked")  # torch's integer bookkeeping buffer: all layers advance in one add_ (_forward)
            res = ops.bn_act(z, mean, invstd, P[prefix + ".1.weight"], P[prefix + ".1.bias"], act, keep, ks, want_planes=want_planes)
            y_act, y = res[0], res[1]
            cache = dict(x=x, z=z, y_act=y_act, wt=wt, mean=mean, invstd=invstd, prefix=prefix, act=act, lo=lo, hi=hi, keep=keep, ks=ks)
            return (y, cache, res[2]) if want_planes else (y, cache)
        scale, shift = ops.fold_batchnorm(P[prefix + ".1.weight"], P[prefix + ".1.bias"], B[prefix + ".1.running_mean"], B[prefix + ".1.running_var"], BN_EPS)
        wp, wt = self._conv_pack(P[prefix + ".0.weight"], scale, need_t=c.save)
        z = conv(wp, shift)
        yp = None
        if act != ops.ACT_NONE or want_planes:
            r = ops.act_fwd(z, act, want_planes=want_planes)
            y, yp = r if want_planes else (r, None)
        else:
            y = z
        cache = dict(x=x, z=z, y_act=y, wt=wt, scale=scale, prefix=prefix, act=act, lo=lo, hi=hi, keep=None, ks=1.0)
        return (y, cache, yp) if want_planes else (y, cache)

    def _conv_dx(self, dz, dzp, wt, key, lo, hi):
        """Input gradient of a Conv1d = the forward conv of dz with taps reversed and W_j transposed (wt [k, Cin, Cout]); on pre-split operands
        when the producer of dz also wrote its planes."""
        if dzp is not None:
            k, cin, cout = wt.shape
            wpp = self._wplanes(key + ".t", wt.reshape(k * cin, cout))
            return ops.conv1d_planes(dzp, _ConvP(wpp, None, cin, cout, k), lo, hi, ops.ACT_NONE, want_f32=True, want_planes=False)[0]
        return ops.conv1d(dz, wt, None, lo, hi)

    def _conv_bn_bwd(self, c, dy, cc):
        G, P = self.G, self.P
        pre = cc["prefix"]
        cout, cin, k = P[pre + ".0.weight"].shape
        pl = ops.planes_enabled() and cout % 32 == 0
        dzp = None
        if cc["act"] != ops.ACT_NONE or cc["keep"] is not None:
            r = ops.act_bwd(dy, cc["y_act"], cc["act"], cc["keep"], cc["ks"], want_planes=pl and not c.train)
            dz, dzp = r if (pl and not c.train) else (r, None)
        else:
            dz = dy
        no_bn = cc.get("no_bn", False)
        if no_bn:
            scale = None
            if pl and dzp is None:
                dzp = ops.pack_planes(dz)
        elif c.train:
            dbeta, dgamma = self._z((cout,)), self._z((cout,))
            ops.colsum(dz, dgamma, y=cc["z"], gamma=cc["invstd"], beta=cc["mean"], mode=3, out_x=dbeta)  # both sums from one pass over dz
            r = ops.bn_bwd(dz, cc["z"], cc["mean"], cc["invstd"], P[pre + ".1.weight"], dbeta, dgamma, want_planes=pl,
                           acc=(G[pre + ".1.bias"], G[pre + ".1.weight"]))  # ... and G += (dbeta, dgamma) in the same launch: 5 -> 2 launches per layer
            dz, dzp = r if pl else (r, None)
            scale = None
        else:
            scale = cc["scale"]
            if pl and dzp is None:
                dzp = ops.pack_planes(dz)

        def dw(dz=dz, scale=scale, eval_affine=not c.train and not no_bn):
            if eval_affine:
                ops.colsum(dz, G[pre + ".1.weight"], y=cc["z"], gamma=P[pre + ".1.weight"], beta=P[pre + ".1.bias"], mode=2, out_x=G[pre + ".1.bias"])
            dwp = self._z((k, cout, cin))
            self._dw_gemm(dz, [(cc["x"], dwp)], taps=(k, cc["lo"], cc["hi"]))
            ops.unpack_conv1d_grad(dwp, G[pre + ".0.weight"], scale)

        self._dw(dw)
        # eval mode folds the running statistics into the taps: those transposed taps are not a function of the parameters alone (no caching)
        if dzp is not None and not c.train and not no_bn:
            kk, ci, co = cc["wt"].shape
            wpp = ops.pack_planes(cc["wt"].reshape(kk * ci, co))
            return ops.conv1d_planes(dzp, _ConvP(wpp, None, ci, co, kk), cc["lo"], cc["hi"], ops.ACT_NONE, want_f32=True, want_planes=False)[0]
        return self._conv_dx(dz, dzp, cc["wt"], pre, cc["lo"], cc["hi"])

    def _conv_bias_relu_fwd(self, x, prefix, lo, hi, need_t=True, xp=None):
        wp, wt = self._conv_pack(self.P[prefix + ".weight"], need_t=need_t)
        k, cout, cin = wp.shape
        if xp is not None:
            wpp = self._wplanes(prefix, wp.reshape(k * cout, cin))
            y = ops.conv1d_planes(xp, _ConvP(wpp, self.P[prefix + ".bias"], cout, cin, k), lo, hi, ops.ACT_RELU, want_f32=True, want_planes=False)[0]
        else:
            y = ops.conv1d(x, wp, self.P[prefix + ".bias"], lo, hi, ops.ACT_RELU)
        return y, dict(x=x, y=y, wt=wt, prefix=prefix, lo=lo, hi=hi)

    def _conv_bias_relu_bwd(self, dy, cc):
        G = self.G
        pre = cc["prefix"]
        cout, cin, k = self.P[pre + ".weight"].shape
        pl = ops.planes_enabled() and cout % 32 == 0
        r = ops.act_bwd(dy, cc["y"], ops.ACT_RELU, want_planes=pl)
        dz, dzp = r if pl else (r, None)

        def dw():
            ops.colsum(dz, G[pre + ".bias"])
            dwp = self._z((k, cout, cin))
            self._dw_gemm(dz, [(cc["x"], dwp)], taps=(k, cc["lo"], cc["hi"]))
            ops.unpack_conv1d_grad(dwp, G[pre + ".weight"])

        self._dw(dw)
        return self._conv_dx(dz, dzp, cc["wt"], pre, cc["lo"], cc["hi"])

    def _predictor_fwd(self, c, hs, name, layers, p_drop, lo, hi, pad, hs_p=None, keeps=None):
        caches, x, xp, out = [], hs, hs_p, None
        for i in range(layers):
            y, cc = self._conv_bias_relu_fwd(x, "%s.conv.%d.0" % (name, i), lo, hi, need_t=c.save, xp=xp)
            last = i == layers - 1
            if keeps is not None:
                keep = keeps[i]
            else:
                keep = self._keep(c, (name, i), tuple(y.shape), 1.0 - p_drop) if (c.train and p_drop > 0) else None
            ks = 1.0 / (1.0 - p_drop) if keep is not None else 1.0
            g, b = self.P["%s.conv.%d.2.weight" % (name, i)], self.P["%s.conv.%d.2.bias" % (name, i)]
            wantp = hs_p is not None and not last and y.shape[1] % 32 == 0
            res = ops.layernorm(y, g, b, LN_EPS, want_y=(not last) and (c.save or not wantp),
                                lin_w=self.P[name + ".linear.weight"].reshape(-1) if last else None,
                                lin_b=self.P[name + ".linear.bias"] if last else None, pad_mask=pad if last else None, keep=keep, keep_scale=ks,
                                want_planes=wantp)
            ln, out = res[0], res[1]
            xp = res[2] if wantp else None
            caches.append((cc, y, i, last, keep, ks))
            x = ln
        return out, caches

    def _pred_fork(self, c):
        """Context in which the predictors' launches go to their own stream, ordered after everything enqueued on the current stream so far.
        Steps with a backward only (c.save): a frozen KD teacher already runs beside the student's three streams, and a fifth active queue
        halves the device's throughput."""
        if self.pstream is None or not c.save:
            return contextlib.nullcontext()
        self.pstream.wait_stream(torch.cuda.current_stream(self.dev))
        return self._pred_ctx()

    @contextlib.contextmanager
    def _pred_ctx(self):
        with torch.cuda.stream(self.pstream):
            try:
                yield
            finally:  # also when the body raises: whatever it enqueued is joined, never left racing the next step's arena clear
                self._pred_ev = torch.cuda.Event()
                self._pred_ev.record(self.pstream)  # (the join waits for the predictors, not for weight gradients queued behind them)

    def _pred_join(self):
        if self._pred_ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._pred_ev)
            self._pred_ev = None

    @contextlib.contextmanager
    def _late_fork(self, c):
        """The KD terms whose gradients the backward needs LATE (encoder taps: at its very end; prenet / LSTM taps: after the postnet's backward) run
        on the weight-gradient stream while the main stream computes the frame-level terms and starts the postnet's backward (FCL_KD_LATE_LOSSES=0:
        in line)."""
        if self.pstream is None or not c.save or not self._late_losses:
            yield
            return
        self.pstream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.pstream):
            try:
                yield
            finally:
                self._late_ev = torch.cuda.Event()
                self._late_ev.record(self.pstream)

    def _late_join(self):
        if self._late_ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._late_ev)
            self._late_ev = None

    def _predictor_bwd(self, d_out, name, caches, pad):
        G, P = self.G, self.P
        dx = None
        for cc, y, i, last, keep, ks in reversed(caches):
            g, b = P["%s.conv.%d.2.weight" % (name, i)], P["%s.conv.%d.2.bias" % (name, i)]
            kw = dict(dgamma=G["%s.conv.%d.2.weight" % (name, i)], dbeta=G["%s.conv.%d.2.bias" % (name, i)], keep=keep, keep_scale=ks)
            if last:
                dy = ops.layernorm_bwd(y, g, b, LN_EPS, lin_w=P[name + ".linear.weight"].reshape(-1), ds=d_out, pad_mask=pad,
                                       dlin_w=G[name + ".linear.weight"].reshape(-1), dlin_b=G[name + ".linear.bias"], **kw)
            else:
                dy = ops.layernorm_bwd(y, g, b, LN_EPS, dy=dx, **kw)
            dx = self._conv_bias_relu_bwd(dy, cc)
        return dx

    # ------------------------------------------------------------------------------------------------ BiLSTM (per-step, saved)
    def _bilstm_fwd(self, x, lens_dev, B, T, save=True, perm=None, xp=None, layer=0):
        """One bidirectional layer (`elayers` > 1: the caller stacks them, encoder_sa.py:96-100).  Returns (out, cache, P32 planes of out or None).
        xp: P32 planes of x (the input projections then run on the pre-split kernels)."""
        P, dev = self.P, self.dev
        H = self.hp.eunits // 2
        L = "_l%d" % layer
        use_p = xp is not None and x.shape[1] % 32 == 0 and (2 * H) % 32 == 0
        wip = [self._wplanes("enc.blstm.weight_ih" + L + sfx, P["enc.blstm.weight_ih" + L + sfx]) for sfx in ("", "_reverse")] if use_p else None
        if not save:  # forward only (the frozen KD teacher): the persistent register-resident kernel of the synthesis path
            g = lambda k: P["enc.blstm." + k.replace("_l0", L)]
            r = ops.bilstm(x, lens_dev, g("weight_ih_l0"), g("weight_hh_l0"), self._bsum(g("bias_ih_l0"), g("bias_hh_l0")),
                           g("weight_ih_l0_reverse"), g("weight_hh_l0_reverse"), self._bsum(g("bias_ih_l0_reverse"), g("bias_hh_l0_reverse")),
                           B, T, algo=3 if H == 256 else 0, status=self.status,  # one group kernel at a time (this engine's stream)
                           x_p=xp if use_p else None, w_ih_p=wip, want_planes=use_p)
            return (r[0], None, r[1]) if use_p else (r, None, None)
        out = torch.empty(B * T, 2 * H, device=dev)
        gx, whh, sv = [], [], []
        for d, sfx in enumerate(("", "_reverse")):
            bias = self._bsum(P["enc.blstm.bias_ih" + L + sfx], P["enc.blstm.bias_hh" + L + sfx])
            if use_p:
                gx.append(ops.linear_planes(xp, wip[d], 4 * H, x.shape[1], bias)[0])
            else:
                gx.append(ops.linear(x, P["enc.blstm.weight_ih" + L + sfx], bias))  # [B*T, 4H]
            whh.append(P["enc.blstm.weight_hh" + L + sfx])
            # gates, c_new, c_old, h_old (t-major); zero-filled: dead cells are never written but are read by the batched weight-gradient GEMM
            sv.append([self._z((T, B, 4 * H))] + [self._z((T, B, H)) for _ in range(3)])
        ops.bilstm_train_fwd(gx, whh, lens_dev, B, T, out, sv, status=self.status)
        return out, dict(x=x, dirs=sv, B=B, T=T, lens=lens_dev, perm=perm, layer=layer), (ops.pack_planes(out) if use_p else None)

    def _bilstm_bwd(self, d_out, c):
        P, G, dev = self.P, self.G, self.dev
        B, T, H = c["B"], c["T"], self.hp.eunits // 2
        L = "_l%d" % c.get("layer", 0)
        dx = self._z(c["x"].shape)
        perm = c["perm"]  # (b, t) row -> t-major row of the saved / gradient tensors
        sfxs = ("", "_reverse")
        dgs = [torch.empty(T, B, 4 * H, device=dev) for _ in sfxs]  # d_out is already zero on padded rows (masked by the caller); dead cells get dg = 0
        ops.bilstm_bptt(c["dirs"], c["lens"], B, T, d_out, [self._wt(P["enc.blstm.weight_hh" + L + sfx]) for sfx in sfxs], dgs, status=self.status)
        for d, sfx in enumerate(sfxs):
            dg2 = dgs[d].reshape(T * B, 4 * H)
            dgx = ops.gather_rows(dg2, perm)  # back to (b, t) rows like x

            def dw(dg2=dg2, dgx=dgx, d=d, sfx=sfx):
                ops.gemm_tn(dg2, c["dirs"][d][3].reshape(T * B, H), G["enc.blstm.weight_hh" + L + sfx])  # one TN GEMM over every (t, b) cell
                ops.gemm_tn(dgx, c["x"], G["enc.blstm.weight_ih" + L + sfx])
                # bias_ih and bias_hh enter the gates as a sum: the same column sums, straight into both accumulators (one launch)
                ops.colsum(dgx, G["enc.blstm.bias_ih" + L + sfx], out_x=G["enc.blstm.bias_hh" + L + sfx])

            self._dw(dw)
            ops.add2d(dx, ops.linear(dgx, self._wt(P["enc.blstm.weight_ih" + L + sfx])))
        return dx

    # ------------------------------------------------------------------------------------------------ index maps (host, integers)
    def _maps(self, c, batch):
        """Integer index maps of one batch: built on the host (build_maps_host: pure numpy, possibly already done by a loader thread), shipped to
        the device in three non-blocking copies from pinned memory and cached on the batch dict (the frozen teacher and the student share them)."""
        dev = self.dev
        cache = batch.get("_fcl_maps") if isinstance(batch, dict) else None
        if cache is not None and cache[0] == str(dev):
            c.__dict__.update(cache[1])
            return
        host = batch.get("_fcl_maps_host") if isinstance(batch, dict) else None
        if host is None:
            host = build_maps_host(batch, self.hp.reduction_factor)
        m = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in host["scalars"].items()}  # a DataLoader hands numpy arrays over as tensors
        up = _RING.upload({"i32": host["i32"], "u8": host["u8"], "pos4": host["pos4"]}, dev)  # pinned staging: the CPU keeps running ahead
        for name in ("i32", "u8"):
            for k, (o, n) in host[name + "_layout"].items():
                m[k] = up[name][o : o + n]
        m["cell_row_i64"] = m["cell_row_i32"].to(torch.int64)
        m["pos4"] = up["pos4"]
        m["cell_valid"] = torch.ones(m["F"], dtype=torch.uint8, device=dev)  # every cell is a valid frame
        if isinstance(batch, dict):
            batch["_fcl_maps"] = (str(dev), m)
        c.__dict__.update(m)

    def _cells(self, c, arr):
        """[lmax, N(compact order), X] injected decoder masks -> step-major cells [F, X]."""
        a = np.asarray(arr)[: c.lmax][:, c.order]
        return _u8(np.concatenate([a[t, : c.live[t]] for t in range(c.lmax)]), self.dev)

    # ------------------------------------------------------------------------------------------------ decoder cell stack, any depth
    def _decoder_cells_fwd(self, c, G0, w0_pos):
        """`dlayers` != 2 (decoder_sa.py:357-369, 500-504: a stack of ZoneOut LSTMCells, cell l > 0 on cell l - 1's NEW state): one fcl_lstm_step_fwd
        per (step, cell) over the step-major cells, saving what the reverse pass needs.  The two-cell recipe has its own C loop
        (fcl_decoder_train_fwd); this is the launch-by-launch form of the same step."""
        hp, P, dev = self.hp, self.P, self.dev
        U, Pn, DL, N = hp.dunits, hp.prenet_units, hp.dlayers, c.N
        lk = lambda l, n: lstm_key(hp, l, n)
        h = [[torch.zeros(N, U, device=dev), torch.zeros(N, U, device=dev)] for _ in range(DL)]
        cst = [torch.zeros(N, U, device=dev) for _ in range(DL)]
        bs = [None] + [self._bsum(P[lk(l, "bias_ih")], P[lk(l, "bias_hh")]) for l in range(1, DL)]
        cur = 0
        for t in range(c.lmax):
            n, o = int(c.live[t]), int(c.offs[t])
            for l in range(DL):
                kw = dict(step=t, zoneout=c.zr, out2=c.h_all[l][o:], out2_row_mul=1, ld2=U, save=[q[o:] for q in c.S[l]])
                if c.zk is not None:
                    kw.update(zone_keep_h=c.zk[l][0][o:], zone_keep_c=c.zk[l][1][o:])
                if l == 0:
                    terms = [(c.p1d[o : o + n], c.w0_pre, Pn), (h[0][cur], c.w0_hh, U)]
                    kw.update(G=G0, g_row_mul=1, rank1_w=w0_pos, dur=c.dur_dev)
                else:
                    terms = [(h[l - 1][cur ^ 1], P[lk(l, "weight_ih")], U), (h[l][cur], P[lk(l, "weight_hh")], U)]
                    kw.update(bias=bs[l])
                ops.lstm_step(terms, n, U, h[l][cur], h[l][cur ^ 1], cst[l], **kw)
            cur ^= 1

    def _decoder_cells_bptt(self, c, dh_last_all):
        """Reverse pass of _decoder_cells_fwd: gate gradients dg[l] [F, 4U] of every cell.  Per step, top cell first: the cell's backward from its
        saved activations (carry from step t + 1 + this step's output gradient), dh carry <- keep path + dg . W_hh, and dg . W_ih as the output
        gradient of the cell below at the same step."""
        hp, P, dev = self.hp, self.P, self.dev
        U, DL, N, F = hp.dunits, hp.dlayers, c.N, c.F
        lk = lambda l, n: lstm_key(hp, l, n)
        dg = [torch.empty(F, 4 * U, device=dev) for _ in range(DL)]
        dh = [torch.zeros(N, U, device=dev) for _ in range(DL)]  # rows that only become live at earlier steps must see zero carries
        dc = [torch.zeros(N, U, device=dev) for _ in range(DL)]
        whh_t = [self._wt(P[lk(l, "weight_hh")]) for l in range(DL)]
        wih_t = [None] + [self._wt(P[lk(l, "weight_ih")]) for l in range(1, DL)]
        for t in range(c.lmax - 1, -1, -1):
            n, o = int(c.live[t]), int(c.offs[t])
            below = dh_last_all[o : o + n]  # output gradient of the top cell at this step (feat_out, KD taps)
            for l in range(DL - 1, -1, -1):
                S = c.S[l]
                zk = (c.zk[l][0][o : o + n], c.zk[l][1][o : o + n]) if c.zk is not None else (None, None)
                dgl, dh_old, dc_old = ops.lstm_cell_bwd(S[0][o : o + n], S[2][o : o + n], S[1][o : o + n], dh[l][:n], dc[l][:n], c.zr, zk[0], zk[1],
                                                        out=(dg[l][o : o + n], torch.empty(n, U, device=dev), torch.empty(n, U, device=dev)), dh_out2=below)
                ops.add2d(dh_old, ops.linear(dgl, whh_t[l]))
                ops.copy2d(dh[l][:n], dh_old)
                ops.copy2d(dc[l][:n], dc_old)
                if l > 0:
                    below = ops.linear(dgl, wih_t[l])
        return dg

    # ------------------------------------------------------------------------------------------------ forward
    def _forward(self, c, batch):
        hp, dev, P = self.hp, self.dev, self.P
        c.bn_run = []
        B, T, L = c.B, c.T, c.L
        O, U, Pn, C = hp.odim, hp.dunits, hp.prenet_units, hp.adim  # C: width of the decoder / predictor input (eunits + spk_embed_dim)
        p_conv = hp.dropout_rate
        drop_conv = c.train and p_conv > 0
        # ---- encoder
        c.xs = batch["xs"][:, :T].to(dev).to(torch.int64).reshape(-1).contiguous()
        pl = ops.planes_enabled()  # forward GEMM operands travel pre-split (P32 planes) next to the fp32 tensors the backward pass keeps
        xp = None
        if pl:
            c.emb, xp = ops.embedding(c.xs, P["enc.embed.weight"], want_planes=True)
        else:
            c.emb = ops.embedding(c.xs, P["enc.embed.weight"])
        x, c.conv_c, c.enc_taps = c.emb, [], [c.emb]
        enc_keeps = (self._keeps(c, [(("enc.convs", i), (B * T, hp.econv_chans), 1.0 - p_conv) for i in range(hp.econv_layers)]) if drop_conv
                     else [None] * hp.econv_layers)
        for i in range(hp.econv_layers):
            keep = enc_keeps[i]
            r = self._conv_bn_fwd(c, x, "enc.convs.%d" % i, c.e_lo, c.e_hi, ops.ACT_RELU, keep, xp=xp, want_planes=pl and not hp.use_residual)
            y, cc, yp = r if len(r) == 3 else (r[0], r[1], None)
            if hp.use_residual:  # convs[i](xs) + xs, after the block's ReLU and Dropout (encoder_sa_kd.py:158-171); the taps are the sums
                y = ops.add_vec(y, x)
                yp = ops.pack_planes(y) if pl else None
            x, xp = y, yp
            c.conv_c.append(cc)
            c.enc_taps.append(x)
        c.hs, c.bl_c, hs_p = self._bilstm_fwd(x, c.lens_dev, B, T, save=c.save, perm=c.perm_tb, xp=xp)
        c.bl_stack = [c.bl_c]
        for l in range(1, hp.elayers):  # stacked layers (encoder_sa.py:96-100): layer l reads [forward | reverse] of layer l - 1
            c.hs, bl_l, hs_p = self._bilstm_fwd(c.hs, c.lens_dev, B, T, save=c.save, perm=c.perm_tb, xp=hs_p, layer=l)
            c.bl_stack.append(bl_l)
        c.hs_enc = c.hs  # the encoder's own output: the KD tap (encoder_sa_kd.py:178-188) and what the BiLSTM's backward receives
        if hp.spk_embed_dim is not None:  # hs <- cat[hs, F.normalize(spembs)] (..._sa.py:555-557); the embeddings are inputs: no gradient leaves here
            if batch.get("spembs") is None:
                raise ValueError("fcl-taco2_amd: the model was built with spk_embed_dim=%d: forward() needs spembs" % hp.spk_embed_dim)
            spk = batch["spembs"].to(dev).float().contiguous()
            c.hs, hs_p = ops.concat_spk(c.hs_enc, spk, T, want_planes=pl and C % 32 == 0)
        # ---- predictors + embeds (their dropout masks: one launch for all of them)
        p_emb = hp.variance_embed_dropout_rate
        pk = {}
        if c.train:
            sites = []
            for nm, layers, chans, pd_ in (("duration_predictor", hp.duration_predictor_layers, hp.duration_predictor_chans, hp.duration_predictor_dropout_rate),
                                           ("pitch_predictor", hp.variance_predictor_layers, hp.variance_predictor_chans, hp.variance_predictor_dropout_rate),
                                           ("energy_predictor", hp.variance_predictor_layers, hp.variance_predictor_chans, hp.variance_predictor_dropout_rate)):
                if pd_ > 0:
                    sites += [((nm, i), (B * T, chans), 1.0 - pd_) for i in range(layers)]
            if p_emb > 0:
                sites += [(("pitch_embed",), (B * T, C), 1.0 - p_emb), (("energy_embed",), (B * T, C), 1.0 - p_emb)]
            pk = dict(zip([st[0] for st in sites], self._keeps(c, sites)))
        pkeeps = lambda nm, layers: [pk.get((nm, i)) for i in range(layers)]
        with self._pred_fork(c):  # (joined at the end of this forward)
            c.d_outs, c.dur_c = self._predictor_fwd(c, c.hs, "duration_predictor", hp.duration_predictor_layers, hp.duration_predictor_dropout_rate,
                                                    c.e_lo, c.e_hi, c.enc_pad, hs_p=hs_p, keeps=pkeeps("duration_predictor", hp.duration_predictor_layers))
            c.p_outs, c.pit_c = self._predictor_fwd(c, c.hs, "pitch_predictor", hp.variance_predictor_layers, hp.variance_predictor_dropout_rate,
                                                    c.e_lo, c.e_hi, c.enc_pad, hs_p=hs_p, keeps=pkeeps("pitch_predictor", hp.variance_predictor_layers))
            c.e_outs, c.en_c = self._predictor_fwd(c, c.hs, "energy_predictor", hp.variance_predictor_layers, hp.variance_predictor_dropout_rate,
                                                   c.e_lo, c.e_hi, c.enc_pad, hs_p=hs_p, keeps=pkeeps("energy_predictor", hp.variance_predictor_layers))
        c.f0 = batch["f0"][:, :T].to(dev).float().reshape(-1).contiguous()
        c.en = batch["energy"][:, :T].to(dev).float().reshape(-1).contiguous()
        c.ds = batch["extras"][:, :T].to(dev).float().reshape(-1).contiguous()
        kk = hp.variance_embed_kernel_size
        att, pe, ee = ops.variance_embed_add(c.hs, c.f0, c.en, P["pitch_embed.0.weight"].reshape(C, kk), P["pitch_embed.0.bias"],
                                             P["energy_embed.0.weight"].reshape(C, kk), P["energy_embed.0.bias"], c.e_lo, c.e_hi, want_embs=True)
        c.emb_keep, c.emb_ks = (None, None), 1.0
        if c.train and p_emb > 0:
            c.emb_keep = (pk[("pitch_embed",)], pk[("energy_embed",)])
            c.emb_ks = 1.0 / (1.0 - p_emb)
            pe, ee = ops.act_fwd(pe, ops.ACT_NONE, c.emb_keep[0], c.emb_ks), ops.act_fwd(ee, ops.ACT_NONE, c.emb_keep[1], c.emb_ks)
            att = ops.add2d(ops.add2d(c.hs.clone(), pe), ee)
        c.p_embs, c.e_embs = pe, ee
        # ---- decoder, teacher forced, step-major cells
        N, F, lmax, live, offs = c.N, c.F, c.lmax, c.live, c.offs
        dpl = pl and all(v % 32 == 0 for v in (C, Pn, U))  # decoder-side GEMMs on pre-split operands (whole 32-column lines)
        att_p = pre_in_p = None
        if dpl:
            c.att_c, att_p = ops.gather_rows(att, c.src_sorted, want_planes=True)  # [N, C] sorted rows
        else:
            c.att_c = ops.gather_rows(att, c.src_sorted)
        c.ys = batch["ys"][:, :L].to(dev).float().reshape(B * L, O).contiguous()
        if dpl:
            c.pre_in, pre_in_p = ops.gather_rows(c.ys, c.prev_frame, want_planes=True)  # [F, O]; idx -1 -> zero row
        else:
            c.pre_in = ops.gather_rows(c.ys, c.prev_frame)
        PL, DL = hp.prenet_layers, hp.dlayers  # (2, 2) in every shipped recipe; other counts: decoder_sa.py:119-158, 357-369 (G18 / G19)
        c.pre_keep = [None] * PL
        c.pks = 1.0
        if hp.dropout_rate > 0:  # the prenet's dropout is on in BOTH modes (decoder_sa.py:156-158)
            c.pks = 1.0 / (1.0 - hp.dropout_rate)
            if c.masks is not None:
                pk = np.asarray(c.masks["prenet"])
                c.pre_keep = [self._cells(c, pk[:, l]) for l in range(PL)]
            else:  # (with the zoneout masks of a train-mode pass: one launch for the decoder's mask tensors)
                sites = [(("prenet", l), (F, Pn), 1.0 - hp.dropout_rate) for l in range(PL)]
                if c.train and float(hp.zoneout_rate) > 0:
                    sites += [(("zoneout", l, j), (F, U), float(hp.zoneout_rate)) for l in range(DL) for j in range(2)]
                dk = self._keeps(c, sites)
                c.pre_keep = list(dk[:PL])
                c.zk_pre = [[dk[PL + 2 * l + j] for j in range(2)] for l in range(DL)] if len(dk) == PL + 2 * DL else None
        # prenet: PL x {Linear -> ReLU -> dropout}; the pre-dropout activations are kept for the backward
        c.pre_act, c.pre_drop = [], []
        x, x_p, kin = c.pre_in, pre_in_p, O
        for l in range(PL):
            wn, bn = "dec.prenet.prenet.%d.0.weight" % l, "dec.prenet.prenet.%d.0.bias" % l
            if dpl:
                a = ops.linear_planes(x_p, self._wplanes(wn, P[wn]), Pn, kin, P[bn], ops.ACT_RELU)[0]
                d, d_p = ops.act_fwd(a, ops.ACT_NONE, c.pre_keep[l], c.pks, want_planes=True)
            else:
                a = ops.linear(x, P[wn], P[bn], ops.ACT_RELU)
                d, d_p = (ops.act_fwd(a, ops.ACT_NONE, c.pre_keep[l], c.pks) if c.pre_keep[l] is not None else a), None
            c.pre_act.append(a)
            c.pre_drop.append(d)
            x, x_p, kin = d, d_p, Pn
        c.p1d, p1d_p = x, x_p  # the prenet's output (what LSTM 0 reads; the KD tap)
        lk = lambda l, n: lstm_key(hp, l, n)
        w_ih0 = P[lk(0, "weight_ih")]
        c.w0_att, c.w0_pre = self._cols(w_ih0, 0, C), self._cols(w_ih0, C, Pn)
        # options off = the term does not exist in the reference (decoder_sa.py:361-365, :397): a zero block keeps the kernels on one shape
        w0_pos = self._cols(w_ih0, C + Pn, 1).reshape(-1) if hp.append_position else self._zeros_const("w0_pos0", (4 * U,))
        c.w0_hh = P[lk(0, "weight_hh")]
        b0s = self._bsum(P[lk(0, "bias_ih")], P[lk(0, "bias_hh")])
        wf = P["dec.feat_out.weight"]
        R = hp.reduction_factor
        if R == 1:
            c.wf_h = self._cols(wf, 0, U)
            c.wf_att = self._cols(wf, U, C) if hp.use_concate else self._zeros_const("wf_att0", (O, C))
        else:  # a step emits R frames (decoder_sa.py:397-398, 512): rows re-ordered frame-major (row j * odim + o <- feat_out row o * R + j) so that a
               # cell's output [R * odim] IS its R consecutive frames of the frame-major buffers
            ldf = wf.shape[1]
            pcols = lambda col0, n: self._form((self._param_ptr[wf.data_ptr()], "cp", col0, n, R), wf, (R, O, n, ldf, R * ldf, 1), base=col0)
            c.wf_h = pcols(0, U)
            c.wf_att = pcols(U, C) if hp.use_concate else self._zeros_const("wf_att0r", (O * R, C))
        OR = O * R
        dpl = dpl and R == 1  # (r > 1: the decoder-side GEMMs stay on the fp32 operands)
        if dpl:
            G0 = ops.linear_planes(att_p, self._wplanes("w0_att", c.w0_att), 4 * U, C, b0s)[0]  # hoisted att_c share of the layer-0 gates
            F0 = ops.linear_planes(att_p, self._wplanes("wf_att", c.wf_att), O, C)[0]
        else:
            G0 = ops.linear(c.att_c, c.w0_att, b0s)
            F0 = ops.linear(c.att_c, c.wf_att)
        c.zr = float(hp.zoneout_rate)
        c.zk = None
        if c.train and c.zr > 0:  # sampled zoneout: mask = 1 keeps the OLD state, P(1) = rate; [layer][h, c] -> [F, U]
            if c.masks is not None:
                zm = np.asarray(c.masks["zoneout"])
                c.zk = [[self._cells(c, zm[:, l, j]) for j in range(2)] for l in range(DL)]
            elif getattr(c, "zk_pre", None) is not None:
                c.zk = c.zk_pre
            else:
                c.zk = [[dk_ for dk_ in self._keeps(c, [(("zoneout", l, j), (F, U), c.zr) for j in range(2)])] for l in range(DL)]
        # per cell: saved (gates, c_new, c_old, h_old) and the zoneout-ed output of every cell
        c.S = [[torch.empty(F, 4 * U, device=dev)] + [torch.empty(F, U, device=dev) for _ in range(3)] for _ in range(DL)]
        c.h_all = [torch.empty(F, U, device=dev) for _ in range(DL)]
        c.S0, c.h0_all = c.S[0], c.h_all[0]
        c.S1, c.h1_all = c.S[-1], c.h_all[-1]  # (the LAST cell: what feat_out reads)
        if DL == 2:
            c.w1_ih, c.w1_hh = P[lk(1, "weight_ih")], P[lk(1, "weight_hh")]
            b1s = self._bsum(P[lk(1, "bias_ih")], P[lk(1, "bias_hh")])
            dec_planes = None
            if dpl and (N * U * 4) % 128 == 0:  # big steps on the LDS-DMA kernels (pre-split operands)
                dec_planes = (p1d_p, self._wplanes("w0_pre", c.w0_pre), self._wplanes("w0_hh", c.w0_hh),
                              self._wplanes("w1_ih", c.w1_ih), self._wplanes("w1_hh", c.w1_hh))
            ops.decoder_train_fwd(c.live_i32, c.p1d, G0, c.w0_pre, c.w0_hh, w0_pos, c.dur_dev, c.w1_ih, c.w1_hh, b1s, c.zr, c.zk, c.S0, c.S1, c.h0_all,
                                  c.h1_all, planes=dec_planes)
        else:
            self._decoder_cells_fwd(c, G0, w0_pos)
        if dpl:
            out_cells = ops.linear_planes(ops.pack_planes(c.h1_all), self._wplanes("wf_h", c.wf_h), O, U)[0]
        else:
            out_cells = ops.linear(c.h1_all, c.wf_h)  # [F, R * odim]: the cell's R frames, frame-major
        ops.add2d(out_cells, ops.gather_rows(F0, c.cell_row_i32))
        xp = None
        if pl and R == 1:
            c.before, xp = ops.gather_rows(out_cells, c.frame_cell, want_planes=True)  # [B*L, O], zero where no cell maps (padding)
        else:  # (r > 1: frame_cell maps GROUPS of R frames to cells; [B * L / R, R * odim] viewed as [B * L, odim])
            c.before = ops.gather_rows(out_cells, c.frame_cell).reshape(B * L, O)
            xp = ops.pack_planes(c.before) if pl else None
        # ---- postnet
        x, c.post_c, c.post_taps = c.before, [], []
        n_post = hp.postnet_layers
        post_keeps = (self._keeps(c, [(("postnet", i), (B * L, hp.odim if i == n_post - 1 else hp.postnet_chans), 1.0 - p_conv) for i in range(n_post)])
                      if drop_conv else [None] * n_post)
        for i in range(n_post):
            cout = hp.odim if i == n_post - 1 else hp.postnet_chans
            keep = post_keeps[i]
            r = self._conv_bn_fwd(c, x, "dec.postnet.postnet.%d" % i, c.f_lo, c.f_hi, ops.ACT_NONE if i == n_post - 1 else ops.ACT_TANH, keep,
                                  xp=xp, want_planes=pl and i < n_post - 1)
            x, cc, xp = r if len(r) == 3 else (r[0], r[1], None)
            c.post_c.append(cc)
            c.post_taps.append(x)
        c.after = ops.add_vec(c.before, x)
        if hp.output_activation is not None:  # decoder_sa.py:538-540: the losses / the knowledge see activated outputs; the postnet read the raw `before`
            c.before_raw_act = (ops.act_fwd(c.before, output_act_code(hp)), ops.act_fwd(c.after, output_act_code(hp)))
            c.before, c.after = c.before_raw_act
        self._pred_join()
        if c.bn_run:
            if self._nbt_flat is not None and len(c.bn_run) == self._nbt_flat.numel():
                self._nbt_flat.add_(1)
            else:
                for k in c.bn_run:
                    self.B[k].add_(1)

    def _knowledge(self, c):
        """The KD teacher's 5-tuple (..._kd_teacher.py:597-603), shaped as the reference's."""
        B, T, L = c.B, c.T, c.L
        e = lambda x: x.reshape(B, T, -1)
        f = lambda x: x.reshape(B, L, -1)
        cells = [ops.gather_rows(x, c.frame_cell) for x in (c.p1d, c.h0_all, c.h1_all)]
        return (f(c.after), f(c.before), [e(t) for t in c.enc_taps] + [e(c.hs_enc)], [f(t) for t in cells + c.post_taps],
                [e(c.d_outs), e(c.p_outs), e(c.e_outs), e(c.p_embs), e(c.e_embs)])

    # ------------------------------------------------------------------------------------------------ losses and their gradients
    def _losses(self, c, teacher_knowledge):
        """Named losses into one device buffer + the gradient every loss term injects at its tap (c.inj[name])."""
        dev, P, G, hp = self.dev, self.P, self.G, self.hp
        nf, ne = c.n_frames * hp.odim, c.n_enc
        sums = self._z((48, 3), torch.float64)
        names = []

        def term(name, a, b, valid, count, w_l1, w_mse, b_log=None, da=None, want_planes=False):
            a2, b2 = (a.reshape(-1, 1), b.reshape(-1, 1)) if a.dim() == 1 else (a, b)
            names.append(name)
            return ops.l1_mse_loss_grad(a2, b2, valid, count * self.accum_grad, w_l1, w_mse, sums[len(names) - 1], da=da,
                                        b_log_offset=b_log, want_planes=want_planes)  # loss sums + d(loss / accum_grad) in one pass

        inj = c.inj = {}
        # use_masking False (the reference's argparse default): Tacotron2Loss, Tacotron2Loss_KD and prosody_criterions average over the PADDED
        # tensors -- every (b, l) / (b, t) row counts, padded targets are 0, `before` is 0 there and `after` is whatever the postnet makes of the
        # zero padding (..._sa.py:60-70, 122-126); the duration loss and Knowledge_loss are masked regardless (..._kd_student.py:719, 165-176)
        um = not hp.use_masking
        fv, nfm = (None, c.after.shape[0] * hp.odim) if um else (c.frame_valid, nf)
        ev, nem = (None, c.p_outs.numel()) if um else (c.enc_valid, ne)
        inj["after"] = term("after", c.after, c.ys, fv, nfm, 1.0, 1.0)
        inj["before"] = term("before", c.before, c.ys, fv, nfm, 1.0, 1.0)
        inj["d_outs"] = term("dur", c.d_outs, c.ds, c.enc_valid, ne, 0.0, 1.0, b_log=1.0)
        inj["p_outs"] = term("pitch", c.p_outs, c.f0, ev, nem, 0.0, 1.0)
        inj["e_outs"] = term("energy", c.e_outs, c.en, ev, nem, 0.0, 1.0)
        if self.role == "student":
            t_after, t_before, t_enc, t_dec, t_pro = teacher_knowledge
            flat = lambda t: t.to(device=dev, dtype=torch.float32).reshape(-1, t.shape[-1]).contiguous()
            if self.share_proj:
                cp, lp, pp = ["enc.convs_proj.0"] * 3, ["dec.lstm_proj"] * 2, ["dec.post_proj"] * 4
            else:
                cp = ["enc.convs_proj.%d" % i for i in range(3)]
                lp, pp = ["dec.lstm0_proj", "dec.lstm1_proj"], ["dec.post%d_proj" % i for i in range(4)]

            def kd(name, s_in, proj, t, valid, nvalid):
                """MSE(s_in . W^T, t) over valid rows: accumulates dW, returns the gradient w.r.t. s_in."""
                w = P[proj + ".weight"]
                n, k = w.shape
                if ops.planes_enabled() and n % 32 == 0 and k % 32 == 0 and s_in.shape[0] >= 4096:
                    # the projections over every frame (7 taps x ~25 k rows x up to [1024, 256]) on pre-split operands: the tap is packed once, the
                    # loss kernel hands its gradient over as planes too (r3 trace: 1.4 ms of the update's critical path on the fp32-operand kernel)
                    s = ops.linear_planes(ops.pack_planes(s_in), self._wplanes(proj + ".weight", w), n, k)[0]
                    ds_, ds_p = term(name, s, t, valid, nvalid * n, 0.0, 1.0, want_planes=True)
                    self._dw(lambda: self._dw_gemm(ds_, [(s_in, G[proj + ".weight"])]))
                    return ops.linear_planes(ds_p, self._wplanes(proj + ".weight.t", self._wt(w)), k, n)[0]
                s = ops.linear(s_in, w)
                ds_ = term(name, s, t, valid, nvalid * s.shape[1], 0.0, 1.0)
                self._dw(lambda: ops.gemm_tn(ds_, s_in, G[proj + ".weight"]))
                return ops.linear(ds_, self._wt(w))

            with self._late_fork(c):  # (joined in _backward after the postnet)
                if self.distill[2]:
                    cellv = c.cell_valid
                    tc = lambda t: ops.gather_rows(flat(t), c.cell_frame)
                    inj["h1"] = kd("dec2", c.h1_all, lp[1], tc(t_dec[2]), cellv, c.n_frames)
                    inj["h0"] = kd("dec1", c.h0_all, lp[0], tc(t_dec[1]), cellv, c.n_frames)
                    inj["p1d"] = kd("dec0", c.p1d, "dec.prenet_proj", tc(t_dec[0]), cellv, c.n_frames)
                if self.distill[1]:
                    inj["enc0"] = kd("enc0", c.enc_taps[0], "enc.embed_proj", flat(t_enc[0]), c.enc_valid, ne)
                    for i in range(3):
                        inj["enc%d" % (i + 1)] = kd("enc%d" % (i + 1), c.enc_taps[1 + i], cp[i], flat(t_enc[1 + i]), c.enc_valid, ne)
                    inj["hs"] = kd("enc4", c.hs, "enc.blstm_proj", flat(t_enc[4]), c.enc_valid, ne)
            if self.distill[0]:
                term("o_after", c.after, flat(t_after), fv, nfm, 1.0, 1.0, da=inj["after"])
                term("o_before", c.before, flat(t_before), fv, nfm, 1.0, 1.0, da=inj["before"])
            if self.distill[2]:
                for i in range(4):
                    inj["post%d" % i] = kd("dec%d" % (3 + i), c.post_taps[i], pp[i], flat(t_dec[3 + i]), c.frame_valid, c.n_frames)
                inj["post4"] = term("dec7", c.post_taps[4], flat(t_dec[7]), c.frame_valid, nf, 0.0, 1.0)
            if self.distill[3]:
                term("pro0", c.d_outs, flat(t_pro[0]).reshape(-1), c.enc_valid, ne, 0.0, 1.0, da=inj["d_outs"])
                term("pro1", c.p_outs, flat(t_pro[1]).reshape(-1), c.enc_valid, ne, 0.0, 1.0, da=inj["p_outs"])
                term("pro2", c.e_outs, flat(t_pro[2]).reshape(-1), c.enc_valid, ne, 0.0, 1.0, da=inj["e_outs"])
                inj["p_embs"] = kd("pro3", c.p_embs, "pemb_proj", flat(t_pro[3]), c.enc_valid, ne)
                inj["e_embs"] = kd("pro4", c.e_embs, "eemb_proj", flat(t_pro[4]), c.enc_valid, ne)
        c.sums, c.loss_names = sums, names

    def _report(self, c):
        self._late_join()  # (a loss-only caller: the sums of the terms on the weight-gradient stream must be complete before they are copied)
        return LossReport(c.sums, c.loss_names, status_dev=self.status)

    # ------------------------------------------------------------------------------------------------ backward
    def _backward(self, c):
        hp, dev, P, G, inj = self.hp, self.dev, self.P, self.G, c.inj
        B, T = c.B, c.T
        U, Pn, C = hp.dunits, hp.prenet_units, hp.adim
        N, F, lmax, live, offs = c.N, c.F, c.lmax, c.live, c.offs
        lk = lambda l, n: lstm_key(hp, l, n)
        # ---- postnet: after = before + postnet(before)
        if hp.output_activation is not None:  # back through output_activation_fn (y = the activated outputs)
            inj["before"] = ops.act_bwd(inj["before"], c.before, output_act_code(hp))
            inj["after"] = ops.act_bwd(inj["after"], c.after, output_act_code(hp))
        with self._pred_fork(c):  # the predictors' backward needs nothing but their loss gradients: beside the whole decoder backward
            d_preds = [self._predictor_bwd(inj["d_outs"].reshape(-1), "duration_predictor", c.dur_c, c.enc_pad),
                       self._predictor_bwd(inj["p_outs"].reshape(-1), "pitch_predictor", c.pit_c, c.enc_pad),
                       self._predictor_bwd(inj["e_outs"].reshape(-1), "energy_predictor", c.en_c, c.enc_pad)]
        d_before = inj["before"]
        ops.add2d(d_before, inj["after"])
        dx = inj["after"]
        for i in range(len(c.post_c) - 1, -1, -1):
            if "post%d" % i in inj:
                ops.add2d(dx, inj["post%d" % i])
            dx = self._conv_bn_bwd(c, dx, c.post_c[i])
        ops.add2d(d_before, dx)
        self._late_join()  # the KD gradients at the prenet / LSTM / encoder taps (computed beside the frame-level terms and the postnet's backward)
        self._launch_bucket(c, 0)
        # ---- decoder BPTT
        R = hp.reduction_factor
        OR = hp.odim * R
        # [F, R * odim]: the gradient of every cell's R frames (r > 1: d_before [B * L, odim] viewed as groups of R frames, cell_frame = group index)
        d_out_cells = ops.gather_rows(d_before if R == 1 else d_before.reshape(-1, OR), c.cell_frame)
        g_wf = G["dec.feat_out.weight"] if R == 1 else self._z((OR, G["dec.feat_out.weight"].shape[1]))  # (r > 1: frame-major rows, un-permuted below)
        self._dw(lambda: ops.gemm_tn(d_out_cells, c.h1_all, g_wf[:, :U]))  # column blocks of the [odim, U + C] gradient are written in place
        dh1_all = ops.linear(d_out_cells, self._wt(c.wf_h))  # [F, U]
        if "h1" in inj:
            ops.add2d(dh1_all, inj["h1"])
        dF0 = self._z((N, OR))
        ops.scatter_add_rows(d_out_cells, c.cell_row_i64, dF0)
        if hp.use_concate:
            self._dw(lambda: ops.gemm_tn(dF0, c.att_c, g_wf[:, U:]))
        if R > 1:  # feat_out.weight row o * R + j <- frame-major row j * odim + o
            if getattr(self, "_wf_unperm", None) is None:
                idx = np.arange(OR)
                self._wf_unperm = torch.from_numpy(((idx % R) * hp.odim + idx // R).astype(np.int32)).to(dev)
            self._dw(lambda: ops.add2d(G["dec.feat_out.weight"], ops.gather_rows(g_wf, self._wf_unperm)))
        d_att_c = ops.linear(dF0, self._wt(c.wf_att))
        DL, PL = hp.dlayers, hp.prenet_layers
        w0_pre_t = self._wt(c.w0_pre)
        bpl = None
        if DL == 2:
            dg0_all, dg1_all = torch.empty(F, 4 * U, device=dev), torch.empty(F, 4 * U, device=dev)
            w1_ih_t, w1_hh_t, w0_hh_t = self._wt(c.w1_ih), self._wt(c.w1_hh), self._wt(c.w0_hh)
            if ops.planes_enabled() and U % 8 == 0:  # the recurrence's GEMMs on pre-split operands: the cell-backward kernel writes dgates as planes too
                dg0_p, dg1_p = ops.planes_empty(F, 4 * U, dev), ops.planes_empty(F, 4 * U, dev)
                bpl = (self._wplanes("w1_ih_t", w1_ih_t), self._wplanes("w1_hh_t", w1_hh_t), self._wplanes("w0_hh_t", w0_hh_t), dg0_p, dg1_p)
            w1_cat_t = torch.cat([w1_hh_t, w1_ih_t], 0)  # [2U, 4U]: both GEMMs that leave layer 1's gate gradients in one launch per step
            w1_cat = (w1_cat_t, self._wplanes("w1_cat_t", w1_cat_t) if bpl is not None else None)
            ops.decoder_bptt(c.live_i32, N, c.S0, c.S1, c.zr, c.zk, dh1_all, inj.get("h0"), w1_ih_t, w1_hh_t, w0_hh_t, dg0_all, dg1_all, planes=bpl,
                             w1_cat=w1_cat)
            dgs = [dg0_all, dg1_all]
        else:  # any other depth: the launch-by-launch reverse pass (G18); KD taps on the cells do not exist there (the reference's KD classes fail)
            dgs = self._decoder_cells_bptt(c, dh1_all)
            dg0_all = dgs[0]
        if bpl is not None:  # [F, P]: gradient w.r.t. the prenet output of every cell
            dp1_all = ops.linear_planes(bpl[3], self._wplanes("w0_pre_t", w0_pre_t), Pn, 4 * U)[0]
        else:
            dp1_all = ops.linear(dg0_all, w0_pre_t)
        g_ih0 = G[lk(0, "weight_ih")]  # [4U, C + P (+ 1)] = [att_c | prenet (| position)]

        def dw_cells():  # weight gradients of the cells from the saved step-major tensors (one TN GEMM each)
            for l in range(DL - 1, 0, -1):
                self._dw_gemm(dgs[l], [(c.h_all[l - 1], G[lk(l, "weight_ih")]), (c.S[l][3], G[lk(l, "weight_hh")])])
            for l, dg in enumerate(dgs):  # bias_ih and bias_hh enter the gates as a sum: identical gradients
                ops.colsum(dg, G[lk(l, "bias_ih")], out_x=G[lk(l, "bias_hh")])
            self._dw_gemm(dg0_all, [(c.S[0][3], G[lk(0, "weight_hh")]), (c.p1d, g_ih0[:, C : C + Pn])])
            if hp.append_position:
                dw0_pos4 = self._z((4 * U, 4))
                ops.gemm_tn(dg0_all, c.pos4, dw0_pos4)
                ops.add2d(g_ih0[:, C + Pn :], dw0_pos4[:, :1])

        self._dw(dw_cells)
        dG0 = self._z((N, 4 * U))
        ops.scatter_add_rows(dg0_all, c.cell_row_i64, dG0)
        self._dw(lambda: ops.gemm_tn(dG0, c.att_c, g_ih0[:, :C]))
        ops.add2d(d_att_c, ops.linear(dG0, self._wt(c.w0_att)))
        # prenet (batched over all cells), last block first
        if "p1d" in inj:
            ops.add2d(dp1_all, inj["p1d"])
        d = dp1_all
        for l in range(PL - 1, -1, -1):
            wn, bn = "dec.prenet.prenet.%d.0.weight" % l, "dec.prenet.prenet.%d.0.bias" % l
            want_p = l > 0 and bpl is not None and Pn % 32 == 0
            r = ops.act_bwd(d, c.pre_act[l], ops.ACT_RELU, c.pre_keep[l], c.pks, want_planes=want_p)
            dz, dz_p = r if want_p else (r, None)
            inp = c.pre_drop[l - 1] if l > 0 else c.pre_in
            self._dw(lambda dz=dz, inp=inp, wn=wn, bn=bn: (ops.gemm_tn(dz, inp, G[wn]), ops.colsum(dz, G[bn])))
            if l > 0:
                d = ops.linear_planes(dz_p, self._wplanes(wn + ".t", self._wt(P[wn])), Pn, Pn)[0] if dz_p is not None else ops.linear(dz, self._wt(P[wn]))
        self._launch_bucket(c, 1)
        # ---- att = hs + p_embs + e_embs
        d_att = ops.gather_rows(d_att_c, c.row_of_enc)  # back to (b, t) rows; rows without a phoneme get 0
        d_hs = d_att.clone()
        kk = hp.variance_embed_kernel_size
        for nm, sig, tap, keep in (("pitch", c.f0, "p_embs", c.emb_keep[0]), ("energy", c.en, "e_embs", c.emb_keep[1])):
            d_e = d_att
            if tap in inj:
                d_e = ops.add2d(d_att.clone(), inj[tap])
            if keep is not None:
                d_e = ops.act_bwd(d_e, None, ops.ACT_NONE, keep, c.emb_ks)
            def dw_embed(nm=nm, sig=sig, d_e=d_e):  # Conv1d(1 -> C, k): weight and bias gradients in one launch
                ops.conv1d_in1_dw(d_e, sig, G[nm + "_embed.0.weight"], G[nm + "_embed.0.bias"], seg_lo=c.e_lo, seg_hi=c.e_hi)

            self._dw(dw_embed)
        # ---- predictors (their backward was enqueued at the top of this function, on their own stream)
        self._pred_join()
        for d_pred in d_preds:
            ops.add2d(d_hs, d_pred)
        self._launch_bucket(c, 2)
        # ---- encoder
        if "hs" in inj:
            ops.add2d(d_hs, inj["hs"])
        if hp.spk_embed_dim is not None:
            d_hs = ops.copy_cols(d_hs, 0, hp.eunits)  # the speaker-embedding columns are inputs; the encoder sees the first eunits only
        d_hs_live = ops.add2d(self._z(d_hs.shape), d_hs, row_valid=c.enc_valid)  # pad_packed_sequence: padded outputs are constants
        dx = d_hs_live
        for bl_l in reversed(c.bl_stack):  # (dead cells get dg = 0, so the gradient handed to the layer below is zero on padded rows as well)
            dx = self._bilstm_bwd(dx, bl_l)
        for i in range(len(c.conv_c) - 1, -1, -1):
            if "enc%d" % (i + 1) in inj:
                ops.add2d(dx, inj["enc%d" % (i + 1)])
            dxi = self._conv_bn_bwd(c, dx, c.conv_c[i])
            dx = ops.add2d(dxi, dx) if self.hp.use_residual else dxi  # the skip path of `convs[i](xs) + xs`
        if "enc0" in inj:
            ops.add2d(dx, inj["enc0"])
        self._dw(lambda: ops.scatter_add_rows(dx, c.xs, G["enc.embed.weight"], skip=0))  # padding_idx = 0 gets no gradient
        self._launch_bucket(c, 3)
        self._join_dw()

    # ------------------------------------------------------------------------------------------------ public API
    def zero_grad(self):
        self.gflat.zero_()

    def _ctx(self, batch, mode, masks, save=True, reduce=True):
        if mode not in ("eval", "train"):
            raise ValueError("mode must be 'eval' or 'train'")
        c = _Ctx()
        c.train, c.masks, c.save, c.reduce = mode == "train", masks, save, reduce
        if self.side is not None:
            # the arena's clear must not depend on the previous step having ENDED with _join_dw() (a step that raised half way leaves weight
            # gradients / predictor launches queued on the side stream): order this step behind everything that stream still holds
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            self._dw_keep.clear()
            self._pred_ev = self._late_ev = None
        self.arena.begin()
        self.forward_count += 1
        c.draw = self.forward_count
        self._maps(c, batch)
        return c

    def knowledge(self, batch, mode="train", masks=None, native=False):
        """Forward only: the frozen KD teacher's 5-tuple (tts_distill.py:159; the reference leaves the teacher in train mode).
        native=True (KDPipeline, when both engines have the native step): a NativeKnowledge -- pointers into the native engine's arena, decoder
        taps cell-major -- instead of the reference-shaped tuple of tensors."""
        if native and mode == "train" and masks is None and self.native is not None:
            return self.native.knowledge(batch)
        with torch.cuda.device(self.dev), ops.gemm_mode(self.amp):
            c = self._ctx(batch, mode, masks, save=False)
            self._forward(c, batch)
            return self._knowledge(c)

    def forward_backward(self, batch, teacher_knowledge=None, mode="eval", masks=None, reduce=True):
        """One micro-batch: named losses (floats) and d(loss / accum_grad) accumulated into the flat gradient buffer.
        reduce: launch the data-parallel gradient all-reduce buckets as backward completes them.  With accum_grad > 1 pass True only for the
        LAST micro-batch of an update (the earlier ones would race the in-flight collective and multiply the traffic); False keeps the
        gradients local (optimizer_step() then averages nothing)."""
        if self.role == "student" and teacher_knowledge is None:
            raise ValueError("the student step needs teacher_knowledge (tts_distill.py:159-161)")
        if mode == "train" and masks is None and self.role != "kd_teacher" and self.native is not None:
            return self.native.forward_backward(batch, teacher_knowledge, reduce)
        if isinstance(teacher_knowledge, NativeKnowledge):
            raise ValueError("a NativeKnowledge can only be consumed by the native step (mode='train', no injected masks)")
        with torch.cuda.device(self.dev), ops.gemm_mode(self.amp):
            c = self._ctx(batch, mode, masks, reduce=reduce)
            self._forward(c, batch)
            self._losses(c, teacher_knowledge)
            self._backward(c)
            return self._report(c)

    def optimizer_step(self):
        """all-reduce (if distributed) + clip_grad_norm_(grad_clip) + NaN guard + Adam on the flat buffers (tts.py:173-182)."""
        with torch.cuda.device(self.dev):
            self.buckets.finish(lambda t, s: ops.scale_(t, s))
            self.gn_sq.zero_()
            ops.sumsq_accum(self.gflat, self.gn_sq)
            self.update_calls += 1
            ops.adam_step(self.pflat, self.gflat, self.mflat, self.vflat, self.gn_sq, self.grad_clip, self.lr, self.betas[0], self.betas[1], self.eps,
                          self.step_dev, self.status, weight_decay=self.weight_decay)  # skipped on the device (counter included) on a NaN / inf norm or a non-zero status word
            if self._native is not None:
                self._native.params_changed()
            self.model.refresh_plan()
        return self.update_calls

    def grad_norm(self):
        return float(torch.sqrt(self.gn_sq).item())

    def train_step(self, batch, teacher_knowledge=None, mode="eval", masks=None):
        self.zero_grad()
        rep = self.forward_backward(batch, teacher_knowledge, mode, masks)
        self.optimizer_step()
        gn_host = torch.empty(1, dtype=torch.float64, pin_memory=True)  # this step's norm, read back without stalling (gn_sq is reused next step)
        gn_host.copy_(self.gn_sq, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        rep["grad_norm"] = lambda: (ev.synchronize(), float(np.sqrt(gn_host.numpy()[0])))[1]
        return rep


KD_TEACHER_AT_DEFAULT = -1  # where the next batch's frozen-teacher forward is enqueued: -1 = at the update's start, 0 .. 3 = behind that backward stage
KD_TEACHER_CUS_DEFAULT = 0  # compute units the frozen teacher's stream may use beside the student's update (0 = all; measured in DESIGN §5)


class KDPipeline(object):
    """The KD update with the frozen teacher one batch ahead on its own HIP stream: teacher(batch i+1) has no dependency on student(batch i)
    (the teacher is frozen; its BatchNorm buffers still advance in batch order), and neither fills the GPU alone, so the two overlap.  The student's
    stream waits on an event recorded after the teacher's forward; the knowledge tensors are handed over with record_stream so the caching
    allocator does not recycle them while the student still reads them.  step() is call-compatible with the sequential pair
    `know = teacher.knowledge(batch); student.train_step(batch, know)`."""

    def __init__(self, teacher_engine, student_engine, mode="train"):
        self.teng, self.eng, self.mode = teacher_engine, student_engine, mode
        # FCL_KD_TEACHER_CUS=n: the frozen teacher's stream dispatches to n compute units only (fcl_stream_create_cus), the student's streams keep all
        n_cus = int(os.environ.get("FCL_KD_TEACHER_CUS", str(KD_TEACHER_CUS_DEFAULT)))
        if n_cus > 0:
            with torch.cuda.device(student_engine.dev):
                h = C.c_void_p()
                _lib.check(_lib.load().fcl_stream_create_cus(n_cus, C.byref(h)))
            self._side_handle = h  # lives as long as the pipeline (the process): the teacher's arena may be read on it until the last update
            self.side = torch.cuda.ExternalStream(h.value, device=student_engine.dev)
        elif int(os.environ.get("FCL_PLACE_STREAMS", "1")):
            # round 6: the frozen teacher's stream is measured onto a compute pipe of its own, apart from the student's stream and the student's weight-gradient
            # stream (ops.stream_apart): on a shared pipe the KD update takes 12.7 ms (with the student's stream) or 9.4 ms (with its weight-gradient stream)
            # instead of 8.4 -- and which pipe a new stream lands on depends on how many streams the process created before
            with torch.cuda.device(student_engine.dev):
                busy = [torch.cuda.current_stream(student_engine.dev)]
                if student_engine.native is not None:
                    busy.append(student_engine.native.side)
                elif getattr(student_engine, "side", None) is not None:
                    busy.append(student_engine.side)
                self.side = ops.stream_apart(busy, device=student_engine.dev, cache=False)  # (a cached one may have been placed against a destroyed engine's stream)
        else:
            self.side = torch.cuda.Stream(device=student_engine.dev)
        self.teacher_cus = n_cus
        self.teacher_at = int(os.environ.get("FCL_KD_TEACHER_AT", str(KD_TEACHER_AT_DEFAULT)))
        self.pending = None  # (batch id, knowledge, event)
        # both engines native (and train form): the knowledge stays in the teacher engine's arena, cell-major (no frame round trip, no torch tensors)
        self.native = mode == "train" and teacher_engine.native is not None and student_engine.native is not None

    def _launch_teacher(self, batch):
        main = torch.cuda.current_stream(self.eng.dev)
        c = _Ctx()
        self.teng._maps(c, batch)  # index maps are allocated and uploaded on the main stream (both engines use them), then cached on the batch
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            know = self.teng.knowledge(batch, mode=self.mode, native=self.native)
            if not isinstance(know, NativeKnowledge):
                for t in (know[0], know[1], *know[2], *know[3], *know[4]):
                    t.record_stream(main)
            ev = torch.cuda.Event()
            ev.record(self.side)
        return id(batch), know, ev

    def step(self, batch, next_batch=None):
        """One student update on `batch`; `next_batch` (if given) starts its teacher forward concurrently with this update: at the update's start
        (teacher_at = -1) or, on the native step, once backward stage `teacher_at` (0 .. 3) of this update has been enqueued -- the teacher's stream then
        waits for that stage, and its forward runs beside the rest of this update and the beginning of the next (its outputs are first read by the next
        update's loss phase)."""
        if self.pending is None or self.pending[0] != id(batch):
            self.pending = self._launch_teacher(batch)
        _, know, ev = self.pending
        main = torch.cuda.current_stream(self.eng.dev)
        if self.teacher_at < 0 or not self.native or next_batch is None:
            self.pending = self._launch_teacher(next_batch) if next_batch is not None else None
            main.wait_event(ev)
            return self.eng.train_step(batch, know, mode=self.mode)
        self.pending = None

        def hook(stage):
            if stage == self.teacher_at:
                self.pending = self._launch_teacher(next_batch)

        main.wait_event(ev)
        self.eng.native.stage_hook = hook
        try:
            rep = self.eng.train_step(batch, know, mode=self.mode)
        finally:
            self.eng.native.stage_hook = None
        if self.pending is None:  # the update did not go through the native stages (accumulation micro-batch, fallback): launch now
            self.pending = self._launch_teacher(next_batch)
        return rep
