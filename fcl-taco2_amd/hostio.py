"""Host -> device hand-over helpers shared by the synthesis (`engine.prepare`) and training (`training.TrainEngine._maps`) sides."""
import torch


class PinnedRing(object):
    """A few fixed page-locked staging buffers, reused round-robin: host -> device hand-over without touching the pinned-memory allocator in the
    training loop (its blocks are only recycled once their copy has completed, i.e. a step later, so per-batch pin_memory() keeps calling
    hipHostMalloc, which is slow and serialises with the GPU).  A slot is rewritten only after the event recorded behind its last copy."""

    def __init__(self, depth=4):
        self.depth, self.slots, self.i = depth, {}, 0

    def stage(self, key, t):
        """Copy the CPU tensor `t` into the next pinned slot of `key`; returns (pinned view, slot) — call done(slot) after enqueuing the copy."""
        ring = self.slots.setdefault(key, [None] * self.depth)
        j = self.i % self.depth
        slot = ring[j]
        need = t.numel() * t.element_size()
        if slot is None or slot["buf"].numel() < need:
            slot = ring[j] = {"buf": torch.empty(max(need * 3 // 2, 1 << 16), dtype=torch.uint8, pin_memory=True), "ev": None}
        elif slot["ev"] is not None:
            slot["ev"].synchronize()  # the copy that last read this slot (depth steps ago): long done in steady state
        view = slot["buf"][:need].view(t.dtype).view(t.shape)
        view.copy_(t)
        return view, slot

    def upload(self, items, dev):
        """{key: cpu tensor} -> {key: device tensor}, non-blocking, one ring position per call."""
        out, used = {}, []
        for k, t in items.items():
            view, slot = self.stage(k, t.contiguous())
            out[k] = view.to(dev, non_blocking=True)
            used.append(slot)
        ev = torch.cuda.Event()
        ev.record()
        for slot in used:
            slot["ev"] = ev
        self.i += 1
        return out
