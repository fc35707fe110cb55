"""Parallel WaveGAN generator on the HIP path (BASELINE configs[4]): mel [T', 80] -> waveform [T' * 256].

The reference hands its mels to the external `parallel-wavegan-decode --checkpoint vocoder/PWG/PWG.pkl` (inference_student.sh:20-23); this module
is that stage, built from the published architecture (kan-bayashi/ParallelWaveGAN `ParallelWaveGANGenerator`, v1 / LJSpeech; restated in
oracle/pwg_oracle.py — no source or vectors in the reference: parity unpinned).  State-dict names and the `inference(c, x=None)` call are that
package's, so its checkpoints (`{"model": {"generator": ...}}`, with or without weight norm) load here.

Batched: utterances are concatenated sample-major; every convolution of the 30-layer residual stack is a K-term of the pre-split-operand GEMM
(csrc/pwg.hip).  No CPU fallback; needs the pre-split path (FCL_PRECISION / FCL_PLANES not 0)."""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib, ops

CONFIG = dict(layers=30, stacks=3, residual_channels=64, gate_channels=128, skip_channels=64, aux_channels=80, aux_context_window=2,
              kernel_size=3, upsample_scales=(4, 4, 4, 4))


def param_spec(cfg=None):
    """Ordered {state_dict name: shape}, weight norm folded (the generator after remove_weight_norm())."""
    cfg = dict(CONFIG, **(cfg or {}))
    R, G, S, A, k = cfg["residual_channels"], cfg["gate_channels"], cfg["skip_channels"], cfg["aux_channels"], cfg["kernel_size"]
    spec = {"first_conv.weight": (R, 1, 1), "first_conv.bias": (R,), "upsample_net.conv_in.weight": (A, A, 2 * cfg["aux_context_window"] + 1)}
    for i, s in enumerate(cfg["upsample_scales"]):
        spec["upsample_net.upsample.up_layers.%d.weight" % (2 * i + 1)] = (1, 1, 1, 2 * s + 1)
    for l in range(cfg["layers"]):
        p = "conv_layers.%d." % l
        spec[p + "conv.weight"], spec[p + "conv.bias"] = (G, R, k), (G,)
        spec[p + "conv1x1_aux.weight"] = (G, A, 1)
        spec[p + "conv1x1_out.weight"], spec[p + "conv1x1_out.bias"] = (R, G // 2, 1), (R,)
        spec[p + "conv1x1_skip.weight"], spec[p + "conv1x1_skip.bias"] = (S, G // 2, 1), (S,)
    spec["last_conv_layers.1.weight"], spec["last_conv_layers.1.bias"] = (S, S, 1), (S,)
    spec["last_conv_layers.3.weight"], spec["last_conv_layers.3.bias"] = (1, S, 1), (1,)
    return spec


def fold_weight_norm(sd):
    """weight_g / weight_v pairs (torch.nn.utils.weight_norm, norm over every dim but 0) -> plain weights; other entries pass through."""
    out = {}
    for k, v in sd.items():
        v = np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v)
        if k.endswith("weight_g"):
            base = k[: -len("_g")]
            vv = sd[base + "_v"]
            vv = np.asarray(vv.detach().cpu().numpy() if torch.is_tensor(vv) else vv, dtype=np.float64)
            norm = np.sqrt((vv.reshape(vv.shape[0], -1) ** 2).sum(axis=1)).reshape([-1] + [1] * (vv.ndim - 1))
            out[base] = (v.astype(np.float64) * vv / norm).astype(np.float32)
        elif not k.endswith("weight_v"):
            out[k] = v
    return out


def unpack_planes(p, cols, chunk_major=False):
    """fp32 value hi + lo of a P32 plane buffer [rows, ceil(cols/32) * 64], row-major or chunk-major ([chunk][row] lines) (tests / debugging: plain
    tensor reshuffling, no arithmetic kernel of ours)."""
    rows = p.shape[0]
    v = (p.reshape(-1, rows, 2, 32).permute(1, 0, 2, 3) if chunk_major else p.reshape(rows, -1, 2, 32)).to(torch.int32)
    f = lambda t: (t << 16).view(torch.float32)
    return (f(v[:, :, 0]) + f(v[:, :, 1])).reshape(rows, -1)[:, :cols].contiguous()


class PWGPlan(object):
    """Device-resident, GEMM-ready weights of one generator: packed taps and their P32 planes, stacked out / skip projections."""

    def __init__(self, state_dict, device, cfg=None):
        if not ops.planes_enabled():
            raise _lib.FclError("fcl-taco2_amd: the vocoder runs on the pre-split operand kernels only (FCL_PRECISION=0 / FCL_PLANES=0 is set)")
        if not str(device).startswith("cuda"):
            raise _lib.FclError("fcl-taco2_amd: PWGPlan needs a GPU device (no CPU fallback)")
        self.cfg = cfg = dict(CONFIG, **(cfg or {}))
        self.device = dev = torch.device(device)
        if "model" in state_dict and "generator" in state_dict["model"]:  # a parallel_wavegan checkpoint
            state_dict = state_dict["model"]["generator"]
        sd = fold_weight_norm(state_dict)
        for k, shp in param_spec(cfg).items():
            if k not in sd or tuple(np.shape(sd[k])) != tuple(shp):
                raise _lib.FclError("fcl-taco2_amd: generator state_dict lacks %s %r (got %r)" % (k, shp, None if k not in sd else np.shape(sd[k])))
        self.R, self.A, self.S, self.k = cfg["residual_channels"], cfg["aux_channels"], cfg["skip_channels"], cfg["kernel_size"]
        if cfg["gate_channels"] != 2 * self.R or self.S != self.R or self.R % 32:
            raise _lib.FclError("fcl-taco2_amd: the vocoder kernels need gate = 2 x residual = 2 x skip channels, a multiple of 32")
        self.hop = int(np.prod(cfg["upsample_scales"]))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        with torch.cuda.device(dev):
            self.first_w, self.first_b = t(sd["first_conv.weight"].reshape(-1)), t(sd["first_conv.bias"])
            self.conv_in = ops.pack_conv1d_weight(t(sd["upsample_net.conv_in.weight"]))  # [k, A, A]
            self.up_w = [t(sd["upsample_net.upsample.up_layers.%d.weight" % (2 * i + 1)].reshape(-1)) for i in range(len(cfg["upsample_scales"]))]
            self.layers = []
            for l in range(cfg["layers"]):
                p = "conv_layers.%d." % l
                wp = ops.pack_conv1d_weight(t(sd[p + "conv.weight"]))  # [k, 2R, R]
                w_os = np.concatenate([sd[p + "conv1x1_out.weight"].reshape(self.R, self.R), sd[p + "conv1x1_skip.weight"].reshape(self.S, self.R)])
                self.layers.append(dict(
                    dilation=2 ** (l % (cfg["layers"] // cfg["stacks"])),
                    w_conv_p=ops.pack_planes(wp.reshape(self.k * 2 * self.R, self.R)), b_conv=t(sd[p + "conv.bias"]),
                    w_aux_p=ops.pack_planes(t(sd[p + "conv1x1_aux.weight"].reshape(2 * self.R, self.A))),
                    w_os_p=ops.pack_planes(t(w_os)), b_os=t(np.concatenate([sd[p + "conv1x1_out.bias"], sd[p + "conv1x1_skip.bias"]]))))
            # every block's conv1x1_aux stacked [layers * 2R, A]: the frame-rate form of the auxiliary term projects the features once for all blocks
            self.w_aux_all_p = ops.pack_planes(t(np.concatenate([sd["conv_layers.%d.conv1x1_aux.weight" % l].reshape(2 * self.R, self.A)
                                                                 for l in range(cfg["layers"])])))
            self.last_w1p = ops.pack_planes(t(sd["last_conv_layers.1.weight"].reshape(self.S, self.S)))
            self.last_b1 = t(sd["last_conv_layers.1.bias"])
            self.last_w2 = t(sd["last_conv_layers.3.weight"].reshape(-1))
            self.last_b2 = float(np.asarray(sd["last_conv_layers.3.bias"]).reshape(-1)[0])


def fused_block(plan):
    """One launch per residual block: the published v1 geometry (64 residual channels, kernel 3, <= 96 auxiliary channels); FCL_PWG_FUSED=0 opts out."""
    return plan.R == 64 and plan.k == 3 and plan.A <= 96 and os.environ.get("FCL_PWG_FUSED", "1") != "0"


def aux_frame_rate(plan):
    """The one-launch block with the auxiliary term evaluated at frame rate (default; FCL_PWG_AUX_FRAME_RATE=0 = planes of the upsampled features)."""
    return fused_block(plan) and plan.hop % 128 == 0 and os.environ.get("FCL_PWG_AUX_FRAME_RATE", "1") != "0"


class ParallelWaveGANGenerator(object):
    """mel -> waveform.  `synthesize(mels)` is the batched entry; `inference(c, x=None)` mirrors the published single-utterance call."""

    def __init__(self, plan):
        self.plan = plan
        self._maps_cache = {}

    # ---- integer index maps of a batch shape (host-built once per tuple of utterance lengths, cached) -------------------------------------------
    def _maps(self, lens):
        key = tuple(lens)
        hit = self._maps_cache.get(key)
        if hit is not None:
            cur = torch.cuda.current_stream(self.plan.device)
            cur.wait_event(hit["ready"])  # built on another stream, possibly: order this one behind it
            if cur.cuda_stream not in hit["streams"]:  # ... and tell the caching allocator that this stream reads the maps too: an eviction while
                hit["streams"].add(cur.cuda_stream)    # its kernels are in flight must not hand the blocks back to the building stream
                for v in hit.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
            return hit
        pl, dev = self.plan, self.plan.device
        ctx = pl.cfg["aux_context_window"]
        lens_np = np.asarray(lens, dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(lens_np)]).astype(np.int64)
        pad_idx, lo, hi, keep = [], [], [], []
        base = 0
        for u, n in enumerate(lens):  # padded utterance = frames clamp(-ctx .. n-1+ctx); conv_in is 'valid', i.e. 'same' evaluated on the interior
            pad_idx.append(offs[u] + np.clip(np.arange(-ctx, n + ctx), 0, n - 1))
            lo.append(np.full(n + 2 * ctx, base))
            hi.append(np.full(n + 2 * ctx, base + n + 2 * ctx))
            keep.append(np.arange(base + ctx, base + ctx + n))
            base += n + 2 * ctx
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        m = dict(offs=offs, pad_idx=i32(np.concatenate(pad_idx)), lo=i32(np.concatenate(lo)), hi=i32(np.concatenate(hi)), keep=i32(np.concatenate(keep)),
                 frame_utt=i32(np.repeat(np.arange(len(lens)), lens_np)), utt_off=i32(offs))
        s_off = torch.from_numpy(offs * pl.hop).to(dev)
        reps = torch.from_numpy(lens_np * pl.hop).to(dev)
        m["seg_lo"] = torch.repeat_interleave(s_off[:-1], reps).to(torch.int32)  # sample range of each row's utterance (integers only)
        m["seg_hi"] = torch.repeat_interleave(s_off[1:], reps).to(torch.int32)
        m["ready"] = torch.cuda.Event()  # the cached maps may be used from other streams later: those wait for this event (no host synchronisation:
        m["ready"].record(torch.cuda.current_stream(dev))  # a driver that sees a new batch shape every call keeps its batches pipelined)
        m["streams"] = {torch.cuda.current_stream(dev).cuda_stream}
        if len(self._maps_cache) >= 4:
            self._maps_cache.clear()
        self._maps_cache[key] = m
        return m

    # ---- feature side: replicate padding + conv_in at frame rate, then 4 x (stretch + smoothing) to sample rate --------------------------------
    def _conv_in(self, mel_rows, mp):
        """replicate padding + conv_in at frame rate -> [sum T', A]"""
        c_pad = ops.gather_rows(mel_rows, mp["pad_idx"])
        c_in = ops.conv1d(c_pad, self.plan.conv_in, None, mp["lo"], mp["hi"])
        return ops.gather_rows(c_in, mp["keep"])

    def _cascade(self, c, mp, want_planes, chunk_major=False):
        """The 4 x (stretch + smoothing) stages on c [frames, C] (C % 4 == 0): planes of the result (want_planes) or the fp32 result."""
        pl, dev = self.plan, self.plan.device
        C_ = c.shape[1]
        frames = int(mp["offs"][-1])
        rate, lib = 1, _lib.load()
        n_st = len(pl.up_w)
        for i, s in enumerate(pl.cfg["upsample_scales"]):
            last = i == n_st - 1
            rows = frames * rate * s
            as_planes = last and want_planes
            out = ops.planes_empty(rows, C_, dev) if as_planes else torch.empty(rows, C_, device=dev)
            _lib.check(lib.fcl_pwg_upsample_stage(c.data_ptr(), mp["frame_utt"].data_ptr(), mp["utt_off"].data_ptr(), frames, rate, s, pl.up_w[i].data_ptr(),
                                                  None if as_planes else out.data_ptr(), out.data_ptr() if as_planes else None, C_, int(chunk_major),
                                                  ops._stream()))
            c, rate = out, rate * s
        return c

    def _upsample(self, mel_rows, mp, chunk_major):
        return self._cascade(self._conv_in(mel_rows, mp), mp, True, chunk_major)

    def _aux_frame_rate(self, mel_rows, mp):
        """The auxiliary term of every block at frame rate (fcl_pwg_layer_t.kp): the upsampling network is linear and a sample of frame f sees
        frames f-2 .. f+2 only, so conv1x1_aux(upsample(c))[m] = sum_g k[m][g] (W_aux c_in[g]).  Returns (coefficient lines [M], planes of W_aux c_in^T
        for all blocks in the two window alignments, lines per row)."""
        pl, dev = self.plan, self.plan.device
        lib = _lib.load()
        frames = int(mp["offs"][-1])
        M = frames * pl.hop
        # k: response of the upsampling network (utterance edges included) to the colour basis e[g][c] = (g mod 5 == c)
        g = torch.arange(frames, device=dev)
        e = (g[:, None] % 5 == torch.arange(8, device=dev)[None, :]).to(torch.float32).contiguous()
        kc = self._cascade(e, mp, False)
        kp = ops.planes_empty(M, 32, dev)
        _lib.check(lib.fcl_pwg_aux_coeff(kc.data_ptr(), M, pl.hop, frames, kp.data_ptr(), ops._stream()))
        # projected features, gate-row major over frames: line q of pt_a = frames [32q, 32q + 32), of pt_b = frames [32q - 16, 32q + 16)
        ld_pt = (frames + 16 + 31) // 32
        n = ld_pt * 32
        cfull = torch.zeros(16 + n, pl.A, device=dev)
        cfull[16 : 16 + frames] = self._conv_in(mel_rows, mp)
        cfull_p = ops.pack_planes(cfull)
        rows = pl.w_aux_all_p.shape[0]
        pt_a = ops.linear_planes(pl.w_aux_all_p, cfull_p[16:], n, pl.A, want_f32=False, want_planes=True)[1]
        pt_b = ops.linear_planes(pl.w_aux_all_p, cfull_p, n, pl.A, want_f32=False, want_planes=True)[1]
        assert pt_a.shape == (rows, ld_pt * 64)
        return kp, pt_a, pt_b, ld_pt

    def synthesize(self, mels, noise=None, seed=0, return_intermediates=False):
        """mels: list of [T'_i, aux] float tensors / arrays.  noise: optional list of [T'_i * hop] arrays (else drawn on the device from `seed`).
        Returns a list of [T'_i * hop] float32 device tensors."""
        dev = self.plan.device
        with torch.cuda.device(dev):
            lens = [int(m.shape[0]) for m in mels]
            mel_rows = torch.cat([torch.as_tensor(m, dtype=torch.float32).to(dev) for m in mels]).contiguous()
            return self.synthesize_packed(mel_rows, lens, noise, seed, return_intermediates)

    def synthesize_packed(self, mel_rows, lens, noise=None, seed=0, return_intermediates=False):
        """The same on utterances already packed row-wise ([sum T', aux] device tensor, e.g. engine.run's output) with their frame counts."""
        pl, dev = self.plan, self.plan.device
        lib = _lib.load()
        with torch.cuda.device(dev):
            lens = [int(n) for n in lens]
            if not lens or min(lens) < 1:
                raise _lib.FclError("fcl-taco2_amd: empty mel")
            if mel_rows.dim() != 2 or mel_rows.shape[1] != pl.A or mel_rows.shape[0] != sum(lens):
                raise _lib.FclError("fcl-taco2_amd: expected [%d, %d] mel rows, got %r" % (sum(lens), pl.A, tuple(mel_rows.shape)))
            if sum(lens) * pl.hop >= 2 ** 31:  # before any int32 index map is built or any kernel launched with the overflowing sizes
                raise _lib.FclError("fcl-taco2_amd: more than 2^31 samples in one vocoder batch")
            mp = self._maps(lens)
            offs = mp["offs"]
            M, R = int(offs[-1]) * pl.hop, pl.R
            fused = fused_block(pl)
            aux_fr = aux_frame_rate(pl)
            if aux_fr:
                cp = None
                kp, pt_a, pt_b, ld_pt = self._aux_frame_rate(mel_rows, mp)
            else:
                cp = self._upsample(mel_rows, mp, fused)  # the one-launch block reads chunk-major planes (contiguous rows per 32-column chunk)
            if noise is None:
                z = torch.empty(M, device=dev)
                _lib.check(lib.fcl_pwg_noise(z.data_ptr(), M, seed & 0xFFFFFFFF, ops._stream()))
            else:
                z = torch.cat([torch.as_tensor(n_, dtype=torch.float32).reshape(-1).to(dev) for n_ in noise]).contiguous()
                if z.numel() != M:
                    raise _lib.FclError("fcl-taco2_amd: noise must hold T' * %d samples per utterance" % pl.hop)
            seg_lo, seg_hi = mp["seg_lo"], mp["seg_hi"]
            # the one-launch blocks carry x as planes only; fp32 x is then just the workspace of the general last stage (not needed for 64 channels)
            x = None if fused and pl.S == 64 else torch.empty(M, R, device=dev)
            xp = ops.planes_empty(M, R, dev)
            _lib.check(lib.fcl_pwg_first_conv(z.data_ptr(), pl.first_w.data_ptr(), pl.first_b.data_ptr(), None if fused else x.data_ptr(), xp.data_ptr(),
                                              M, R, int(fused), ops._stream()))
            skips = torch.empty(M, R, device=dev)
            gp = ops.planes_empty(M, R, dev)  # unfused: the gate's planes; fused: the second x buffer (blocks ping-pong between xp and gp)
            zbuf = obuf = None
            if not fused:
                zbuf, obuf = torch.empty(M, 2 * R, device=dev), torch.empty(M, 2 * R, device=dev)
            taps = []
            for l, L in enumerate(pl.layers):
                a = _lib.PwgLayer()
                a.m, a.r, a.aux, a.ksize, a.dilation, a.first_layer = M, R, pl.A, pl.k, L["dilation"], int(l == 0)
                a.seg_lo, a.seg_hi = seg_lo.data_ptr(), seg_hi.data_ptr()
                a.x, a.xp = None if x is None else x.data_ptr(), xp.data_ptr()
                a.w_conv_p, a.b_conv, a.w_aux_p = L["w_conv_p"].data_ptr(), L["b_conv"].data_ptr(), L["w_aux_p"].data_ptr()
                if aux_fr:
                    row0 = l * 2 * R * ld_pt * 64  # this block's gate rows of the stacked projection (int16 elements)
                    a.kp, a.pt_a, a.pt_b = kp.data_ptr(), pt_a.data_ptr() + 2 * row0, pt_b.data_ptr() + 2 * row0
                    a.ld_pt, a.hop = ld_pt, pl.hop
                else:
                    a.cp = cp.data_ptr()
                a.w_os_p, a.b_os, a.skips = L["w_os_p"].data_ptr(), L["b_os"].data_ptr(), skips.data_ptr()
                if fused:
                    a.xp_out = gp.data_ptr()
                else:
                    a.z, a.gp, a.o = zbuf.data_ptr(), gp.data_ptr(), obuf.data_ptr()
                _lib.check(lib.fcl_pwg_layer_fwd(C.byref(a), ops._stream()))
                if fused:
                    xp, gp = gp, xp
                if return_intermediates:
                    taps.append(unpack_planes(xp, R, chunk_major=True) if fused else x.clone())
            wav = torch.empty(M, device=dev)
            _lib.check(lib.fcl_pwg_last_fwd(skips.data_ptr(), math.sqrt(1.0 / len(pl.layers)), pl.last_w1p.data_ptr(), pl.last_b1.data_ptr(),
                                            pl.last_w2.data_ptr(), pl.last_b2, gp.data_ptr(), None if x is None else x.data_ptr(), wav.data_ptr(), M, pl.S,
                                            ops._stream()))
            outs = [wav[int(offs[i]) * pl.hop : int(offs[i + 1]) * pl.hop] for i in range(len(lens))]
            if return_intermediates:
                return outs, dict(z=z, taps=taps, skips=skips, seg=(seg_lo, seg_hi))
            return outs

    def inference(self, c, x=None):
        """ParallelWaveGANGenerator.inference(c, x): c [T', aux] (x: optional noise [T' * hop]) -> waveform [T' * hop, 1]."""
        return self.synthesize([c], None if x is None else [x])[0].reshape(-1, 1)
