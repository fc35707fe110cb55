"""CPU restatement of the Parallel WaveGAN generator -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

PARITY UNPINNED.  The reference has no vocoder source: it shells out to the external `parallel-wavegan-decode --checkpoint vocoder/PWG/PWG.pkl`
(inference_student.sh:20-23, inference_teacher.sh:20-23, README.md:46) and ships neither the package, the checkpoint nor a single waveform, so no
golden vector of the reference can pin this file.  It restates the PUBLISHED architecture of kan-bayashi/ParallelWaveGAN
(`parallel_wavegan/models/parallel_wavegan.py:ParallelWaveGANGenerator`, `layers/residual_block.py:ResidualBlock`,
`layers/upsample.py:ConvInUpsampleNetwork / UpsampleNetwork / Stretch2d`) in its LJSpeech v1 configuration, which matches the reference's features
(80 mels, hop 256 at 22 050 Hz: preprocess.py:252-257): in/out channels 1, 30 layers in 3 stacks (dilation 2^(l mod 10)), kernel 3, residual / skip
channels 64, gate channels 128, aux channels 80, aux_context_window 2, upsample scales [4, 4, 4, 4], weight norm folded, dropout 0, bias everywhere
except conv_in, the 1x9 smoothing convolutions and conv1x1_aux.  State-dict names are that package's.  plain torch, fp32.
"""
import math

import torch
import torch.nn.functional as F

CONFIG = dict(layers=30, stacks=3, residual_channels=64, gate_channels=128, skip_channels=64, aux_channels=80, aux_context_window=2,
              kernel_size=3, upsample_scales=(4, 4, 4, 4))


def param_spec(cfg=None):
    """Ordered {state_dict name: shape} of the generator with weight norm folded (one `weight` per convolution)."""
    cfg = dict(CONFIG, **(cfg or {}))
    R, G, S, A, k = cfg["residual_channels"], cfg["gate_channels"], cfg["skip_channels"], cfg["aux_channels"], cfg["kernel_size"]
    spec = {"first_conv.weight": (R, 1, 1), "first_conv.bias": (R,),
            "upsample_net.conv_in.weight": (A, A, 2 * cfg["aux_context_window"] + 1)}
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
    """torch.nn.utils.weight_norm parametrisation -> plain weights: w = g * v / ||v|| with the norm over every dim but 0 (the checkpoint
    layout of a generator saved before remove_weight_norm())."""
    out = {}
    for k, v in sd.items():
        if k.endswith("weight_g"):
            base = k[: -len("_g")]
            vv = sd[base + "_v"]
            norm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (vv.dim() - 1))
            out[base] = v * vv / norm
        elif not k.endswith("weight_v"):
            out[k] = v
    return out


def upsample(sd, c, cfg=None):
    """ConvInUpsampleNetwork: c [B, aux, T' + 2*ctx] (already replicate-padded) -> [B, aux, T' * prod(scales)]."""
    cfg = dict(CONFIG, **(cfg or {}))
    c = F.conv1d(c, sd["upsample_net.conv_in.weight"])  # no padding, no bias: consumes the context frames
    c = c.unsqueeze(1)
    for i, s in enumerate(cfg["upsample_scales"]):
        c = F.interpolate(c, scale_factor=(1, s), mode="nearest")  # Stretch2d
        c = F.conv2d(c, sd["upsample_net.upsample.up_layers.%d.weight" % (2 * i + 1)], padding=(0, s))
    return c.squeeze(1)


def generator_forward(sd, z, c_up, cfg=None, return_taps=False):
    """z [B, 1, T] noise, c_up [B, aux, T] upsampled features -> waveform [B, 1, T]."""
    cfg = dict(CONFIG, **(cfg or {}))
    half = cfg["gate_channels"] // 2
    lps = cfg["layers"] // cfg["stacks"]
    x = F.conv1d(z, sd["first_conv.weight"], sd["first_conv.bias"])
    skips = 0
    taps = []
    for l in range(cfg["layers"]):
        p = "conv_layers.%d." % l
        d = 2 ** (l % lps)
        res = x
        h = F.conv1d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=(cfg["kernel_size"] - 1) // 2 * d, dilation=d)
        a = F.conv1d(c_up, sd[p + "conv1x1_aux.weight"])
        g = torch.tanh(h[:, :half] + a[:, :half]) * torch.sigmoid(h[:, half:] + a[:, half:])
        s = F.conv1d(g, sd[p + "conv1x1_skip.weight"], sd[p + "conv1x1_skip.bias"])
        x = (F.conv1d(g, sd[p + "conv1x1_out.weight"], sd[p + "conv1x1_out.bias"]) + res) * math.sqrt(0.5)
        skips = skips + s
        if return_taps:
            taps.append(x)
    skips = skips * math.sqrt(1.0 / cfg["layers"])
    y = F.conv1d(torch.relu(skips), sd["last_conv_layers.1.weight"], sd["last_conv_layers.1.bias"])
    y = F.conv1d(torch.relu(y), sd["last_conv_layers.3.weight"], sd["last_conv_layers.3.bias"])
    return (y, taps, skips) if return_taps else y


def inference(sd, mel, z=None, cfg=None):
    """ParallelWaveGANGenerator.inference: mel [T', aux] -> waveform [T' * hop].  z: noise [T' * hop] (drawn from torch's generator when None)."""
    cfg = dict(CONFIG, **(cfg or {}))
    hop = 1
    for s in cfg["upsample_scales"]:
        hop *= s
    c = torch.as_tensor(mel, dtype=torch.float32).t().unsqueeze(0)
    c = F.pad(c, (cfg["aux_context_window"], cfg["aux_context_window"]), mode="replicate")
    T = (c.shape[2] - 2 * cfg["aux_context_window"]) * hop
    z = torch.randn(1, 1, T) if z is None else torch.as_tensor(z, dtype=torch.float32).reshape(1, 1, T)
    with torch.no_grad():
        return generator_forward(sd, z, upsample(sd, c, cfg), cfg).reshape(-1)
