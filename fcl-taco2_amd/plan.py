"""SynthesisPlan — device-resident, kernel-ready weights built once from a reference-named state_dict.

Packing (all by libfcl_hip kernels, not torch ops):
  * Conv1d+BatchNorm(eval) -> tap-major [k, Cout, Cin] weights with the BN scale folded in, BN shift as bias
    (reference encoder_sa.py:61-78, decoder_sa.py:199-263);
  * decoder LSTMCell-0 `weight_ih` split into its att_c / prenet / position column blocks and `feat_out.weight`
    into its lstm / att_c blocks, so the att_c parts are hoisted out of the time loop;
  * bias_ih + bias_hh summed.
LSTM / Linear matrices are used in place: torch's [out, in] layout is already the K-contiguous W the GEMM wants.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops
from .hparams import lstm_key, output_act_code, param_spec

BN_EPS = 1e-5
LN_EPS = 1e-12


class ConvPack(object):
    __slots__ = ("wp", "bias", "k", "cin", "cout", "wpp")  # wpp: P32 planes of the packed taps viewed as [k*Cout, Cin] (None: fp32 path only)


def _conv_planes(c):
    c.wpp = ops.pack_planes(c.wp.reshape(c.k * c.cout, c.cin)) if ops.planes_enabled() else None
    return c


class PredictorPack(object):
    __slots__ = ("convs", "ln", "lin_w", "lin_b")


class PredictorGroup(object):
    """G predictors of one shape stacked for the grouped launches (ops.conv1d_planes_group / ops.layernorm_group): the first layer's taps stacked
    along Cout (all predictors read the same hs: ONE Conv1d with G * Cout output channels), the other layers' planes group-major."""
    __slots__ = ("names", "G", "layers", "cin", "chans", "k", "w0", "b0", "wpp", "bias", "gamma", "beta", "lin_w", "lin_b")


def _predictor_group(preds, names):
    """None unless the predictors share one geometry (the shipped recipes': 2 layers, 384 channels, kernel 3) and the pre-split path is on."""
    p0 = preds[0]
    same = all(len(p.convs) == len(p0.convs) and all((a.cin, a.cout, a.k) == (b.cin, b.cout, b.k) for a, b in zip(p.convs, p0.convs)) for p in preds)
    c0 = p0.convs[0]
    if not (ops.planes_enabled() and same and len(p0.convs) >= 1 and c0.cout % 32 == 0 and c0.cin <= 384 and c0.cout <= 384 and c0.k >= 3 and c0.wpp is not None):
        return None
    g = PredictorGroup()
    g.names, g.G, g.layers, g.cin, g.chans, g.k = list(names), len(preds), len(p0.convs), c0.cin, c0.cout, c0.k
    # layer 0: [k, G * Cout, Cin] taps -> one ConvPack-like pair (planes of the stacked taps, stacked bias)
    w0 = torch.cat([p.convs[0].wp for p in preds], dim=1).contiguous()
    g.w0 = ConvPack()
    g.w0.k, g.w0.cout, g.w0.cin, g.w0.wp = c0.k, g.G * c0.cout, c0.cin, w0
    g.w0.bias = torch.cat([p.convs[0].bias for p in preds]).contiguous()
    _conv_planes(g.w0)
    g.b0 = g.w0.bias
    # layers >= 1: group-major planes [G][k * Cout][Cin planes], bias [G * Cout]
    g.wpp = [torch.cat([p.convs[i].wpp for p in preds], dim=0).contiguous() for i in range(1, g.layers)]
    g.bias = [torch.cat([p.convs[i].bias for p in preds]).contiguous() for i in range(1, g.layers)]
    g.gamma = [torch.cat([p.ln[i][0] for p in preds]).contiguous() for i in range(g.layers)]
    g.beta = [torch.cat([p.ln[i][1] for p in preds]).contiguous() for i in range(g.layers)]
    g.lin_w = torch.cat([p.lin_w for p in preds]).contiguous()
    g.lin_b = torch.cat([p.lin_b.reshape(-1) for p in preds]).contiguous()
    return g


class DecoderPack(object):
    """Holds the ctypes struct and keeps alive every tensor it points to."""

    def __init__(self):
        self.struct = _lib.DecoderWeights()
        self.keep = []


def _dev(sd, k, device):
    v = sd[k]
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(np.ascontiguousarray(v))
    return v.detach().to(device=device, dtype=torch.float32).contiguous()


class SynthesisPlan(object):
    def __init__(self, state_dict, hp, device="cuda:0"):
        hp.check_supported()
        _lib.load()
        self.hp = hp
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.FclError("fcl-taco2_amd: SynthesisPlan needs a GPU device (no CPU fallback)")
        if not torch.cuda.is_available():
            raise _lib.FclError("fcl-taco2_amd: no HIP GPU is available and the product path has no CPU fallback")
        missing = [k for k in param_spec(hp) if k not in state_dict]
        if missing:
            raise KeyError("state_dict is missing reference keys: %s ..." % missing[:4])
        g = lambda k: _dev(state_dict, k, self.device)
        with torch.cuda.device(self.device):
            self.embed = g("enc.embed.weight")
            self.enc_convs = [self._conv_bn(g, "enc.convs.%d" % i) for i in range(hp.econv_layers)]
            self.blstm_layers = []  # `elayers` stacked bidirectional layers (encoder_sa.py:96-100); layer l > 0 reads [forward | reverse] of layer l - 1
            for l in range(hp.elayers):
                bl = {}
                for sfx, tag in (("", "f"), ("_reverse", "r")):
                    bl["w_ih_" + tag] = g("enc.blstm.weight_ih_l%d%s" % (l, sfx))
                    bl["w_hh_" + tag] = g("enc.blstm.weight_hh_l%d%s" % (l, sfx))
                    bl["b_" + tag] = ops.add_vec(g("enc.blstm.bias_ih_l%d%s" % (l, sfx)), g("enc.blstm.bias_hh_l%d%s" % (l, sfx)))
                    bl["w_ih_p_" + tag] = ops.pack_planes(bl["w_ih_" + tag]) if ops.planes_enabled() else None
                self.blstm_layers.append(bl)
            self.blstm = self.blstm_layers[0]
            self.duration = self._predictor(g, "duration_predictor", hp.duration_predictor_layers)
            self.pitch = self._predictor(g, "pitch_predictor", hp.variance_predictor_layers)
            self.energy = self._predictor(g, "energy_predictor", hp.variance_predictor_layers)
            # grouped forms: pitch + energy (forced durations) and duration + pitch + energy (predicted durations) as one launch per layer
            self.group_pe = _predictor_group([self.pitch, self.energy], ["pitch", "energy"])
            self.group_dpe = _predictor_group([self.duration, self.pitch, self.energy], ["duration", "pitch", "energy"])
            self.pitch_embed_w = g("pitch_embed.0.weight").reshape(hp.adim, -1).contiguous()
            self.pitch_embed_b = g("pitch_embed.0.bias")
            self.energy_embed_w = g("energy_embed.0.weight").reshape(hp.adim, -1).contiguous()
            self.energy_embed_b = g("energy_embed.0.bias")
            self.postnet = [self._conv_bn(g, "dec.postnet.postnet.%d" % i) for i in range(hp.postnet_layers)]
            self.decoder = self._decoder(g)
            # student KD projections (bias-free Linear student->teacher width), present only in KD-training state_dicts
            self.proj = {k[: -len(".weight")]: g(k) for k in state_dict
                         if k.endswith("_proj.weight") or ".convs_proj." in k}
            torch.cuda.synchronize(self.device)

    def _conv_bn(self, g, prefix):
        w = g(prefix + ".0.weight")
        c = ConvPack()
        c.cout, c.cin, c.k = w.shape
        if self.hp.use_batch_norm:
            scale, shift = ops.fold_batchnorm(g(prefix + ".1.weight"), g(prefix + ".1.bias"), g(prefix + ".1.running_mean"),
                                              g(prefix + ".1.running_var"), BN_EPS)
        else:  # `--use-batch-norm false`: Conv1d(no bias) -> activation (encoder_sa.py:78-90, decoder_sa.py:219-232, :250-262)
            scale, shift = None, torch.zeros(c.cout, device=self.device, dtype=torch.float32)
        c.wp = ops.pack_conv1d_weight(w, scale)
        c.bias = shift
        return _conv_planes(c)

    def _predictor(self, g, prefix, layers):
        p = PredictorPack()
        p.convs, p.ln = [], []
        for i in range(layers):
            w = g("%s.conv.%d.0.weight" % (prefix, i))
            c = ConvPack()
            c.cout, c.cin, c.k = w.shape
            c.wp = ops.pack_conv1d_weight(w, None)
            c.bias = g("%s.conv.%d.0.bias" % (prefix, i))
            p.convs.append(_conv_planes(c))
            p.ln.append((g("%s.conv.%d.2.weight" % (prefix, i)), g("%s.conv.%d.2.bias" % (prefix, i))))
        p.lin_w = g(prefix + ".linear.weight").reshape(-1).contiguous()
        p.lin_b = g(prefix + ".linear.bias")
        return p

    def _decoder(self, g):
        hp = self.hp
        C, P, U, O = hp.adim, hp.prenet_units, hp.dunits, hp.odim  # C: att_c width = eunits (+ spk_embed_dim)
        d = DecoderPack()
        s = d.struct
        s.c, s.p, s.u, s.odim = C, P, U, O
        lk = lambda l, n: g(lstm_key(hp, l, n))
        w_ih0 = lk(0, "weight_ih")  # [4U, C + P (+ 1)] = [att_c | prenet (| position: --append-position)]
        wf = g("dec.feat_out.weight")  # [odim * r, U (+ C)]  = [lstm (| att_c: --use-concate)]
        R = int(hp.reduction_factor)
        if R > 1:  # row o * r + j = bin o of the step's frame j (decoder_sa.py:512 `.view(B, odim, -1)`): frame-major rows j * odim + o, so that a
                   # step's output is r consecutive frames of the frame-major mel buffer (fcl_decoder_weights_t.reduction_factor)
            wf = wf.reshape(O, R, -1).permute(1, 0, 2).reshape(O * R, -1).contiguous()
        zeros = lambda *shape: torch.zeros(*shape, device=self.device, dtype=torch.float32)
        t = dict(
            prenet_w0=g("dec.prenet.prenet.0.0.weight"), prenet_b0=g("dec.prenet.prenet.0.0.bias"),
            w0_att=ops.copy_cols(w_ih0, 0, C), w0_pre=ops.copy_cols(w_ih0, C, P),
            # options off = the term is absent in the reference: a zero block here (same kernels, one shape)
            w0_pos=ops.copy_cols(w_ih0, C + P, 1).reshape(-1) if hp.append_position else zeros(4 * U),
            w0_hh=lk(0, "weight_hh"),
            b0=ops.add_vec(lk(0, "bias_ih"), lk(0, "bias_hh")),
            wf_h=ops.copy_cols(wf, 0, U), wf_att=ops.copy_cols(wf, U, C) if hp.use_concate else zeros(O * R, C),
        )
        # the shipped structure: two prenet blocks, two cells; other counts (decoder_sa.py:119-158, 357-369) fill the optional third slots / leave the
        # second ones empty and run the loop launch by launch on the fp32 operands (fcl_decoder_weights_t.prenet_layers / dlayers)
        self.generic_decoder = (hp.prenet_layers, hp.dlayers) != (2, 2) or R > 1
        for l in range(1, hp.prenet_layers):
            t["prenet_w%d" % l], t["prenet_b%d" % l] = g("dec.prenet.prenet.%d.0.weight" % l), g("dec.prenet.prenet.%d.0.bias" % l)
        for l in range(1, hp.dlayers):
            t["w%d_ih" % l], t["w%d_hh" % l] = lk(l, "weight_ih"), lk(l, "weight_hh")
            t["b%d" % l] = ops.add_vec(lk(l, "bias_ih"), lk(l, "bias_hh"))
        if self.generic_decoder:
            s.prenet_layers, s.dlayers, s.reduction_factor = hp.prenet_layers, hp.dlayers, R
        for k, v in t.items():
            setattr(s, k, v.data_ptr())
            d.keep.append(v)
        if self.generic_decoder:
            s.zoneout_rate = float(hp.zoneout_rate)
            s.prenet_dropout = float(hp.dropout_rate)
            s.out_act = output_act_code(hp)
            return d
        # fragment-major bf16x3 planes for the small-tile decoder kernels
        import os
        if os.environ.get("FCL_PRECISION", "1") != "0" and U % 16 == 0 and all(x % 8 == 0 for x in (C, P, O)):
            for k in ("prenet_w0", "prenet_w1", "w0_pre", "w0_hh", "w1_ih", "w1_hh", "wf_h"):
                hi, lo = ops.pack_frag_bf16(t[k])
                setattr(s, k + "_hi", hi.data_ptr())
                setattr(s, k + "_lo", lo.data_ptr())
                d.keep += [hi, lo]
            if ops.planes_enabled() and all(x % 32 == 0 for x in (C, P, U)):  # P32 planes for the LDS-DMA GEMM / LSTM-step kernels
                for k in ("w0_att", "wf_att", "w0_pre", "w0_hh", "w1_ih", "w1_hh"):
                    pl = ops.pack_planes(t[k])
                    setattr(s, k + "_p", pl.data_ptr())
                    d.keep.append(pl)
                # the step's weights in the consumption order of the persistent row-tile decoder kernel (csrc/decoder_tile.hip; 0 bytes: shape not covered)
                lib = _lib.load()
                nbytes = lib.fcl_decoder_stream_bytes(ctypes.byref(s))
                if nbytes:
                    stream = torch.empty(nbytes, device=self.device, dtype=torch.uint8)
                    ops.check(lib.fcl_decoder_stream_pack(ctypes.byref(s), stream.data_ptr(), nbytes, ops._stream()))
                    s.stream = stream.data_ptr()
                    d.keep.append(stream)
        if os.environ.get("FCL_PRECISION", "1") == "0" and U == 256 and P == 256 and 64 < O <= 96:
            # exact-fp32 mode: register-resident fp32 weight fragments for the fused feat/prenet launch (csrc/decoder_step.hip feat_prenet_split_f32_kernel)
            for k in ("wf_h", "prenet_w0", "prenet_w1", "w0_pre", "w0_hh", "w1_ih", "w1_hh"):
                ff = ops.pack_frag_f32(t[k])
                setattr(s, k + "_ff", ff.data_ptr())
                d.keep.append(ff)
        s.zoneout_rate = float(hp.zoneout_rate)
        s.prenet_dropout = float(hp.dropout_rate)
        s.out_act = output_act_code(hp)  # output_activation_fn on the fed-back frame (decoder_sa.py:614-617)
        return d
