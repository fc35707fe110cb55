"""fcl_oracle — CPU restatement of the FCL-taco2 hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file, and
only as the checker / the timed CPU baseline.  The product package (`fcl-taco2_amd/`) never imports it.

What it is: this build's own functional restatement, in plain `torch` CPU fp32 ops, of the reference's
mel-synthesis / teacher-forced forward maths, operating directly on a reference-named state_dict
(`{name: torch.Tensor}`).  Every function cites the reference lines it follows (paths relative to
/root/reference).  ESPnet helpers the reference calls (not vendored there; "espnet 8.0.0",
README.md:14) are restated from ESPnet v0.8's published source.

Pinning: the reference has no tests or golden vectors.  This oracle is pinned against
OUTPUTS OF THE REFERENCE ITSELF, produced in the survey container by `oracle/gen_golden.py` (which
imports /root/reference through `oracle/espnet_shim`) and committed under `tests/golden/`;
`tests/test_oracle_golden.py` checks every set.  The ESPnet boundary itself is unpinned by any
reference test ("parity unpinned" there): the goldens pin reference+shim, not ESPnet's own binaries.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default (reference encoder_sa.py:70, decoder_sa.py:214)
LN_EPS = 1e-12  # espnet LayerNorm (variance_predictor.py:62)


# ----------------------------------------------------------------------------- ESPnet helpers
def pad_list(xs, pad_value):
    """espnet nets_utils.pad_list: stack on new dim 0, right-pad dim 0 to the longest."""
    n = len(xs)
    m = max(x.size(0) for x in xs)
    out = xs[0].new_full((n, m) + tuple(xs[0].shape[1:]), pad_value)
    for i, x in enumerate(xs):
        out[i, : x.size(0)] = x
    return out


def make_pad_mask(lengths):
    """espnet nets_utils.make_pad_mask: bool [B, max(lengths)], True where idx >= length."""
    lengths = [int(l) for l in lengths]
    ar = torch.arange(max(lengths)).unsqueeze(0)
    return ar >= torch.tensor(lengths).unsqueeze(1)


def make_non_pad_mask(lengths):
    return ~make_pad_mask(lengths)


# ----------------------------------------------------------------------------- building blocks
def batch_norm_eval(y, sd, p):
    """Eval-mode BatchNorm1d on [B, C, T] with running statistics."""
    g, b = sd[p + ".weight"], sd[p + ".bias"]
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    return (y - rm[None, :, None]) / torch.sqrt(rv[None, :, None] + BN_EPS) * g[None, :, None] + b[None, :, None]


def batch_norm_train(y, sd, p):
    """Train-mode BatchNorm1d: batch statistics over (B, T) incl. padded positions."""
    mean = y.mean(dim=(0, 2), keepdim=True)
    var = y.var(dim=(0, 2), unbiased=False, keepdim=True)
    g, b = sd[p + ".weight"], sd[p + ".bias"]
    return (y - mean) / torch.sqrt(var + BN_EPS) * g[None, :, None] + b[None, :, None]


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(a)


def _drop(x, keep, p):
    """Inverted dropout with an explicit {0,1} keep mask (same maths as F.dropout(training=True))."""
    if keep is None:
        return x
    return x * keep.to(x.dtype) * (1.0 / (1.0 - p))


def encoder_convs(sd, x_bct, n_layers, bn_train=False, keeps=None, p=0.5, residual=False):
    """H2 — reference encoder_sa.py:61-78,136-140 / encoder_sa_kd.py:158-171.

    3 x {Conv1d(k5, pad 2, no bias) -> BatchNorm1d -> ReLU -> Dropout}.  Padded positions are NOT
    masked between layers (the "padding leak").  Returns the list of per-layer outputs."""
    taps = []
    x = x_bct
    for i in range(n_layers):
        w = sd["enc.convs.%d.0.weight" % i]
        y = F.conv1d(x, w, None, 1, (w.shape[2] - 1) // 2)
        if "enc.convs.%d.1.weight" % i in sd:  # `use_batch_norm` (encoder_sa.py:63-90): without it the block has no normalisation layer
            y = (batch_norm_train if bn_train else batch_norm_eval)(y, sd, "enc.convs.%d.1" % i)
        y = _drop(torch.relu(y), None if keeps is None else keeps[i], p)
        x = y + x if residual else y  # use_residual: convs[i](x) + x, after the block's ReLU and Dropout (encoder_sa_kd.py:158-171, 213-214)
        taps.append(x)
    return taps


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTMCell maths, gate order i,f,g,o (rows [0:U],[U:2U],[2U:3U],[3U:4U])."""
    gates = torch.addmm(b_ih + b_hh, x, w_ih.t()) + h @ w_hh.t()
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def blstm_packed(sd, x_btc, ilens, prefix="enc.blstm", layer=0):
    """H3 — one layer of the bidirectional nn.LSTM over pack_padded_sequence (encoder_sa.py:98-100,143-146).

    Restated as an explicit masked time loop: the forward direction runs t = 0..len-1, the reverse
    direction starts from zero state at t = len-1; outputs past each length are zero
    (pad_packed_sequence).  `elayers` > 1 (blstm_stack): layer l reads the [forward | reverse] outputs of layer l - 1
    (torch.nn.LSTM(num_layers=elayers, bidirectional=True): parameters `..._l<l>[_reverse]`)."""
    B, T, _ = x_btc.shape
    lens = torch.tensor([int(l) for l in ilens])
    outs = []
    for sfx, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
        w_ih, w_hh = sd[prefix + ".weight_ih_l%d%s" % (layer, sfx)], sd[prefix + ".weight_hh_l%d%s" % (layer, sfx)]
        b_ih, b_hh = sd[prefix + ".bias_ih_l%d%s" % (layer, sfx)], sd[prefix + ".bias_hh_l%d%s" % (layer, sfx)]
        H = w_hh.shape[1]
        h = x_btc.new_zeros(B, H)
        c = x_btc.new_zeros(B, H)
        out = x_btc.new_zeros(B, T, H)
        gx = x_btc @ w_ih.t() + (b_ih + b_hh)  # all steps at once
        for t in order:
            live = (t < lens).unsqueeze(1)
            gates = gx[:, t] + h @ w_hh.t()
            i, f, g, o = gates.chunk(4, dim=1)
            c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h2 = torch.sigmoid(o) * torch.tanh(c2)
            h = torch.where(live, h2, h)
            c = torch.where(live, c2, c)
            out[:, t] = torch.where(live, h2, torch.zeros_like(h2))
        outs.append(out)
    return torch.cat(outs, dim=2)


def blstm_stack(sd, x_btc, ilens, n_layers=1, prefix="enc.blstm"):
    """`elayers` stacked bidirectional layers (encoder_sa.py:96-100: nn.LSTM(iunits, eunits // 2, elayers, batch_first=True, bidirectional=True);
    no dropout between layers: the reference passes none)."""
    for l in range(n_layers):
        x_btc = blstm_packed(sd, x_btc, ilens, prefix, l)
    return x_btc


def encoder_forward(sd, hp, xs, ilens, bn_train=False, keeps=None):
    """Batched Encoder.forward (encoder_sa_kd.py:144-197): returns (enc_out [B,T,C], taps) where taps =
    [embed, conv0, conv1, conv2] as [B,T,C] (un-projected; student projections are applied by callers).
    keeps: optional per-layer dropout keep masks [B,T,C] (train form)."""
    emb = F.embedding(xs, sd["enc.embed.weight"], padding_idx=0)  # H1
    kt = None if keeps is None else [_t(k).transpose(1, 2) for k in keeps]
    taps = encoder_convs(sd, emb.transpose(1, 2), hp.econv_layers, bn_train, kt, hp.dropout_rate, residual=hp.use_residual)
    enc = blstm_stack(sd, taps[-1].transpose(1, 2), ilens, getattr(hp, "elayers", 1))
    return enc, [emb] + [t.transpose(1, 2) for t in taps]


def encoder_inference(sd, hp, x):
    """Encoder.inference (encoder_sa_kd.py:199-224): one utterance, no padding."""
    enc, _ = encoder_forward(sd, hp, x.unsqueeze(0), [x.numel()])
    return enc[0]


def predictor_trunk(sd, prefix, hs_btc, n_layers, keeps=None, p=0.0):
    """H4/H5 trunk — n x {Conv1d(k3,pad1,bias) -> ReLU -> LayerNorm(channels, eps 1e-12) -> Dropout}
    -> Linear(->1).  variance_predictor.py:48-66,86-90; ESPnet DurationPredictor._forward."""
    x = hs_btc.transpose(1, 2)
    for i in range(n_layers):
        w, b = sd["%s.conv.%d.0.weight" % (prefix, i)], sd["%s.conv.%d.0.bias" % (prefix, i)]
        x = torch.relu(F.conv1d(x, w, b, 1, (w.shape[2] - 1) // 2))
        x = F.layer_norm(x.transpose(1, 2), (w.shape[0],), sd["%s.conv.%d.2.weight" % (prefix, i)],
                         sd["%s.conv.%d.2.bias" % (prefix, i)], LN_EPS).transpose(1, 2)
        x = _drop(x, None if keeps is None else keeps[i], p)
    return F.linear(x.transpose(1, 2), sd[prefix + ".linear.weight"], sd[prefix + ".linear.bias"])  # [B,T,1]


def duration_predictor(sd, hp, hs_btc, pad_mask, inference=False, keeps=None):
    """H4 — ESPnet DurationPredictor.forward / .inference (call sites ..._kd_student.py:716,825).
    inference: clamp(round(exp(x) - 1.0), min=0).long(); torch.round is half-to-even."""
    kt = None if keeps is None else [_t(k).transpose(1, 2) for k in keeps]
    y = predictor_trunk(sd, "duration_predictor", hs_btc, hp.duration_predictor_layers, kt, hp.duration_predictor_dropout_rate).squeeze(-1)
    if inference:
        y = torch.clamp(torch.round(y.exp() - 1.0), min=0).long()
    if pad_mask is not None:
        y = y.masked_fill(pad_mask, 0.0)
    return y


def duration_round(logits):
    """The INT step of H4 on its own (used by the G4 crafted-logit golden)."""
    return torch.clamp(torch.round(logits.exp() - 1.0), min=0).long()


def variance_predictor(sd, hp, name, hs_btc, pad_mask, keeps=None):
    """H5 — VariancePredictor.forward (variance_predictor.py:74-95): [B,T,1], masked_fill 0."""
    kt = None if keeps is None else [_t(k).transpose(1, 2) for k in keeps]
    y = predictor_trunk(sd, name + "_predictor", hs_btc, hp.variance_predictor_layers, kt, hp.variance_predictor_dropout_rate)
    if pad_mask is not None:
        y = y.masked_fill(pad_mask.unsqueeze(-1), 0.0)
    return y


def variance_embed(sd, name, v_bt1, keep=None, p=0.5):
    """pitch_embed / energy_embed: Conv1d(1->C, k9, pad 4, bias) (+Dropout in train) — ..._kd_student.py:567-600."""
    w, b = sd[name + "_embed.0.weight"], sd[name + "_embed.0.bias"]
    return _drop(F.conv1d(v_bt1.transpose(1, 2), w, b, 1, (w.shape[2] - 1) // 2).transpose(1, 2), None if keep is None else _t(keep), p)


def position_table(ds_nonzero):
    """H10 position table — ..._kd_student.py:845-851: position[p, t] = t / d_p (fp32), zero padded."""
    rows = [torch.arange(int(d), dtype=torch.float32) / torch.tensor(int(d)) for d in ds_nonzero]
    return pad_list(rows, 0)


def prenet(sd, x, keep_pair=None, p=0.5, n_layers=2):
    """H6 — Prenet.forward (decoder_sa.py:119-158): n_layers x {Linear -> ReLU -> dropout(always on)} (2 in every shipped recipe).
    keep_pair: None = dropout disabled (rate 0), else n_layers {0,1} masks [N, P]."""
    for l in range(n_layers):
        x = torch.relu(F.linear(x, sd["dec.prenet.prenet.%d.0.weight" % l], sd["dec.prenet.prenet.%d.0.bias" % l]))
        if keep_pair is not None and p > 0:
            x = _drop(x, keep_pair[l], p)
    return x


def zoneout(old, new, rate, keep_old=None):
    """ZoneOutCell._zoneout (decoder_sa.py:82-96): eval = rate*old + (1-rate)*new; train = mask select
    (mask=1 keeps the OLD state, P(mask=1)=rate)."""
    if keep_old is None:
        return rate * old + (1 - rate) * new
    m = keep_old.to(old.dtype)
    return m * old + (1 - m) * new


def postnet(sd, hp, x_bcl, bn_train=False, keeps=None, p=0.5):
    """H11 — Postnet.forward (decoder_sa_kd.py:334-352): 5 x Conv1d(k5,no bias)+BN, tanh on all but
    the last, dropout train-only.  Returns the list of the 5 layer outputs."""
    outs = []
    x = x_bcl
    n = hp.postnet_layers
    for l in range(n):
        w = sd["dec.postnet.postnet.%d.0.weight" % l]
        y = F.conv1d(x, w, None, 1, (w.shape[2] - 1) // 2)
        if "dec.postnet.postnet.%d.1.weight" % l in sd:  # `use_batch_norm` (decoder_sa.py:203-263)
            y = (batch_norm_train if bn_train else batch_norm_eval)(y, sd, "dec.postnet.postnet.%d.1" % l)
        if l != n - 1:
            y = torch.tanh(y)
        x = _drop(y, None if keeps is None else keeps[l], p)
        outs.append(x)
    return outs


def _out_act(hp, x):
    """output_activation_fn (decoder_sa.py:353-360: getattr(torch.nn.functional, name)); None = identity."""
    name = getattr(hp, "output_activation", None)
    return x if name is None else getattr(F, name)(x)


def decoder_loop(sd, hp, att_c, position, n_steps, teacher_ys=None, prenet_keep=None, zone_keep=None):
    """H6-H8 hot loop — Decoder.inference decoder_sa_kd.py:742-778 (free-running, teacher_ys None) or
    Decoder.forward :572-625 (teacher forced: prev_out = y_t).

    att_c [N, C] (already hs + p_embs + e_embs), position [N, >=n_steps].
    prenet_keep: None (dropout off) or uint8 [n_steps, prenet_layers, N, P]; zone_keep: None (eval zoneout) or
    uint8 [n_steps, dlayers, 2(h,c), N, U].  `dlayers` cells (decoder_sa.py:357-369, 500-504: cell l > 0 reads cell l - 1's new state;
    feat_out reads the LAST cell), `prenet_layers` prenet blocks.
    Returns outs [N, odim, n_steps * reduction_factor], prenet_outs [N, n_steps, P], lstm0 [N, n_steps, U], last cell [N, n_steps, U]."""
    N = att_c.shape[0]
    U, zr = hp.dunits, hp.zoneout_rate
    # decoder_sa.py:366-369: the cell is wrapped in ZoneOutCell (parameters under `.cell`) only for a positive rate
    pat = "dec.lstm.%d.cell.%s" if zr > 0.0 else "dec.lstm.%d.%s"
    DL, PL = getattr(hp, "dlayers", 2), getattr(hp, "prenet_layers", 2)
    W = [[sd[pat % (l, k)] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")] for l in range(DL)]
    wf = sd["dec.feat_out.weight"]
    z = [att_c.new_zeros(N, U) for _ in range(DL)]
    c = [att_c.new_zeros(N, U) for _ in range(DL)]
    prev = att_c.new_zeros(N, hp.odim)
    outs, pres, l0, l1 = [], [], [], []
    for t in range(n_steps):
        if isinstance(prenet_keep, str):  # "rng": the reference's always-on F.dropout with fresh Bernoulli masks
            kp = [(torch.rand(N, hp.prenet_units) >= hp.dropout_rate) for _ in range(PL)] if hp.dropout_rate > 0 else None
        else:
            kp = None if prenet_keep is None else [_t(k) for k in prenet_keep[t]]
        pre = prenet(sd, prev, kp, hp.dropout_rate if kp is not None else 0.5, PL)
        pres.append(pre)
        base_cat = [att_c, pre]
        if getattr(hp, "append_position", True):  # decoder_sa.py:494-498 / :594-597
            base_cat.append(position[:, t].reshape(-1, 1))
        xs = torch.cat(base_cat, dim=1)
        for l in range(DL):
            inp = xs if l == 0 else z[l - 1]
            h2, c2 = lstm_cell(inp, z[l], c[l], *W[l])
            zk = None if zone_keep is None else zone_keep[t, l]
            z[l] = zoneout(z[l], h2, zr, None if zk is None else zk[0])
            c[l] = zoneout(c[l], c2, zr, None if zk is None else zk[1])
        l0.append(z[0])
        l1.append(z[-1])
        out = F.linear(torch.cat([z[-1], att_c], dim=1) if getattr(hp, "use_concate", True) else z[-1], wf)  # H8, no bias (decoder_sa.py:505-511)
        out = out.view(N, hp.odim, -1)  # [N, odim, reduction_factor]: a step emits r frames, feat_out row o * r + j = bin o of frame j (decoder_sa.py:512, 611)
        outs.append(out)
        # decoder_sa.py:513 / :614-617: teacher forcing feeds the ground-truth frame, free running the LAST of the step's r frames, activated
        prev = _out_act(hp, out[:, :, -1]) if teacher_ys is None else teacher_ys[:, t]
    return (torch.cat(outs, dim=2), torch.stack(pres, dim=1), torch.stack(l0, dim=1), torch.stack(l1, dim=1))


def decoder_inference(sd, hp, h, ds, p_embs, e_embs, prenet_keep=None):
    """Decoder.inference (decoder_sa_kd.py:707-800) for one utterance: returns (after [L,odim], before [L,odim])."""
    h = h + p_embs + e_embs
    ds = ds.view(-1)
    nz = ds[ds.ne(0)]
    if nz.shape[0] != h.shape[0]:
        # decoder_sa_kd.py:739 — a zero predicted duration crashes reference inference
        raise AssertionError("zero duration: ds_nonzeros.shape[0] != hs.shape[0]")
    r = getattr(hp, "reduction_factor", 1)
    position = position_table(nz)  # ..._sa.py:665-669: t / d in STEPS (the converter's training table is t / (r d): tts.py:256-258)
    n_steps = int(ds.max())
    outs, _, _, _ = decoder_loop(sd, hp, h, position, n_steps, None, prenet_keep)
    segs = [outs[p, :, : r * int(nz[p])] for p in range(h.shape[0])]  # H10 re-assembly (:782-790): r frames per step (decoder_sa.py:573, 627)
    before = torch.cat(segs, dim=-1).unsqueeze(0)  # [1, odim, L]
    after = _out_act(hp, before + postnet(sd, hp, before)[-1])  # decoder_sa.py:635-636 (`before` is not returned by the reference: kept raw)
    return after[0].t(), before[0].t()


def inference(sd, hp, x, dur=None, f0=None, energy=None, prenet_keep=None, spemb=None):
    """Tacotron2_sa.inference (..._kd_student.py:804-863 == teacher ..._sa.py:624-683).

    prenet_keep None => prenet dropout disabled (the golden sets G1/G2 run the reference with
    dropout_rate=0); otherwise the injected keep masks (G3).  Returns a dict of intermediates."""
    h = encoder_inference(sd, hp, x)
    if getattr(hp, "spk_embed_dim", None) is not None:  # ..._sa.py:636-638
        h = torch.cat([h, F.normalize(spemb, dim=0).unsqueeze(0).expand(h.size(0), -1)], dim=-1)
    pad = make_pad_mask([h.shape[0]])
    if dur is not None:
        d_outs = dur.reshape(-1).long()
    else:
        d_outs = duration_predictor(sd, hp, h.unsqueeze(0), pad, inference=True).squeeze(0).long()
    if f0 is not None:
        p_outs, e_outs = f0.unsqueeze(0), energy.unsqueeze(0)
    else:
        p_outs = variance_predictor(sd, hp, "pitch", h.unsqueeze(0), pad)
        e_outs = variance_predictor(sd, hp, "energy", h.unsqueeze(0), pad)
    p_embs = variance_embed(sd, "pitch", p_outs).squeeze(0)
    e_embs = variance_embed(sd, "energy", e_outs).squeeze(0)
    after, before = decoder_inference(sd, hp, h, d_outs, p_embs, e_embs, prenet_keep)
    return dict(h=h, d_outs=d_outs, p_outs=p_outs[0], e_outs=e_outs[0], p_embs=p_embs, e_embs=e_embs,
                before=before, after=after)


# ----------------------------------------------------------------------------- host batch layout (H15)
def convert_batch(xs, ys, ds, f0, energy, reduction_factor=1):
    """CustomConverter.__call__ (tts.py:215-306) restated with the same start/end arithmetic
    (start=int(sum(ds[:it])) * r, end=int(sum(ds[:it+1])) * r, :250-251; ds_nonzeros = int(d * r) :256; position = t / (end - start) :258).

    Inputs: lists of numpy arrays xs [T] int64, ys [L,odim] f32, ds [T,1] float, f0/energy [T,1] f32."""
    import numpy as np

    ilens = torch.tensor([x.shape[0] for x in xs]).long()
    olens = torch.tensor([y.shape[0] for y in ys]).long()
    new_ys, nzm, dsnz, pos = [], [], [], []
    for ib in range(len(xs)):
        d = np.asarray(ds[ib]).reshape(-1)
        csum = np.concatenate([[0.0], np.cumsum(d.astype(np.float64))])
        row = []
        for it in range(int(ilens[ib])):
            start, end = int(csum[it]) * reduction_factor, int(csum[it + 1]) * reduction_factor
            if start != end:
                new_ys.append(torch.from_numpy(ys[ib][start:end]).float())
                row.append(1)
                dsnz.append(int(d[it] * reduction_factor))
                pos.append(torch.arange(end - start, dtype=torch.float32) / (end - start))
            else:
                row.append(0)
        nzm.append(torch.tensor(row))
    batch = dict(
        xs=pad_list([torch.from_numpy(x).long() for x in xs], 0),
        ilens=ilens,
        ys=pad_list([torch.from_numpy(y).float() for y in ys], 0),
        olens=olens,
        extras=pad_list([torch.from_numpy(np.asarray(e)).float() for e in ds], 0),
        new_ys=pad_list(new_ys, 0),
        non_zero_lens_mask=pad_list(nzm, 0),
        ds_nonzeros=torch.tensor(dsnz),
        position=pad_list(pos, 0),
        f0=pad_list([torch.from_numpy(a).float() for a in f0], 0),
        energy=pad_list([torch.from_numpy(a).float() for a in energy], 0),
    )
    batch["output_masks"] = make_non_pad_mask(batch["ds_nonzeros"])
    return batch


# ----------------------------------------------------------------------------- teacher-forced forward
def decoder_forward(sd, hp, hs, olens, new_ys, non_zero_lens_mask, ds_nonzeros, output_masks, position,
                    p_embs, e_embs, prenet_keep=None, zone_keep=None, bn_train=False, post_keeps=None):
    """Decoder.forward (decoder_sa_kd.py:523-704), un-projected taps.

    Returns after [B,L,odim], before [B,L,odim], taps = [prenet [B,L,P], lstm0, lstm1 [B,L,U],
    conv0..conv4 [B,L,Cp|odim]]."""
    hs = hs + p_embs + e_embs
    att_c = hs[non_zero_lens_mask.eq(1)]  # H9 row compaction, row-major over (b, t)
    assert att_c.shape[0] == len(ds_nonzeros)
    r = getattr(hp, "reduction_factor", 1)
    # decoder_sa.py:487-489: every r-th frame (the LAST of each group of r) is the teacher-forced input of the next step; one step per group.
    # The converter's position table is indexed by the STEP (position[:, itt], :494-498): step t reads t / (r d)
    ys_in = new_ys[:, r - 1 :: r] if r > 1 else new_ys
    outs, pres, l0, l1 = decoder_loop(sd, hp, att_c, position, ys_in.shape[1], ys_in, prenet_keep, zone_keep)
    ylens = [int(l) for l in olens]

    def regroup(x_nlc):  # [N, Lseg, C] -> [B, L, C] : mask-select then split by ylens then pad (:634-655)
        flat = x_nlc[output_masks]
        segs, s = [], 0
        for l in ylens:
            segs.append(flat[s : s + l])
            s += l
        return pad_list(segs, 0)

    before = regroup(outs.transpose(1, 2))  # [B, L, odim]
    post = postnet(sd, hp, before.transpose(1, 2), bn_train, None if post_keeps is None else [_t(k).transpose(1, 2) for k in post_keeps],
                   hp.dropout_rate)
    after = _out_act(hp, before + post[-1].transpose(1, 2))  # decoder_sa.py:538-540 / decoder_sa_kd.py:698-700: both outputs, after the postnet
    before = _out_act(hp, before)
    # (the step-level taps are the KD classes' business; with r > 1 a step is r frames and the plain teacher class has no use for them)
    taps = ([regroup(pres), regroup(l0), regroup(l1)] if r == 1 else [None, None, None]) + [p.transpose(1, 2) for p in post]
    return after, before, taps


def _masked_mean_l1_mse(a, b, mask):
    if mask is not None:
        a, b = a.masked_select(mask), b.masked_select(mask)
    return (a - b).abs().mean(), ((a - b) ** 2).mean()


def taco2_loss(after, before, ys, olens, use_masking=True):
    """Tacotron2Loss (..._sa.py:26-82): L1(after)+L1(before), MSE(after)+MSE(before); use_masking False = means over the padded tensors."""
    m = make_non_pad_mask(olens).unsqueeze(-1) if use_masking else None
    l1a, ma = _masked_mean_l1_mse(after, ys, m)
    l1b, mb = _masked_mean_l1_mse(before, ys, m)
    return l1a + l1b, ma + mb


def knowledge_loss(student, teacher, lens):
    """Knowledge_loss (..._kd_student.py:134-179): sum over items of masked MSE."""
    loss = 0.0
    m = make_non_pad_mask(lens).unsqueeze(-1)
    for s, t in zip(student, teacher):
        loss = loss + _masked_mean_l1_mse(s, t, m)[1]
    return loss


def masks_from_sequence(seq, hp):
    """Names the keep masks a train-mode forward() of the reference consumes, in its call order (..._sa.py:553-590, decoder_sa.py:472-531):
    encoder conv dropouts, duration / pitch / energy predictor dropouts, pitch / energy embed dropouts, then per decoder step
    {prenet x2, zoneout layer0 (h, c), layer1 (h, c)}, then the postnet dropouts.  seq: list of uint8 arrays in the reference's own
    layouts ([B,C,T] for conv-side masks); returned in row-major [B,T,C] layouts."""
    import numpy as np

    it = iter(seq)
    bct = lambda: np.ascontiguousarray(np.transpose(next(it), (0, 2, 1)))
    m = {"enc.convs": [bct() for _ in range(hp.econv_layers)] if hp.dropout_rate > 0 else None}
    m["duration_predictor"] = [bct() for _ in range(hp.duration_predictor_layers)]
    m["pitch_predictor"] = [bct() for _ in range(hp.variance_predictor_layers)]
    m["energy_predictor"] = [bct() for _ in range(hp.variance_predictor_layers)]
    m["pitch_embed"], m["energy_embed"] = bct(), bct()
    rest = list(it)
    n_post = hp.postnet_layers if hp.dropout_rate > 0 else 0
    dec, post = rest[: len(rest) - n_post], rest[len(rest) - n_post:]
    DL, PL = getattr(hp, "dlayers", 2), getattr(hp, "prenet_layers", 2)
    per = (PL if hp.dropout_rate > 0 else 0) + 2 * DL
    assert len(dec) % per == 0
    steps = len(dec) // per
    pk, zk = [], []
    for t in range(steps):
        g = dec[t * per : (t + 1) * per]
        if hp.dropout_rate > 0:
            pk.append(np.stack(g[:PL]))
            g = g[PL:]
        zk.append(np.stack([np.stack(g[2 * l : 2 * l + 2]) for l in range(DL)]))
    m["prenet"] = np.stack(pk) if pk else None  # [steps, prenet_layers, N, P]
    m["zoneout"] = np.stack(zk)  # [steps, layer, (h, c), N, U]
    m["postnet"] = [np.ascontiguousarray(np.transpose(a, (0, 2, 1))) for a in post] if n_post else None
    return m


def model_forward(sd, hp, batch, role, teacher_hp=None, share_proj=True, teacher_knowledge=None,
                  prenet_keep=None, bn_train=False, masks=None):
    """Tacotron2_sa.forward.  Default: eval-dropout-off mode (all nn.Dropout inactive, zoneout eval form).
    masks (see masks_from_sequence) + bn_train=True: the train-mode graph with every Bernoulli draw injected.

    role: "teacher" (..._sa.py:520-622 -> dict of named losses), "kd_teacher"
    (..._kd_teacher.py:521-603 -> 5-tuple), "student" (..._kd_student.py:673-802 -> dict of losses)."""
    xs, ilens, ys, olens = batch["xs"], batch["ilens"], batch["ys"], batch["olens"]
    xs = xs[:, : int(max(ilens))]
    ys = ys[:, : int(max(olens))]
    mk = masks or {}
    hs, enc_taps = encoder_forward(sd, hp, xs, ilens, bn_train, mk.get("enc.convs"))
    hs_enc = hs  # the encoder-KD tap is the encoder's own output (encoder_sa_kd.py:178-188)
    if getattr(hp, "spk_embed_dim", None) is not None:  # ..._sa.py:555-557
        hs = torch.cat([hs, F.normalize(batch["spembs"]).unsqueeze(1).expand(-1, hs.size(1), -1)], dim=-1)
    ds = batch["extras"].squeeze(-1)
    pad = make_pad_mask(ilens)
    d_outs = duration_predictor(sd, hp, hs, pad, keeps=mk.get("duration_predictor"))  # log domain, masked_fill 0
    nonpad = ~pad
    dur_loss = ((d_outs.masked_select(nonpad) - torch.log(ds.masked_select(nonpad).float() + 1.0)) ** 2).mean()
    p_outs = variance_predictor(sd, hp, "pitch", hs, pad, mk.get("pitch_predictor"))
    e_outs = variance_predictor(sd, hp, "energy", hs, pad, mk.get("energy_predictor"))
    m1 = nonpad.unsqueeze(-1) if hp.use_masking else None  # prosody_criterions follows use_masking (..._sa.py:122-126); the duration loss does not
    pitch_loss = _masked_mean_l1_mse(p_outs, batch["f0"][:, : int(max(ilens))], m1)[1]
    energy_loss = _masked_mean_l1_mse(e_outs, batch["energy"][:, : int(max(ilens))], m1)[1]
    p_embs = variance_embed(sd, "pitch", batch["f0"], mk.get("pitch_embed"), hp.variance_embed_dropout_rate)  # ground-truth f0/energy feed the embeds
    e_embs = variance_embed(sd, "energy", batch["energy"], mk.get("energy_embed"), hp.variance_embed_dropout_rate)
    after, before, dec_taps = decoder_forward(
        sd, hp, hs, olens, batch["new_ys"], batch["non_zero_lens_mask"], batch["ds_nonzeros"],
        batch["output_masks"], batch["position"], p_embs, e_embs, prenet_keep if masks is None else mk.get("prenet"),
        None if mk.get("zoneout") is None else _t(mk["zoneout"]), bn_train, mk.get("postnet"))
    if role == "kd_teacher":
        return after, before, enc_taps + [hs_enc], dec_taps, [d_outs.unsqueeze(-1), p_outs, e_outs, p_embs, e_embs]
    if getattr(hp, "reduction_factor", 1) > 1:  # ..._sa.py:594-598: the target is cut to whole groups of r frames
        r = hp.reduction_factor
        olens = torch.tensor([int(o) - int(o) % r for o in olens])
        ys = ys[:, : int(max(olens))]
    l1, mse = taco2_loss(after, before, ys, olens, hp.use_masking)
    rep = dict(l1_loss=l1, mse_loss=mse, dur_loss=dur_loss, pitch_loss=pitch_loss, energy_loss=energy_loss)
    loss = l1 + mse + dur_loss + pitch_loss + energy_loss
    if role == "student":
        t_after, t_before, t_enc, t_dec, t_pro = teacher_knowledge
        lin = lambda x, k: F.linear(x, sd[k])
        if share_proj:
            cp = ["enc.convs_proj.0.weight"] * 3
            lp = ["dec.lstm_proj.weight"] * 2
            pp = ["dec.post_proj.weight"] * 4
        else:
            cp = ["enc.convs_proj.%d.weight" % i for i in range(3)]
            lp = ["dec.lstm0_proj.weight", "dec.lstm1_proj.weight"]
            pp = ["dec.post%d_proj.weight" % i for i in range(4)]
        s_enc = [lin(enc_taps[0], "enc.embed_proj.weight")] + [lin(enc_taps[1 + i], cp[i]) for i in range(3)] \
            + [lin(hs_enc, "enc.blstm_proj.weight")]
        s_dec = [lin(dec_taps[0], "dec.prenet_proj.weight"), lin(dec_taps[1], lp[0]), lin(dec_taps[2], lp[1])] \
            + [lin(dec_taps[3 + i], pp[i]) for i in range(4)] + [dec_taps[7]]
        s_pro = [d_outs.unsqueeze(-1), p_outs, e_outs, lin(p_embs, "pemb_proj.weight"), lin(e_embs, "eemb_proj.weight")]
        mo = make_non_pad_mask(olens).unsqueeze(-1) if hp.use_masking else None  # Tacotron2Loss_KD (..._kd_student.py:120-125)
        o_l1 = _masked_mean_l1_mse(after, t_after, mo)[0] + _masked_mean_l1_mse(before, t_before, mo)[0]
        o_mse = _masked_mean_l1_mse(after, t_after, mo)[1] + _masked_mean_l1_mse(before, t_before, mo)[1]
        enc_l = knowledge_loss(s_enc, t_enc, ilens)
        dec_l = knowledge_loss(s_dec, t_dec, olens)
        pro_l = knowledge_loss(s_pro, t_pro, ilens)
        rep.update(output_l1_loss=o_l1, output_mse_loss=o_mse, encoder_loss=enc_l, decoder_loss=dec_l, prosody_loss=pro_l)
        loss = loss + o_l1 + o_mse + enc_l + dec_l + pro_l
    rep["loss"] = loss
    rep["_after"], rep["_before"] = after, before
    return rep


def synthesize_batch(sd, hp, xs, ds, prenet_keep_fn=None):
    """Batched free-running synthesis == B independent inference() calls.
    This is also the workload timed as bench.py's cpu_baseline ("port")."""
    mels = []
    for i, (x, d) in enumerate(zip(xs, ds)):
        keep = None if prenet_keep_fn is None else prenet_keep_fn(i, x, d)
        mels.append(inference(sd, hp, x, dur=d, prenet_keep=keep)["after"])
    return mels
