"""Teacher-forced training step on the HIP path: forward with saved activations, backward, clip + Adam.

Restates the reference's `CustomUpdater.update_core` (tts.py:137-179; KD: tts_distill.py:143-182):
    loss = model(**batch).mean() / accum_grad ; loss.backward() ; clip_grad_norm_(params, grad_clip) ; skip if NaN ; Adam.step()
with every FLOP in libfcl_hip.so (forward GEMMs reuse the synthesis kernels; gradients use the primitives of backward.hip).
torch supplies device memory only; there is no autograd graph.

Regulariser semantics implemented so far — the EVALUATION form of every stochastic layer, i.e. exactly the graph the
reference builds under `model.eval()` (BatchNorm = affine map with running statistics, nn.Dropout off, zoneout in its
expectation form decoder_sa.py:96, prenet dropout through explicit masks or off).  This is the configuration the real
reference's gradients are pinned on (tests/golden/g5_teacher_train.npz).  Train-mode BatchNorm statistics and sampled
dropout / zoneout masks are the next increment.

Layouts: encoder rows (b, t) -> b*T + t, frame rows (b, l) -> b*L + l (both zero padded like the reference's batch), decoder
cells in STEP-MAJOR order: phoneme rows sorted by duration (descending), cell (t, m) at offset[t] + m with offset[t] = sum of
live rows of the earlier steps, so the live rows of every step are one contiguous slice of every saved tensor.
"""
import numpy as np
import torch

from . import ops
from .plan import BN_EPS, LN_EPS


def _i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


class TrainEngine(object):
    def __init__(self, model, lr=1e-3, eps=1e-6, betas=(0.9, 0.999), grad_clip=1.0):
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise RuntimeError("fcl-taco2_amd: TrainEngine needs the model on a GPU (no CPU fallback)")
        if model.role != "teacher":
            raise NotImplementedError("fcl-taco2_amd: TrainEngine covers the teacher step; the KD student step is the next increment")
        self.model, self.hp, self.dev = model, model.hp, p0.device
        self.P = {k: v.data for k, v in model.named_parameters()}  # master weights: the module's own storage
        self.B = {k: v for k, v in model.named_buffers()}
        self.G = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.lr, self.eps, self.betas, self.grad_clip = lr, eps, betas, grad_clip
        self.step_count = 0
        self.gn_sq = torch.zeros(1, dtype=torch.float64, device=self.dev)

    # ------------------------------------------------------------------------------------------------ layers
    def _wt(self, w):
        return ops.transpose2d(w.contiguous())

    def _conv_pack(self, w, scale=None):
        wp = ops.pack_conv1d_weight(w, scale)  # [k, Cout, Cin]
        k = wp.shape[0]
        wt = torch.stack([ops.transpose2d(wp[k - 1 - j]) for j in range(k)]).contiguous()  # [k, Cin, Cout], taps reversed
        return wp, wt

    def _conv_bn_fwd(self, x, prefix, lo, hi, act):
        """Conv1d(no bias) -> eval BatchNorm folded -> act.  Keeps z (pre-activation) for the gamma gradient."""
        P, B = self.P, self.B
        scale, shift = ops.fold_batchnorm(P[prefix + ".1.weight"], P[prefix + ".1.bias"], B[prefix + ".1.running_mean"], B[prefix + ".1.running_var"], BN_EPS)
        wp, wt = self._conv_pack(P[prefix + ".0.weight"], scale)
        z = ops.conv1d(x, wp, shift, lo, hi, ops.ACT_NONE)
        y = ops.act_fwd(z, act) if act != ops.ACT_NONE else z
        return y, dict(x=x, z=z, y=y, wt=wt, scale=scale, prefix=prefix, act=act, lo=lo, hi=hi)

    def _conv_bn_bwd(self, dy, c):
        G, P = self.G, self.P
        pre = c["prefix"]
        dz = ops.act_bwd(dy, c["y"], c["act"]) if c["act"] != ops.ACT_NONE else dy
        ops.colsum(dz, G[pre + ".1.bias"])
        ops.colsum(dz, G[pre + ".1.weight"], y=c["z"], gamma=P[pre + ".1.weight"], beta=P[pre + ".1.bias"], mode=2)
        w = P[pre + ".0.weight"]
        cout, cin, k = w.shape
        dwp = torch.zeros(k, cout, cin, device=self.dev)
        for j in range(k):
            ops.gemm_tn(dz, c["x"], dwp[j], shift=j - (k - 1) // 2, seg_lo=c["lo"], seg_hi=c["hi"])
        ops.unpack_conv1d_grad(dwp, G[pre + ".0.weight"], c["scale"])
        return ops.conv1d(dz, c["wt"], None, c["lo"], c["hi"])

    def _conv_bias_relu_fwd(self, x, prefix, lo, hi):
        wp, wt = self._conv_pack(self.P[prefix + ".weight"])
        y = ops.conv1d(x, wp, self.P[prefix + ".bias"], lo, hi, ops.ACT_RELU)
        return y, dict(x=x, y=y, wt=wt, prefix=prefix, lo=lo, hi=hi)

    def _conv_bias_relu_bwd(self, dy, c):
        G = self.G
        pre = c["prefix"]
        dz = ops.act_bwd(dy, c["y"], ops.ACT_RELU)
        ops.colsum(dz, G[pre + ".bias"])
        cout, cin, k = self.P[pre + ".weight"].shape
        dwp = torch.zeros(k, cout, cin, device=self.dev)
        for j in range(k):
            ops.gemm_tn(dz, c["x"], dwp[j], shift=j - (k - 1) // 2, seg_lo=c["lo"], seg_hi=c["hi"])
        ops.unpack_conv1d_grad(dwp, G[pre + ".weight"])
        return ops.conv1d(dz, c["wt"], None, c["lo"], c["hi"])

    def _predictor_fwd(self, hs, name, layers, lo, hi, pad):
        caches, x = [], hs
        for i in range(layers):
            y, cc = self._conv_bias_relu_fwd(x, "%s.conv.%d.0" % (name, i), lo, hi)
            last = i == layers - 1
            g, b = self.P["%s.conv.%d.2.weight" % (name, i)], self.P["%s.conv.%d.2.bias" % (name, i)]
            ln, out = ops.layernorm(y, g, b, LN_EPS, want_y=not last, lin_w=self.P[name + ".linear.weight"].reshape(-1) if last else None,
                                    lin_b=self.P[name + ".linear.bias"] if last else None, pad_mask=pad if last else None)
            caches.append((cc, y, i, last))
            x = ln
        return out, caches

    def _predictor_bwd(self, d_out, name, caches, pad):
        G, P = self.G, self.P
        dx = None
        for cc, y, i, last in reversed(caches):
            g, b = P["%s.conv.%d.2.weight" % (name, i)], P["%s.conv.%d.2.bias" % (name, i)]
            kw = dict(dgamma=G["%s.conv.%d.2.weight" % (name, i)], dbeta=G["%s.conv.%d.2.bias" % (name, i)])
            if last:
                dy = ops.layernorm_bwd(y, g, b, LN_EPS, lin_w=P[name + ".linear.weight"].reshape(-1), ds=d_out, pad_mask=pad,
                                       dlin_w=G[name + ".linear.weight"].reshape(-1), dlin_b=G[name + ".linear.bias"], **kw)
            else:
                dy = ops.layernorm_bwd(y, g, b, LN_EPS, dy=dx, **kw)
            dx = self._conv_bias_relu_bwd(dy, cc)
        return dx

    def _linear_bwd(self, dz, x, wname, bname=None):
        """dW += dz^T x ; db += colsum(dz) ; returns dz . W."""
        ops.gemm_tn(dz, x, self.G[wname])
        if bname:
            ops.colsum(dz, self.G[bname])
        return ops.linear(dz, self._wt(self.P[wname]))

    # ------------------------------------------------------------------------------------------------ BiLSTM (per-step, saved)
    def _bilstm_fwd(self, x, lens_dev, B, T):
        P, dev = self.P, self.dev
        H = self.hp.eunits // 2
        out = torch.zeros(B * T, 2 * H, device=dev)
        cache = []
        for d, sfx in enumerate(("", "_reverse")):
            w_ih, w_hh = P["enc.blstm.weight_ih_l0" + sfx], P["enc.blstm.weight_hh_l0" + sfx]
            bias = ops.add_vec(P["enc.blstm.bias_ih_l0" + sfx], P["enc.blstm.bias_hh_l0" + sfx])
            gx = ops.linear(x, w_ih, bias)  # [B*T, 4H]
            h = [torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)]
            c = torch.zeros(B, H, device=dev)
            sv = [torch.empty(T, B, 4 * H, device=dev)] + [torch.empty(T, B, H, device=dev) for _ in range(3)]  # gates, c_new, c_old, h_old
            cur = 0
            order = range(T) if d == 0 else range(T - 1, -1, -1)
            for t in order:
                ops.lstm_step([(h[cur], w_hh, H)], B, H, h[cur], h[cur ^ 1], c, G=gx, g_row_mul=T, g_row_add=t, step=t, row_len=lens_dev,
                              out2=out, out2_row_mul=T, out2_row_add=t, ld2=2 * H, out2_col_off=d * H, save=[s[t] for s in sv])
                cur ^= 1
            cache.append((sv, order))
        return out, dict(x=x, dirs=cache, B=B, T=T, lens=lens_dev)

    def _bilstm_bwd(self, d_out, c):
        P, G, dev = self.P, self.G, self.dev
        B, T, H = c["B"], c["T"], self.hp.eunits // 2
        dx = torch.zeros_like(c["x"])
        d3 = d_out.reshape(B, T, 2 * H)
        for d, sfx in enumerate(("", "_reverse")):
            sv, order = c["dirs"][d]
            w_ih, w_hh = P["enc.blstm.weight_ih_l0" + sfx], P["enc.blstm.weight_hh_l0" + sfx]
            whh_t = self._wt(w_hh)  # [H, 4H]
            dgx = torch.zeros(B * T, 4 * H, device=dev)
            dh_carry, dc_carry = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
            dgx3 = dgx.reshape(B, T, 4 * H)
            for t in reversed(list(order)):
                ops.add2d(dh_carry, d3[:, t, d * H : (d + 1) * H])  # d_out is already zero on padded rows (masked by the caller)
                dgates, dh_old, dc_old = ops.lstm_cell_bwd(sv[0][t], sv[2][t], sv[1][t], dh_carry, dc_carry, 0.0, row_len=c["lens"], step=t)
                dh_carry = ops.add2d(ops.linear(dgates, whh_t), dh_old)
                dc_carry = dc_old
                ops.gemm_tn(dgates, sv[3][t], G["enc.blstm.weight_hh_l0" + sfx])
                ops.copy2d(dgx3[:, t], dgates)  # Gx rows are (b, t)
            ops.gemm_tn(dgx, c["x"], G["enc.blstm.weight_ih_l0" + sfx])
            ops.colsum(dgx, G["enc.blstm.bias_ih_l0" + sfx])
            ops.colsum(dgx, G["enc.blstm.bias_hh_l0" + sfx])
            ops.add2d(dx, ops.linear(dgx, self._wt(w_ih)))
        return dx

    # ------------------------------------------------------------------------------------------------ the step
    def zero_grad(self):
        for g in self.G.values():
            g.zero_()

    def forward_backward(self, batch, prenet_keep=None):
        """Teacher loss (…_sa.py:601-613) and all parameter gradients into self.G.  Returns the named losses (floats)."""
        hp, dev, P, G = self.hp, self.dev, self.P, self.G
        ilens = [int(v) for v in batch["ilens"]]
        olens = [int(v) for v in batch["olens"]]
        B, T, L = len(ilens), max(ilens), max(olens)
        O, U, Pn, C = hp.odim, hp.dunits, hp.prenet_units, hp.eunits
        drop_p = hp.dropout_rate if prenet_keep is not None else 0.0
        kscale = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        with torch.cuda.device(dev):
            self.zero_grad()
            # ---- index maps (host, integers) ---------------------------------------------------------------
            rows = np.arange(B * T)
            b_of = rows // T
            lens_np = np.asarray(ilens)
            e_lo, e_hi = _i32(b_of * T, dev), _i32(b_of * T + T, dev)
            pad_np = (rows % T) >= lens_np[b_of]
            enc_pad = torch.from_numpy(pad_np.astype(np.uint8)).to(dev)
            enc_valid = torch.from_numpy((~pad_np).astype(np.uint8)).to(dev)
            frows = np.arange(B * L)
            f_lo, f_hi = _i32((frows // L) * L, dev), _i32((frows // L) * L + L, dev)
            fvalid_np = (frows % L) < np.asarray(olens)[frows // L]
            frame_valid = torch.from_numpy(fvalid_np.astype(np.uint8)).to(dev)
            nzm = np.asarray(batch["non_zero_lens_mask"])[:, :T] != 0
            dsn = np.asarray(batch["ds_nonzeros"]).astype(np.int64)
            src = np.flatnonzero(nzm.reshape(-1))
            N = src.shape[0]
            b_row = src // T
            excl = np.cumsum(dsn) - dsn
            first = np.concatenate([[0], np.cumsum(np.bincount(b_row, minlength=B))[:-1]])
            foff = b_row * L + (excl - excl[first[b_row]])
            order = np.argsort(-dsn, kind="stable")
            dur_s = dsn[order]
            foff_s = foff[order]
            lmax = int(dur_s[0])
            live = (dur_s[None, :] > np.arange(lmax)[:, None]).sum(1)
            offs = np.concatenate([[0], np.cumsum(live)])  # step-major cell offsets
            F = int(offs[-1])
            cell_row = np.concatenate([np.arange(n) for n in live])  # sorted-row index of every cell
            cell_t = np.repeat(np.arange(lmax), live)
            cell_frame = foff_s[cell_row] + cell_t  # frame row (b*L + l) of every cell
            frame_cell = np.full(B * L, -1, dtype=np.int64)
            frame_cell[cell_frame] = np.arange(F)
            prev_frame = np.where(cell_t > 0, cell_frame - 1, -1)  # teacher-forced input y_{t-1}; zero row at t = 0

            # ---- encoder ---------------------------------------------------------------------------------------
            xs = batch["xs"][:, :T].to(dev).to(torch.int64).reshape(-1).contiguous()
            emb = ops.embedding(xs, P["enc.embed.weight"])
            x, conv_c = emb, []
            for i in range(hp.econv_layers):
                x, cc = self._conv_bn_fwd(x, "enc.convs.%d" % i, e_lo, e_hi, ops.ACT_RELU)
                conv_c.append(cc)
            lens_dev = _i32(lens_np, dev)
            hs, bl_c = self._bilstm_fwd(x, lens_dev, B, T)
            # ---- predictors + embeds -----------------------------------------------------------------------------
            d_outs, dur_c = self._predictor_fwd(hs, "duration_predictor", hp.duration_predictor_layers, e_lo, e_hi, enc_pad)
            p_outs, pit_c = self._predictor_fwd(hs, "pitch_predictor", hp.variance_predictor_layers, e_lo, e_hi, enc_pad)
            e_outs, en_c = self._predictor_fwd(hs, "energy_predictor", hp.variance_predictor_layers, e_lo, e_hi, enc_pad)
            f0 = batch["f0"][:, :T].to(dev).float().reshape(-1).contiguous()
            en = batch["energy"][:, :T].to(dev).float().reshape(-1).contiguous()
            ds = batch["extras"][:, :T].to(dev).float().reshape(-1).contiguous()
            kk = hp.variance_embed_kernel_size
            att, _, _ = ops.variance_embed_add(hs, f0, en, P["pitch_embed.0.weight"].reshape(C, kk), P["pitch_embed.0.bias"],
                                               P["energy_embed.0.weight"].reshape(C, kk), P["energy_embed.0.bias"], e_lo, e_hi)
            # ---- decoder, teacher forced, step-major cells ---------------------------------------------------------
            att_c = ops.gather_rows(att, _i32(src[order], dev))  # [N, C] sorted rows
            ys = batch["ys"][:, :L].to(dev).float().reshape(B * L, O).contiguous()
            pre_in = ops.gather_rows(ys, _i32(prev_frame, dev))  # [F, O]; idx -1 -> zero row
            k0 = k1 = None
            if prenet_keep is not None:  # [lmax, 2, N(compact order), P] -> cells
                pk = np.asarray(prenet_keep)[:lmax][:, :, order, :]
                k0 = torch.from_numpy(np.ascontiguousarray(np.concatenate([pk[t, 0, : live[t]] for t in range(lmax)]))).to(dev)
                k1 = torch.from_numpy(np.ascontiguousarray(np.concatenate([pk[t, 1, : live[t]] for t in range(lmax)]))).to(dev)
            w0n, b0n, w1n, b1n = ["dec.prenet.prenet.%d.0.%s" % (l, s) for l in (0, 1) for s in ("weight", "bias")]
            p0 = ops.linear(pre_in, P[w0n], P[b0n], ops.ACT_RELU)  # pre-dropout activations are kept
            p0d = ops.act_fwd(p0, ops.ACT_NONE, k0, kscale) if k0 is not None else p0
            p1 = ops.linear(p0d, P[w1n], P[b1n], ops.ACT_RELU)
            p1d = ops.act_fwd(p1, ops.ACT_NONE, k1, kscale) if k1 is not None else p1
            w_ih0 = P["dec.lstm.0.cell.weight_ih"]
            w0_att, w0_pre = ops.copy_cols(w_ih0, 0, C), ops.copy_cols(w_ih0, C, Pn)
            w0_pos = ops.copy_cols(w_ih0, C + Pn, 1).reshape(-1)
            w0_hh = P["dec.lstm.0.cell.weight_hh"]
            b0s = ops.add_vec(P["dec.lstm.0.cell.bias_ih"], P["dec.lstm.0.cell.bias_hh"])
            w1_ih, w1_hh = P["dec.lstm.1.cell.weight_ih"], P["dec.lstm.1.cell.weight_hh"]
            b1s = ops.add_vec(P["dec.lstm.1.cell.bias_ih"], P["dec.lstm.1.cell.bias_hh"])
            wf = P["dec.feat_out.weight"]
            wf_h, wf_att = ops.copy_cols(wf, 0, U), ops.copy_cols(wf, U, C)
            G0 = ops.linear(att_c, w0_att, b0s)  # hoisted att_c share of the layer-0 gates
            F0 = ops.linear(att_c, wf_att)
            dur_dev = _i32(dur_s, dev)
            S0 = [torch.empty(F, 4 * U, device=dev)] + [torch.empty(F, U, device=dev) for _ in range(3)]  # gates, c_new, c_old, h_old
            S1 = [torch.empty(F, 4 * U, device=dev)] + [torch.empty(F, U, device=dev) for _ in range(3)]
            h0_all, h1_all = torch.empty(F, U, device=dev), torch.empty(F, U, device=dev)  # zoneout-ed outputs per cell
            h0 = [torch.zeros(N, U, device=dev), torch.zeros(N, U, device=dev)]
            h1 = [torch.zeros(N, U, device=dev), torch.zeros(N, U, device=dev)]
            c0, c1 = torch.zeros(N, U, device=dev), torch.zeros(N, U, device=dev)
            zr = float(hp.zoneout_rate)
            cur = 0
            for t in range(lmax):
                n, o = int(live[t]), int(offs[t])
                sl = slice(o, o + n)
                ops.lstm_step([(p1d[sl], w0_pre, Pn), (h0[cur], w0_hh, U)], n, U, h0[cur], h0[cur ^ 1], c0, G=G0, rank1_w=w0_pos, dur=dur_dev,
                              step=t, zoneout=zr, out2=h0_all[sl], out2_row_mul=1, ld2=U, save=[s[sl] for s in S0])
                ops.lstm_step([(h0[cur ^ 1], w1_ih, U), (h1[cur], w1_hh, U)], n, U, h1[cur], h1[cur ^ 1], c1, bias=b1s, step=t, zoneout=zr,
                              out2=h1_all[sl], out2_row_mul=1, ld2=U, save=[s[sl] for s in S1])
                cur ^= 1
            cell_row_dev = _i32(cell_row, dev)
            F0_cells = ops.gather_rows(F0, cell_row_dev)
            out_cells = ops.linear(h1_all, wf_h)
            ops.add2d(out_cells, F0_cells)
            before = ops.gather_rows(out_cells, _i32(frame_cell, dev))  # [B*L, O], zero where no cell maps (padding)
            # ---- postnet ----------------------------------------------------------------------------------------------
            x, post_c = before, []
            n_post = hp.postnet_layers
            for i in range(n_post):
                x, cc = self._conv_bn_fwd(x, "dec.postnet.postnet.%d" % i, f_lo, f_hi, ops.ACT_NONE if i == n_post - 1 else ops.ACT_TANH)
                post_c.append(cc)
            after = ops.add_vec(before, x)
            # ---- losses (Tacotron2Loss + duration + pitch + energy) -----------------------------------------------------
            nf, ne = float(fvalid_np.sum()) * O, float((~pad_np).sum())
            sums = torch.zeros(5, 3, dtype=torch.float64, device=dev)
            ops.masked_l1_mse(after, ys, frame_valid, sums[0])
            ops.masked_l1_mse(before, ys, frame_valid, sums[1])
            ops.masked_l1_mse(d_outs.reshape(-1, 1), ds.reshape(-1, 1), enc_valid, sums[2], b_log_offset=1.0)
            ops.masked_l1_mse(p_outs.reshape(-1, 1), f0.reshape(-1, 1), enc_valid, sums[3])
            ops.masked_l1_mse(e_outs.reshape(-1, 1), en.reshape(-1, 1), enc_valid, sums[4])
            # ================================================= backward =================================================
            d_after = ops.l1_mse_grad(after, ys, frame_valid, nf, 1.0, 1.0)
            d_before = ops.l1_mse_grad(before, ys, frame_valid, nf, 1.0, 1.0)
            ops.add2d(d_before, d_after)  # residual path of after = before + postnet(before)
            dx = d_after
            for cc in reversed(post_c):
                dx = self._conv_bn_bwd(dx, cc)
            ops.add2d(d_before, dx)
            # ---- decoder BPTT ----------------------------------------------------------------------------------------------
            d_out_cells = ops.gather_rows(d_before, _i32(cell_frame, dev))  # [F, O]
            g_wf = G["dec.feat_out.weight"]
            ops.gemm_tn(d_out_cells, h1_all, g_wf[:, :U])  # column blocks of the [odim, U + C] gradient are written in place
            dh1_all = ops.linear(d_out_cells, self._wt(wf_h))  # [F, U]
            dF0 = torch.zeros(N, O, device=dev)
            cell_row64 = torch.from_numpy(cell_row.astype(np.int64)).to(dev)
            ops.scatter_add_rows(d_out_cells, cell_row64, dF0)
            ops.gemm_tn(dF0, att_c, g_wf[:, U:])
            d_att_c = ops.linear(dF0, self._wt(wf_att))
            w1ih_t, w1hh_t, w0hh_t, w0pre_t = self._wt(w1_ih), self._wt(w1_hh), self._wt(w0_hh), self._wt(w0_pre)
            dg0_all, dg1_all = torch.empty(F, 4 * U, device=dev), torch.empty(F, 4 * U, device=dev)
            dp1_all = torch.empty(F, Pn, device=dev)
            ch0, cc0 = torch.zeros(N, U, device=dev), torch.zeros(N, U, device=dev)  # carries: grads w.r.t. the state entering step t+1
            ch1, cc1 = torch.zeros(N, U, device=dev), torch.zeros(N, U, device=dev)
            tmp_h, tmp_c = torch.empty(N, U, device=dev), torch.zeros(N, U, device=dev)
            for t in range(lmax - 1, -1, -1):  # live rows only grow as t falls, so carries of newly-live rows are still zero
                n, o = int(live[t]), int(offs[t])
                sl = slice(o, o + n)
                ops.add2d(ch1[:n], dh1_all[sl])
                ops.lstm_cell_bwd(S1[0][sl], S1[2][sl], S1[1][sl], ch1[:n], cc1[:n], zr, out=(dg1_all[sl], tmp_h[:n], tmp_c[:n]))
                ops.linear(dg1_all[sl], w1hh_t, out=ch1[:n])
                ops.add2d(ch1[:n], tmp_h[:n])
                cc1, tmp_c = tmp_c, cc1
                ops.add2d(ch0[:n], ops.linear(dg1_all[sl], w1ih_t))
                ops.lstm_cell_bwd(S0[0][sl], S0[2][sl], S0[1][sl], ch0[:n], cc0[:n], zr, out=(dg0_all[sl], tmp_h[:n], tmp_c[:n]))
                ops.linear(dg0_all[sl], w0hh_t, out=ch0[:n])
                ops.add2d(ch0[:n], tmp_h[:n])
                cc0, tmp_c = tmp_c, cc0
                ops.linear(dg0_all[sl], w0pre_t, out=dp1_all[sl])
            # weight gradients of the two cells from the saved step-major tensors (one TN GEMM each)
            ops.gemm_tn(dg1_all, h0_all, G["dec.lstm.1.cell.weight_ih"])
            ops.gemm_tn(dg1_all, S1[3], G["dec.lstm.1.cell.weight_hh"])
            for nm in ("bias_ih", "bias_hh"):
                ops.colsum(dg1_all, G["dec.lstm.1.cell." + nm])
                ops.colsum(dg0_all, G["dec.lstm.0.cell." + nm])
            ops.gemm_tn(dg0_all, S0[3], G["dec.lstm.0.cell.weight_hh"])
            g_ih0 = G["dec.lstm.0.cell.weight_ih"]  # [4U, C + P + 1] = [att_c | prenet | position]
            ops.gemm_tn(dg0_all, p1d, g_ih0[:, C : C + Pn])
            pos_np = np.zeros((F, 4), dtype=np.float32)
            pos_np[:, 0] = cell_t.astype(np.float32) / dur_s[cell_row].astype(np.float32)  # the position input t/d, padded to 4 columns
            dw0_pos4 = torch.zeros(4 * U, 4, device=dev)
            ops.gemm_tn(dg0_all, torch.from_numpy(pos_np).to(dev), dw0_pos4)
            ops.add2d(g_ih0[:, C + Pn :], dw0_pos4[:, :1])
            dG0 = torch.zeros(N, 4 * U, device=dev)
            ops.scatter_add_rows(dg0_all, cell_row64, dG0)
            ops.gemm_tn(dG0, att_c, g_ih0[:, :C])
            ops.add2d(d_att_c, ops.linear(dG0, self._wt(w0_att)))
            # prenet (batched over all cells)
            dz1 = ops.act_bwd(dp1_all, p1, ops.ACT_RELU, k1, kscale)
            dp0 = self._linear_bwd(dz1, p0d, w1n, b1n)
            dz0 = ops.act_bwd(dp0, p0, ops.ACT_RELU, k0, kscale)
            ops.gemm_tn(dz0, pre_in, G[w0n])
            ops.colsum(dz0, G[b0n])
            # ---- att = hs + p_emb + e_emb -----------------------------------------------------------------------------------
            inv = np.full(B * T, -1, dtype=np.int64)
            inv[src[order]] = np.arange(N)
            d_att = ops.gather_rows(d_att_c, _i32(inv, dev))  # scatter back to (b, t) rows; rows without a phoneme get 0
            d_hs = d_att.clone()
            for nm, sig in (("pitch", f0), ("energy", en)):
                ops.colsum(d_att, G[nm + "_embed.0.bias"])
                sig4 = torch.zeros(B * T, 4, device=dev)
                ops.copy2d(sig4[:, :1], sig.reshape(-1, 1))
                gw = G[nm + "_embed.0.weight"].reshape(C, kk)
                for j in range(kk):
                    tmp = torch.zeros(C, 4, device=dev)
                    ops.gemm_tn(d_att, sig4, tmp, shift=j - (kk - 1) // 2, seg_lo=e_lo, seg_hi=e_hi)
                    ops.add2d(gw[:, j : j + 1], tmp[:, :1])
            # ---- predictors ----------------------------------------------------------------------------------------------------
            d_d = ops.l1_mse_grad(d_outs, ds, enc_valid, ne, 0.0, 1.0, b_log_offset=1.0).reshape(-1)
            d_p = ops.l1_mse_grad(p_outs, f0, enc_valid, ne, 0.0, 1.0).reshape(-1)
            d_e = ops.l1_mse_grad(e_outs, en, enc_valid, ne, 0.0, 1.0).reshape(-1)
            ops.add2d(d_hs, self._predictor_bwd(d_d, "duration_predictor", dur_c, enc_pad))
            ops.add2d(d_hs, self._predictor_bwd(d_p, "pitch_predictor", pit_c, enc_pad))
            ops.add2d(d_hs, self._predictor_bwd(d_e, "energy_predictor", en_c, enc_pad))
            # ---- encoder -----------------------------------------------------------------------------------------------------------
            d_hs_live = ops.add2d(torch.zeros_like(d_hs), d_hs, row_valid=enc_valid)  # pad_packed_sequence: padded outputs are constants
            dx = self._bilstm_bwd(d_hs_live, bl_c)
            for cc in reversed(conv_c):
                dx = self._conv_bn_bwd(dx, cc)
            ops.scatter_add_rows(dx, xs, G["enc.embed.weight"], skip=0)  # padding_idx = 0 gets no gradient
            host = sums.cpu().numpy()
        l1 = host[0, 0] / host[0, 2] + host[1, 0] / host[1, 2]
        mse = host[0, 1] / host[0, 2] + host[1, 1] / host[1, 2]
        rep = dict(l1_loss=l1, mse_loss=mse, dur_loss=host[2, 1] / host[2, 2], pitch_loss=host[3, 1] / host[3, 2], energy_loss=host[4, 1] / host[4, 2])
        rep["loss"] = sum(rep.values())
        return rep

    def optimizer_step(self):
        """clip_grad_norm_(grad_clip) + NaN guard + Adam, all on the stream (tts.py:173-182).  Returns the step count."""
        with torch.cuda.device(self.dev):
            self.gn_sq.zero_()
            for g in self.G.values():
                ops.sumsq_accum(g.reshape(-1), self.gn_sq)
            self.step_count += 1
            for k, p in self.P.items():
                ops.adam_step(p.reshape(-1), self.G[k].reshape(-1), self.m[k].reshape(-1), self.v[k].reshape(-1), self.gn_sq, self.grad_clip, self.lr,
                              self.betas[0], self.betas[1], self.eps, self.step_count)
            self.model.refresh_plan()
        return self.step_count

    def grad_norm(self):
        return float(torch.sqrt(self.gn_sq).item())

    def train_step(self, batch, prenet_keep=None):
        rep = self.forward_backward(batch, prenet_keep)
        self.optimizer_step()
        rep["grad_norm"] = self.grad_norm()
        return rep
