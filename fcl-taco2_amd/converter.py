"""CustomConverter — vectorised host batch layout.

Same inputs and outputs as the reference's `tts.py:CustomConverter.__call__` (:215-306; duplicated in
tts_distill.py:205-309) — `batch = [(xs, ys, spembs, extras, f0, energy)]` in, dict of tensors out — but the
O(T^2) Python loops (`start = int(sum(ds[:it]))` per phoneme, :249-258) become one exclusive cumsum per
utterance and a few fancy-indexing gathers.  Durations are integer-valued floats, so every index is exact:
outputs are bit-identical to the reference's (tests/golden/g4_integer.npz was produced by the real class)."""
import numpy as np
import torch


def _pad_list(arrs, dtype, trailing=()):
    n = len(arrs)
    m = max((a.shape[0] for a in arrs), default=0)
    out = np.zeros((n, m) + tuple(trailing), dtype=dtype)
    for i, a in enumerate(arrs):
        out[i, : a.shape[0]] = a
    return out


class CustomConverter(object):
    def __init__(self, reduction_factor=1, use_fe_condition=False, append_position=False):
        if int(reduction_factor) < 1:
            raise ValueError("reduction_factor must be >= 1")
        self.reduction_factor = int(reduction_factor)  # tts.py:250-258: segments, ds_nonzeros and the position table are in FRAMES = r x the durations
        self.use_fe_condition = use_fe_condition
        self.append_position = append_position

    def __call__(self, batch, device=torch.device("cpu")):
        assert len(batch) == 1  # batch should be located in list (tts.py:227)
        xs, ys, spembs, extras, f0, energy = batch[0]
        ilens = np.array([x.shape[0] for x in xs], dtype=np.int64)
        olens = np.array([y.shape[0] for y in ys], dtype=np.int64)
        new = {
            "xs": torch.from_numpy(_pad_list([np.asarray(x, dtype=np.int64) for x in xs], np.int64)).to(device),
            "ilens": torch.from_numpy(ilens).to(device),
            "ys": torch.from_numpy(_pad_list([np.asarray(y, dtype=np.float32) for y in ys], np.float32, ys[0].shape[1:])).to(device),
            "olens": torch.from_numpy(olens).to(device),
        }
        if spembs is not None:
            new["spembs"] = torch.from_numpy(np.array(spembs)).float().to(device)
        if extras is not None:
            odim = ys[0].shape[1]
            seg_rows, seg_len, masks = [], [], []
            for ib in range(len(xs)):
                d = np.asarray(extras[ib], dtype=np.float64).reshape(-1)[: ilens[ib]]
                edges = np.concatenate([[0.0], np.cumsum(d)]).astype(np.int64) * self.reduction_factor  # int(sum(ds[:it])) * r for every it at once
                length = edges[1:] - edges[:-1]
                nz = length != 0
                masks.append(nz.astype(np.int64))
                seg_rows.append((ib, edges[:-1][nz], length[nz]))
                seg_len.append(length[nz])
            ds_nonzeros = np.concatenate(seg_len) if seg_len else np.zeros(0, np.int64)
            n, lmax = ds_nonzeros.shape[0], int(ds_nonzeros.max()) if ds_nonzeros.size else 0
            t = np.arange(lmax, dtype=np.int64)[None, :]
            valid = t < ds_nonzeros[:, None]  # == make_non_pad_mask(ds_nonzeros)
            new_ys = np.zeros((n, lmax, odim), dtype=np.float32)
            row = 0
            for ib, starts, lengths in seg_rows:  # one gather per utterance
                k = starts.shape[0]
                idx = np.minimum(starts[:, None] + t, max(int(olens[ib]) - 1, 0))
                seg = np.asarray(ys[ib], dtype=np.float32)[idx]  # [k, lmax, odim]
                seg[~valid[row : row + k]] = 0.0
                new_ys[row : row + k] = seg
                row += k
            new["extras"] = torch.from_numpy(_pad_list([np.asarray(e, dtype=np.float32) for e in extras], np.float32, np.asarray(extras[0]).shape[1:])).to(device)
            new["new_ys"] = torch.from_numpy(new_ys).to(device)
            new["non_zero_lens_mask"] = torch.from_numpy(_pad_list(masks, np.int64))
            new["ds_nonzeros"] = torch.from_numpy(ds_nonzeros).to(device)
            new["output_masks"] = torch.from_numpy(valid).to(device)
            if self.append_position:
                pos = np.where(valid, t.astype(np.float32) / ds_nonzeros[:, None].astype(np.float32), np.float32(0.0)).astype(np.float32)
                new["position"] = torch.from_numpy(pos)
            if self.use_fe_condition:
                new["f0"] = torch.from_numpy(_pad_list([np.asarray(a, dtype=np.float32) for a in f0], np.float32, np.asarray(f0[0]).shape[1:]))
                new["energy"] = torch.from_numpy(_pad_list([np.asarray(a, dtype=np.float32) for a in energy], np.float32, np.asarray(energy[0]).shape[1:]))
        return new
