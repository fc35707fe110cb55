"""Stand-in for espnet.nets.pytorch_backend.nets_utils (ESPnet v0.8 semantics, restated)."""
import torch


def pad_list(xs, pad_value):
    """Stack along a new dim 0; right-pad dim 0 of every item to the longest with pad_value."""
    n_batch = len(xs)
    max_len = max(x.size(0) for x in xs)
    pad = xs[0].new(n_batch, max_len, *xs[0].size()[1:]).fill_(pad_value)
    for i in range(n_batch):
        pad[i, : xs[i].size(0)] = xs[i]
    return pad


def make_pad_mask(lengths, xs=None, length_dim=-1):
    """Bool [B, max(lengths)], True where idx >= length. Built on CPU (callers .to(device))."""
    if not isinstance(lengths, list):
        lengths = lengths.tolist()
    bs = int(len(lengths))
    maxlen = int(max(lengths))
    seq_range = torch.arange(0, maxlen, dtype=torch.int64)
    seq_range_expand = seq_range.unsqueeze(0).expand(bs, maxlen)
    seq_length_expand = seq_range_expand.new(lengths).unsqueeze(-1)
    return seq_range_expand >= seq_length_expand


def make_non_pad_mask(lengths, xs=None, length_dim=-1):
    return ~make_pad_mask(lengths, xs, length_dim)
