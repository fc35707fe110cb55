"""Import-only stand-in: the reference imports AttForwardTA (decoder_sa.py:11) and never uses it."""
import torch


class AttForwardTA(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("unused by the reference path")
