"""Import-safe placeholder: run.py:15-26 imports `model.pepnet.PEPNet` at module import time, but PEPNet is not on the hot path
this build accelerates (out of scope — not named by the north star; §8f row N4)."""
import torch.nn as nn


class PEPNet(nn.Module):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("PEPNet is outside the MI355X hot path of this build (see DESIGN.md, Out of scope)")
