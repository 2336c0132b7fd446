"""Import-safe placeholder: run.py:15-26 imports `model.autoint.AutoInt` at module import time, but AutoInt is not on the hot path
this build accelerates (out of scope — not named by the north star; §8f row N4)."""
import torch.nn as nn


class AutoInt(nn.Module):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("AutoInt is outside the MI355X hot path of this build (see DESIGN.md, Out of scope)")
