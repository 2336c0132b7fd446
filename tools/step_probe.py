#!/usr/bin/env python
"""Times every launch of the C2 training plan on its own (the same launch issued back to back, chip otherwise idle):
    python tools/step_probe.py [--reps 200] [--filter cgc]
One line per launch of plan.fwd_steps / plan.bwd_steps: microseconds per launch."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdcmdr_amd import _lib as L  # noqa: E402
from cdcmdr_amd.model.ple import PLE  # noqa: E402
from cdcmdr_amd.optim import FusedAdam  # noqa: E402
from cdcmdr_amd.trainer import TrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--filter", default="")
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--vocab", type=int, default=100000)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
fd = [args.vocab] * 26
with torch.device(dev):
    m = PLE(fd, 16, 3, 2, 2, ((256, 128), (64,)), (64, 32), 0.2)
m.set_precision("bf16")
opt = FusedAdam(m, table_mode="lazy")
ts = TrainStep(m, opt, args.batch, mode="multi", use_graph=False)
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.integers(0, args.vocab, size=(args.batch, 26)).astype(np.int32)).to(dev)
X[:, 10] %= 3
y = torch.from_numpy(rng.integers(0, 2, size=args.batch).astype(np.int16)).to(dev)
g = X[:, 10].long()
for _ in range(3):
    ts.step(X, y, g)
torch.cuda.synchronize()
names = []
L.PROFILE = names
ts.step(X, y, g)
torch.cuda.synchronize()
L.PROFILE = None
order = [n for n, *_ in names]
plan = ts.plan
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
steps = [("fwd", f) for f in plan.fwd_steps] + [("bwd", f) for f in plan.bwd_steps]
for kind, fn in steps:
    rec = []
    L.PROFILE = rec
    fn(st)
    torch.cuda.synchronize()
    L.PROFILE = None
    name = rec[0][0] if rec else "?"
    if args.filter and args.filter not in name:
        continue
    for _ in range(10):
        fn(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn(st)
    e1.record()
    torch.cuda.synchronize()
    print(f"{kind} {name:40s} {e0.elapsed_time(e1) / args.reps * 1e3:8.2f} us")
