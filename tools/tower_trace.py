#!/usr/bin/env python
"""Phase stamps of the fused tower launches (csrc/tower.hip built with -DTW_TRACE=<1 + workgroup>): one C2 step, then the stamps the
chosen workgroup's first thread left in the workspace header, as microseconds from the launch's first stamp.
    CDC_EXTRA_HIPCC_FLAGS=-DTW_TRACE=1 python tools/tower_trace.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdcmdr_amd import plan as P  # noqa: E402
from cdcmdr_amd.model.ple import PLE  # noqa: E402
from cdcmdr_amd.optim import FusedAdam  # noqa: E402
from cdcmdr_amd.trainer import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
B, V = 4096, 100000
torch.manual_seed(0)
with torch.device(dev):
    m = PLE([V] * 26, 16, 3, 2, 2, ((256, 128), (64,)), (64, 32), 0.2)
m.set_precision("bf16")
opt = FusedAdam(m, table_mode="lazy")
ts = TrainStep(m, opt, B, mode="multi", use_graph=False)
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.integers(0, V, size=(B, 26)).astype(np.int32)).to(dev)
X[:, 10] %= 3
y = torch.from_numpy(rng.integers(0, 2, size=B).astype(np.int16)).to(dev)
g = X[:, 10].long()
chain = [op for op in ts.plan.ops if isinstance(op, P.TowerChain)][0]
names = {0: "tiles issued", 1: "tiles landed", 2: "L1 done, stats published, arrived F1", 3: "wide term done", 4: "F1 passed", 5: "stats 1 gathered",
         6: "L2 done, stats published", 7: "F2 passed", 8: "stats 2 gathered", 9: "head done",
         16: "entry", 17: "logit grads", 18: "L2 pieces, sums published, arrived B3", 19: "wide grads done", 20: "B3 passed", 21: "sums 2 gathered",
         22: "dZ2 out", 23: "dA1, L1 pieces, sums published", 24: "B4 passed", 25: "sums 1 gathered", 26: "dZ1 out", 27: "dX out", 28: "final sums"}
for rep in range(6):
    ts.step(X, y, g)
    torch.cuda.synchronize()
    st = chain.ws[4096:4096 + 32 * 8].view(torch.int64).cpu().numpy()
    if rep < 3:
        continue
    for lo, hi, what in ((0, 16, "forward"), (16, 32, "backward")):
        t0 = st[hi - 1] if st[hi - 1] else st[lo]          # slot 15 / 31: the wall clock at the body's first instruction
        print(f"-- {what}")
        prev = t0
        for i in range(lo, hi - 1):
            if st[i]:
                print(f"  {names.get(i, i):48s} {(st[i] - t0) / 100.0:7.2f} us   (+{(st[i] - prev) / 100.0:5.2f})")
                prev = st[i]
