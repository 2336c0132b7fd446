#!/usr/bin/env python
"""Phase stamps of the fused PLE expert-pair forward launch (csrc/pair.hip built with -DPAIR_TRACE=1): a few C2 steps, then the stamps
thread 0 of workgroup 0 left, as microseconds from the launch's first stamp.
    CDC_EXTRA_HIPCC_FLAGS=-DPAIR_TRACE=1 python tools/mid_trace.py"""
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

dev = torch.device("cuda:0")
B, V = 4096, 100000
torch.manual_seed(0)
with torch.device(dev):
    m = PLE([V] * 26, 16, 3, 2, 2, ((256, 128), (64,)), (64, 32), 0.2)
m.set_precision("bf16")
opt = FusedAdam(m, table_mode="lazy")
GRAPH = os.environ.get("PAIR_TRACE_GRAPH", "0") == "1"            # 1: the launches replayed from a hipGraph (kernel arguments then live where the graph put them)
ts = TrainStep(m, opt, B, mode="multi", use_graph=GRAPH, overlap=False)
rng = np.random.default_rng(0)
X = torch.from_numpy(rng.integers(0, V, size=(B, 26)).astype(np.int32)).to(dev)
X[:, 10] %= 3
y = torch.from_numpy(rng.integers(0, 2, size=B).astype(np.int16)).to(dev)
g = X[:, 10].long()
lib = L.load()
fn = C.CDLL(lib._name).cdc_debug_pair_stamps
names = {0: "first instruction", 1: "descriptors read, biases fetched", 2: "phase 1 (K loop) done, this wave", 3: "barrier", 4: "epilogue 1 done, this wave",
         5: "W2 landed, barrier", 6: "hidden tile stored", 7: "phase 2 done", 8: "accumulators -> LDS tile, barrier", 9: "epilogue 2, stores issued"}
for rep in range(8):
    ts.step(X, y, g)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    fn(buf)
    st = np.array(list(buf), dtype=np.int64)
    if rep < 6:
        continue
    for lo, hi, what in ((0, 16, "forward"),):
        t0 = st[lo]
        print(f"-- {what}")
        prev = t0
        for i in sorted(range(lo, hi), key=lambda k: st[k]):
            if st[i]:
                print(f"  {str(names.get(i, i)):48s} {(st[i] - t0) / 100.0:7.2f} us   (+{(st[i] - prev) / 100.0:5.2f})")
                prev = st[i]

# the same launch issued back to back (instruction cache and argument block warm): stamps of the last of 20
st_ = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for kind, fns, lo, hi in (("forward", ts.plan.fwd_steps, 0, 16),):
    for f in fns:
        rec = []
        L.PROFILE = rec
        f(st_)
        torch.cuda.synchronize()
        L.PROFILE = None
        if not rec or "pair_fwd" not in rec[0][0]:
            continue
        for _ in range(20):
            f(st_)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 16)()
        fn(buf)
        st = np.array(list(buf), dtype=np.int64)
        print(f"-- {kind}, back to back")
        t0 = prev = st[lo]
        for i in sorted(range(lo, hi), key=lambda k: st[k]):
            if st[i]:
                print(f"  {str(names.get(i, i)):48s} {(st[i] - t0) / 100.0:7.2f} us   (+{(st[i] - prev) / 100.0:5.2f})")
                prev = st[i]
