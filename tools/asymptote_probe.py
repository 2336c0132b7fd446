#!/usr/bin/env python
"""Large-batch asymptote of the contraction launches and of the gather: the C2 model's forward/backward
launches timed one by one (the same launch issued back to back on an idle chip) at B = 4096 (the benchmark batch), 16384,
32768 and 65536 — same layer shapes, only the row count grows (above 32768, the per-field sort limit of the table gradient, the
plan is built by a forward call alone and its backward launches are timed on whatever the buffers hold).  Shows which launches are latency-bound at B = 4096 (their TFLOP/s or
GB/s keeps rising with B) and where the kernels level off.

    python tools/asymptote_probe.py > profiles/round3/asymptote.txt
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdcmdr_amd import _lib as L  # noqa: E402
from cdcmdr_amd.model.ple import PLE  # noqa: E402

dev = torch.device("cuda:0")
V, F_, D = 100_000, 26, 16
torch.manual_seed(0)
with torch.device(dev):
    m = PLE([V] * F_, D, 3, 2, 2, ((256, 128), (64,)), (64, 32), 0.2)
m.set_precision("bf16")
m.train()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print("# launch                          B      us/launch   TFLOP/s (bf16 MFMA peak 2500)   or GB/s (HBM peak 8000)")
for B in (4096, 16384, 32768, 65536):
    rng = np.random.default_rng(0)
    X = torch.from_numpy(rng.integers(0, V, size=(B, F_)).astype(np.int32)).to(dev)
    out = m(X)
    if B <= 32768:
        out.backward(torch.ones_like(out))
    torch.cuda.synchronize()
    plan = m.plan_holder(B).plan
    rows = []
    for kind, steps in (("fwd", plan.fwd_steps), ("bwd", plan.bwd_steps)):
        for i, fn in enumerate(steps):
            rec = []
            L.PROFILE = rec
            fn(st)
            torch.cuda.synchronize()
            L.PROFILE = None
            if not rec:
                continue
            name, _, _, flops, nbytes = rec[0]
            if not ("glinear" in name or "gather" in name or "cgc_mid" in name or "gate_pool" in name):
                continue
            reps = 50
            for _ in range(5):
                fn(st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn(st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            if "gather" in name:
                nb = B * F_ * (D * 4 + 4 + D * 4 + D * 2)          # row read + id + fp32 row written + bf16 shadow written
                rate = f"{nb / us / 1e3:9.0f} GB/s"
            else:
                rate = f"{flops / us / 1e6:9.1f} TFLOP/s"
            rows.append((f"{kind}#{i:02d} {name}", us, rate, flops))
    tot_us = sum(r[1] for r in rows if "gather" not in r[0])
    tot_fl = sum(r[3] for r in rows if "gather" not in r[0])
    for nm, us, rate, _ in rows:
        print(f"{nm:34s} {B:6d} {us:10.2f}   {rate}")
    print(f"{'all contraction launches':34s} {B:6d} {tot_us:10.2f}   {tot_fl / tot_us / 1e6:9.1f} TFLOP/s = {tot_fl / tot_us / 1e6 / 2500 * 100:.1f} % of the bf16 MFMA peak")
    print()
    del plan
    m._cache().clear()
    torch.cuda.empty_cache()
