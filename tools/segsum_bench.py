"""Development probe: cdc_embed_segment_sum on the bench's batch shape (26 fields x vocab 1M, B=4096) with and without the
3-value domain field — is the launch as long as its longest segment?"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdcmdr_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
B, F, D = 4096, 26, 16
rng = np.random.default_rng(0)


def run(name, card):
    idx = np.stack([rng.integers(0, c, size=B) + 1_000_000 * f for f, c in enumerate(card)], axis=1).astype(np.int32)
    d_idx = torch.from_numpy(idx).to(dev)
    uniq = torch.empty((F, B), dtype=torch.int32, device=dev)
    seg = torch.empty((F, B + 1), dtype=torch.int32, device=dev)
    perm = torch.empty((F, B), dtype=torch.int32, device=dev)
    cnt = torch.zeros(F, dtype=torch.int32, device=dev)
    scratch = torch.empty(2 * F * B, dtype=torch.int64, device=dev)
    g = torch.randn(B, F * D, device=dev)
    srt = torch.empty(F * B * D, device=dev)
    rg = torch.empty(F * B * D, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.cdc_embed_sort_dedupe(d_idx.data_ptr(), uniq.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(),
                                      scratch.data_ptr(), B, F, s), "sort")

    def once():
        lib.cdc_embed_segment_sum(g.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(), srt.data_ptr(), rg.data_ptr(), B, F, D, s)

    for _ in range(10):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        once()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / 200 * 1e3:7.2f} us (gather + sum)", flush=True)


run("26 x 1M", [1_000_000] * 26)
run("25 x 1M + one 3-value field", [1_000_000] * 10 + [3] + [1_000_000] * 15)
run("25 x 1M + one 100-value field", [1_000_000] * 10 + [100] + [1_000_000] * 15)
run("25 x 1M + one 1000-value field", [1_000_000] * 10 + [1000] + [1_000_000] * 15)
run("26 x 1000", [1000] * 26)
