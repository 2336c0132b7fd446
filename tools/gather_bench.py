"""cdc_embed_gather_fwd alone at growing batch sizes (26 fields x vocab 1M x D=16, uniform ids): the BASELINE batch of 4096
moves 14.6 MB — a launch-latency-sized job — so the kernel's streaming rate only shows at larger batches.
Algorithmic bytes per sample: F*(D*4+4) read + F*D*4 written = 3432 B.   python tools/gather_bench.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdcmdr_amd import _lib as L  # noqa: E402


def main():
    lib = L.load()
    dev = torch.device("cuda:0")
    F, V, D = 26, 1_000_000, 16
    table = torch.randn(F * V, D, device=dev)
    offs = (torch.arange(F, dtype=torch.int32, device=dev) * V)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for B in (4096, 32768, 262144, 1048576):
        ids = torch.randint(0, V, (B, F), dtype=torch.int32, device=dev)
        out = torch.empty((B, F * D), dtype=torch.float32, device=dev)
        for _ in range(3):
            L.check(lib.cdc_embed_gather_fwd(ids.data_ptr(), offs.data_ptr(), table.data_ptr(), out.data_ptr(), None, None, B, F, D, F * V, s), "gather")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            lib.cdc_embed_gather_fwd(ids.data_ptr(), offs.data_ptr(), table.data_ptr(), out.data_ptr(), None, None, B, F, D, F * V, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = B * F * (D * 4 + 4 + D * 4)
        print(f"B {B:8d}: {ms * 1e3:9.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / ms / 1e6:7.1f} GB/s  ({nbytes / ms / 1e6 / 8000 * 100:.0f} % of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
