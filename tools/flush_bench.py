"""Micro-benchmark of the lazy-table replay kernels (whole-table flush of a given gap, catch-up of one batch's rows).
Run on the GPU box:  python tools/flush_bench.py [--rows 26000000] [--gap 64]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdcmdr_amd import _lib as L  # noqa: E402
from cdcmdr_amd.optim import FusedAdam  # noqa: E402


class _M(torch.nn.Module):
    def __init__(self, R, D, dev):
        super().__init__()
        self.embedding = torch.nn.Module()
        self.embedding.embedding_dict = torch.nn.Embedding(R, D, device=dev)
        torch.nn.init.normal_(self.embedding.embedding_dict.weight, std=0.01)

    def regularized_parameters(self):
        return [(self.embedding.embedding_dict.weight, 0.0, 1e-5)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=26_000_000)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--gap", type=int, default=64)
    ap.add_argument("--step", type=int, default=5000)
    ap.add_argument("--fast", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    m = _M(args.rows, args.dim, dev)
    opt = FusedAdam(m, table_mode="lazy", fast_replay=bool(args.fast), flush_every=0)
    opt.table_m.normal_(0, 1e-7)
    opt.table_v.fill_(4e-14)
    for rep in range(3):
        opt.step_dev.fill_(args.step)
        opt.table_last.fill_(args.step - args.gap)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.flush_table()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        n = args.rows * args.dim * args.gap
        print(f"flush gap {args.gap}: {ms:.3f} ms  {n / ms / 1e6:.1f} G element-steps/s  "
              f"({ms * 1e-3 * 256 * 64 * 2.4e9 / n:.1f} lane-cycles per element-step at 2.4 GHz)", flush=True)


if __name__ == "__main__":
    main()
