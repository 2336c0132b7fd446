"""ms/step of the fast path on the other BASELINE.json configurations (parity-test cases, not the bench line): a sanity
check that nothing falls off a cliff at their sizes.  Single GPU; the 8-GPU configurations run their per-GPU share.
  python tools/bench_configs.py [c1 c3 c4 c5]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdcmdr_amd.optim import FusedAdam  # noqa: E402
from cdcmdr_amd.synth import make_dataset  # noqa: E402
from cdcmdr_amd.trainer import TrainStep  # noqa: E402


def run(name, model, mode, B, field_dims, n_domain, domain_idx, steps=60, warm=30, pool=16):
    dev = torch.device("cuda:0")
    opt = FusedAdam(model, table_mode="lazy")
    ts = TrainStep(model, opt, B, mode=mode, use_graph=True)
    X, y = make_dataset(B * pool, field_dims, n_domain=n_domain, domain_idx=domain_idx, seed=1)
    Xd = torch.from_numpy(X).to(dev).view(pool, B, -1)
    yd = torch.from_numpy(y).to(dev).view(pool, B)
    gd = Xd[:, :, domain_idx].to(torch.int64) if mode in ("multi", "star") else None
    for i in range(warm):
        ts.step(Xd[i % pool], yd[i % pool], None if gd is None else gd[i % pool], next_X=Xd[(i + 1) % pool])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        ts.step(Xd[i % pool], yd[i % pool], None if gd is None else gd[i % pool], next_X=Xd[(i + 1) % pool])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ts.check_ids()
    print(f"{name}: {ms:.3f} ms/step  {B / ms * 1e3 / 1e6:.2f} M samples/s  loss {float(ts.loss.item()):.4f}  "
          f"table {model.embedding.embedding_dict.weight.numel() * 4 / 2**30:.1f} GiB", flush=True)
    del ts, opt
    torch.cuda.empty_cache()


def main():
    which = sys.argv[1:] or ["c1", "c3", "c5", "c4"]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if "c1" in which:
        from cdcmdr_amd.model.dcn import DCN
        fd = [10_000] * 13
        with torch.device(dev):
            m = DCN(fd, 16, 3, (256, 128, 64), dropout=0.2)
        run("C1 DCN 13x10k D16 B1024", m, "single", 1024, fd, 1, 0)
        del m
    if "c3" in which:
        from cdcmdr_amd.model.mmoe import MMoE
        fd = [1_000_000] * 26
        with torch.device(dev):
            m = MMoE(fd, 16, 3, 8, (256, 128, 64), (64, 32), dropout=0.2)
        run("C3 MMoE-8 26x1M D16 B8192/8 GPUs -> 1024 per GPU", m, "multi", 1024, fd, 3, 10)
        run("C3 MMoE-8 26x1M D16 B8192 on one GPU", m, "multi", 8192, fd, 3, 10)
        del m
    if "c5" in which:
        from cdcmdr_amd.model.star import STAR
        fd = [1_000_000] * 26
        fd[10] = 30
        with torch.device(dev):
            m = STAR(fd, 16, 30, (256, 128, 64), domain_idx=10, dropout=0.2)
        run("C5 STAR-30 26x1M D16 B16384/8 GPUs -> 2048 per GPU", m, "star", 2048, fd, 30, 10)
        run("C5 STAR-30 26x1M D16 B16384 on one GPU", m, "star", 16384, fd, 30, 10)
        del m
    if "c4" in which:
        from cdcmdr_amd.model.ple import PLE
        fd = [10_000_000] * 26
        fd[10] = 30
        with torch.device(dev):
            m = PLE(fd, 32, 4, 2, 2, ((256, 128), (64,)), (64, 32), dropout=0.2)
        run("C4 CDC-PLE base (4 clusters) 26x10M D32 B8192/8 GPUs -> 1024 per GPU, whole 31 GiB table on one GPU", m, "multi", 1024, fd, 4, 10,
            steps=20, warm=10, pool=8)
        del m


if __name__ == "__main__":
    main()
