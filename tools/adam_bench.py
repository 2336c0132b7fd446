"""Development probe: times cdc_adam_multi back to back on synthetic tensor sets to see what a launch is waiting for
(the tensor search over the argument block, the regularisation atomics, or plain memory traffic)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdcmdr_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
step_dev = torch.ones(1, dtype=torch.int32, device=dev)
scalars = torch.rand(4096 * 2, dtype=torch.float32, device=dev) * 1e-3 + 1e-3
reg = torch.zeros(2, dtype=torch.float64, device=dev)


def make(sizes, l2):
    a = L.AdamArgs()
    a.n_tensors = len(sizes)
    a.lerp_w, a.beta2, a.one_minus_beta2, a.eps, a.weight_decay = 0.1, 0.99, 0.01, 1e-8, 1e-8
    a.step_scalars, a.n_scalars = scalars.data_ptr(), 4096
    a.grad_scale, a.step_dev, a.reg_sum = 1.0, step_dev.data_ptr(), reg.data_ptr()
    keep = []
    for i, n in enumerate(sizes):
        ts = [torch.randn(n, device=dev) * 0.01 for _ in range(4)]
        ts[3].abs_()
        keep.append(ts)
        T = a.t[i]
        T.w, T.g, T.m, T.v, T.n, T.l2 = ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), n, l2
    return a, keep


def timeit(name, sizes, l2, reps=200):
    a, keep = make(sizes, l2)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(10):
        lib.cdc_adam_multi(C.byref(a), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.cdc_adam_multi(C.byref(a), s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    n = sum(sizes)
    print(f"{name:50s} {len(sizes):3d} tensors {n:9d} elems  {us:7.2f} us  {n * 28 / us / 1e6:7.2f} TB/s", flush=True)


timeit("one tensor 0.6M, l2>0 (150 atomics)", [600_000], 1e-5)
timeit("one tensor 0.6M, l2=0 (no atomics)", [600_000], 0.0)
timeit("one tensor 2.4M, l2=0", [2_400_000], 0.0)
timeit("one tensor 2.4M, l2>0", [2_400_000], 1e-5)
timeit("48 tensors of 256, l2=0", [256] * 48, 0.0)
timeit("48 tensors of 256, l2>0", [256] * 48, 1e-5)
timeit("40 x 256 then 8 x 106496 (PLE level 1 like), l2>0", [256] * 40 + [106496] * 8, 1e-5)
timeit("8 x 106496 then 40 x 256, l2>0", [106496] * 8 + [256] * 40, 1e-5)
timeit("8 x 106496, l2=0", [106496] * 8, 0.0)
