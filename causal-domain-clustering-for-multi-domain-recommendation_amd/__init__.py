"""MI355X-native hot path of the multi-domain CTR training step.

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C-ABI of include/cdcmdr.h
  _lib.py    ctypes binding (no fallback: ops raise if libcdcmdr.so is missing)
  plan.py    static launch plans (forward / backward kernel sequences over preallocated buffers)
  model/     host-side mirror of the reference's model/ registry (same class names, ctor arguments,
             forward() signatures and state_dict keys)
  optim.py   table + dense-parameter Adam with the reference's dense-Adam/L2 semantics
  trainer.py the step driver reproducing run.py:470-497 (Run.train)
  dist.py    data-parallel wrapper (RCCL via torch.distributed)
"""
__version__ = "0.1.0"
