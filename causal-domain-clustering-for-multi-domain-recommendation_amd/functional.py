"""Bridges static plans (plan.py) to the nn.Module surface: plan caching per module and ONE
autograd.Function per forward() call, so an unchanged reference trainer (run.py:481-492:
model(X) -> criterion -> loss.backward() -> torch.optim.Adam.step()) runs on the HIP kernels.

The fast path (trainer.py) bypasses autograd and drives plan.forward()/plan.backward() directly.
"""
import ctypes as C

import torch

from . import _lib as L
from . import plan as P


class PlanCache:
    """Plans of one module keyed by (tag, batch, mode, precision, dropout) and by the storage addresses
    of the module's parameters/buffers (a .to()/.cuda() or a re-assigned .data invalidates the plans)."""

    MAX_PLANS = 24      # a plan owns every activation and gradient buffer of its batch size: bound what stays resident
                        # (loops over many batch sizes — ragged epoch tails, CDC's concatenated domain batches — evict the
                        # least recently used plan; anything still holding an evicted plan keeps it alive and valid)

    def __init__(self):
        self.plans = {}

    def get(self, module, tag, B, build):
        sig = tuple(t.data_ptr() for t in list(module.parameters()) + list(module.buffers()))
        key = (tag, B, module.training, getattr(module, "precision", "bf16"), float(getattr(module, "dropout_p", 0.0)))
        hit = self.plans.get(key)
        if hit is not None and hit[0] == sig:
            self.plans[key] = self.plans.pop(key)          # most recently used last
            return hit[1]
        plan = build()
        self.plans.pop(key, None)
        self.plans[key] = (sig, plan)
        while len(self.plans) > self.MAX_PLANS:
            self.plans.pop(next(iter(self.plans)))
        return plan

    def clear(self):
        self.plans.clear()


def table_dense_grad(plan, emb_op, table):
    """aten::embedding_dense_backward as the reference's nn.Embedding (model/layer.py:140) produces it:
    a dense [R, D] gradient, rows summed in ascending batch order."""
    lib = plan.lib
    B, F, D = plan.B, emb_op.F, emb_op.D
    if B > L.SORT_MAX_ROWS:
        raise RuntimeError(f"batch {B} exceeds the per-field sort limit {L.SORT_MAX_ROWS}")
    dev = plan.device
    uniq = torch.empty((F, B), dtype=torch.int32, device=dev)
    seg = torch.empty((F, B + 1), dtype=torch.int32, device=dev)
    perm = torch.empty((F, B), dtype=torch.int32, device=dev)
    cnt = torch.empty((F,), dtype=torch.int32, device=dev)
    grad = torch.zeros_like(table)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scratch = torch.empty((2 * F * B,), dtype=torch.int64, device=dev) if B > 1024 else None
    L.check(lib.cdc_embed_sort_dedupe(emb_op.idx.data_ptr(), uniq.data_ptr(), seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(),
                                      None if scratch is None else scratch.data_ptr(), B, F, s), "embed_sort_dedupe")
    g = emb_op.out.grad
    assert g.ld == F * D
    scratch = torch.empty((F * B * D,), dtype=torch.float32, device=dev)
    rowgrad = torch.empty((F * B * D,), dtype=torch.float32, device=dev)
    L.check(lib.cdc_embed_segment_sum(g.ptr, seg.data_ptr(), perm.data_ptr(), cnt.data_ptr(), scratch.data_ptr(), rowgrad.data_ptr(),
                                      B, F, D, s), "embed_segment_sum")
    L.check(lib.cdc_embed_grad_dense(rowgrad.data_ptr(), uniq.data_ptr(), cnt.data_ptr(), grad.data_ptr(),
                                     B, F, D, table.shape[0], s), "embed_grad_dense")
    return grad


class PlanFunction(torch.autograd.Function):
    """forward(plan_holder, n_int_inputs, *inputs_and_params)"""

    @staticmethod
    def forward(ctx, holder, *tensors):
        plan = holder.plan
        n_in = len(holder.inputs)
        ins, params = tensors[:n_in], tensors[n_in:]
        for dst, src in zip(holder.inputs, ins):
            if isinstance(dst, P.Buf):
                dst.tensor().copy_(src)
            else:
                dst.copy_(src)
        plan.forward()
        ctx.holder = holder
        outs = tuple(o.tensor().clone() for o in holder.outputs)
        extra = tuple(t.clone() for t in holder.extra_outputs)
        ctx.n_out = len(outs)
        ctx.mark_non_differentiable(*extra)
        res = outs + extra
        return res if len(res) > 1 else res[0]

    @staticmethod
    def backward(ctx, *gouts):
        holder = ctx.holder
        plan = holder.plan
        for o, g in zip(holder.outputs, gouts[:ctx.n_out]):
            og = o.grad.tensor()
            if g is None:
                og.zero_()
            else:
                og.copy_(g)
        plan.backward()
        grads = [None]                       # holder
        for dst in holder.inputs:
            if isinstance(dst, P.Buf) and dst.root.dtype == torch.float32 and holder.input_needs_grad.get(id(dst), False):
                grads.append(dst.grad.tensor().clone())
            else:
                grads.append(None)
        for p in holder.params:
            if holder.table is not None and p is holder.table:
                grads.append(table_dense_grad(plan, holder.emb_op, p) if p.requires_grad else None)
            else:
                g = plan.param_grads.get(id(p))
                grads.append(None if (g is None or not p.requires_grad) else g.clone())
        return tuple(grads)


class PlanHolder:
    """A finalised plan plus what the autograd bridge needs to know about it."""

    def __init__(self, plan, inputs, outputs, emb_op=None, extra_outputs=()):
        self.plan = plan
        self.inputs = inputs                  # list of Buf (float inputs) or torch int tensors (ids, groups)
        self.outputs = outputs                # list of Buf
        self.extra_outputs = list(extra_outputs)   # non-differentiable torch tensors returned as well
        self.emb_op = emb_op
        self.table = emb_op.table if emb_op is not None else None
        self.input_needs_grad = {}
        ps = [plan._param_refs[k] for k in plan.param_grads]
        if self.table is not None:
            ps.append(self.table)
        self.params = ps

    def run(self, *inputs):
        for dst, src in zip(self.inputs, inputs):
            if isinstance(dst, P.Buf):
                self.input_needs_grad[id(dst)] = bool(getattr(src, "requires_grad", False))
        return PlanFunction.apply(self, *inputs, *self.params)
