"""Progressive Layered Extraction (PLE) on the HIP hot path.

Mirror of the reference's model/ple.py (PLE: lines 9-70, CGC: lines 73-125): same constructor
arguments, parameter names and forward(x) -> [B, n_tower] probabilities.  Per CGC level every expert's
layer and every gate go out as ONE grouped MFMA launch; softmax + pooling is one row-wise kernel.
"""
import torch
import torch.nn as nn

from .. import plan as P
from .layer import BaseModel, MultiLayerPerceptron, CrossNetwork, mlp_stack, _reg_filter


class CGC(nn.Module):
    """One extraction level: task-specific + shared experts, one gate per task (+ a shared gate below the top level).
    Reference: model/ple.py:73-125."""

    def __init__(self, cur_level, n_level, n_task, n_expert_specific, n_expert_shared, input_dims, expert_dims, dropout=0.2):
        super().__init__()
        self.cur_level, self.n_level, self.n_task = cur_level, n_level, n_task
        self.n_expert_specific, self.n_expert_shared = n_expert_specific, n_expert_shared
        self.n_expert_all = n_expert_specific * n_task + n_expert_shared
        self.experts_specific = nn.ModuleList(
            MultiLayerPerceptron(input_dims, expert_dims, dropout, output_layer=False, bn=False)
            for _ in range(n_task * n_expert_specific))
        self.experts_shared = nn.ModuleList(
            MultiLayerPerceptron(input_dims, expert_dims, dropout, output_layer=False, bn=False)
            for _ in range(n_expert_shared))
        self.gates_specific = nn.ModuleList([
            nn.Sequential(nn.Linear(input_dims, n_expert_specific + n_expert_shared), nn.Softmax(dim=1))
            for _ in range(n_task)])
        if cur_level < n_level:
            self.gate_shared = nn.Sequential(nn.Linear(input_dims, self.n_expert_all), nn.Softmax(dim=1))
        self.out_dim = expert_dims[-1]

    def describe(self, plan, x_list):
        ns, nsh, nt = self.n_expert_specific, self.n_expert_shared, self.n_task
        experts = list(self.experts_specific) + list(self.experts_shared)
        inputs = [x_list[i // ns] for i in range(nt * ns)] + [x_list[-1]] * nsh
        gates = [{"x": x_list[i], "w": g[0].weight, "b": g[0].bias} for i, g in enumerate(self.gates_specific)]
        has_shared_gate = self.cur_level < self.n_level
        if has_shared_gate:
            gates.append({"x": x_list[-1], "w": self.gate_shared[0].weight, "b": self.gate_shared[0].bias})
        # the last expert layer writes straight into the [B, n_expert_all * H] buffer the pooling kernel reads
        outs, gate_logits = _expert_stack(plan, experts, inputs, gates, self.out_dim)
        shared_idx = [nt * ns + k for k in range(nsh)]
        pool_gates = [(gate_logits[i], [i * ns + k for k in range(ns)] + shared_idx) for i in range(nt)]
        if has_shared_gate:
            pool_gates.append((gate_logits[nt], list(range(self.n_expert_all))))
        pool = P.GatePool(plan, outs, self.n_expert_all, self.out_dim, pool_gates)
        return pool.outs


def _expert_stack(plan, experts, inputs, gates, out_dim):
    """mlp_stack whose last layer lands in one contiguous [B, n_expert*H] buffer (experts in list order)."""
    outs, gate_logits = mlp_stack(plan, experts, inputs, extra_groups=gates)
    first = outs[0]
    # mlp_stack allocates each depth as one root buffer with expert i at columns [i*H, (i+1)*H)
    assert all(o.root is first.root for o in outs) and first.col0 == 0
    whole = P.Buf(first.root, first.rows, len(experts) * out_dim, first.ld, 0, plan)
    whole.mask = (first.mask[0], whole.cols) if first.mask is not None else None
    return whole, gate_logits


class PLE(BaseModel):
    """Reference: model/ple.py:9-70."""

    def __init__(self, feature_dims, embed_dim, n_tower, n_expert_specific, n_expert_shared, expert_dims, tower_dims,
                 dropout=0.2, config=None, l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5,
                 model_name='ple'):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.model_name = model_name
        self.n_level = len(expert_dims)
        self.n_tower = n_tower
        self.dropout_p = float(dropout)
        self.use_dcn = getattr(config, 'use_dcn', False)
        self.use_atten = getattr(config, 'use_atten', False)
        if self.use_dcn:
            # the reference raises at the first forward with use_dcn=True (CrossNetwork returns [B,E], added in
            # place to a [B,1] logit: model/layer.py:53-54); the parameters are still created, as there.
            self.cn = CrossNetwork(self.embed_output_dim, config.n_cross_layers)
        if self.use_atten:
            self.build_atten(config, dropout)
        self.cgc_layers = nn.ModuleList(
            CGC(i + 1, self.n_level, n_tower, n_expert_specific, n_expert_shared,
                self.embed_output_dim if i == 0 else expert_dims[i - 1][-1], expert_dims[i], dropout)
            for i in range(self.n_level))
        self.towers, self.towers_linear, self.output_layers = self.build_tower_output(
            n_tower, expert_dims[-1][-1], tower_dims, dropout)
        self.add_regularization_weight(_reg_filter(self.cgc_layers), l2=l2_reg_dnn)
        self.add_regularization_weight(_reg_filter(self.towers), l2=l2_reg_dnn)
        if self.use_dcn:
            self.add_regularization_weight(_reg_filter(self.cn), l2=l2_reg_cross)

    def describe(self, plan, emb):
        if self.use_dcn:
            raise RuntimeError("use_dcn=True cannot run: the reference adds a [B,E] cross output in place to a [B,1] "
                               "logit (model/layer.py:53-54) and raises; so do we")
        E = emb.out
        inputs = [E] * (self.n_tower + 1)
        for cgc in self.cgc_layers:
            inputs = cgc.describe(plan, inputs)
        # a level's pooled outputs feed the next level's experts and gates and nothing else (ple.py:54-57): pooling, the next
        # level's contractions and its pooling go out as one launch where the shapes allow (csrc/cgc.hip)
        P.fuse_cgc_mid(plan)
        # the two layers of a level's experts (no BatchNorm between them, ple.py:84-87) as one forward launch (csrc/pair.hip)
        P.fuse_expert_pair(plan)
        others = []                                                      # the wide term (ple.py:61) is formed inside the head launch
        if self.use_atten:
            others.append(self.describe_atten(plan, E))                  # ple.py:65-67
        out = plan.new(self.n_tower)
        self.describe_towers(plan, inputs[:self.n_tower], others, out, wide_in=E)
        return [out], [], []

    def forward(self, x):
        return self.plan_holder(x.shape[0]).run(x.to(torch.int32))
