"""Multi-gate Mixture-of-Experts on the HIP hot path.

Mirror of the reference's model/mmoe.py:10-74: same constructor arguments, parameter names and
forward(x) -> [B, n_tower] probabilities.  All experts advance layer by layer in one grouped MFMA launch
(+ one BatchNorm launch); the gates ride in the first launch; softmax + pooling is one row-wise kernel.
"""
import torch
import torch.nn as nn

from .. import plan as P
from .layer import BaseModel, MultiLayerPerceptron, CrossNetwork, mlp_stack, _reg_filter


class MMoE(BaseModel):
    def __init__(self, feature_dims, embed_dim, n_tower, n_expert, expert_dims, tower_dims, dropout=0.2, config=None,
                 l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5, model_name='mmoe'):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.config = config
        self.model_name = model_name
        self.n_tower = n_tower
        self.n_expert = n_expert
        self.dropout_p = float(dropout)
        self.use_dcn = getattr(config, 'use_dcn', False)
        self.use_atten = getattr(config, 'use_atten', False)
        if self.use_dcn:
            self.cn = CrossNetwork(self.embed_output_dim, config.n_cross_layers)
        if self.use_atten:
            self.build_atten(config, dropout)
        self.experts = nn.ModuleList(MultiLayerPerceptron(self.embed_output_dim, expert_dims, dropout, output_layer=False)
                                     for _ in range(n_expert))
        self.gates = nn.ModuleList([nn.Sequential(nn.Linear(self.embed_output_dim, n_expert), nn.Softmax(dim=1))
                                    for _ in range(self.n_tower)])
        self.towers, self.towers_linear, self.output_layers = self.build_tower_output(n_tower, expert_dims[-1], tower_dims, dropout)
        self.expert_out = expert_dims[-1]
        self.add_regularization_weight(_reg_filter(self.experts), l2=l2_reg_dnn)
        self.add_regularization_weight(_reg_filter(self.towers), l2=l2_reg_dnn)
        if self.use_dcn:
            self.add_regularization_weight(_reg_filter(self.cn), l2=l2_reg_cross)

    def describe(self, plan, emb):
        if self.use_dcn:
            raise RuntimeError("use_dcn=True cannot run: the reference adds a [B,E] cross output in place to a [B,1] "
                               "logit (model/layer.py:53-54) and raises; so do we")
        E = emb.out
        gates = [{"x": E, "w": g[0].weight, "b": g[0].bias} for g in self.gates]
        outs, gate_logits = mlp_stack(plan, list(self.experts), [E] * self.n_expert, extra_groups=gates)
        first = outs[0]
        experts = P.Buf(first.root, first.rows, self.n_expert * self.expert_out, first.ld, 0, plan)
        pool = P.GatePool(plan, experts, self.n_expert, self.expert_out, [(lg, list(range(self.n_expert))) for lg in gate_logits])
        others = []                                                      # the wide term (mmoe.py:62) is formed inside the head launch
        if self.use_atten:
            others.append(self.describe_atten(plan, E))                  # mmoe.py:68-70
        out = plan.new(self.n_tower)
        self.describe_towers(plan, pool.outs, others, out, wide_in=E)
        return [out], [], []

    def forward(self, x):
        return self.plan_holder(x.shape[0]).run(x.to(torch.int32))
