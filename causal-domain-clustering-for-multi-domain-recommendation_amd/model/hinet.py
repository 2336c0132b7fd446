"""HiNet on the HIP hot path.  Mirror of the reference's model/hinet.py:8-92 (SEI + HiNet):
    every domain's SEI block and one shared SEI block = 4 expert MLPs + softmax gate over the flattened embeddings;
    con_feas = the row's own domain's block, san_feas = all domain blocks mixed by a gate on the domain embedding;
    y = sigmoid(tower_linear(tower(cat[shared, con, san])) + linear(e) [+ attention branch]).
All (n_tower + 1) x 4 experts and the gates run as grouped launches; the pooling is the gate-pool kernel twice."""
import torch
from torch import nn

from .. import plan as P
from .layer import BaseModel, CrossNetwork, MultiLayerPerceptron, mlp_stack, _reg_filter


class SEI(nn.Module):
    """model/hinet.py:8-21 (parameter container; the forward is part of HiNet's plan)"""

    def __init__(self, input_dim, hidden_dims=(64, 32), expert_num=4, dropout=0.2):
        super().__init__()
        self.expert_num = expert_num
        self.experts = nn.ModuleList([MultiLayerPerceptron(input_dim, hidden_dims, dropout, output_layer=False) for _ in range(expert_num)])
        self.gate = nn.Sequential(nn.Linear(input_dim, self.expert_num), nn.Softmax(dim=1))


class HiNet(BaseModel):
    def __init__(self, feature_dims, embed_dim=10, n_tower=6, sei_dims=None, tower_dims=None, domain_idx=None, device='cpu',
                 dropout=0.2, config=None, l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.model_name = 'hinet'
        self.n_tower = n_tower
        self.device = device
        self.domain_idx = domain_idx
        self.dropout_p = float(dropout)
        self.use_dcn = getattr(config, 'use_dcn', False)
        self.use_atten = getattr(config, 'use_atten', False)
        if self.use_dcn:
            self.cn = CrossNetwork(self.embed_output_dim, config.n_cross_layers)
        if self.use_atten:
            self.build_atten(config, dropout)
        self.specific_seis = nn.ModuleList([SEI(self.embed_output_dim, hidden_dims=sei_dims, dropout=dropout) for _ in range(n_tower)])
        self.shared_seis = SEI(self.embed_output_dim, hidden_dims=sei_dims, dropout=dropout)
        self.san_gate = nn.Sequential(nn.Linear(embed_dim, n_tower), nn.Softmax(dim=1))
        self.tower = MultiLayerPerceptron(sei_dims[-1] * 3, tower_dims, dropout, output_layer=False)
        self.tower_linear = nn.Linear(tower_dims[-1], 1, bias=False)
        self.output_layer = nn.Sigmoid()
        self.sei_out = sei_dims[-1]
        if self.use_dcn:
            self.add_regularization_weight(_reg_filter(self.cn), l2=l2_reg_cross)
        for part in (self.specific_seis, self.shared_seis, self.san_gate, self.tower):
            self.add_regularization_weight(_reg_filter(part), l2=l2_reg_dnn)

    def describe(self, plan, emb):
        if self.use_dcn:
            raise RuntimeError("use_dcn=True cannot run: the reference adds a [B,E] cross output in place to a [B,1] logit "
                               "(model/hinet.py:82-90) and raises; so do we")
        E, n, H, D = emb.out, self.n_tower, self.sei_out, self.embed_dim
        seis = list(self.specific_seis) + [self.shared_seis]
        mlps = [ex for s in seis for ex in s.experts]
        ne = seis[0].expert_num
        gates = [{"x": E, "w": s.gate[0].weight, "b": s.gate[0].bias} for s in seis]
        outs, gate_logits = mlp_stack(plan, mlps, [E] * len(mlps), extra_groups=gates)
        first = outs[0]
        experts = P.Buf(first.root, first.rows, len(mlps) * H, first.ld, 0, plan)
        feature = plan.new(3 * H)                                       # cat([shared_feas, con_feas, san_feas], dim=1)
        spec = plan.new(n * H)                                          # the domain blocks side by side
        pooled = [spec.slice(i * H, (i + 1) * H) for i in range(n)] + [feature.slice(0, H)]
        P.GatePool(plan, experts, len(mlps), H, [(gate_logits[i], list(range(i * ne, (i + 1) * ne))) for i in range(n + 1)], outs=pooled)
        sel = P.SelectByGroup(plan, spec, n, H, out=feature.slice(H, 2 * H))
        dom = E.slice(self.domain_idx * D, (self.domain_idx + 1) * D)   # embed_x[:, domain_idx, :]
        san_logits = P.GLinear(plan, [{"x": dom, "w": self.san_gate[0].weight, "b": self.san_gate[0].bias}]).outs[0]
        P.GatePool(plan, spec, n, H, [(san_logits, list(range(n)))], outs=[feature.slice(2 * H, 3 * H)])
        tower_out, _ = mlp_stack(plan, [self.tower], [feature])
        others = [self.linear.describe(plan, E)]
        if self.use_atten:
            others.append(self.describe_atten(plan, E))
        out = plan.new(1)
        P.RowDot(plan, [{"x": tower_out[0], "w": self.tower_linear.weight, "b": None, "out": out}], addends=others, sigmoid=True)
        return [out], [sel.group], []

    def forward(self, x, x_group, targets=None):
        pred = self.plan_holder(x.shape[0]).run(x.to(torch.int32), x_group.reshape(-1).to(torch.int64))
        return pred.squeeze(1), targets
