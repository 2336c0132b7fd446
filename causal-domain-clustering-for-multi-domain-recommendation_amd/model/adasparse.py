"""AdaSparse on the HIP hot path.  Mirror of the reference's model/adasparse.py:16-116 (DNN_w_Pruner + AdaSparse):
every DNN layer's output is scaled by a per-sample, per-unit pruning gate computed from the layer input and the (detached)
domain embedding, gates below the threshold are cut to zero:
    fc = linear_i(h);  pi = 2*sigmoid(pruner_i(cat[h, dom]));  pi[pi <= 0.25] = 0;  h = dropout(relu(bn_i(fc * pi)))
    y = sigmoid(dnn_linear(h) + linear(e) [+ attention branch])"""
import torch
from torch import nn

from .. import plan as P
from .layer import BaseModel, CrossNetwork, _bn_seg, _reg_filter


class DNN_w_Pruner(nn.Module):
    """model/adasparse.py:16-65 (parameter container; the forward is part of AdaSparse's plan)"""

    def __init__(self, inputs_dim, hidden_units, domain_emb_dim, init_std=0.0001, use_bn=False, dropout_rate=0):
        super().__init__()
        if len(hidden_units) == 0:
            raise ValueError("hidden_units is empty!!")
        self.dropout_rate = dropout_rate
        self.use_bn = use_bn
        dims = [inputs_dim] + list(hidden_units)
        self.linears = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.pruners = nn.ModuleList([nn.Linear(dims[i] + domain_emb_dim, dims[i + 1]) for i in range(len(dims) - 1)])
        if self.use_bn:
            self.bn = nn.ModuleList([nn.BatchNorm1d(dims[i + 1]) for i in range(len(dims) - 1)])
        for name, tensor in self.linears.named_parameters():
            if 'weight' in name:
                nn.init.normal_(tensor, mean=0, std=init_std)
        self.alpha, self.beta, self.epsilon = 1, 2.0, 0.25


class AdaSparse(BaseModel):
    def __init__(self, feature_dims, embed_dim, hidden_dims, domain_idx=None, dropout=0.2, config=None,
                 l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.model_name = 'adasparse'
        self.domain_idx = domain_idx
        self.dropout_p = float(dropout)
        self.use_dcn = getattr(config, 'use_dcn', False)
        self.use_atten = getattr(config, 'use_atten', False)
        if self.use_dcn:
            self.cn = CrossNetwork(self.embed_output_dim, config.n_cross_layers)
        if self.use_atten:
            self.build_atten(config, dropout)
        self.dnn = DNN_w_Pruner(self.embed_output_dim, hidden_dims, domain_emb_dim=embed_dim, use_bn=True, dropout_rate=dropout)
        self.dnn_linear = nn.Linear(hidden_dims[-1], 1)
        self.output_layer = nn.Sigmoid()
        if self.use_dcn:
            self.add_regularization_weight(_reg_filter(self.cn), l2=l2_reg_cross)
        self.add_regularization_weight(_reg_filter(self.dnn), l2=l2_reg_dnn)

    def describe(self, plan, emb):
        if self.use_dcn:
            raise RuntimeError("use_dcn=True cannot run: the reference adds a [B,E] cross output in place to a [B,1] logit "
                               "(model/adasparse.py:103-112) and raises; so do we")
        E, D = emb.out, self.embed_dim
        dom = E.slice(self.domain_idx * D, (self.domain_idx + 1) * D)       # embed_x[:, domain_idx, :].detach()
        dnn = self.dnn
        h = E
        for i, (lin, pruner) in enumerate(zip(dnn.linears, dnn.pruners)):
            K = lin.weight.shape[1]
            cat = plan.new(K + D)                                           # torch.cat([deep_input, domain_embs], dim=1)
            P.CopyCols(plan, h, cat.slice(0, K))
            P.CopyCols(plan, dom, cat.slice(K, K + D), detach=True)
            both = P.GLinear(plan, [{"x": h, "w": lin.weight, "b": lin.bias}, {"x": cat, "w": pruner.weight, "b": pruner.bias}])
            gated = P.SigmoidGate(plan, both.outs[0], both.outs[1], beta=dnn.beta, alpha=dnn.alpha, eps=dnn.epsilon).out
            post = plan.new(lin.weight.shape[0])
            seg = _bn_seg(gated, dnn.bn[i], out=post)
            P.BatchNorm(plan, [seg], relu=True, dropout=True)               # bn -> relu -> dropout (adasparse.py:58-63)
            h = post
        others = [self.linear.describe(plan, E)]
        if self.use_atten:
            others.append(self.describe_atten(plan, E))
        out = plan.new(1)
        P.RowDot(plan, [{"x": h, "w": self.dnn_linear.weight, "b": self.dnn_linear.bias, "out": out}], addends=others, sigmoid=True)
        return [out], [], []

    def forward(self, x):
        return self.plan_holder(x.shape[0]).run(x.to(torch.int32)).squeeze(1)
