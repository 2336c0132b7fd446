"""ADL on the HIP hot path.  Mirror of the reference's model/adl.py:12-126:
    routing (no gradient): coefficients = softmax(e @ centers^T); tower = argmax; the centres move towards the normalised
    coefficient-weighted sum of the batch (rate 0.9) — model/adl.py:62-79, including its quirk that every routing iteration
    scores against the SAME old centres;
    every sample runs through ITS tower's MLP (rows partitioned by tower: per-tower BatchNorm statistics), the output layer
    is the tower's Linear(-> 1) multiplied element-wise with the shared one (biases added), plus linear(e) [+ attention];
    training returns the predictions in tower order with the targets permuted alike, `is_training=False` the predictions
    in batch order.
The routing is three small torch matmuls inside the launch sequence (no gradient, [B,E]x[E,n_tower]); the partition, the
per-tower layers with ragged row groups, the fused output weights and the row dot reuse STAR's machinery."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import plan as P
from .layer import BaseModel, CrossNetwork, MultiLayerPerceptron, _bn_seg, _reg_filter

MAX_GROUPED_TOWERS = 32


class ADL(BaseModel):
    def __init__(self, feature_dims, embed_dim, n_tower, tower_dims, domain_idx=None, dropout=0.2, l2_reg_embedding=1e-5,
                 l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5, dlm_iters=3, dlm_update_rate=0.9, device=None, config=None):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.model_name = 'adl'
        self.n_tower = self.cluster_num = n_tower
        self.domain_idx, self.device = domain_idx, device
        self.dlm_iters, self.dlm_update_rate = dlm_iters, dlm_update_rate
        self.dropout_p = float(dropout)
        # a plain attribute in the reference (not a parameter or buffer: it is not part of the state_dict)
        self.cluster_centers = torch.randn((self.cluster_num, self.embed_output_dim)).to(self.device)
        self.use_dcn = getattr(config, 'use_dcn', False)
        self.use_atten = getattr(config, 'use_atten', False)
        if self.use_dcn:
            self.cn = CrossNetwork(self.embed_output_dim, config.n_cross_layers)
        if self.use_atten:
            self.build_atten(config, dropout)
        self.domain_mlps = nn.ModuleList([MultiLayerPerceptron(self.embed_output_dim, tower_dims, dropout, output_layer=False)
                                          for _ in range(n_tower)])
        self.domain_mlps_linears = nn.ModuleList([nn.Linear(tower_dims[-1], 1) for _ in range(n_tower)])
        self.shared_mlps = MultiLayerPerceptron(self.embed_output_dim, tower_dims, dropout, output_layer=False)
        self.shared_mlps_linear = nn.Linear(tower_dims[-1], 1)
        self.output_layers = nn.ModuleList([nn.Sigmoid() for _ in range(n_tower)])
        if self.use_dcn:
            self.add_regularization_weight(_reg_filter(self.cn), l2=l2_reg_cross)
        self.add_regularization_weight(_reg_filter(self.domain_mlps), l2=l2_reg_dnn)
        self.add_regularization_weight(_reg_filter(self.shared_mlps), l2=l2_reg_dnn)

    def to(self, *args, **kwargs):
        out = super().to(*args, **kwargs)
        dev = self.embedding.embedding_dict.weight.device
        self.cluster_centers = self.cluster_centers.to(dev)       # the reference creates it on `device` directly
        return out

    def DLM_routing(self, embed_x):
        """model/adl.py:62-79, as written there (every iteration scores against the centres from before the call)"""
        with torch.no_grad():
            for _ in range(self.dlm_iters):
                coeff = F.softmax(torch.matmul(embed_x, self.cluster_centers.t()), dim=1)
                tmp = F.normalize(torch.matmul(coeff.t(), embed_x), p=2, dim=1)
            self.cluster_centers.copy_(F.normalize(self.dlm_update_rate * self.cluster_centers + (1 - self.dlm_update_rate) * tmp,
                                                   p=2, dim=1))
        return coeff

    def describe(self, plan, emb, is_training=True):
        if self.use_dcn:
            raise RuntimeError("use_dcn=True cannot run: the reference adds a [B,E] cross output in place to a [B,1] logit "
                               "(model/adl.py:98-115) and raises; so do we")
        n, E = self.n_tower, emb.out
        if n > MAX_GROUPED_TOWERS:
            raise NotImplementedError(f"ADL supports up to {MAX_GROUPED_TOWERS} towers per launch")
        route = P.HostStep(plan)                                  # filled in below: needs the partition op's group tensor
        part = P.GroupPartition(plan, E, n)
        et = E.tensor()

        def routing(part=part, et=et):
            part.group.copy_(torch.argmax(self.DLM_routing(et), dim=1))

        route.fwd = routing
        X, ro = part.out, part.row_offsets
        others = [self.linear.describe(plan, X)]
        if self.use_atten:
            others.append(self.describe_atten(plan, X))
        cur = X
        for i in range(len(self.domain_mlps[0].hidden)):
            layers = [m.hidden[i] for m in self.domain_mlps]
            width = layers[0][0].weight.shape[0]
            pre = plan.new(width)
            P.GLinear(plan, [{"x": cur, "w": lin.weight, "b": lin.bias, "out": pre} for lin, _ in layers], row_offsets=ro)
            post = plan.new(width)
            segs = [_bn_seg(pre, norm, out=post, row_group=g) for g, (_, norm) in enumerate(layers)]
            P.BatchNorm(plan, segs, relu=True, dropout=True, row_offsets=ro)       # BatchNorm skipped for a one-row tower
            cur = post
        wl = P.StarFuse(plan, [l.weight for l in self.domain_mlps_linears], self.shared_mlps_linear.weight, "mul")
        bl = P.StarFuse(plan, [l.bias for l in self.domain_mlps_linears], self.shared_mlps_linear.bias, "add")
        out = plan.new(1)
        P.RowDot(plan, [{"x": cur, "w": wl.views[g], "b": bl.views[g], "out": out} for g in range(n)], addends=others, sigmoid=True,
                 row_offsets=ro)
        if is_training:
            return [out], [part.group], [part.order]
        ys = torch.zeros((plan.B, 1), dtype=torch.float32, device=plan.device)
        ot = out.tensor()
        P.HostStep(plan, fwd=lambda ys=ys, ot=ot, part=part: ys.index_copy_(0, part.order.long(), ot))   # ys_tensor[mask] = ...
        return [out], [part.group], [part.order, ys]

    def forward(self, x, group=None, targets=None, is_training=True):
        B = x.shape[0]
        dummy = torch.zeros(B, dtype=torch.int64, device=x.device)             # the routing launch overwrites the group ids
        holder = self.plan_holder(B, tag="train" if is_training else "infer", is_training=is_training)
        if is_training:
            pred, order = holder.run(x.to(torch.int32), dummy)
            return pred, targets[order.long()]
        _, _, ys = holder.run(x.to(torch.int32), dummy)
        return ys
