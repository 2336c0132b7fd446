"""STAR (star-topology adaptive recommender) on the HIP hot path.

Mirror of the reference's model/star.py (STAR: lines 12-114, MDR_BatchNorm: lines 117-187): same constructor
arguments, parameter names and the two forward conventions
    forward(x)                      -> [B, n_tower]: every tower over the full batch (eval path run.py:669, CDC)
    forward(x, x_group, targets=y)  -> ([B,1] in ascending-group order, targets in the same order)  (run.py:477-480)
Per domain the effective parameters are W_domain * W_shared, b_domain + b_shared, gamma_d * gamma_s, beta_d + beta_s;
they are fused for ALL domains in one launch each, then every domain's layer runs as one group of a grouped MFMA
launch (ragged row ranges in grouped mode)."""
import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _NormBase

from .. import plan as P
from .layer import BaseModel, CrossNetwork, DNN, _reg_filter

MAX_GROUPED_TOWERS = 32


class MDR_BatchNorm(_NormBase):
    """Parameter container of the partitioned normalisation (model/star.py:117-187): affine terms are combined with the
    shared ones by the caller; the batch-of-one skip (star.py:134-135) lives in the BatchNorm kernel."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, device=None, dtype=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats, device=device, dtype=dtype)

    def _check_input_dim(self, input):
        if input.dim() != 2 and input.dim() != 3:
            raise ValueError("expected 2D or 3D input (got {}D input)".format(input.dim()))

    def forward(self, input, shared_weight, shared_bias):
        raise RuntimeError("MDR_BatchNorm runs inside STAR's launch plan (one kernel for all domains), not stand-alone")


class STAR(BaseModel):
    def __init__(self, feature_dims, embed_dim, n_tower, tower_dims, domain_idx=None, dropout=0.2, config=None,
                 l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5, device=None):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        self.model_name = 'star'
        self.n_tower = n_tower
        self.domain_idx = domain_idx
        self.device = device
        self.dropout_p = float(dropout)
        self.use_dcn = getattr(config, 'use_dcn', False)
        self.use_atten = getattr(config, 'use_atten', False)
        if self.use_dcn:
            self.cn = CrossNetwork(self.embed_output_dim, config.n_cross_layers)
        if self.use_atten:
            self.build_atten(config, dropout)
        self.shared_bn_weight = nn.Parameter(torch.ones(self.embed_output_dim))
        self.shared_bn_bias = nn.Parameter(torch.zeros(self.embed_output_dim))
        self.domain_norm = nn.ModuleList([MDR_BatchNorm(self.embed_output_dim) for _ in range(n_tower)])
        self.domain_dnns = nn.ModuleList([DNN(self.embed_output_dim, tower_dims, dropout_rate=dropout) for _ in range(n_tower)])
        self.domain_dnn_linears = nn.ModuleList([nn.Linear(tower_dims[-1], 1) for _ in range(n_tower)])
        self.shared_dnn = DNN(self.embed_output_dim, tower_dims, dropout_rate=dropout)
        self.shared_dnn_linear = nn.Linear(tower_dims[-1], 1)
        self.output_layers = nn.ModuleList([nn.Sigmoid() for _ in range(n_tower)])
        self.tower_dims = tuple(tower_dims)
        if self.use_dcn:
            self.add_regularization_weight(_reg_filter(self.cn), l2=l2_reg_cross)
        self.add_regularization_weight(_reg_filter(self.domain_dnns), l2=l2_reg_dnn)
        self.add_regularization_weight(_reg_filter(self.shared_dnn), l2=l2_reg_dnn)

    # ------------------------------------------------------------------------------------------------
    def describe(self, plan, emb, grouped=False):
        if self.use_dcn:
            raise RuntimeError("use_dcn=True cannot run: the reference adds a [B,E] cross output in place to a [B,1] "
                               "logit (model/star.py:103-107) and raises; so do we")
        n, Ed = self.n_tower, self.embed_output_dim
        E = emb.out
        ins, extra, ro = [], [], None
        if grouped:
            if n > MAX_GROUPED_TOWERS:
                raise NotImplementedError(f"grouped STAR forward supports up to {MAX_GROUPED_TOWERS} towers per launch")
            part = P.GroupPartition(plan, E, n)
            X, ro = part.out, part.row_offsets
            ins, extra = [part.group], [part.order]
        else:
            X = E
        wide = self.linear.describe(plan, X)
        atten = self.describe_atten(plan, X) if self.use_atten else None     # star.py:70-72 (per row: the partitioned rows do)

        def per_domain(buf, g, width):
            return buf if grouped else buf.slice(g * width, (g + 1) * width)

        # partitioned normalisation: gamma_d * gamma_s, beta_d + beta_s (star.py:169-176)
        gam = P.StarFuse(plan, [dn.weight for dn in self.domain_norm], self.shared_bn_weight, "mul")
        bet = P.StarFuse(plan, [dn.bias for dn in self.domain_norm], self.shared_bn_bias, "add")
        h = plan.new(Ed if grouped else n * Ed)
        segs = []
        for g, dn in enumerate(self.domain_norm):
            segs.append({"x": X, "out": per_domain(h, g, Ed), "gamma": gam.views[g].tensor, "beta": bet.views[g].tensor,
                         "dgamma": gam.views[g].grad_tensor, "dbeta": bet.views[g].grad_tensor,
                         "running_mean": dn.running_mean, "running_var": dn.running_var,
                         "num_batches_tracked": dn.num_batches_tracked, "row_group": g})
        P.BatchNorm(plan, segs, relu=False, dropout=False, row_offsets=ro)
        cur, cur_w = h, Ed
        for i, width in enumerate(self.tower_dims):
            W = P.StarFuse(plan, [d.linears[i].weight for d in self.domain_dnns], self.shared_dnn.linears[i].weight, "mul")
            b = P.StarFuse(plan, [d.linears[i].bias for d in self.domain_dnns], self.shared_dnn.linears[i].bias, "add")
            pre = plan.new(width if grouped else n * width)
            P.GLinear(plan, [{"x": per_domain(cur, g, cur_w), "w": W.views[g], "b": b.views[g], "out": per_domain(pre, g, width)}
                             for g in range(n)], row_offsets=ro)
            post = plan.new(width if grouped else n * width)
            segs = []
            for g, d in enumerate(self.domain_dnns):
                bn = d.bn[i]
                segs.append({"x": per_domain(pre, g, width), "out": per_domain(post, g, width), "gamma": bn.weight, "beta": bn.bias,
                             "gamma_param": bn.weight, "beta_param": bn.bias, "running_mean": bn.running_mean,
                             "running_var": bn.running_var, "num_batches_tracked": bn.num_batches_tracked, "row_group": g})
            P.BatchNorm(plan, segs, relu=True, dropout=True, row_offsets=ro, skip_le1=True)
            cur, cur_w = post, width
        wl = P.StarFuse(plan, [l.weight for l in self.domain_dnn_linears], self.shared_dnn_linear.weight, "mul")
        bl = P.StarFuse(plan, [l.bias for l in self.domain_dnn_linears], self.shared_dnn_linear.bias, "add")
        out = plan.new(1 if grouped else n)
        for c0 in range(0, n, 32):
            P.RowDot(plan, [{"x": per_domain(cur, g, cur_w), "w": wl.views[g], "b": bl.views[g],
                             "out": out if grouped else out.slice(g, g + 1)} for g in range(c0, min(n, c0 + 32))],
                     addends=[wide] if atten is None else [wide, atten], sigmoid=True, row_offsets=ro)
        return [out], ins, extra

    def forward(self, x, x_group=None, targets=None):
        B = x.shape[0]
        if x_group is None:
            return self.plan_holder(B, tag="all").run(x.to(torch.int32))
        holder = self.plan_holder(B, tag="grouped", grouped=True)
        pred, order = holder.run(x.to(torch.int32), x_group.reshape(-1).to(torch.int64))
        return pred, targets[order.long()]
