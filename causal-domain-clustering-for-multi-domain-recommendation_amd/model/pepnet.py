"""PEPNet / EPNet on the HIP hot path.  Mirror of the reference's model/pepnet.py:9-180 (PEPNet, GateNN, PPNetBlock):
    EPNet:  w = 2*sigmoid(GateNN(cat[e.detach(), domain_embed]));  e' = e * w
    PPNet:  per tower layer l: gates_l = 2*sigmoid(GateNN_l(cat[e.detach(), e'])) split per tower;
            h_t <- layer_l(h_t * gates_l[t]) where layer_l = Linear -> BatchNorm -> ReLU -> Dropout is ONE module shared by all
            towers (the reference builds `[one_tower_layer] * n_tower`): the weights see every tower's gradient, the
            BatchNorm statistics are the tower's own batch statistics and the running statistics are updated once per tower,
            in tower order.
    y_t = sigmoid(ppnet_linear_t(h_t) + linear(e) [+ attention branch])
Also the CDC base options 'pepnet' / 'epnet' (model/cdc.py:43-50)."""
import torch
from torch import nn

from .. import plan as P
from .layer import BaseModel, CrossNetwork, MultiLayerPerceptron, mlp_stack, _reg_filter


class GateNN(nn.Module):
    """model/pepnet.py:117-134 (parameter container: same Sequential indices, hence the same state_dict keys)"""

    def __init__(self, input_dim, hidden_dim=None, output_dim=None, dropout=0.0, batch_norm=False):
        super().__init__()
        if hidden_dim is None:
            hidden_dim = output_dim
        layers = [nn.Linear(input_dim, hidden_dim)]
        if batch_norm:
            raise NotImplementedError("GateNN(batch_norm=True) is never constructed by the reference")
        layers.append(nn.ReLU())
        if dropout > 0:
            layers.append(nn.Dropout(p=dropout))
        layers.append(nn.Linear(hidden_dim, output_dim))
        layers.append(nn.Sigmoid())
        self.gate = nn.Sequential(*layers)

    def describe_logits(self, plan, x):
        """the pre-sigmoid output; the caller applies `2 * sigmoid(.)` together with the product it feeds"""
        hidden = P.GLinear(plan, [{"x": x, "w": self.gate[0].weight, "b": self.gate[0].bias}], relu=True, dropout=True).outs[0]
        last = self.gate[-2]
        return P.GLinear(plan, [{"x": hidden, "w": last.weight, "b": last.bias}]).outs[0]


class PPNetBlock(nn.Module):
    """model/pepnet.py:137-180 (parameter container)"""

    def __init__(self, input_dim, gate_input_dim, tower_dims, gate_hidden_dim=None, n_tower=None, dropout=0.0, output_layer=False):
        super().__init__()
        if output_layer:
            raise NotImplementedError("PPNetBlock(output_layer=True) is never constructed by the reference")
        self.n_tower, self.n_layer = n_tower, len(tower_dims)
        self.gate_layers, self.tower_layers = nn.ModuleList(), nn.ModuleList()
        dims = (input_dim,) + tuple(tower_dims)
        for idx in range(self.n_layer):
            dense = [nn.Linear(dims[idx], dims[idx + 1]), nn.BatchNorm1d(dims[idx + 1]), nn.ReLU()]
            if dropout > 0:
                dense.append(nn.Dropout(p=dropout))
            one = nn.Sequential(*dense)
            self.tower_layers.append(nn.ModuleList([one] * self.n_tower))          # ONE module, n_tower references
            self.gate_layers.append(GateNN(gate_input_dim + input_dim, gate_hidden_dim, output_dim=dims[idx] * self.n_tower))
        self.dims = dims


class PEPNet(BaseModel):
    def __init__(self, feature_dims, embed_dim, n_tower, tower_dims, gate_hidden_dim=64, domain_idx=None, use_ppnet=True, dropout=0.2,
                 config=None, l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5):
        super().__init__(feature_dims, embed_dim, l2_reg_embedding=l2_reg_embedding, l2_reg_linear=l2_reg_linear)
        if use_ppnet:
            self.model_name = 'pepnet' if n_tower > 1 else 'pepnet-single'
        else:
            self.model_name = 'epnet' if n_tower > 1 else 'epnet-single'
        self.n_tower, self.domain_idx, self.use_ppnet = n_tower, domain_idx, use_ppnet
        self.dropout_p = float(dropout)
        self.use_dcn = getattr(config, 'use_dcn', False)
        self.use_atten = getattr(config, 'use_atten', False)
        if self.use_dcn:
            self.cn = CrossNetwork(self.embed_output_dim, config.n_cross_layers)
        if self.use_atten:
            self.build_atten(config, dropout)
        Ed = self.embed_output_dim
        self.epnet = GateNN(Ed + embed_dim, gate_hidden_dim, Ed, dropout=dropout)
        if use_ppnet:
            self.ppnet = PPNetBlock(input_dim=Ed, gate_input_dim=Ed, tower_dims=tuple(tower_dims), gate_hidden_dim=gate_hidden_dim,
                                    n_tower=n_tower, dropout=dropout, output_layer=False)
            self.ppnet_linears = nn.ModuleList([nn.Linear(tower_dims[-1], 1, bias=False) for _ in range(n_tower)])
            self.output_layers = nn.ModuleList([nn.Sigmoid() for _ in range(n_tower)])
        elif n_tower > 1:
            self.towers = nn.ModuleList(MultiLayerPerceptron(Ed, tower_dims, dropout, output_layer=False) for _ in range(n_tower))
            self.ppnet_linears = nn.ModuleList([nn.Linear(tower_dims[-1], 1, bias=False) for _ in range(n_tower)])
            self.output_layers = nn.ModuleList([nn.Sigmoid() for _ in range(n_tower)])
        else:
            self.towers = MultiLayerPerceptron(Ed, tower_dims, dropout, output_layer=False)
            self.ppnet_linears = nn.Linear(tower_dims[-1], 1, bias=False)
            self.output_layers = nn.Sigmoid()
        self.add_regularization_weight(_reg_filter(self.epnet), l2=l2_reg_dnn)
        if self.use_dcn:
            self.add_regularization_weight(_reg_filter(self.cn), l2=l2_reg_cross)
        self.add_regularization_weight(_reg_filter(self.ppnet if use_ppnet else self.towers), l2=l2_reg_dnn)

    # ------------------------------------------------------------------------------------------------
    def _describe_ppnet(self, plan, E, epnet_out):
        """PPNetBlock.forward (model/pepnet.py:169-180) with the shared tower layers."""
        pp, n, Ed = self.ppnet, self.n_tower, self.embed_output_dim
        gate_in = plan.new(2 * Ed)                                          # cat([feature_emb.detach(), gate_emb], dim=-1)
        P.CopyCols(plan, E, gate_in.slice(0, Ed), detach=True)
        P.CopyCols(plan, epnet_out, gate_in.slice(Ed, 2 * Ed))
        cur = [E] * n
        dev = plan.device
        for idx in range(pp.n_layer):
            lin, bn = pp.tower_layers[idx][0][0], pp.tower_layers[idx][0][1]
            K, N = pp.dims[idx], pp.dims[idx + 1]
            logits = pp.gate_layers[idx].describe_logits(plan, gate_in)      # [B, K * n]
            gated = [P.SigmoidGate(plan, cur[t], logits.slice(t * K, (t + 1) * K)).out for t in range(n)]
            pre = P.GLinear(plan, [{"x": gated[t], "w": lin.weight, "b": lin.bias} for t in range(n)]).outs
            # one BatchNorm module applied n times: own batch statistics per tower, shared affine parameters and running stats
            dg = [torch.zeros(N, device=dev) for _ in range(n)]
            db = [torch.zeros(N, device=dev) for _ in range(n)]
            if plan.training:
                rm = [torch.zeros(N, device=dev) for _ in range(n)]
                rv = [torch.zeros(N, device=dev) for _ in range(n)]
            else:
                rm, rv = [bn.running_mean] * n, [bn.running_var] * n
            gg, gb = plan.param_grad(bn.weight), plan.param_grad(bn.bias)
            if plan._claim_param(bn.weight) or plan._claim_param(bn.bias):
                raise RuntimeError("a shared PPNet BatchNorm appears twice in one plan")

            def sum_grads(gg=gg, gb=gb, dg=dg, db=db):                      # fixed tower order: deterministic
                gg.copy_(dg[0]); gb.copy_(db[0])
                for t in range(1, len(dg)):
                    gg.add_(dg[t]); gb.add_(db[t])

            def zero_scratch(rm=rm, rv=rv):
                for t in range(len(rm)):
                    rm[t].zero_(); rv[t].zero_()

            def update_running(bn=bn, rm=rm, rv=rv):
                # the kernel left momentum * statistic in the zeroed scratch; apply the towers' updates one after the other
                for t in range(len(rm)):
                    bn.running_mean.mul_(1.0 - 0.1).add_(rm[t])
                    bn.running_var.mul_(1.0 - 0.1).add_(rv[t])
                bn.num_batches_tracked.add_(len(rm))

            P.HostStep(plan, bwd=sum_grads)                                  # runs after the BatchNorm backward (reverse order)
            if plan.training:
                P.HostStep(plan, fwd=zero_scratch)
            post = [plan.new(N) for _ in range(n)]
            segs = [{"x": pre[t], "out": post[t], "gamma": bn.weight, "beta": bn.bias, "dgamma": dg[t], "dbeta": db[t],
                     "running_mean": rm[t], "running_var": rv[t], "num_batches_tracked": None, "row_group": 0} for t in range(n)]
            P.BatchNorm(plan, segs, relu=True, dropout=True)
            if plan.training:
                P.HostStep(plan, fwd=update_running)
            cur = post
        return cur

    def describe(self, plan, emb):
        if self.use_dcn:
            raise RuntimeError("use_dcn=True cannot run: the reference adds a [B,E] cross output in place to a [B,1] logit "
                               "(model/pepnet.py:83-101) and raises; so do we")
        E, D, Ed, n = emb.out, self.embed_dim, self.embed_output_dim, self.n_tower
        dom = E.slice(self.domain_idx * D, (self.domain_idx + 1) * D)
        ep_in = plan.new(Ed + D)                                            # cat([embed_x.detach(), domain_embed], dim=-1)
        P.CopyCols(plan, E, ep_in.slice(0, Ed), detach=True)
        P.CopyCols(plan, dom, ep_in.slice(Ed, Ed + D))
        epnet_out = P.SigmoidGate(plan, E, self.epnet.describe_logits(plan, ep_in)).out      # embed_x * (sigmoid(.) * 2)
        others = [self.linear.describe(plan, E)]
        if self.use_atten:
            others.append(self.describe_atten(plan, E))
        if self.use_ppnet:
            tops = self._describe_ppnet(plan, E, epnet_out)
            linears = list(self.ppnet_linears)
        elif n > 1:
            tops, _ = mlp_stack(plan, list(self.towers), [epnet_out] * n)
            linears = list(self.ppnet_linears)
        else:
            tops, _ = mlp_stack(plan, [self.towers], [epnet_out])
            linears = [self.ppnet_linears]
        out = plan.new(len(linears))
        P.RowDot(plan, [{"x": tops[t], "w": linears[t].weight, "b": None, "out": out.slice(t, t + 1)} for t in range(len(linears))],
                 addends=others, sigmoid=True)
        return [out], [], []

    def forward(self, x):
        pred = self.plan_holder(x.shape[0]).run(x.to(torch.int32))
        if not self.use_ppnet and self.n_tower == 1:
            return pred.squeeze(1)                                            # model/pepnet.py:112
        return pred
