"""Layer library of the multi-domain CTR models, MI355X-native.

Same class names, constructor arguments, parameter names (state_dict keys) and forward() conventions
as the reference's model/layer.py, but no ATen compute: every forward() describes its dataflow to a
static launch plan (plan.py) whose steps are the hand-written gfx950 kernels behind include/cdcmdr.h.

Reference citations (file:line under the reference repository):
  BaseModel                model/layer.py:10-112
  FeaturesLinear           model/layer.py:115-126
  FeaturesEmbedding        model/layer.py:129-157
  MultiLayerPerceptron     model/layer.py:178-206
  DNN                      model/layer.py:209-300
  CrossNetwork             model/layer.py:303-329
  CrossNetV2 / CrossNetMix model/layer.py:332-407
"""
import os

import numpy as np
import torch
from torch import nn

from .. import plan as P
from ..functional import PlanCache, PlanHolder


class HipModule(nn.Module):
    """Common plumbing: arithmetic precision of the MFMA contractions and the plan cache."""

    precision = "bf16"      # "bf16": bf16 MFMA operands / fp32 accumulate;  "f32": exact fp32 MFMA

    def _cache(self):
        c = self.__dict__.get("_plan_cache")
        if c is None:
            c = PlanCache()
            self.__dict__["_plan_cache"] = c
        return c

    def set_precision(self, precision):
        assert precision in ("bf16", "f32")
        for m in self.modules():
            if isinstance(m, HipModule):
                m.precision = precision
                m._cache().clear()
        return self

    def _new_plan(self, B, device, seed=0):
        return P.Plan(device, B, precision=self.precision, training=self.training,
                      dropout=float(getattr(self, "dropout_p", 0.0)), seed=seed)


# --------------------------------------------------------------------------------------------------
# embedding + wide term
# --------------------------------------------------------------------------------------------------
class FeaturesEmbedding(HipModule):
    """One flat table for all fields; forward adds the per-field offsets in x's dtype (int32) and gathers.
    Reference: model/layer.py:129-157."""

    def __init__(self, field_dims, embed_dim):
        super().__init__()
        self.field_num = len(field_dims)
        self.output_dim0 = self.field_num
        self.embed_dim = embed_dim
        total = int(np.sum(np.asarray(field_dims, dtype=np.int64)))
        self.embedding_dict = nn.Embedding(total, embed_dim)          # N(0,1) init, as the reference
        self.offsets = np.array((0, *np.cumsum(np.asarray(field_dims, dtype=np.int64))[:-1]), dtype=np.longlong)

    def offsets_device(self, device):
        cached = self.__dict__.get("_offsets_dev")
        if cached is None or cached.device != torch.device(device):
            # x.new_tensor(offsets) in the reference takes x's dtype: int32 (wrapping on overflow)
            cached = torch.from_numpy(self.offsets.astype(np.int64)).to(torch.int32).to(device)
            self.__dict__["_offsets_dev"] = cached
        return cached

    def describe(self, plan):
        table = self.embedding_dict.weight
        return P.EmbedGather(plan, table, self.offsets_device(table.device), self.field_num, self.embed_dim)

    def forward(self, x, squeeze_dim=False):
        B = x.shape[0]
        dev = self.embedding_dict.weight.device

        def build():
            plan = self._new_plan(B, dev)
            op = self.describe(plan)
            plan.finalize([op.out])
            return PlanHolder(plan, [op.ids], [op.out], emb_op=op)

        holder = self._cache().get(self, "emb", B, build)
        out = holder.run(x.to(torch.int32))
        return out if squeeze_dim else out.view(B, self.field_num, self.embed_dim)


class FeaturesLinear(HipModule):
    """Wide term: one output unit over the flattened embeddings. Reference: model/layer.py:115-126."""

    def __init__(self, field_dims, output_dim=1, sigmoid=False):
        super().__init__()
        if output_dim != 1:
            raise NotImplementedError("the reference only ever builds FeaturesLinear with output_dim=1")
        self.fc = nn.Linear(field_dims, output_dim, bias=True)
        self.sigmoid = sigmoid

    def describe(self, plan, x, out=None, addends=()):
        op = P.RowDot(plan, [{"x": x, "w": self.fc.weight, "b": self.fc.bias, "out": out}], addends=addends,
                      sigmoid=self.sigmoid)
        return op.outs[0]

    def forward(self, x):
        return _run_float_module(self, "lin", x, lambda plan, xb: [self.describe(plan, xb)])


def _run_float_module(mod, tag, x, describe):
    """Standalone forward of a layer module on a float input [B, K]."""
    B, K = x.shape
    dev = x.device

    def build():
        plan = mod._new_plan(B, dev)
        xb = plan.new(K)
        outs = describe(plan, xb)
        plan.finalize(outs)
        return PlanHolder(plan, [xb], outs)

    holder = mod._cache().get(mod, (tag, K), B, build)
    return holder.run(x.contiguous().float())


# --------------------------------------------------------------------------------------------------
# MLPs
# --------------------------------------------------------------------------------------------------
class MultiLayerPerceptron(HipModule):
    """[Linear, (BatchNorm1d), ReLU, Dropout] x n (+ Linear -> 1). BatchNorm is skipped for a batch of one row.
    Reference: model/layer.py:178-206 (module order inside `layers` fixes the state_dict indices)."""

    def __init__(self, input_dim, layer_dims, dropout, output_layer=True, bn=True):
        super().__init__()
        self.layers = nn.ModuleList()
        self.use_bn = bn
        self.dropout_p = float(dropout)
        self.hidden = []                   # (linear, bn or None) per hidden layer
        for layer_dim in layer_dims:
            lin = nn.Linear(input_dim, layer_dim)
            self.layers.append(lin)
            norm = None
            if bn:
                norm = nn.BatchNorm1d(layer_dim)
                self.layers.append(norm)
            self.layers.append(nn.ReLU())
            self.layers.append(nn.Dropout(p=dropout))
            self.hidden.append((lin, norm))
            input_dim = layer_dim
        self.out_linear = None
        if output_layer:
            lin = nn.Linear(input_dim, 1)
            self.layers.append(lin)
            self.__dict__["out_linear"] = lin     # alias only: parameters are registered under `layers`

    def describe(self, plan, x):
        outs, _ = mlp_stack(plan, [self], [x])
        return outs[0]

    def forward(self, x):
        return _run_float_module(self, "mlp", x, lambda plan, xb: [self.describe(plan, xb)])


def _bn_seg(x, norm, out=None, row_group=0):
    return {"x": x, "out": out, "gamma": norm.weight, "beta": norm.bias, "gamma_param": norm.weight,
            "beta_param": norm.bias, "running_mean": norm.running_mean, "running_var": norm.running_var,
            "num_batches_tracked": norm.num_batches_tracked, "row_group": row_group}


def mlp_stack(plan, mlps, inputs, extra_groups=(), final_addends=(), final_sigmoid=False, final_outs=None, last_outs=None,
              head_out=None, head_wide=None):
    """Runs structurally identical MultiLayerPerceptrons side by side: one grouped-linear launch (and one
    BatchNorm launch) per layer depth for all of them.

    extra_groups: extra linear groups (dicts x/w/b/out) riding in the first launch (gates).
    last_outs: where the last hidden layer's activations go (e.g. columns of a concat buffer) instead of a new buffer.
    Returns (outputs per mlp, outputs of the extra groups)."""
    n = len(mlps)
    cur = list(inputs)
    extra_outs = []
    depth = len(mlps[0].hidden)
    if depth == 0 and extra_groups:
        op = P.GLinear(plan, [dict(g) for g in extra_groups])
        extra_outs = op.outs
    for j in range(depth):
        width = mlps[0].hidden[j][0].out_features
        use_bn = mlps[0].use_bn
        pre = plan.new(n * width)
        dst = None
        if j == depth - 1 and last_outs is not None:
            dst = list(last_outs)
        groups = []
        for i, m in enumerate(mlps):
            lin = m.hidden[j][0]
            o = pre.slice(i * width, (i + 1) * width) if (use_bn or dst is None) else dst[i]
            groups.append({"x": cur[i], "w": lin.weight, "b": lin.bias, "out": o})
        n_extra = 0
        # The gates read the level's input like the first layer, but the first layer's launch is exactly one round of tiles at C2
        # (8 experts x 256 columns x 4096 rows = 512 tiles of 128 x 128 on 256 CUs x 2) and the gates' 128 nearly empty tiles
        # made it a second round: 28.9 us against 13.6 us for the second layer's launch, which has 256 slots to spare.  So with
        # two or more layers the gates ride in the LAST layer's forward launch; their grad-input stays with the first layer's
        # (further reduction segments of the same output: adopt_bwd_x below).
        gate_layer = depth - 1 if (depth >= 2 and plan.use_g2) else 0
        if j == gate_layer and extra_groups:
            for g in extra_groups:
                g = dict(g)
                g["act_cols"] = 0
                groups.append(g)
                n_extra += 1
        if use_bn:
            op = P.GLinear(plan, groups, relu=False, dropout=False)
            post = None
            if dst is None:
                post = plan.new(n * width)
                dst = [post.slice(i * width, (i + 1) * width) for i in range(n)]
            segs = [_bn_seg(pre.slice(i * width, (i + 1) * width), m.hidden[j][1], out=dst[i]) for i, m in enumerate(mlps)]
            if j < depth - 1 and width % 8 == 0 and post is not None:
                for sg in segs:                       # read by the next layer's contractions only (see the BatchNorm-free case below)
                    sg["half_only"] = True
            P.BatchNorm(plan, segs, relu=True, dropout=True)
            cur = dst
        else:
            # hidden activations of a BatchNorm-free stack are read by the next layer's contractions only (forward, grad-weight,
            # and as the activation mask of its grad-input): with the bf16-shadow path they never exist in fp32
            if j < depth - 1 and width % 8 == 0 and dst is None:
                for g in groups[:n]:
                    g["half_only"] = True
            op = P.GLinear(plan, groups, relu=True, dropout=True)
            cur = [op.outs[i] for i in range(n)]
        if j == 0:
            first_op = op
        if n_extra:
            extra_outs = op.outs[n:]
            if op is not first_op:
                first_op.adopt_bwd_x(op, op.groups[n:])
    if mlps[0].out_linear is not None and head_out is not None:
        # the output layers of all towers, the wide term and the sigmoid in one launch per direction (csrc/head.hip)
        towers = [{"x": cur[i], "w": m.out_linear.weight, "b": m.out_linear.bias} for i, m in enumerate(mlps)]
        P.TowerHead(plan, towers, head_out, wide=head_wide, addends=final_addends, sigmoid=final_sigmoid)
        return [head_out.slice(i, i + 1) for i in range(n)], extra_outs
    if mlps[0].out_linear is not None:
        groups = []
        for i, m in enumerate(mlps):
            lin = m.out_linear
            groups.append({"x": cur[i], "w": lin.weight, "b": lin.bias,
                           "out": None if final_outs is None else final_outs[i]})
        op = P.RowDot(plan, groups, addends=final_addends, sigmoid=final_sigmoid)
        cur = op.outs
    return cur, extra_outs


def activation_layer(act_name):
    """Reference: model/layer.py:209-238 (DeepCTR-style activation factory)."""
    if isinstance(act_name, str):
        name = act_name.lower()
        if name == "sigmoid":
            return nn.Sigmoid()
        if name == "linear":
            return nn.Identity()
        if name == "relu":
            return nn.ReLU(inplace=True)
        if name == "prelu":
            return nn.PReLU()
        raise NotImplementedError(act_name)
    if isinstance(act_name, type) and issubclass(act_name, nn.Module):
        return act_name()
    raise NotImplementedError


class DNN(HipModule):
    """DeepCTR-style MLP with separate `linears` / `bn` / `activation_layers` lists (used by STAR).
    Reference: model/layer.py:241-300."""

    def __init__(self, inputs_dim, hidden_units, activation="relu", dropout_rate=0, use_bn=True):
        super().__init__()
        if len(hidden_units) == 0:
            raise ValueError("hidden_units is empty!!")
        if str(activation).lower() != "relu":
            raise NotImplementedError("the HIP path implements the relu DNN the reference models use")
        self.dropout_rate = dropout_rate
        self.dropout_p = float(dropout_rate)
        self.dropout = nn.Dropout(dropout_rate)
        self.use_bn = use_bn
        dims = [inputs_dim] + list(hidden_units)
        self.linears = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        if self.use_bn:
            self.bn = nn.ModuleList([nn.BatchNorm1d(dims[i + 1]) for i in range(len(dims) - 1)])
        self.activation_layers = nn.ModuleList([activation_layer(activation) for _ in range(len(dims) - 1)])

    def describe(self, plan, x):
        cur = x
        for i, lin in enumerate(self.linears):
            if self.use_bn:
                op = P.GLinear(plan, [{"x": cur, "w": lin.weight, "b": lin.bias}])
                bn = P.BatchNorm(plan, [_bn_seg(op.outs[0], self.bn[i])], relu=True, dropout=True)
                cur = bn.outs[0]
            else:
                op = P.GLinear(plan, [{"x": cur, "w": lin.weight, "b": lin.bias}], relu=True, dropout=True)
                cur = op.outs[0]
        return cur

    def forward(self, inputs):
        return _run_float_module(self, "dnn", inputs, lambda plan, xb: [self.describe(plan, xb)])


# --------------------------------------------------------------------------------------------------
# cross networks
# --------------------------------------------------------------------------------------------------
class CrossNetwork(HipModule):
    """DCN-v1: x_{l+1} = x0 * (x_l . w_l) + b_l + x_l.  Reference: model/layer.py:303-329."""

    def __init__(self, input_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.w = nn.ModuleList([nn.Linear(input_dim, 1, bias=False) for _ in range(num_layers)])
        self.b = nn.ParameterList([nn.Parameter(torch.zeros((input_dim,))) for _ in range(num_layers)])

    def describe(self, plan, x0, out=None):
        cur = x0
        for i in range(self.num_layers):
            last = i == self.num_layers - 1
            dst = out if (last and out is not None) else None
            op = P.CrossLayer(plan, x0, cur, self.w[i].weight, self.b[i], out=dst)
            cur = op.out
        return cur

    def forward(self, x):
        return _run_float_module(self, "cross", x, lambda plan, xb: [self.describe(plan, xb)])


class CrossNetV2(HipModule):
    """DCN-v2 full-matrix cross: x_{l+1} = x0 * (W_l x_l) + b_l + x_l.  Reference: model/layer.py:332-343."""

    def __init__(self, input_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.w = nn.ModuleList([nn.Linear(input_dim, input_dim, bias=False) for _ in range(num_layers)])
        self.b = nn.ParameterList([nn.Parameter(torch.zeros((input_dim,))) for _ in range(num_layers)])

    def describe(self, plan, x0, out=None):
        cur = x0
        for i in range(self.num_layers):
            u = P.GLinear(plan, [{"x": cur, "w": self.w[i].weight, "b": None}]).outs[0]
            dst = out if (i == self.num_layers - 1) else None
            cur = P.CrossCombine(plan, x0, u, b2=self.b[i], r=cur, out=dst).out
        return cur

    def forward(self, x):
        return _run_float_module(self, "crossv2", x, lambda plan, xb: [self.describe(plan, xb)])


class CrossNetMix(HipModule):
    """DCN-v2 mixture of low-rank experts.  Reference: model/layer.py:346-407.
    Per layer: gate score x_l.g_k per expert (the n gates are shared by every layer), expert
    x0 * (U_k tanh(C_k tanh(V_k^T x_l)) + bias_l), softmax-weighted mixture, residual."""

    def __init__(self, input_dim, num_layers=2, low_rank=32, num_experts=4):
        super().__init__()
        self.num_layers, self.num_experts, self.low_rank, self.input_dim = num_layers, num_experts, low_rank, input_dim
        self.u_list = nn.ParameterList([nn.Parameter(nn.init.xavier_normal_(torch.empty(num_experts, input_dim, low_rank)))
                                        for _ in range(num_layers)])
        self.v_list = nn.ParameterList([nn.Parameter(nn.init.xavier_normal_(torch.empty(num_experts, input_dim, low_rank)))
                                        for _ in range(num_layers)])
        self.c_list = nn.ParameterList([nn.Parameter(nn.init.xavier_normal_(torch.empty(num_experts, low_rank, low_rank)))
                                        for _ in range(num_layers)])
        self.gating = nn.ModuleList([nn.Linear(input_dim, 1, bias=False) for _ in range(num_experts)])
        self.bias = nn.ParameterList([nn.Parameter(nn.init.zeros_(torch.empty(input_dim, 1))) for _ in range(num_layers)])

    def describe(self, plan, x0, out=None):
        if plan.B == 1:
            # model/layer.py:406 `x_l.squeeze()` drops the batch dimension too; every caller then fails (golden g6)
            raise IndexError("Dimension out of range (CrossNetMix squeezes a batch of one to 1-D, as the reference does)")
        n, r, E = self.num_experts, self.low_rank, self.input_dim
        xl = x0
        for i in range(self.num_layers):
            scores = plan.new(n)
            P.RowDot(plan, [{"x": xl, "w": self.gating[k].weight, "b": None, "out": scores.slice(k, k + 1)} for k in range(n)])
            v_pre = plan.new(n * r)
            P.MatmulRight(plan, [{"x": xl, "m": P.PView(self.v_list[i], k), "out": v_pre.slice(k * r, (k + 1) * r)} for k in range(n)])
            v = P.Tanh(plan, v_pre).out
            c_pre = plan.new(n * r)
            P.GLinear(plan, [{"x": v.slice(k * r, (k + 1) * r), "w": P.PView(self.c_list[i], k), "b": None,
                              "out": c_pre.slice(k * r, (k + 1) * r)} for k in range(n)])
            c = P.Tanh(plan, c_pre).out
            uv = plan.new(n * E)
            P.GLinear(plan, [{"x": c.slice(k * r, (k + 1) * r), "w": P.PView(self.u_list[i], k), "b": None,
                              "out": uv.slice(k * E, (k + 1) * E)} for k in range(n)])
            dot = P.CrossCombine(plan, x0, uv, b1=self.bias[i], n_rep=n).out
            moe = P.GatePool(plan, dot, n, E, [(scores, list(range(n)))]).outs[0]
            dst = out if (i == self.num_layers - 1) else None
            xl = P.AddOut(plan, moe, xl, out=dst).out
        return xl

    def forward(self, x):
        return _run_float_module(self, "crossmix", x, lambda plan, xb: [self.describe(plan, xb)])


# --------------------------------------------------------------------------------------------------
# BaseModel
# --------------------------------------------------------------------------------------------------
class BaseModel(HipModule):
    """Embedding + wide term + regularisation registry + tower helpers shared by every model.
    Reference: model/layer.py:10-112."""

    def __init__(self, feature_dims, embed_dim, l2_reg_embedding=1e-5, l2_reg_linear=1e-5):
        super().__init__()
        self.feature_dims = feature_dims
        self.embedding = FeaturesEmbedding(feature_dims, embed_dim)
        self.embed_output_dim = self.embedding.output_dim0 * embed_dim
        self.embed_dim = embed_dim
        self.field_num = self.embedding.field_num
        self.linear = FeaturesLinear(self.embed_output_dim)
        self.is_concat_linear_cn = None
        self.reg_loss = torch.zeros((1,))
        self.regularization_weight = []
        self.add_regularization_weight(self.embedding.embedding_dict.parameters(), l2=l2_reg_embedding)
        self.add_regularization_weight(_reg_filter(self.linear), l2=l2_reg_linear)

    # ---- regularisation (model/layer.py:86-112) ----------------------------------------------------
    def add_regularization_weight(self, weight_list, l1=0.0, l2=0.0):
        if isinstance(weight_list, torch.nn.parameter.Parameter):
            weight_list = [weight_list]
        else:
            weight_list = list(weight_list)
        self.regularization_weight.append((weight_list, l1, l2))

    def regularized_parameters(self):
        """[(parameter, l1, l2)] over the registry, (name, parameter) tuples unwrapped."""
        out = []
        for weight_list, l1, l2 in self.regularization_weight:
            for w in weight_list:
                out.append((w[1] if isinstance(w, tuple) else w, l1, l2))
        return out

    def get_regularization_loss(self, device):
        """sum over the registry of l1*|w| + l2*w^2, shape (1,), differentiable (run.py:489 adds it to the loss).
        Drop-in path only: the fused trainer folds the same term into its optimiser kernels."""
        total = torch.zeros((1,), device=device)
        for p, l1, l2 in self.regularized_parameters():
            if l1 > 0:
                total = total + torch.sum(l1 * torch.abs(p))
            if l2 > 0:
                total = total + torch.sum(l2 * torch.square(p))
        return total

    # ---- towers (model/layer.py:35-56) ---------------------------------------------------------------
    def build_tower_output(self, n_tower, tower_input_dim, tower_dims, dropout):
        towers = nn.ModuleList(
            MultiLayerPerceptron(tower_input_dim, tower_dims, dropout, output_layer=True) for _ in range(self.n_tower))
        output_layers = nn.ModuleList([nn.Sigmoid() for _ in range(n_tower)])
        return towers, None, output_layers

    def describe_towers(self, plan, tower_inputs, other_outs, out, wide_in=None):
        """tower MLP -> `+= other` for every other logit -> sigmoid -> column i of `out` [B, n_tower].
        wide_in: the embeddings buffer when the caller has NOT described the wide term itself: it is then formed inside the
        fused head launch (csrc/head.hip) together with the towers' output layers; other_outs are the further logits."""
        import os
        n = len(self.towers)
        # the fused head's backward keeps one weight-gradient column per tower input, bias and wide input in LDS: 8 waves x width
        # floats in 64 KB (csrc/head.hip: head_width) — wider models (e.g. 39 fields x emb_dim 64) take the row-dot launches
        head_cols = sum(t.out_linear.weight.numel() + 1 for t in self.towers if t.out_linear is not None)
        head_cols += 0 if wide_in is None else wide_in.cols + 1
        fused = (wide_in is not None and n <= P.L.HEAD_MAX_TOWERS and len(other_outs) <= 2 and head_cols <= 2048 and
                 all(t.out_linear is not None for t in self.towers))
        if wide_in is not None and not fused:
            other_outs = [self.linear.describe(plan, wide_in)] + list(other_outs)
        if fused:
            wide = {"x": wide_in, "w": self.linear.fc.weight, "b": self.linear.fc.bias}
            mlp_stack(plan, list(self.towers), tower_inputs, final_addends=other_outs, final_sigmoid=True, head_out=out, head_wide=wide)
            # linear, BatchNorm, linear, BatchNorm and the head of all towers as ONE launch per direction where the shapes are the
            # reference's (csrc/tower.hip; training plans on one GPU)
            P.fuse_tower(plan)
            return out
        finals = [out.slice(i, i + 1) for i in range(n)]
        mlp_stack(plan, list(self.towers), tower_inputs, final_addends=other_outs, final_sigmoid=True, final_outs=finals)
        return out

    def build_atten(self, config, dropout):
        """Parameters of the attention branch, named and initialised like the reference's (model/layer.py:58-69):
        a token embedding D -> A, att_layer_num x nn.MultiheadAttention(A, heads) (used as parameter containers: the forward
        runs on the HIP plan), the optional residual projection, and the final Linear(F*A -> 1, no bias)."""
        atten_embed_dim = getattr(config, 'atten_embed_dim', self.embed_dim)
        self.atten_embedding = nn.Linear(self.embed_dim, atten_embed_dim)
        self.atten_output_dim = self.embedding.output_dim0 * atten_embed_dim
        self.att_res = config.att_res
        self.self_attns = nn.ModuleList([nn.MultiheadAttention(config.atten_embed_dim, config.att_head_num, dropout=dropout)
                                         for _ in range(config.att_layer_num)])
        if self.att_res:
            self.V_res_embedding = nn.Linear(self.embed_dim, atten_embed_dim)
        self.atten_linear = nn.Linear(self.atten_output_dim, 1, bias=False)
        self.att_head_num = config.att_head_num

    def describe_atten(self, plan, E):
        """atten_forward (model/layer.py:71-84) on the plan: [B, F*D] -> logit [B, 1]."""
        flat = self.describe_atten_features(plan, E, self.att_head_num)
        out = plan.new(1)
        P.RowDot(plan, [{"x": flat, "w": self.atten_linear.weight, "b": None, "out": out}])
        return out

    def describe_atten_features(self, plan, E, n_head, out=None):
        """the interacting-layer stack up to relu(...).view(-1, F*A) (model/layer.py:72-83, model/autoint.py:49-60)"""
        F_, D = self.field_num, self.embed_dim
        A = self.atten_embedding.weight.shape[0]
        tokens = P.Reshape(plan, E, plan.B * F_, D).out                       # embed_x.reshape(-1, field_num, embed_dim)
        groups = [{"x": tokens, "w": self.atten_embedding.weight, "b": self.atten_embedding.bias}]
        if self.att_res:
            groups.append({"x": tokens, "w": self.V_res_embedding.weight, "b": self.V_res_embedding.bias})
        first = P.GLinear(plan, groups, M=plan.B * F_)
        cur = first.outs[0]
        for attn in self.self_attns:                                          # q = k = v = the running token tensor
            qkv = P.GLinear(plan, [{"x": cur, "w": attn.in_proj_weight, "b": attn.in_proj_bias}], M=plan.B * F_).outs[0]
            ctx = P.AttnCore(plan, qkv, F_, n_head).out
            cur = P.GLinear(plan, [{"x": ctx, "w": attn.out_proj.weight, "b": attn.out_proj.bias}], M=plan.B * F_).outs[0]
        if self.att_res:
            act = P.AddRelu(plan, cur, first.outs[1]).out
        else:
            zero = plan.new(A, rows=plan.B * F_)
            zero.root.zero_()
            act = P.AddRelu(plan, cur, zero).out
        return P.Reshape(plan, act, plan.B, F_ * A, out=out).out             # .contiguous().view(-1, atten_output_dim)

    # ---- plan plumbing for whole models ----------------------------------------------------------------
    def plan_holder(self, B, tag="fwd", **kw):
        """The (cached) plan of this model at batch size B in its current train/eval mode."""
        dev = self.embedding.embedding_dict.weight.device

        def build():
            plan = self._new_plan(B, dev, seed=int(getattr(self, "seed", 0)))
            emb = self.embedding.describe(plan)
            outs, ins, extra = self.describe(plan, emb, **kw)
            plan.finalize(outs)
            return PlanHolder(plan, [emb.ids] + ins, outs, emb_op=emb, extra_outputs=extra)

        return self._cache().get(self, (tag, tuple(sorted(kw.items()))), B, build)


def _reg_filter(module):
    """named parameters whose name has 'weight' and not 'bn' — the reference's filter (e.g. model/ple.py:42-45).
    BatchNorm gammas inside MultiLayerPerceptron are named layers.N.weight and therefore ARE regularised."""
    return [(n, p) for n, p in module.named_parameters() if "weight" in n and "bn" not in n]
