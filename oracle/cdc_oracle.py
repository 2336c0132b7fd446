"""CPU oracle of the multi-domain CTR training hot path — TEST INFRASTRUCTURE ONLY.

A from-scratch functional restatement (numpy for the integer index math, plain torch fp32 CPU ops for
the floating-point math, torch.autograd for derivatives) of what the reference computes on this path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path
(the package's HIP kernels) never does.

Parity pin: tests/test_oracle_golden.py checks every function here against the golden vectors in
tests/golden/*.npz, which tools/make_golden.py captured by importing the reference's own model/
package in the build container (the reference has no tests or fixtures of its own: SURVEY.md §4).

Every function takes the model's parameters as a flat dict keyed by the reference's state_dict names,
so reference checkpoints and the HIP modules' state_dict() are interchangeable inputs.
Reference citations are file:line under the reference repository.
"""
import math
import re

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
MATMUL_BF16 = False  # True / "exact": restate the bf16 MFMA path (operands of every contraction rounded to bf16, fp32 accumulate)
DROPOUT_P = 0.0     # parity runs use 0; bench.py's cpu_baseline sets the reference default (0.2) to time the same work


# --------------------------------------------------------------------------------------------------
# embedding  (model/layer.py:129-157)
# --------------------------------------------------------------------------------------------------
def field_offsets(field_dims):
    """model/layer.py:141-144: offsets[f] = sum of the cardinalities of the fields before f."""
    fd = np.asarray(field_dims, dtype=np.int64)
    return np.concatenate([[0], np.cumsum(fd)[:-1]]).astype(np.int64)


def gather_index(x_i32, field_dims):
    """model/layer.py:152: x + x.new_tensor(offsets): the sum is formed in x's dtype, int32, and wraps."""
    x = np.asarray(x_i32, dtype=np.int32)
    off = field_offsets(field_dims).astype(np.int32)           # new_tensor -> int32 (wraps like torch)
    with np.errstate(over="ignore"):
        return (x.astype(np.int32) + off[None, :]).astype(np.int32)


def embed(table, x_i32, field_dims):
    """model/layer.py:153-155: row lookup then flatten to [B, F*D]."""
    idx = gather_index(x_i32, field_dims)
    if idx.min() < 0 or idx.max() >= table.shape[0]:
        raise IndexError("index out of range in self")
    rows = table[torch.from_numpy(idx.astype(np.int64))]       # [B, F, D]
    return rows.flatten(1)


# --------------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------------
def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _Bf16Matmul(torch.autograd.Function):
    """y = x @ w^T exactly as the bf16 MFMA kernels contract it: operands rounded to bf16 (round-to-nearest-even) in the
    forward AND in both backward contractions (dX = bf(dY) @ bf(W), dW = bf(dY)^T @ bf(X)).  Accumulation: fp32 in the library's
    order (MATMUL_BF16 = True), or exact (fp64, MATMUL_BF16 = "exact") — the products of bf16 operands are exact in fp32, so
    the exact sum is the value every fp32 summation order (MKL's, the MFMA tiles') approximates, and the better centre to
    compare against: tools/grad_err_report.py measures the HIP path 2-3x closer to it than to the fp32-accumulated form."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        if MATMUL_BF16 == "exact":
            return (_bf(x).double() @ _bf(w).double().t()).float()
        return _bf(x) @ _bf(w).t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if MATMUL_BF16 == "exact":
            return (_bf(dy).double() @ _bf(w).double()).float(), (_bf(dy).double().t() @ _bf(x).double()).float()
        return _bf(dy) @ _bf(w), _bf(dy).t() @ _bf(x)


class _Bf16Bias(torch.autograd.Function):
    """+ b of a bf16-contracted layer: the bias gradient is the column sum of the SAME bf16-rounded dY the weight gradient
    contracts (csrc/gemm2.hip forms it with one more MFMA against a ones fragment), not of the fp32 dY."""

    @staticmethod
    def forward(ctx, y, b):
        return y + b

    @staticmethod
    def backward(ctx, dy):
        return dy, (_bf(dy).double().sum(0).float() if MATMUL_BF16 == "exact" else _bf(dy).sum(0))


def _mm(x, w, b, mfma=True):
    """x @ w^T + b; `mfma=False` marks the single-output layers the HIP path runs as fp32 row dot products."""
    if MATMUL_BF16 and mfma:
        y = _Bf16Matmul.apply(x, w)
        return y if b is None else _Bf16Bias.apply(y, b)
    y = x @ w.t()
    return y if b is None else y + b


def _lin(x, sd, prefix, mfma=True):
    """nn.Linear"""
    return _mm(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), mfma)


def _bn(x, sd, prefix, training, stats_out, gamma=None, beta=None):
    """nn.BatchNorm1d / F.batch_norm: batch mean and biased variance in training, running stats in eval;
    running stats updated with momentum 0.1 and the unbiased variance (functional: new stats go to stats_out)."""
    g = sd[prefix + ".weight"] if gamma is None else gamma
    b = sd[prefix + ".bias"] if beta is None else beta
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training:
        n = x.shape[0]
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        if stats_out is not None:
            unb = var * (n / (n - 1)) if n > 1 else var
            stats_out[prefix + ".running_mean"] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean).detach()
            stats_out[prefix + ".running_var"] = ((1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unb).detach()
            stats_out[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = rm, rv
    return (x - mean) / torch.sqrt(var + BN_EPS) * g + b


def _mlp_layer_ids(sd, prefix):
    """indices N of `prefix.layers.N.weight` entries, with whether each is a Linear (2-D) or a BatchNorm (1-D)."""
    pat = re.compile(re.escape(prefix) + r"\.layers\.(\d+)\.weight$")
    ids = sorted(int(m.group(1)) for k in sd for m in [pat.match(k)] if m)
    return [(i, sd[f"{prefix}.layers.{i}.weight"].dim()) for i in ids]


def mlp(x, sd, prefix, training, stats_out=None, output_layer=False):
    """MultiLayerPerceptron.forward (model/layer.py:197-206) with dropout p = 0: Linear -> (BN unless the batch
    has one row) -> ReLU per hidden layer; with output_layer the trailing Linear -> 1 has no activation."""
    layers = _mlp_layer_ids(sd, prefix)
    lin_pos = [k for k, (_, dim) in enumerate(layers) if dim == 2]
    for n, k in enumerate(lin_pos):
        idx = layers[k][0]
        is_out = output_layer and n == len(lin_pos) - 1
        y = _lin(x, sd, f"{prefix}.layers.{idx}", mfma=not is_out)
        if is_out:
            return y
        if k + 1 < len(layers) and layers[k + 1][1] == 1 and x.shape[0] != 1:
            y = _bn(y, sd, f"{prefix}.layers.{layers[k + 1][0]}", training, stats_out)
        x = torch.relu(y)
        if DROPOUT_P > 0 and training:
            x = F.dropout(x, DROPOUT_P, True)
    return x


def wide_logit(e, sd):
    """FeaturesLinear (model/layer.py:122-126)."""
    return _lin(e, sd, "linear.fc", mfma=False)


ATT_HEADS = 2       # config.att_head_num (config.py:27) — not recoverable from a state_dict


def atten_features(e, sd):
    """model/layer.py:72-83 (= model/autoint.py:49-60) with dropout 0: token embedding D -> A, the stack of
    nn.MultiheadAttention layers (q = k = v; in_proj, q scaled by dh^-1/2, softmax over the keys, out_proj), the optional
    residual projection of the raw tokens, ReLU, flatten to [B, F*A].  Written out with plain matmuls."""
    D = sd["embedding.embedding_dict.weight"].shape[1]
    B = e.shape[0]
    tok = e.reshape(B, -1, D)                                                    # [B, F, D]
    cur = tok @ sd["atten_embedding.weight"].t() + sd["atten_embedding.bias"]    # [B, F, A]
    A = cur.shape[-1]
    dh = A // ATT_HEADS
    n_layer = _count(sd, r"self_attns\.(\d+)\.")
    for i in range(n_layer):
        qkv = cur @ sd[f"self_attns.{i}.in_proj_weight"].t() + sd[f"self_attns.{i}.in_proj_bias"]
        q, k, v = [t.reshape(B, -1, ATT_HEADS, dh).transpose(1, 2) for t in qkv.split(A, dim=-1)]       # [B, H, F, dh]
        p = torch.softmax((q * dh ** -0.5) @ k.transpose(-1, -2), dim=-1)
        ctx = (p @ v).transpose(1, 2).reshape(B, -1, A)
        cur = ctx @ sd[f"self_attns.{i}.out_proj.weight"].t() + sd[f"self_attns.{i}.out_proj.bias"]
    if "V_res_embedding.weight" in sd:
        cur = cur + (tok @ sd["V_res_embedding.weight"].t() + sd["V_res_embedding.bias"])
    return torch.relu(cur).reshape(B, -1)


def atten_logit(e, sd):
    """BaseModel.atten_forward (model/layer.py:71-84): the features through Linear(F*A -> 1, no bias)."""
    return atten_features(e, sd) @ sd["atten_linear.weight"].t()


def autoint_forward(sd, x_i32, field_dims, training=True, stats_out=None):
    """AutoInt.forward (model/autoint.py:48-65)."""
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    final = torch.cat([atten_features(e, sd), mlp(e, sd, "dnn", training, stats_out)], dim=1)
    return torch.sigmoid(final @ sd["dnn_linear.weight"].t() + wide_logit(e, sd)).squeeze(1)


def other_logits(e, sd):
    """the `other_outs` every tower adds to its logit: the wide term, and the attention branch when the model has one"""
    outs = [wide_logit(e, sd)]
    if "atten_embedding.weight" in sd:
        outs.append(atten_logit(e, sd))
    return outs


def towers(tower_inputs, other_outs, sd, training, stats_out, prefix="towers"):
    """BaseModel.tower_forward (model/layer.py:48-56)."""
    ys = []
    for i, t_in in enumerate(tower_inputs):
        logit = mlp(t_in, sd, f"{prefix}.{i}", training, stats_out, output_layer=True)
        for o in other_outs:
            logit = logit + o
        ys.append(torch.sigmoid(logit))
    return torch.cat(ys, dim=1)


def _count(sd, pattern):
    pat = re.compile(pattern)
    return len({m.group(1) for k in sd for m in [pat.match(k)] if m})


# --------------------------------------------------------------------------------------------------
# PLE  (model/ple.py:50-70, 96-125)
# --------------------------------------------------------------------------------------------------
def cgc(x_list, sd, prefix, n_task, training):
    n_spec_total = _count(sd, re.escape(prefix) + r"\.experts_specific\.(\d+)\.")
    n_shared = _count(sd, re.escape(prefix) + r"\.experts_shared\.(\d+)\.")
    ns = n_spec_total // n_task
    spec = [mlp(x_list[i // ns], sd, f"{prefix}.experts_specific.{i}", training) for i in range(n_spec_total)]
    shared = [mlp(x_list[-1], sd, f"{prefix}.experts_shared.{k}", training) for k in range(n_shared)]
    outs = []
    for i in range(n_task):
        gate = torch.softmax(_lin(x_list[i], sd, f"{prefix}.gates_specific.{i}.0"), dim=1)
        cat = torch.stack(spec[i * ns:(i + 1) * ns] + shared, dim=1)        # [B, ns+nsh, H]
        outs.append((gate.unsqueeze(-1) * cat).sum(1))
    if f"{prefix}.gate_shared.0.weight" in sd:
        gate = torch.softmax(_lin(x_list[-1], sd, f"{prefix}.gate_shared.0"), dim=1)
        cat = torch.stack(spec + shared, dim=1)
        outs.append((gate.unsqueeze(-1) * cat).sum(1))
    return outs


def ple_forward(sd, x_i32, field_dims, n_tower, training=True, stats_out=None, prefix=""):
    sd = _strip(sd, prefix)
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    n_level = _count(sd, r"cgc_layers\.(\d+)\.")
    inputs = [e] * (n_tower + 1)
    for lvl in range(n_level):
        inputs = cgc(inputs, sd, f"cgc_layers.{lvl}", n_tower, training)
    return towers(inputs[:n_tower], other_logits(e, sd), sd, training, _prefixed(stats_out, prefix))


# --------------------------------------------------------------------------------------------------
# MMoE  (model/mmoe.py:53-74)
# --------------------------------------------------------------------------------------------------
def mmoe_forward(sd, x_i32, field_dims, n_tower, training=True, stats_out=None, prefix=""):
    sd = _strip(sd, prefix)
    so = _prefixed(stats_out, prefix)
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    n_expert = _count(sd, r"experts\.(\d+)\.")
    experts = torch.stack([mlp(e, sd, f"experts.{k}", training, so) for k in range(n_expert)], dim=1)   # [B, n_e, H]
    tin = []
    for i in range(n_tower):
        gate = torch.softmax(_lin(e, sd, f"gates.{i}.0"), dim=1)
        tin.append((gate.unsqueeze(-1) * experts).sum(1))
    return towers(tin, other_logits(e, sd), sd, training, so)


# --------------------------------------------------------------------------------------------------
# cross networks + DCN / DCNv2  (model/layer.py:303-407, model/dcn.py:36-43, model/dcnv2.py:59-70)
# --------------------------------------------------------------------------------------------------
def cross_network(x, sd, prefix):
    n = _count(sd, re.escape(prefix) + r"\.w\.(\d+)\.")
    x0 = x
    for i in range(n):
        xw = x @ sd[f"{prefix}.w.{i}.weight"].t()                  # [B,1]
        x = x0 * xw + sd[f"{prefix}.b.{i}"] + x
    return x


def cross_network_v2(x, sd, prefix):
    n = _count(sd, re.escape(prefix) + r"\.w\.(\d+)\.")
    x0 = x
    for i in range(n):
        x = x0 * (x @ sd[f"{prefix}.w.{i}.weight"].t()) + sd[f"{prefix}.b.{i}"] + x
    return x


def cross_network_mix(x, sd, prefix):
    """model/layer.py:372-407, incl. the trailing squeeze() that drops the batch dim when B == 1."""
    n_layers = _count(sd, re.escape(prefix) + r"\.u_list\.(\d+)$")
    n_exp = sd[f"{prefix}.u_list.0"].shape[0]
    x0 = x
    xl = x
    for i in range(n_layers):
        U, V, Cm = sd[f"{prefix}.u_list.{i}"], sd[f"{prefix}.v_list.{i}"], sd[f"{prefix}.c_list.{i}"]
        bias = sd[f"{prefix}.bias.{i}"].squeeze(1)
        outs, scores = [], []
        for k in range(n_exp):
            scores.append(xl @ sd[f"{prefix}.gating.{k}.weight"].t())          # [B,1]
            v = torch.tanh(xl @ V[k])                                           # [B,r]
            v = torch.tanh(v @ Cm[k].t())
            uv = v @ U[k].t()                                                   # [B,E]
            outs.append(x0 * (uv + bias))
        outs = torch.stack(outs, 2)                                             # [B,E,n_exp]
        gate = torch.softmax(torch.stack(scores, 1), dim=1)                     # [B,n_exp,1]
        xl = torch.matmul(outs, gate).squeeze(2) + xl
    return xl.squeeze()


def dcn_forward(sd, x_i32, field_dims, training=True, stats_out=None):
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    cn = cross_network(e, sd, "cn")
    deep = mlp(e, sd, "mlp", training, stats_out)
    stack = torch.cat([cn, deep], dim=1)
    return torch.sigmoid(wide_logit(e, sd) + stack @ sd["mlp_linear.weight"].t()).squeeze(1)


def fm_term(e3):
    """FactorizationMachine(reduce_sum=True) (model/layer.py:160-175) on [B, F, D]."""
    return 0.5 * ((e3.sum(dim=1) ** 2 - (e3 ** 2).sum(dim=1)).sum(dim=1, keepdim=True))


def deepfm_forward(sd, x_i32, field_dims, training=True, stats_out=None):
    """DeepFM.forward (model/dfm.py:29-35): sigmoid(linear(e) + fm(e) + mlp(e)), the MLP ending in Linear(-> 1)."""
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    D = sd["embedding.embedding_dict.weight"].shape[1]
    deep = mlp(e, sd, "mlp", training, stats_out, output_layer=True)
    return torch.sigmoid(wide_logit(e, sd) + fm_term(e.reshape(e.shape[0], -1, D)) + deep).squeeze(1)


def _sei(e, sd, prefix, training, stats_out):
    """SEI.forward (model/hinet.py:16-21): expert MLPs mixed by a softmax gate."""
    n = _count(sd, re.escape(prefix) + r"\.experts\.(\d+)\.")
    fea = torch.stack([mlp(e, sd, f"{prefix}.experts.{k}", training, stats_out) for k in range(n)], dim=1)
    gate = torch.softmax(_lin(e, sd, f"{prefix}.gate.0"), dim=1)
    return (gate.unsqueeze(-1) * fea).sum(1)


def hinet_forward(sd, x_i32, field_dims, x_group, domain_idx, training=True, stats_out=None):
    """HiNet.forward (model/hinet.py:61-92)."""
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    D = sd["embedding.embedding_dict.weight"].shape[1]
    n_tower = _count(sd, r"specific_seis\.(\d+)\.")
    spec = [_sei(e, sd, f"specific_seis.{i}", training, stats_out) for i in range(n_tower)]
    shared = _sei(e, sd, "shared_seis", training, stats_out)
    dom = e[:, domain_idx * D:(domain_idx + 1) * D]
    san_gate = torch.softmax(_lin(dom, sd, "san_gate.0"), dim=1)
    stacked = torch.stack(spec, dim=1)
    san = (san_gate.unsqueeze(-1) * stacked).sum(1)
    g = torch.as_tensor(x_group).reshape(-1)
    onehot = torch.stack([(g == i).float() for i in range(n_tower)], dim=1)
    con = (onehot.unsqueeze(-1) * stacked).sum(1)
    tower = mlp(torch.cat([shared, con, san], dim=1), sd, "tower", training, stats_out)
    logit = tower @ sd["tower_linear.weight"].t()
    for o in other_logits(e, sd):
        logit = logit + o
    return torch.sigmoid(logit).squeeze(1)


def adasparse_forward(sd, x_i32, field_dims, domain_idx, training=True, stats_out=None):
    """AdaSparse.forward + DNN_w_Pruner.forward (model/adasparse.py:45-65,94-116): per layer fc * pi with
    pi = 2*sigmoid(pruner(cat[h, dom.detach()])), pi <= 0.25 cut to 0 (no gradient through the cut), then BN, ReLU."""
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    D = sd["embedding.embedding_dict.weight"].shape[1]
    dom = e[:, domain_idx * D:(domain_idx + 1) * D].detach()
    h = e
    for i in range(_count(sd, r"dnn\.linears\.(\d+)\.")):
        fc = _lin(h, sd, f"dnn.linears.{i}")
        pi = 2.0 * torch.sigmoid(_lin(torch.cat([h, dom], dim=1), sd, f"dnn.pruners.{i}"))
        pi = torch.where(pi.abs() - 0.25 <= 0, torch.zeros_like(pi), pi)
        h = torch.relu(_bn(fc * pi, sd, f"dnn.bn.{i}", training, stats_out))
    logit = _lin(h, sd, "dnn_linear", mfma=False)
    for o in other_logits(e, sd):
        logit = logit + o
    return torch.sigmoid(logit).squeeze(1)


def _gate_nn(x, sd, prefix):
    """GateNN.forward (model/pepnet.py:117-134) with dropout 0: Linear, ReLU, Linear, Sigmoid, times 2."""
    ids = sorted(int(m.group(1)) for k in sd for m in [re.match(re.escape(prefix) + r"\.gate\.(\d+)\.weight$", k)] if m)
    h = torch.relu(_lin(x, sd, f"{prefix}.gate.{ids[0]}"))
    return 2.0 * torch.sigmoid(_lin(h, sd, f"{prefix}.gate.{ids[1]}"))


def pepnet_forward(sd, x_i32, field_dims, domain_idx, n_tower, training=True, stats_out=None):
    """PEPNet.forward + PPNetBlock.forward (model/pepnet.py:74-114,169-180).  PPNet's tower layers are ONE module applied to
    every tower in turn: shared weights, the tower's own batch statistics, running statistics updated once per tower in
    order (each update starts from the previous tower's result)."""
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    D = sd["embedding.embedding_dict.weight"].shape[1]
    dom = e[:, domain_idx * D:(domain_idx + 1) * D]
    ep = e * _gate_nn(torch.cat([e.detach(), dom], dim=-1), sd, "epnet")
    others = other_logits(e, sd)
    tops = []
    if any(k.startswith("ppnet.") for k in sd):
        n_layer = _count(sd, r"ppnet\.gate_layers\.(\d+)\.")
        cur = [e] * n_tower
        gate_in = torch.cat([e.detach(), ep], dim=-1)
        live = dict(sd)                                            # running statistics move from tower to tower
        for l in range(n_layer):
            gws = torch.chunk(_gate_nn(gate_in, sd, f"ppnet.gate_layers.{l}"), n_tower, dim=1)
            nxt = []
            for t in range(n_tower):
                y = _lin(cur[t] * gws[t], sd, f"ppnet.tower_layers.{l}.0.0")
                upd = {}
                y = _bn(y, live, f"ppnet.tower_layers.{l}.0.1", training, upd)
                live.update(upd)
                nxt.append(torch.relu(y))
            cur = nxt
            if stats_out is not None and training:
                for t in range(n_tower):                           # the state_dict lists the shared module once per tower
                    for k in ("running_mean", "running_var", "num_batches_tracked"):
                        stats_out[f"ppnet.tower_layers.{l}.{t}.1.{k}"] = live[f"ppnet.tower_layers.{l}.0.1.{k}"]
        tops = cur
        lin_names = [f"ppnet_linears.{t}" for t in range(n_tower)]
    elif n_tower > 1:
        tops = [mlp(ep, sd, f"towers.{t}", training, stats_out) for t in range(n_tower)]
        lin_names = [f"ppnet_linears.{t}" for t in range(n_tower)]
    else:
        tops = [mlp(ep, sd, "towers", training, stats_out)]
        lin_names = ["ppnet_linears"]
    ys = []
    for top, name in zip(tops, lin_names):
        logit = top @ sd[name + ".weight"].t()
        for o in others:
            logit = logit + o
        ys.append(torch.sigmoid(logit))
    pred = torch.cat(ys, dim=1)
    return pred.squeeze(1) if (n_tower == 1 and not any(k.startswith("ppnet.") for k in sd)) else pred


def adl_forward(sd, x_i32, field_dims, centers, n_tower, targets=None, is_training=True, training=True, stats_out=None,
                dlm_iters=3, rate=0.9):
    """ADL.forward + DLM_routing (model/adl.py:62-126).  Returns (pred, grouped targets, new centres) when is_training,
    (ys in batch order, None, new centres) otherwise."""
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    with torch.no_grad():
        ed = e.detach()
        for _ in range(dlm_iters):                                 # every iteration scores against the incoming centres
            coeff = torch.softmax(ed @ centers.t(), dim=1)
            tmp = F.normalize(coeff.t() @ ed, p=2, dim=1)
        new_centers = F.normalize(rate * centers + (1 - rate) * tmp, p=2, dim=1)
    tower_of = torch.argmax(coeff, dim=1)
    others = other_logits(e, sd)
    ys, ts = [], []
    out = torch.zeros((e.shape[0], 1))
    for t in range(n_tower):
        mask = tower_of == t
        h = mlp(e[mask], sd, f"domain_mlps.{t}", training, stats_out) if int(mask.sum()) > 0 else torch.zeros((0, sd[f"domain_mlps_linears.{t}.weight"].shape[1]))
        w = sd[f"domain_mlps_linears.{t}.weight"] * sd["shared_mlps_linear.weight"]
        b = sd[f"domain_mlps_linears.{t}.bias"] + sd["shared_mlps_linear.bias"]
        logit = h @ w.t() + b
        for o in others:
            logit = logit + o[mask]
        y = torch.sigmoid(logit)
        if is_training:
            ys.append(y)
            ts.append(None if targets is None else torch.as_tensor(targets)[mask])
        else:
            out[mask] = y.detach()
    if is_training:
        return torch.cat(ys, dim=0), (None if targets is None else torch.cat(ts, dim=0)), new_centers
    return out, None, new_centers


def dcnv2_forward(sd, x_i32, field_dims, training=True, stats_out=None, model_structure="parallel"):
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    if "crossnet.u_list.0" in sd:
        cross = cross_network_mix(e, sd, "crossnet")
        if cross.dim() == 1:
            cross = cross.unsqueeze(0) if e.shape[0] == 1 else cross
    else:
        cross = cross_network_v2(e, sd, "crossnet")
    if model_structure == "crossnet_only":
        final = cross
    elif model_structure == "stacked":
        final = mlp(cross, sd, "dnn", training, stats_out)
    else:
        final = torch.cat([cross, mlp(e, sd, "dnn", training, stats_out)], dim=1)
    return torch.sigmoid(final @ sd["dnn_linear.weight"].t() + wide_logit(e, sd)).squeeze(1)


# --------------------------------------------------------------------------------------------------
# STAR  (model/star.py:62-114, 133-181)
# --------------------------------------------------------------------------------------------------
def _star_tower(h, sd, g, training, stats_out):
    n_layers = _count(sd, r"shared_dnn\.linears\.(\d+)\.")
    if h.shape[0] != 1:                                             # MDR_BatchNorm returns its input for one row
        gamma = sd[f"domain_norm.{g}.weight"] * sd["shared_bn_weight"]
        beta = sd[f"domain_norm.{g}.bias"] + sd["shared_bn_bias"]
        h = _bn(h, sd, f"domain_norm.{g}", training, stats_out, gamma=gamma, beta=beta)
    for i in range(n_layers):
        w = sd[f"domain_dnns.{g}.linears.{i}.weight"] * sd[f"shared_dnn.linears.{i}.weight"]
        b = sd[f"domain_dnns.{g}.linears.{i}.bias"] + sd[f"shared_dnn.linears.{i}.bias"]
        h = _mm(h, w, b)                                            # (bf16 restatement: the FUSED weight is what gets rounded)
        if h.shape[0] > 1:
            h = _bn(h, sd, f"domain_dnns.{g}.bn.{i}", training, stats_out)
        h = torch.relu(h)
    w = sd[f"domain_dnn_linears.{g}.weight"] * sd["shared_dnn_linear.weight"]
    b = sd[f"domain_dnn_linears.{g}.bias"] + sd["shared_dnn_linear.bias"]
    return h @ w.t() + b


def star_forward(sd, x_i32, field_dims, n_tower, x_group=None, targets=None, training=True, stats_out=None, prefix=""):
    """x_group None: every tower over the full batch -> [B, n_tower]; else rows are partitioned by group id
    (ascending group, original order inside a group) -> ([B,1] group-ordered, targets group-ordered)."""
    sd = _strip(sd, prefix)
    so = _prefixed(stats_out, prefix)
    e = embed(sd["embedding.embedding_dict.weight"], x_i32, field_dims)
    wide = sum(other_logits(e, sd))                                 # star.py:66-72: wide term (+ attention branch)
    ys, ts = [], []
    for g in range(n_tower):
        if x_group is None:
            h, w_g, t_g = e, wide, targets
        else:
            mask = (torch.as_tensor(x_group) == g).reshape(-1)
            h, w_g = e[mask], wide[mask]
            t_g = None if targets is None else torch.as_tensor(targets)[mask]
        if h.shape[0] == 0 and training:
            # F.batch_norm on an empty training batch: no rows -> nothing to emit (see tests for the statistics)
            ys.append(torch.zeros((0, 1)))
            ts.append(t_g)
            continue
        ys.append(torch.sigmoid(_star_tower(h, sd, g, training, so) + w_g))
        ts.append(t_g)
    if x_group is None:
        return torch.cat(ys, dim=1)
    return torch.cat(ys, dim=0), (None if targets is None else torch.cat(ts, dim=0))


# --------------------------------------------------------------------------------------------------
# CDC.forward  (model/cdc.py:95-111)
# --------------------------------------------------------------------------------------------------
def cdc_forward(base_forward, x_i32, domain2group, domain_idx, mode="split", domain_i=None):
    y = base_forward(x_i32)
    if mode == "warmup":
        return y.mean(dim=1)
    d2g = torch.as_tensor(domain2group, dtype=torch.int64)
    if domain_i is None:
        groups = d2g[torch.as_tensor(np.asarray(x_i32)[:, domain_idx].astype(np.int64))]
        return y.gather(1, groups.unsqueeze(1))
    return y[:, int(d2g[domain_i])]


# --------------------------------------------------------------------------------------------------
# loss, regularisation, optimiser  (run.py:484-492, 720-723; model/layer.py:96-112)
# --------------------------------------------------------------------------------------------------
class _BceMean(torch.autograd.Function):
    """torch.nn.BCELoss(reduction='mean') on probabilities (aten binary_cross_entropy): forward with the log terms clamped at
    -100; backward (x - t) / max((1 - x) * x, 1e-12) / n — finite at saturated probabilities, where differentiating the
    clamped logs would give 0 * inf."""

    @staticmethod
    def forward(ctx, p, y):
        ctx.save_for_backward(p, y)
        return (-(y * torch.clamp(torch.log(p), min=-100.0) + (1 - y) * torch.clamp(torch.log1p(-p), min=-100.0))).mean()

    @staticmethod
    def backward(ctx, g):
        p, y = ctx.saved_tensors
        return g * (p - y) / torch.clamp((1 - p) * p, min=1e-12) / p.numel(), None


def bce_mean(p, y):
    return _BceMean.apply(p, y.float())


def reg_loss(sd, l2_by_name):
    """sum_w l2_w * sum(w^2) over the registered tensors -> shape (1,) (model/layer.py:96-112)."""
    total = torch.zeros((1,))
    for name, l2 in l2_by_name.items():
        if l2 > 0:
            total = total + torch.sum(l2 * torch.square(sd[name]))
    return total


def reg_names(names, model_kind):
    """Which state_dict entries the reference registers for L2 (BaseModel.__init__ + each model's filters).
    The filter is `'weight' in name and 'bn' not in name` on names RELATIVE to the filtered sub-module, so
    BatchNorm gammas of MultiLayerPerceptron (`layers.N.weight`) are included, DNN's `bn.N.weight` are not."""
    out = []
    for n in names:
        base = n.split("base_model_instance.")[-1]
        if base == "embedding.embedding_dict.weight" or base == "linear.fc.weight":
            out.append(n)
            continue
        if "running_" in base or "num_batches" in base:
            continue
        top = base.split(".")[0]
        rel = base[len(top) + 1:]
        if model_kind == "ple" and top in ("cgc_layers", "towers") and "weight" in rel and "bn" not in rel:
            out.append(n)
        elif model_kind == "mmoe" and top in ("experts", "towers") and "weight" in rel and "bn" not in rel:
            out.append(n)
        elif model_kind == "dcn" and top in ("mlp", "cn") and "weight" in rel and "bn" not in rel:
            out.append(n)
        elif model_kind == "deepfm" and top == "mlp" and "weight" in rel and "bn" not in rel:
            out.append(n)
        elif model_kind == "autoint" and top == "dnn" and "weight" in rel and "bn" not in rel:
            out.append(n)
        elif model_kind == "adasparse" and top == "dnn" and "weight" in rel and "bn" not in rel:
            out.append(n)
        elif model_kind == "pepnet" and top in ("epnet", "ppnet", "towers") and "weight" in rel and "bn" not in rel:
            if not (top == "ppnet" and re.match(r"tower_layers\.\d+\.[1-9]\d*\.", rel)):      # the shared module counts once
                out.append(n)
        elif model_kind == "adl" and top in ("domain_mlps", "shared_mlps") and "weight" in rel and "bn" not in rel:
            out.append(n)
        elif model_kind == "hinet" and top in ("specific_seis", "shared_seis", "san_gate", "tower") and "weight" in rel and "bn" not in rel:
            out.append(n)
        elif model_kind == "dcnv2":
            if top == "dnn" and "weight" in rel and "bn" not in rel:
                out.append(n)
            elif base == "dnn_linear.weight" or re.match(r"crossnet\.(u_list|v_list|c_list)\.\d+$", base):
                out.append(n)
        elif model_kind == "star" and top in ("domain_dnns", "shared_dnn") and "weight" in rel and "bn" not in rel:
            out.append(n)
    return out


def adam_scalars(step, lr=1e-3, beta1=0.9, beta2=0.99):
    """step_size and sqrt(bias_correction2) as torch/optim/adam.py forms them (Python doubles)."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    return lr / bc1, math.sqrt(bc2)


def adam_step(w, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=1e-8):
    """One torch.optim.Adam step (single-tensor CPU path), functional: returns (w, m, v)."""
    if weight_decay != 0:
        g = g.add(w, alpha=weight_decay)
    m = m.lerp(g, 1 - beta1)
    v = v.mul(beta2).addcmul(g, g, value=1 - beta2)
    step_size, bc2s = adam_scalars(step, lr, beta1, beta2)
    denom = (v.sqrt() / bc2s).add(eps)
    w = w.addcdiv(m, denom, value=-step_size)
    return w, m, v


# --------------------------------------------------------------------------------------------------
# metrics  (run.py:682-711: sklearn roc_auc_score / log_loss)
# --------------------------------------------------------------------------------------------------
def auc(targets, scores):
    """Tie-aware ROC AUC (Mann-Whitney U with mid-ranks) == sklearn.metrics.roc_auc_score."""
    t = np.asarray(targets).astype(np.int64).reshape(-1)
    s = np.asarray(scores, dtype=np.float64).reshape(-1)
    n_pos = int(t.sum())
    n_neg = t.size - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(s, kind="mergesort")
    ss = s[order]
    ranks = np.empty(t.size, dtype=np.float64)
    i = 0
    while i < t.size:
        j = i
        while j + 1 < t.size and ss[j + 1] == ss[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return (ranks[t == 1].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def logloss(targets, scores, eps=None):
    """sklearn.metrics.log_loss for binary labels, as run.py:686,701 calls it.  scikit-learn is a third-party dependency
    that is not part of /root/reference; the arithmetic restated here is that of the version the golden vectors were
    captured with (1.7.2, sklearn/metrics/_classification.py): the probability matrix [1-p, p] is formed IN THE DTYPE OF
    THE PREDICTIONS (float32 for the reference's `.cpu().numpy()` tensors), clipped to [eps, 1-eps] with that dtype's
    machine epsilon, and only the logarithm (scipy xlogy against int64 labels) and the mean are float64."""
    t = np.asarray(targets).astype(np.int64).reshape(-1)
    s = np.asarray(scores).reshape(-1)
    if s.dtype not in (np.float64, np.float32, np.float16):
        s = s.astype(np.float64)
    e = np.finfo(s.dtype).eps if eps is None else s.dtype.type(eps)
    one = s.dtype.type(1)
    p = np.clip(s, e, one - e).astype(np.float64)
    q = np.clip(one - s, e, one - e).astype(np.float64)
    return float(-np.where(t == 1, np.log(p), np.log(q)).mean())


# --------------------------------------------------------------------------------------------------
def _strip(sd, prefix):
    if not prefix:
        return sd
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


class _Prefixed(dict):
    def __init__(self, target, prefix):
        super().__init__()
        self.target, self.prefix = target, prefix

    def __setitem__(self, k, v):
        self.target[self.prefix + k] = v


def _prefixed(stats_out, prefix):
    if stats_out is None or not prefix:
        return stats_out
    return _Prefixed(stats_out, prefix)
