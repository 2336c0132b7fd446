import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from helpers import make_ids
from cdcmdr_amd import plan as P
from cdcmdr_amd.model.ple import PLE
FD = [1000] * 26
cuda = torch.device("cuda:0")
B, n_tower, dropout = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 3, float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
rng = np.random.default_rng(B)
x = torch.from_numpy(make_ids(rng, B, FD)).to(cuda)
gout = torch.randn((B, n_tower), generator=torch.Generator().manual_seed(7)).to(cuda)
res = {}
for fused in (False, True):
    torch.manual_seed(0)
    m = PLE(FD, 16, n_tower, 2, 2, ((256, 128), (64,)), (64, 32), dropout=dropout).to(cuda).set_precision("bf16")
    m.seed = 1234
    P.TowerChain.enabled = fused
    m.train()
    out = m(x)
    holder = m.plan_holder(B)
    plan = holder.plan
    chain = [op for op in plan.ops if isinstance(op, P.TowerChain)]
    if chain:
        c = chain[0]; l1, b1, l2, b2, head = c.l1, c.b1, c.l2, c.b2, c.head
    else:
        i = max(k for k, op in enumerate(plan.ops) if isinstance(op, P.TowerHead))
        l1, b1, l2, b2, head = plan.ops[i - 4:i + 1]
    out.backward(gout)
    torch.cuda.synchronize()
    d = {}
    d["dX"] = l1.groups[0]["x"].grad.root.detach().clone().float()
    d["dz1h"] = plan._shadow_root(l1.groups[0]["y"].grad.root).detach().clone().float()[:B, :192]
    d["dz2h"] = plan._shadow_root(l2.groups[0]["y"].grad.root).detach().clone().float()[:B, :96]
    d["dE"] = holder.emb_op.out.grad.tensor().detach().clone()
    for k, p in m.named_parameters():
        if p.grad is not None:
            d["g:" + k] = p.grad.detach().clone()
    res[fused] = d
for k in res[False]:
    a, b = res[False][k].double(), res[True][k].double()
    rel = float((a - b).norm() / max(float(a.norm()), 1e-30))
    nz = float(((a - b).abs() > 0).double().mean())
    print(f"{k:60s} rel {rel:.3e}  max|d| {float((a-b).abs().max()):.3e}  frac differing {nz:.4f}  norm {float(a.norm()):.3e}")
