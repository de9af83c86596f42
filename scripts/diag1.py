import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch, numpy as np
import iq_oracle as O
from conftest import load_golden
import vit_vs_raw_iq_amd as P
d = torch.device("cuda:0")
def build(kind, kw, p=0.0):
    cls = P.AMCTransformerViT if kind == "vit" else P.AMCTransformerRawIQ
    return cls(drop_prob=p, device="cuda", **kw)
for name in ["rawiq_C_L2", "vit_tiny224_L2"]:
    kind, kw, z = load_golden(name)
    cfg = O.OracleConfig(kind=kind, drop_prob=0.0, **kw); sd = O.init_state(cfg, int(z["seed"]))
    m = build(kind, kw); m.load_state_dict(sd); m.to(d).train()
    x = torch.from_numpy(z["x"]).to(d); y = torch.from_numpy(z["y"]).to(d)
    loss = torch.nn.functional.cross_entropy(m(x), y, label_smoothing=0.1); loss.backward()
    _, _, gref = O.loss_and_grads(cfg, sd, torch.from_numpy(z["x"]), torch.from_numpy(z["y"]), 0.1)
    print(name)
    for k, p in m.named_parameters():
        r = gref[k].double(); g = p.grad.cpu().double()
        print(f"  {k:55s} rel {(g-r).norm().item()/(r.norm().item()+1e-12):.4f}  |ref| {r.norm().item():.4g}")
# directional derivative, central difference
for pdrop in (0.0, 0.2):
    kind, kw, z = load_golden("rawiq_C_L2")
    cfg = O.OracleConfig(kind=kind, drop_prob=0.0, **kw); sd = O.init_state(cfg, int(z["seed"]))
    m = build(kind, kw, pdrop); m.load_state_dict(sd); m.to(d).train()
    x = torch.from_numpy(z["x"]).to(d); y = torch.from_numpy(z["y"]).to(d)
    plan = m.native_plan()
    loss0 = torch.nn.functional.cross_entropy(m(x), y); loss0.backward()
    st = plan.step
    grads = [p.grad.clone() for p in m.parameters()]
    gn2 = sum(float(g.double().pow(2).sum()) for g in grads)
    for scale in (2e-2, 5e-3, 2e-3):
        eps = scale / math.sqrt(gn2)
        with torch.no_grad():
            for p, g in zip(m.parameters(), grads): p.add_(g, alpha=eps)
            plan.step = st - 1; lp = torch.nn.functional.cross_entropy(m(x), y).item()
            for p, g in zip(m.parameters(), grads): p.add_(g, alpha=-2*eps)
            plan.step = st - 1; lm = torch.nn.functional.cross_entropy(m(x), y).item()
            for p, g in zip(m.parameters(), grads): p.add_(g, alpha=eps)
        print(f"p={pdrop} scale {scale}: central diff {(lp-lm):.5f}  predicted {2*eps*gn2:.5f}  (loss0 {loss0.item():.4f} lp {lp:.4f} lm {lm:.4f})")
