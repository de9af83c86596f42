"""Model-level A/B of the FFN chain: run with IQ_TUNE_FFN_CHAIN=0 and =1, save logits + flat gradient; `cmp` compares."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load("/tmp/chain_ab_0.pt"), torch.load("/tmp/chain_ab_1.pt")
    print("logits max abs diff", (a["logits"] - b["logits"]).abs().max().item())
    print("grad rel diff", ((a["g"] - b["g"]).norm() / a["g"].norm()).item())
    for k in a["per"]:
        r = ((a["per"][k] - b["per"][k]).norm() / (a["per"][k].norm() + 1e-30)).item()
        if r > 2e-3: print(f"   {k}: rel {r:.4g}")
    sys.exit(0)
import vit_vs_raw_iq_amd as P
d = torch.device("cuda:0")
L_ = int(sys.argv[2]) if len(sys.argv) > 2 else 12
kw = dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=192, n_head=3, n_layers=L_, ffn_hidden=768)
torch.manual_seed(11)
m = P.AMCTransformerViT(drop_prob=0.0, device="cuda", **kw).to(d).train()
g = torch.Generator().manual_seed(12)
B = 256
x = torch.randn(B, 1, 224, 224, generator=g).to(d); y = torch.randint(0, 19, (B,), generator=g).to(d)
out = m(x)
torch.nn.functional.cross_entropy(out, y, label_smoothing=0.1).backward()
per = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters()}
torch.save({"logits": out.detach().cpu(), "g": torch.cat([v.reshape(-1) for v in per.values()]), "per": per}, f"/tmp/chain_ab_{sys.argv[1]}.pt")
print("saved", sys.argv[1])
