import os, sys, math
sys.path.insert(0, os.getcwd())
import torch
import vit_vs_raw_iq_amd as P
from vit_vs_raw_iq_amd.trainer import FusedTrainer
torch.manual_seed(0)
dev = torch.device("cuda:0")
m = P.AMCTransformerViT(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=192, n_head=3,
                        n_layers=12, ffn_hidden=768, drop_prob=0.1, device="cuda").to(dev).train()
tr = FusedTrainer(m, lr=3e-4, weight_decay=1e-3, betas=(0.9, 0.99), label_smoothing=0.1, max_norm=1.0, use_graph=True, dropout_seed=7)
g = torch.Generator(device=dev).manual_seed(1)
# a learnable synthetic task: class = argmax of 19 fixed random projections of the frame
proj = torch.randn(19, 224 * 224, device=dev, generator=g)
def batch():
    x = torch.randn(256, 1, 224, 224, device=dev, generator=g)
    y = (x.view(256, -1) @ proj.t()).argmax(1)
    return x, y
for step in range(400):
    x, y = batch()
    tr.step(x, y)
    if (step + 1) % 100 == 0:
        loss, acc, n = tr.read_stats()
        print(f"step {step+1}: mean loss {loss:.4f} acc {acc:.4f} over {n} frames", flush=True)
        assert math.isfinite(loss)
p = torch.cat([q.detach().float().flatten() for q in m.parameters()])
assert torch.isfinite(p).all()
print("soak ok; |params| =", p.norm().item())
