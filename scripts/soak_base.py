import os, sys, math
sys.path.insert(0, os.getcwd())
import torch
import vit_vs_raw_iq_amd as P
from vit_vs_raw_iq_amd.trainer import FusedTrainer
torch.manual_seed(0)
dev = torch.device("cuda:0")
m = P.AMCTransformerViT(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=768, n_head=12,
                        n_layers=4, ffn_hidden=3072, drop_prob=0.1, device="cuda").to(dev).train()
tr = FusedTrainer(m, lr=1e-4, weight_decay=1e-3, betas=(0.9, 0.99), label_smoothing=0.1, max_norm=1.0, use_graph=True, dropout_seed=7)
g = torch.Generator(device=dev).manual_seed(1)
proj = torch.randn(19, 224 * 224, device=dev, generator=g)
xs = torch.randn(2048, 1, 224, 224, device=dev, generator=g)
ys = (xs.view(2048, -1) @ proj.t()).argmax(1)
losses = []
for step in range(120):
    i = (step % 8) * 256
    tr.step(xs[i:i + 256], ys[i:i + 256])
    if (step + 1) % 40 == 0:
        loss, acc, n = tr.read_stats()
        losses.append(loss)
        print(f"step {step+1}: mean loss {loss:.4f} acc {acc:.4f} over {n} frames", flush=True)
        assert math.isfinite(loss)
assert losses[-1] < losses[0], losses
print("ViT-Base (4 layers, 256 frames: gemm_big / wgrad_big) memorises 2048 frames: ok")
