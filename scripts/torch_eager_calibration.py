"""Calibration: the reference's op sequence written with stock torch.nn ops, timed on this GPU.

Not the reference's code (which cannot travel to the GPU box) and not the oracle: a plain nn.Module restatement of the
same post-norm encoder (hand-rolled LayerNorm eps 1e-12, materialised S x S softmax attention, ReLU FFN, dropout),
label-smoothed CE, clip_grad_norm_(1.0), AdamW(betas 0.9/0.99, wd 1e-3) -- what a user of the reference gets from
PyTorch-ROCm's own kernels (rocBLAS / hipBLASLt / elementwise) on the same box, in fp32 (the reference's dtype) and
under bf16 autocast.  Config B of bench.py: ViT-Tiny/16, 224x224, 1 channel, 19 classes, 256 frames, dropout 0.1.

usage: python scripts/torch_eager_calibration.py [batch] [steps]
"""
import math, sys, time
import torch
import torch.nn as nn


class LN(nn.Module):
    def __init__(self, d, eps=1e-12):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(d)); self.beta = nn.Parameter(torch.zeros(d)); self.eps = eps

    def forward(self, x):
        mean = x.mean(-1, keepdim=True)
        var = x.var(-1, unbiased=False, keepdim=True)
        return self.gamma * ((x - mean) / torch.sqrt(var + self.eps)) + self.beta


class Layer(nn.Module):
    def __init__(self, d, h, f, p):
        super().__init__()
        self.h = h
        self.wq, self.wk, self.wv, self.wo = (nn.Linear(d, d) for _ in range(4))
        self.n1, self.n2 = LN(d), LN(d)
        self.l1, self.l2 = nn.Linear(d, f), nn.Linear(f, d)
        self.d1, self.d2, self.d3 = nn.Dropout(p), nn.Dropout(p), nn.Dropout(p)

    def forward(self, x):
        B, S, D = x.shape
        sp = lambda t: t.view(B, S, self.h, D // self.h).transpose(1, 2)
        q, k, v = sp(self.wq(x)), sp(self.wk(x)), sp(self.wv(x))
        score = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(D // self.h), dim=-1)
        a = (score @ v).transpose(1, 2).contiguous().view(B, S, D)
        x = self.n1(self.d1(self.wo(a)) + x)
        y = self.l2(self.d3(torch.relu(self.l1(x))))
        return self.n2(self.d2(y) + x)


class ViT(nn.Module):
    def __init__(self, img=224, patch=16, d=192, h=3, L=12, f=768, ncls=19, p=0.1):
        super().__init__()
        self.proj = nn.Conv2d(1, d, patch, patch)
        self.cls = nn.Parameter(torch.zeros(1, 1, d))
        S = (img // patch) ** 2 + 1
        pe = torch.zeros(S, d); pos = torch.arange(S).float().unsqueeze(1); i2 = torch.arange(0, d, 2).float()
        pe[:, 0::2] = torch.sin(pos / 10000 ** (i2 / d)); pe[:, 1::2] = torch.cos(pos / 10000 ** (i2 / d))
        self.register_buffer("pe", pe)
        self.drop = nn.Dropout(p)
        self.layers = nn.ModuleList(Layer(d, h, f, p) for _ in range(L))
        self.norm = LN(d); self.head = nn.Linear(d, ncls)

    def forward(self, x):
        x = self.proj(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls.expand(x.shape[0], -1, -1), x], 1)
        x = self.drop(x + self.pe)
        for l in self.layers:
            x = l(x)
        return self.head(self.norm(x[:, 0]))


def run(batch, steps, autocast):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = ViT().to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-3, betas=(0.9, 0.99))
    crit = nn.CrossEntropyLoss(label_smoothing=0.1)
    x = torch.randn(batch, 1, 224, 224, device=dev); y = torch.randint(0, 19, (batch,), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            loss = crit(m(x).float(), y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"torch eager {'bf16 autocast' if autocast else 'fp32'}: batch {batch}: {dt * 1e3:8.2f} ms/step  {batch / dt:9.1f} frames/s", flush=True)


if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    run(batch, steps, False)
    run(batch, steps, True)
