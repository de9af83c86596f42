"""Working counterpart of Transformer_Thesis/hyperparameter_tuning.py.

The reference script cannot run (SyntaxError at :4, :18; pyswarms absent), so this honours what it specifies:
  * particle = [model_type, d_model, n_head, n_layers, ffn_hidden, drop_prob, lr, batch, patch|segment]  (:10-16,:89-92)
  * bounds [0,32,2,1,64,0.0,1e-5,16,4] .. [1,512,16,8,2048,0.4,5e-3,128,64]                                (:107-130)
  * global-best PSO, 18 particles x 25 iterations, c1 = c2 = 1.5, w = 0.6                                    (:134-144)
  * fitness = -(validation accuracy) after ONE Adam(lr) step with plain cross entropy                        (:56-84)
  * the two constructor call sites, verbatim keyword arguments                                               (:22-34,:41-54)
Raw PSO coordinates crash the reference's head split (view at multi_head_attention.py:38); here they are snapped
to what the kernels run: head dim in {16,32,64}, d_model = n_head*dh, ffn_hidden % 8 == 0, patch/segment a divisor.
Particles are independent models => task parallel: rank r evaluates particles r, r+W, ... and the W ranks
all-gather the 18 fitness scalars (no other collective).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

MIN_BOUNDS = np.array([0, 32, 2, 1, 64, 0.0, 1e-5, 16, 4], dtype=np.float64)
MAX_BOUNDS = np.array([1, 512, 16, 8, 2048, 0.4, 5e-3, 128, 64], dtype=np.float64)


def _divisors(n, lo, hi):
    return [d for d in range(max(1, lo), hi + 1) if n % d == 0]


def snap(params: Sequence[float], vit_cfg: Dict, rawiq_cfg: Dict) -> Dict:
    """Particle -> a configuration the native path accepts (deterministic)."""
    model_type = int(round(float(params[0]))) & 1
    n_head = int(np.clip(round(float(params[2])), 1, 16))
    dh = min((16, 32, 64), key=lambda c: abs(c - float(params[1]) / n_head))
    while n_head * dh > 512 and n_head > 1:
        n_head -= 1
    d_model = n_head * dh
    out = dict(model_type=model_type, d_model=d_model, n_head=n_head,
               n_layers=int(np.clip(round(float(params[3])), 1, 8)),
               ffn_hidden=int(np.clip(round(float(params[4]) / 8) * 8, 64, 2048)),
               drop_prob=float(np.clip(params[5], 0.0, 0.4)), lr=float(np.clip(params[6], 1e-5, 5e-3)),
               batch=int(np.clip(round(float(params[7])), 16, 128)))
    want = float(params[8])
    if model_type == 0:
        cands = sorted(set(_divisors(vit_cfg["img_h"], 2, 64)) & set(_divisors(vit_cfg["img_w"], 2, 64)))
    else:
        cands = _divisors(rawiq_cfg["seq_length"], 4, 64)
    out["patch_or_segment"] = min(cands, key=lambda c: abs(c - want))
    return out


def build_models(params, rawiq_cfg, vit_cfg):
    """hyperparameter_tuning.py:8-54 with snapped values."""
    s = snap(params, vit_cfg, rawiq_cfg)
    if s["model_type"] == 0:
        from .ViT.models.amc_transformer import AMCTransformer
        return AMCTransformer(in_channels=vit_cfg["in_channels"], img_size_h=vit_cfg["img_h"],
                              img_size_w=vit_cfg["img_w"], patch_size=s["patch_or_segment"],
                              num_classes=vit_cfg["num_classes"], d_model=s["d_model"], n_head=s["n_head"],
                              n_layers=s["n_layers"], ffn_hidden=s["ffn_hidden"], drop_prob=s["drop_prob"],
                              device=vit_cfg["device"]), s
    from .transformer_rawIQ.models.transformer_rawIQ import AMCTransformer
    return AMCTransformer(in_channels=rawiq_cfg["in_channels"], seq_length=rawiq_cfg["seq_length"],
                          num_classes=rawiq_cfg["num_classes"], d_model=s["d_model"], n_head=s["n_head"],
                          n_layers=s["n_layers"], ffn_hidden=s["ffn_hidden"], drop_prob=s["drop_prob"],
                          device=rawiq_cfg["device"], use_cls_token=True, embedding_type="segment",
                          segment_size=s["patch_or_segment"]), s


def fast_train(model, train, val, lr, batch_size, device) -> float:
    """hyperparameter_tuning.py:56-84: one Adam step on one batch, then validation accuracy."""
    (xtr, ytr), (xva, yva) = train, val
    model = model.to(device)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    loss_fn = torch.nn.CrossEntropyLoss()
    model.train()
    xb, yb = xtr[:batch_size].to(device), ytr[:batch_size].to(device)
    opt.zero_grad()
    loss_fn(model(xb), yb).backward()
    opt.step()
    model.eval()
    correct = total = 0
    with torch.no_grad():
        for i in range(0, xva.shape[0], 256):
            pred = model(xva[i:i + 256].to(device)).argmax(dim=1)
            correct += int((pred == yva[i:i + 256].to(device)).sum())
            total += min(256, xva.shape[0] - i)
    return correct / max(total, 1)


def fitness_function(X, data_vit, data_rawiq, rawiq_cfg, vit_cfg, device) -> np.ndarray:
    """Scores for all particles; evaluated task-parallel over ranks when torch.distributed is initialised."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    scores = np.zeros(len(X), dtype=np.float64)
    for i in range(rank, len(X), world):
        model, s = build_models(X[i], rawiq_cfg, vit_cfg)
        train, val = data_vit if s["model_type"] == 0 else data_rawiq
        scores[i] = -fast_train(model, train, val, s["lr"], s["batch"], device)
        del model
    if world > 1:
        t = torch.from_numpy(scores).to(device)
        dist.all_reduce(t)                     # disjoint supports: sum == gather
        scores = t.cpu().numpy()
    return scores


def run_pso(fitness: Callable[[np.ndarray], np.ndarray], n_particles: int = 18, iters: int = 25, c1: float = 1.5,
            c2: float = 1.5, w: float = 0.6, seed: int = 0, bounds: Tuple[np.ndarray, np.ndarray] = (MIN_BOUNDS, MAX_BOUNDS)):
    """Minimal global-best PSO (the reference asks pyswarms.GlobalBestPSO for exactly this update)."""
    lo, hi = bounds
    rng = np.random.default_rng(seed)
    x = rng.uniform(lo, hi, size=(n_particles, len(lo)))
    v = rng.uniform(-(hi - lo), hi - lo, size=x.shape) * 0.1
    pbest, pcost = x.copy(), fitness(x)
    g = int(np.argmin(pcost))
    gbest, gcost = pbest[g].copy(), float(pcost[g])
    for _ in range(iters):
        r1, r2 = rng.random(x.shape), rng.random(x.shape)
        v = w * v + c1 * r1 * (pbest - x) + c2 * r2 * (gbest - x)
        x = np.clip(x + v, lo, hi)
        cost = fitness(x)
        better = cost < pcost
        pbest[better], pcost[better] = x[better], cost[better]
        g = int(np.argmin(pcost))
        if pcost[g] < gcost:
            gbest, gcost = pbest[g].copy(), float(pcost[g])
    return gcost, gbest
