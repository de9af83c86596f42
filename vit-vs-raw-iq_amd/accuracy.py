"""Top-1 accuracy of the MI355X path on the shared synthetic IQ task (`data.accuracy_task`): the build's half of SURVEY
8(d) metric (2).  The training loop is the reference's step (V/training/train.py:185-207: CE with label smoothing 0.1,
clip 1.0, AdamW betas (0.9, 0.99)) through `FusedTrainer`; evaluation is `validate_epoch`'s argmax accuracy
(V/training/train.py:223-260).  The CPU restatement's half lives in oracle/accuracy_oracle.py (test infrastructure)."""
from __future__ import annotations

import torch

from . import _native as N


def _top1(model, x, y, device, batch=500):
    correct = 0
    with torch.no_grad():
        for i in range(0, x.shape[0], batch):
            correct += int((model(x[i:i + batch].to(device)).argmax(1).cpu() == y[i:i + batch]).sum())
    return correct / x.shape[0]


def train_and_score(task: dict, state_dict: dict, device="cuda:0", dropout_seed: int = 5, use_graph: bool = False) -> dict:
    """Train from `state_dict` (the seeded initial values both sides start from) for the task's fixed step budget on the
    GPU; -> {"train", "heldout", "fresh"} accuracies."""
    from .modules import AMCTransformerViT, AMCTransformerRawIQ
    from .trainer import FusedTrainer
    device = torch.device(device)
    if device.type != "cuda":
        raise N.IqError("accuracy.train_and_score runs on an MI355X only (no CPU fallback)")
    h = task["hyper"]
    cls = AMCTransformerViT if task["kind"] == "vit" else AMCTransformerRawIQ
    model = cls(drop_prob=h["drop_prob"], device="cuda", **task["kw"])
    model.load_state_dict(state_dict)
    model.to(device).train()
    tr = FusedTrainer(model, lr=h["lr"], weight_decay=task["wd"], betas=(0.9, 0.99), label_smoothing=0.1, max_norm=1.0,
                      dropout_seed=dropout_seed, use_graph=use_graph)
    xtr, ytr = task["xtr"].to(device), task["ytr"].to(device)
    n, bs = xtr.shape[0], h["batch"]
    for s in range(h["steps"]):
        i = (s * bs) % n
        tr.step(xtr[i:i + bs], ytr[i:i + bs])
    model.eval()
    out = {"train": _top1(model, task["xtr"], task["ytr"], device), "heldout": _top1(model, task["xte"], task["yte"], device),
           "fresh": _top1(model, task["xfresh"], task["yfresh"], device)}
    del tr, model
    return out
