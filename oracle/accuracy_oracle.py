"""CPU-restatement half of the accuracy reproduction (SURVEY 8(d) metric (2)).  TEST INFRASTRUCTURE ONLY, like
iq_oracle.py: imported by tests/ and by bench.py's cpu_baseline leg, never by the product package.

Same task definition (`vit_vs_raw_iq_amd.data.accuracy_task`: frames, split, seeds, step budget), same initial state,
same loop as the reference's train_epoch body (V/training/train.py:185-207 via iq_oracle.train_step), fp32 on the host
cores with torch's own dropout RNG."""
import torch

import iq_oracle as O


def initial_state(task: dict):
    cfg = O.OracleConfig(kind=task["kind"], drop_prob=task["hyper"]["drop_prob"], **task["kw"])
    return cfg, O.init_state(cfg, task["hyper"]["init_seed"])


def _top1(cfg, sd, x, y, batch=500):
    correct = 0
    with torch.no_grad():
        for i in range(0, x.shape[0], batch):
            correct += int((O.model_forward(cfg, sd, x[i:i + batch]).argmax(1) == y[i:i + batch]).sum())
    return correct / x.shape[0]


def train_and_score(task: dict, torch_seed: int = 0) -> dict:
    h = task["hyper"]
    cfg, sd = initial_state(task)
    sd = {k: v.clone() for k, v in sd.items()}
    st = O.adamw_init(sd)
    torch.manual_seed(torch_seed)
    n, bs = task["xtr"].shape[0], h["batch"]
    for s in range(h["steps"]):
        i = (s * bs) % n
        O.train_step(cfg, sd, st, task["xtr"][i:i + bs], task["ytr"][i:i + bs], lr=h["lr"], weight_decay=task["wd"],
                     smoothing=0.1, max_norm=1.0, train=True)
    return {"train": _top1(cfg, sd, task["xtr"], task["ytr"]), "heldout": _top1(cfg, sd, task["xte"], task["yte"]),
            "fresh": _top1(cfg, sd, task["xfresh"], task["yfresh"])}
