"""Checkpoint interchange.

Same dict layout as the reference's save_checkpoint / load_checkpoint (Transformer_Thesis/ViT/training/utils.py:
573-618): {'epoch', 'model_state_dict', 'optimizer_state_dict', 'val_loss', 'history', ['scheduler_state_dict'],
['config']}.  `model_state_dict` has the reference's keys and shapes (fp32), so reference-trained .pth files load
here and files written here load in the reference.  The fused trainer keeps AdamW moments in flat fp32 buffers;
`optimizer_state_dict` converts them to / from torch.optim.AdamW's own state_dict format (per-parameter
'step' / 'exp_avg' / 'exp_avg_sq' in model.parameters() order), so either side can resume the other's run.
Under DDP only rank 0 writes.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional

import torch
import torch.distributed as dist


def optimizer_state_dict(trainer) -> Dict:
    plan = trainer.plan
    named = list(trainer.model.named_parameters())
    m_views = plan.grad_views(trainer.exp_avg, named)
    v_views = plan.grad_views(trainer.exp_avg_sq, named)
    step = torch.tensor(float(trainer.steps))
    state = {i: {"step": step.clone(), "exp_avg": m.detach().cpu().clone(), "exp_avg_sq": v.detach().cpu().clone()}
             for i, (m, v) in enumerate(zip(m_views, v_views))}
    group = {"lr": trainer.lr, "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": trainer.wd,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "decoupled_weight_decay": True, "params": list(range(len(named)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(trainer, sd: Dict) -> None:
    plan = trainer.plan
    named = list(trainer.model.named_parameters())
    m_views = plan.grad_views(trainer.exp_avg, named)
    v_views = plan.grad_views(trainer.exp_avg_sq, named)
    ids = sd["param_groups"][0]["params"]
    if len(ids) != len(named):
        raise ValueError(f"optimizer state has {len(ids)} parameters, model has {len(named)}")
    steps = 0
    with torch.no_grad():
        trainer.exp_avg.zero_()
        trainer.exp_avg_sq.zero_()
        for pos, pid in enumerate(ids):
            st = sd["state"].get(pid)
            if st is None:
                continue
            m_views[pos].copy_(st["exp_avg"].to(m_views[pos].device).view_as(m_views[pos]))
            v_views[pos].copy_(st["exp_avg_sq"].to(v_views[pos].device).view_as(v_views[pos]))
            steps = max(steps, int(float(st["step"])))
    g = sd["param_groups"][0]
    trainer.wd, trainer.eps, trainer.betas = g["weight_decay"], g["eps"], tuple(g["betas"])
    trainer.set_lr(g["lr"])
    trainer.steps = steps
    trainer.plan.step = steps
    trainer.dyn[1] = float(steps)
    trainer.plan.ctr_value = -1       # the device-side dropout step is re-synchronised by the next step()
    trainer._graphs = None            # hyper-parameters baked into a captured graph may have changed


def save_checkpoint(filepath, model, trainer=None, optimizer=None, scheduler=None, epoch: int = 0,
                    val_loss: float = float("inf"), history: Optional[Dict] = None, config: Optional[Dict] = None):
    if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
        return
    ckpt = {"epoch": epoch,
            "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
            "optimizer_state_dict": optimizer_state_dict(trainer) if trainer is not None else optimizer.state_dict(),
            "val_loss": val_loss, "history": history if history is not None else {}}
    if scheduler is not None:
        ckpt["scheduler_state_dict"] = scheduler.state_dict()
    if config is not None:
        ckpt["config"] = config
    Path(filepath).parent.mkdir(parents=True, exist_ok=True)
    torch.save(ckpt, filepath)


def load_checkpoint(filepath, model, trainer=None, optimizer=None, scheduler=None, trust: bool = False) -> Dict:
    """A checkpoint holds tensors, numbers, strings, lists and dicts only, so it is read with the loader that executes
    nothing from the file (weights_only=True) -- reference-trained and third-party .pth files are not trusted code.
    `trust=True` falls back to full unpickling for a file you wrote yourself that carries other Python objects."""
    try:
        ckpt = torch.load(filepath, map_location="cpu", weights_only=not trust)
    except Exception as exc:       # pickle.UnpicklingError from the weights_only loader
        if trust:
            raise
        raise RuntimeError(
            f"{filepath}: the safe loader (weights_only=True) refused this checkpoint: {exc}\n"
            "A checkpoint whose `history` / `config` hold other Python objects (numpy scalars, paths, classes; the "
            "reference's save_checkpoint stores whatever it is given) is only readable by unpickling, which executes "
            "code from the file.  If you wrote the file yourself, pass trust=True; nothing is unpickled silently."
        ) from exc
    model.load_state_dict(ckpt["model_state_dict"])
    if trainer is not None:
        trainer.plan.mark_dirty()
        trainer.plan.ensure(trainer.device)      # pushes the loaded fp32 values into the bf16 shadows, optimizer state or not
    if "optimizer_state_dict" in ckpt:
        if trainer is not None:
            load_optimizer_state_dict(trainer, ckpt["optimizer_state_dict"])
        elif optimizer is not None:
            optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return ckpt
