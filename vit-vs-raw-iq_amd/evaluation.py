"""Evaluation + report writer.

Counterpart of evaluate_model_with_confusion (reference: Transformer_Thesis/ViT/training/utils.py:284-466).
The model runs on the MI355X path; predictions are reduced to a K x K confusion matrix ON DEVICE (one
bincount per batch, no per-batch host sync -- the reference copies every batch to numpy, :311-320) and the
text report is written in the exact layout of :392-401 so that Transformer_Thesis/compare_models.py
(regexes at :39, :44, :49) parses it unchanged.  Per-SNR accuracy uses the reference's +-0.5 dB windows at
-8 / 0 / +8 dB (:349-377).  The per-class table reproduces sklearn.metrics.classification_report(digits=4)
from the confusion matrix (checked against sklearn in tests/test_host_cpu.py).  Plots are not produced.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

TARGET_SNRS = (-8, 0, 8)


def classification_report_text(cm: np.ndarray, class_names: Sequence[str], digits: int = 4) -> str:
    """sklearn-compatible classification report from a confusion matrix (rows = true, cols = predicted)."""
    cm = np.asarray(cm, dtype=np.float64)
    support = cm.sum(axis=1)
    pred = cm.sum(axis=0)
    tp = np.diag(cm)
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = np.where(pred > 0, tp / pred, 0.0)
        recall = np.where(support > 0, tp / support, 0.0)
        f1 = np.where(precision + recall > 0, 2 * precision * recall / (precision + recall), 0.0)
    total = support.sum()
    accuracy = tp.sum() / total if total > 0 else 0.0
    headers = ["precision", "recall", "f1-score", "support"]
    longest_last_line_heading = "weighted avg"
    name_width = max(len(cn) for cn in class_names)
    width = max(name_width, len(longest_last_line_heading), digits)
    head_fmt = "{:>{width}s} " + " {:>9}" * len(headers)
    report = head_fmt.format("", *headers, width=width)
    report += "\n\n"
    row_fmt = "{:>{width}s} " + " {:>9.{digits}f}" * 3 + " {:>9}\n"
    for i, name in enumerate(class_names):
        report += row_fmt.format(name, precision[i], recall[i], f1[i], int(support[i]), width=width, digits=digits)
    report += "\n"
    row_fmt_accuracy = "{:>{width}s} " + " {:>9.{digits}}" * 2 + " {:>9.{digits}f}" + " {:>9}\n"
    report += row_fmt_accuracy.format("accuracy", "", "", accuracy, int(total), width=width, digits=digits)
    macro = (precision.mean(), recall.mean(), f1.mean())
    w = support / total if total > 0 else support
    weighted = ((precision * w).sum(), (recall * w).sum(), (f1 * w).sum())
    report += row_fmt.format("macro avg", *macro, int(total), width=width, digits=digits)
    report += row_fmt.format("weighted avg", *weighted, int(total), width=width, digits=digits)
    return report


def write_report(path: Path, prefix: str, overall: float, snr_acc: Dict[int, float], body: str) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w") as f:
        f.write(f"Classification Report - {prefix.capitalize()} Set\n")
        f.write("=" * 80 + "\n\n")
        f.write(f"Overall Accuracy: {overall * 100:.2f}%\n\n")
        f.write("Accuracy by SNR:\n")
        for snr, acc in snr_acc.items():
            f.write(f"  SNR {snr:+3d} dB: {acc * 100:.2f}%\n")
        f.write("\n" + "=" * 80 + "\n\n")
        f.write(body)


@torch.no_grad()
def evaluate_model_with_confusion(model: torch.nn.Module, dataloader: Iterable, device: torch.device,
                                  class_names: List[str], save_dir: Path, prefix: str = "test") -> Dict:
    """`dataloader` yields (inputs, labels, snrs) like the reference's DataLoader (V/training/utils.py:311)."""
    device = torch.device(device)
    K = len(class_names)
    was_training = model.training
    model.eval()
    cm = torch.zeros(K * K, dtype=torch.int64, device=device)
    cm_snr = {s: torch.zeros(K * K, dtype=torch.int64, device=device) for s in TARGET_SNRS}
    preds, labels_all, snrs_all = [], [], []
    for x, y, z in dataloader:
        x = x.to(device, non_blocking=True)
        y = y.to(device, non_blocking=True).long()
        z = z.to(device, non_blocking=True).float()
        p = model(x).argmax(1)
        idx = y * K + p
        cm += torch.bincount(idx, minlength=K * K)
        for s in TARGET_SNRS:
            m = (z - s).abs() <= 0.5
            cm_snr[s] += torch.bincount(idx[m], minlength=K * K)
        preds.append(p)
        labels_all.append(y)
        snrs_all.append(z)
    model.train(was_training)
    cm_np = cm.view(K, K).cpu().numpy()                       # single host transfer
    overall = float(np.trace(cm_np) / max(cm_np.sum(), 1))
    snr_acc = {}
    for s in TARGET_SNRS:
        c = cm_snr[s].view(K, K).cpu().numpy()
        if c.sum() > 0:
            snr_acc[s] = float(np.trace(c) / c.sum())
    save_dir = Path(save_dir)
    write_report(save_dir / f"{prefix}_classification_report.txt", prefix, overall, snr_acc,
                 classification_report_text(cm_np, class_names, digits=4))
    return {"overall_accuracy": overall, "snr_accuracies": snr_acc, "confusion_matrix": cm_np,
            "predictions": torch.cat(preds).cpu().numpy() if preds else np.zeros(0, np.int64),
            "labels": torch.cat(labels_all).cpu().numpy() if labels_all else np.zeros(0, np.int64),
            "snrs": torch.cat(snrs_all).cpu().numpy() if snrs_all else np.zeros(0, np.float32)}
