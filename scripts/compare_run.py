#!/usr/bin/env python3
"""End-to-end counterpart of the reference's ViT-vs-raw-IQ comparison (BASELINE.json configs[4]).

Trains BOTH model families on the same seeded synthetic IQ set with the fused native trainer, evaluates each on the
held-out split with evaluate_model_with_confusion, and writes the two classification reports at the exact relative
paths Transformer_Thesis/compare_models.py hard-codes (compare_models.py:402-403):

    <out>/ViT/result/checkpoints/production_v2/evaluation/test_classification_report.txt
    <out>/transformer_rawIQ/result/checkpoints/exp_L9_H8_F1024_W1e-3/evaluation/test_classification_report.txt

so `cd <out> && python /path/to/compare_models.py` parses them unchanged (its regexes at :39, :44, :49).  Checkpoints
are saved next to them in the reference's dict format.  One process per GPU under torch.distributed (data parallel,
RCCL): rank r trains on its shard, rank 0 evaluates and writes.

  python scripts/compare_run.py --out /tmp/cmp --frames 6000 --epochs 3 [--vit-geometry default|tiny224]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VIT_REPORT_DIR = os.path.join("ViT", "result", "checkpoints", "production_v2", "evaluation")
RAW_REPORT_DIR = os.path.join("transformer_rawIQ", "result", "checkpoints", "exp_L9_H8_F1024_W1e-3", "evaluation")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--frames", type=int, default=6000)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--lr", type=float, default=2e-4, help="1e-3 collapses both post-norm stacks to one class (measured)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--small", action="store_true", help="2-layer d64 models (tests)")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import vit_vs_raw_iq_amd as P
    from vit_vs_raw_iq_amd import data as D
    from vit_vs_raw_iq_amd.checkpoint import save_checkpoint
    from vit_vs_raw_iq_amd.evaluation import evaluate_model_with_confusion
    from vit_vs_raw_iq_amd.trainer import FusedTrainer, shard_indices

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("compare_run.py needs an MI355X (no CPU fallback)")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("IQ_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    classes = list(D.CLASSES)
    X, Y, Z = D.make_dataset(a.frames, seed=a.seed, classes=classes, snrs_db=(-8.0, 0.0, 8.0, 20.0), n_symbols=1024)
    idx = D.split_indices(Y, Z, sorted(set(Y.tolist())))         # stratified 70 / 15 / 15 (V/dataloader/utils.py:58-148)
    tr_idx, te_idx = np.asarray(idx[0]), np.asarray(idx[2])
    stats = D.normalization_stats(X, tr_idx, seed=49, n_subset=min(5000, len(tr_idx)))
    Yt, Zt = torch.from_numpy(Y), torch.from_numpy(Z)

    k = len(classes)
    if a.small:
        vit_kw = dict(in_channels=1, img_size_h=32, img_size_w=64, patch_size=4, num_classes=k, d_model=64, n_head=4,
                      n_layers=2, ffn_hidden=128)
        raw_kw = dict(in_channels=2, seq_length=1024, num_classes=k, d_model=64, n_head=4, n_layers=2, ffn_hidden=128,
                      use_cls_token=True, embedding_type="segment", segment_size=16)
    else:
        # V/training/train.py:83-88 defaults; R/result/checkpoints/exp_L9_H8_F1024_W1e-3/config.json
        vit_kw = dict(in_channels=1, img_size_h=32, img_size_w=64, patch_size=4, num_classes=k, d_model=128, n_head=8,
                      n_layers=6, ffn_hidden=512)
        raw_kw = dict(in_channels=2, seq_length=1024, num_classes=k, d_model=256, n_head=8, n_layers=9, ffn_hidden=1024,
                      use_cls_token=True, embedding_type="segment", segment_size=16)
    runs = [("ViT", P.AMCTransformerViT, vit_kw, "vit", 0.1, 1e-3, VIT_REPORT_DIR),
            ("transformer_rawIQ", P.AMCTransformerRawIQ, raw_kw, "rawiq", 0.1, 1e-3, RAW_REPORT_DIR)]
    summary = {}
    for name, cls, kw, layout, drop, wd, rep_dir in runs:
        frames = torch.from_numpy(D.preprocess_reference(X, stats, layout))
        torch.manual_seed(a.seed)
        model = cls(drop_prob=drop, device="cuda", **kw).to(dev).train()
        tr = FusedTrainer(model, lr=a.lr, weight_decay=wd, betas=(0.9, 0.99), label_smoothing=0.1, max_norm=1.0,
                          dropout_seed=a.seed)
        history = {"train_loss": [], "train_acc": []}
        t0 = time.time()
        for epoch in range(a.epochs):
            order = tr_idx[shard_indices(len(tr_idx), rank, world, epoch, seed=a.seed).numpy()]
            for i in range(0, len(order) - a.batch + 1, a.batch):
                b = order[i:i + a.batch]
                tr.step(frames[b].to(dev, non_blocking=True), Yt[b].to(dev, non_blocking=True))
            loss, acc, _ = tr.read_stats()
            history["train_loss"].append(loss)
            history["train_acc"].append(acc)
            if rank == 0:
                print(f"[{name}] epoch {epoch + 1}/{a.epochs}: loss {loss:.4f} acc {acc:.4f} ({time.time() - t0:.1f} s)", flush=True)
        if rank == 0:
            loader = [(frames[te_idx[i:i + 512]], Yt[te_idx[i:i + 512]], Zt[te_idx[i:i + 512]]) for i in range(0, len(te_idx), 512)]
            res = evaluate_model_with_confusion(model, loader, dev, classes, os.path.join(a.out, rep_dir), prefix="test")
            save_checkpoint(os.path.join(a.out, os.path.dirname(rep_dir), "model_final.pth"), model, trainer=tr,
                            epoch=a.epochs, val_loss=float(history["train_loss"][-1]), history=history,
                            config={k_.upper(): v for k_, v in kw.items()})
            summary[name] = {"overall_accuracy": res["overall_accuracy"],
                             "snr_accuracies": {str(s): v for s, v in res["snr_accuracies"].items()},
                             "report": os.path.join(a.out, rep_dir, "test_classification_report.txt")}
        if world > 1:
            dist.barrier()
        del tr, model
    if rank == 0:
        print(json.dumps(summary))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
