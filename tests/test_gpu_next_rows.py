"""GPU (MI355X): SURVEY 8(f) components end to end -- evaluation + report, checkpoint interchange, sweep."""
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def small_task(n=600):
    from vit_vs_raw_iq_amd import data as D
    classes = ["BPSK", "QPSK", "16QAM", "OOK"]
    X, Y, Z = D.make_dataset(n, seed=7, classes=classes, snrs_db=(-8.0, 0.0, 8.0), n_symbols=1024)
    mean, std = D.zscore_stats(X)
    return classes, torch.from_numpy(D.to_rawiq(X, mean, std)), torch.from_numpy(Y), torch.from_numpy(Z)


def small_model(k, drop=0.1):
    import vit_vs_raw_iq_amd as P
    return P.AMCTransformerRawIQ(in_channels=2, seq_length=1024, num_classes=k, d_model=64, n_head=4, n_layers=2,
                                 ffn_hidden=128, drop_prob=drop, device="cuda", segment_size=16)


def test_evaluation_writes_the_reference_report(tmp_path):
    from vit_vs_raw_iq_amd.evaluation import evaluate_model_with_confusion
    from vit_vs_raw_iq_amd.trainer import FusedTrainer
    d = dev()
    classes, x, y, z = small_task()
    torch.manual_seed(0)
    m = small_model(len(classes)).to(d).train()
    tr = FusedTrainer(m, lr=2e-3, weight_decay=1e-4)
    for s in range(60):
        i = (s * 100) % 500
        tr.step(x[i:i + 100].to(d), y[i:i + 100].to(d))
    loader = [(x[i:i + 128], y[i:i + 128], z[i:i + 128]) for i in range(0, 600, 128)]
    res = evaluate_model_with_confusion(m, loader, d, classes, tmp_path / "evaluation", prefix="test")
    assert set(res) == {"overall_accuracy", "snr_accuracies", "confusion_matrix", "predictions", "labels", "snrs"}
    assert res["confusion_matrix"].sum() == 600 and res["predictions"].shape == (600,)
    assert abs(res["overall_accuracy"] - float((res["predictions"] == res["labels"]).mean())) < 1e-12
    assert set(res["snr_accuracies"]) == {-8, 0, 8}
    for s_, a_ in res["snr_accuracies"].items():
        mask = np.abs(res["snrs"] - s_) <= 0.5
        assert abs(a_ - float((res["predictions"][mask] == res["labels"][mask]).mean())) < 1e-12
    text = (tmp_path / "evaluation" / "test_classification_report.txt").read_text()
    assert abs(float(re.search(r'Overall Accuracy:\s+([\d.]+)%', text).group(1)) - res["overall_accuracy"] * 100) < 0.006
    assert len(re.findall(r'SNR\s+([-+]\d+)\s+dB:\s+([\d.]+)%', text)) == 3
    assert m.training                                                        # mode restored


def test_checkpoint_round_trip_and_torch_interchange(tmp_path):
    from vit_vs_raw_iq_amd.checkpoint import save_checkpoint, load_checkpoint
    from vit_vs_raw_iq_amd.trainer import FusedTrainer
    d = dev()
    classes, x, y, _ = small_task(300)
    xb, yb = x[:100].to(d), y[:100].to(d)

    def fresh():
        torch.manual_seed(1)
        m = small_model(len(classes), drop=0.0).to(d).train()
        return m, FusedTrainer(m, lr=1e-3, weight_decay=1e-2)

    m1, t1 = fresh()
    for _ in range(3):
        t1.step(xb, yb)
    save_checkpoint(tmp_path / "ck.pth", m1, trainer=t1, epoch=7, val_loss=1.25, history={"train_loss": [1.0]},
                    config={"D_MODEL": 64})
    for _ in range(2):
        t1.step(xb, yb)
    ref = {k: v.detach().cpu().clone() for k, v in m1.state_dict().items()}
    # resume in a fresh process-equivalent
    m2, t2 = fresh()
    ck = load_checkpoint(tmp_path / "ck.pth", m2, trainer=t2)
    assert ck["epoch"] == 7 and ck["val_loss"] == 1.25 and ck["config"] == {"D_MODEL": 64}
    assert set(ck) >= {"epoch", "model_state_dict", "optimizer_state_dict", "val_loss", "history"}
    assert t2.steps == 3
    for _ in range(2):
        t2.step(xb, yb)
    for k, v in m2.state_dict().items():
        assert torch.equal(v.cpu(), ref[k]), k                                # bit-identical continuation
    # the optimizer state is torch.optim.AdamW's own format: the reference's loader accepts it
    m3 = small_model(len(classes), drop=0.0)
    opt = torch.optim.AdamW(m3.parameters(), lr=1e-4, weight_decay=1e-3, betas=(0.9, 0.99))
    raw = torch.load(tmp_path / "ck.pth", map_location="cpu", weights_only=False)
    m3.load_state_dict(raw["model_state_dict"])
    opt.load_state_dict(raw["optimizer_state_dict"])
    st = opt.state[next(iter(m3.parameters()))]
    assert float(st["step"]) == 3.0 and st["exp_avg"].abs().sum() > 0
    # and a torch-written optimizer state loads into the fused trainer
    m4, t4 = fresh()
    torch.save({"epoch": 1, "model_state_dict": m3.state_dict(), "optimizer_state_dict": opt.state_dict(),
                "val_loss": 0.0, "history": {}}, tmp_path / "torch.pth")
    load_checkpoint(tmp_path / "torch.pth", m4, trainer=t4)
    assert t4.steps == 3 and t4.lr == 1e-3 and t4.exp_avg.abs().sum().item() > 0


def test_sweep_runs_task_parallel_driver_single_rank():
    from vit_vs_raw_iq_amd import sweep as SW
    from vit_vs_raw_iq_amd import data as D
    d = dev()
    classes = ["BPSK", "QPSK", "16QAM", "OOK"]
    X, Y, _ = D.make_dataset(320, seed=3, classes=classes, snrs_db=(8.0,), n_symbols=1024)
    mean, std = D.zscore_stats(X)
    raw = torch.from_numpy(D.to_rawiq(X, mean, std))
    img = torch.from_numpy(D.to_vit_images(X, mean, std))
    Yt = torch.from_numpy(Y)
    data_raw = ((raw[:256], Yt[:256]), (raw[256:], Yt[256:]))
    data_vit = ((img[:256], Yt[:256]), (img[256:], Yt[256:]))
    vit_cfg = dict(in_channels=1, img_h=32, img_w=64, num_classes=4, device="cuda")
    raw_cfg = dict(in_channels=2, seq_length=1024, num_classes=4, device="cuda")
    lo, hi = SW.MIN_BOUNDS.copy(), SW.MAX_BOUNDS.copy()
    hi[1], hi[3], hi[4] = 128, 2, 256                       # keep the test quick
    evaluated = []

    def fit(Xp):
        s = SW.fitness_function(Xp, data_vit, data_raw, raw_cfg, vit_cfg, d)
        evaluated.append(s.copy())
        return s

    cost, best = SW.run_pso(fit, n_particles=6, iters=2, seed=0, bounds=(lo, hi))
    assert len(evaluated) == 3 and all(e.shape == (6,) for e in evaluated)
    assert -1.0 <= cost <= 0.0 and np.all(best >= lo) and np.all(best <= hi)
    assert cost == min(e.min() for e in evaluated)


@pytest.mark.gpu
def test_device_input_pipeline_matches_cpu_preprocessing_bitwise():
    """raw frames -> pinned H2D -> iq_frames_preprocess == the reference's per-frame CPU arithmetic."""
    import numpy as np, torch
    from vit_vs_raw_iq_amd import data as D
    X, Y, Z = D.make_dataset(700, seed=5)
    st = D.normalization_stats(X, np.arange(len(X)), seed=49, n_subset=500)
    pv = D.DeviceInputPipeline(st, "vit", batch=256)
    pr = D.DeviceInputPipeline(st, "rawiq", batch=256)
    ps = D.DeviceInputPipeline(st, "vit", batch=256, h=32, w=32)        # BASELINE configs[0]: first 512 I and Q samples
    for lo in (0, 256, 512):                                            # three batches: both staging slots get reused
        fr = X[lo:lo + 256] if lo < 512 else X[lo:]                     # last batch is short (188 frames)
        assert torch.equal(pv(fr).cpu(), torch.from_numpy(D.preprocess_reference(fr, st, "vit")))
        assert torch.equal(pr(fr).cpu(), torch.from_numpy(D.preprocess_reference(fr, st, "rawiq")))
        assert torch.equal(ps(fr).cpu(), torch.from_numpy(D.preprocess_reference(fr, st, "vit", 32, 32)))
    # submit / get split: the copy of batch k+1 overlaps whatever runs between them
    pv.submit(X[:256]); a = pv.get(); pv.submit(X[256:512]); b = pv.get()
    assert torch.equal(a.cpu(), torch.from_numpy(D.preprocess_reference(X[:256], st, "vit")))
    assert torch.equal(b.cpu(), torch.from_numpy(D.preprocess_reference(X[256:512], st, "vit")))
    with pytest.raises(ValueError):
        D.DeviceInputPipeline(st, "vit", batch=8, h=64, w=64)           # needs 2048 samples per channel


def test_compare_run_writes_both_reports_where_compare_models_reads_them(tmp_path):
    """scripts/compare_run.py trains both families, evaluates, and writes the two reports at the relative paths
    Transformer_Thesis/compare_models.py:402-403 hard-codes; its three regexes (:39, :44, :49) must parse them."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    dev()
    out = tmp_path / "cmp"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "compare_run.py"), "--out", str(out), "--frames",
                        "2280", "--epochs", "2", "--batch", "64", "--small"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    summary = json.loads(r.stdout.strip().splitlines()[-1])
    paths = {"ViT": out / "ViT/result/checkpoints/production_v2/evaluation/test_classification_report.txt",
             "transformer_rawIQ": out / "transformer_rawIQ/result/checkpoints/exp_L9_H8_F1024_W1e-3/evaluation/test_classification_report.txt"}
    for name, path in paths.items():
        content = path.read_text()
        overall = float(re.search(r'Overall Accuracy:\s+([\d.]+)%', content).group(1))
        assert abs(overall - 100 * summary[name]["overall_accuracy"]) < 0.006
        snr = {int(s): float(a) for s, a in re.findall(r'SNR\s+([-+]\d+)\s+dB:\s+([\d.]+)%', content)}
        assert set(snr) == {-8, 0, 8}
        rows = {}
        for line in content.split('\n'):
            mt = re.match(r'^\s*(\w+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)', line)
            if mt and mt.group(1) not in ('accuracy', 'macro', 'weighted'):
                rows[mt.group(1)] = int(mt.group(5))
        assert len(rows) == 19 and sum(rows.values()) > 300
        assert (path.parent.parent / "model_final.pth").exists()
