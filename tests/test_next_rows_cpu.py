"""CPU: host logic of the SURVEY 8(f) components (report format, PSO, particle snapping)."""
import re

import numpy as np
import pytest

import vit_vs_raw_iq_amd  # noqa: F401  (registers the package)
from vit_vs_raw_iq_amd import evaluation as E
from vit_vs_raw_iq_amd import sweep as SW
from vit_vs_raw_iq_amd import data as D


def test_report_body_equals_sklearn_and_compare_models_regexes_parse_it(tmp_path):
    from sklearn.metrics import classification_report, confusion_matrix
    rng = np.random.default_rng(0)
    names = D.CLASSES
    y = rng.integers(0, len(names), 5000)
    p = np.where(rng.random(5000) < 0.6, y, rng.integers(0, len(names), 5000))
    cm = confusion_matrix(y, p, labels=list(range(len(names))))
    body = E.classification_report_text(cm, names, digits=4)
    assert body == classification_report(y, p, target_names=names, digits=4)
    path = tmp_path / "ViT" / "test_classification_report.txt"
    E.write_report(path, "test", 0.6202, {-8: 0.1344, 0: 0.5231, 8: 0.9672}, body)
    content = path.read_text()
    # the three regexes of Transformer_Thesis/compare_models.py:39,44,49
    assert float(re.search(r'Overall Accuracy:\s+([\d.]+)%', content).group(1)) == 62.02
    snr = {int(s): float(a) for s, a in re.findall(r'SNR\s+([-+]\d+)\s+dB:\s+([\d.]+)%', content)}
    assert snr == {-8: 13.44, 0: 52.31, 8: 96.72}
    rows = {}
    for line in content.split('\n'):
        m = re.match(r'^\s*(\w+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)', line)
        if m and m.group(1) not in ('accuracy', 'macro', 'weighted'):
            rows[m.group(1)] = int(m.group(5))
    assert set(rows) == set(names) and sum(rows.values()) == 5000
    # header lines identical to the reference's writer (V/training/utils.py:393-400)
    assert content.startswith("Classification Report - Test Set\n" + "=" * 80 + "\n\nOverall Accuracy: 62.02%\n\nAccuracy by SNR:\n  SNR  -8 dB: 13.44%\n  SNR  +0 dB: 52.31%\n")


def test_pso_minimises_a_known_function():
    lo, hi = np.full(3, -5.0), np.full(3, 5.0)
    cost, best = SW.run_pso(lambda x: ((x - 1.5) ** 2).sum(axis=1), n_particles=18, iters=60, seed=1, bounds=(lo, hi))
    assert cost < 1e-3 and np.allclose(best, 1.5, atol=0.05)


def test_particles_snap_to_runnable_configurations():
    vit = dict(in_channels=1, img_h=32, img_w=64, num_classes=19, device="cpu")
    raw = dict(in_channels=2, seq_length=1024, num_classes=19, device="cpu")
    rng = np.random.default_rng(3)
    for _ in range(200):
        x = rng.uniform(SW.MIN_BOUNDS, SW.MAX_BOUNDS)
        s = SW.snap(x, vit, raw)
        assert s["d_model"] % s["n_head"] == 0 and s["d_model"] // s["n_head"] in (16, 32, 64)
        assert s["d_model"] <= 512 and s["ffn_hidden"] % 8 == 0 and 1 <= s["n_layers"] <= 8
        if s["model_type"] == 0:
            assert 32 % s["patch_or_segment"] == 0 and 64 % s["patch_or_segment"] == 0
        else:
            assert 1024 % s["patch_or_segment"] == 0
        model, _ = SW.build_models(x, raw, vit)          # the reference's two constructor call sites
        assert sum(p.numel() for p in model.parameters()) > 0


def test_synthetic_dataset_fields_and_transforms():
    X, Y, Z = D.make_dataset(190, seed=1)
    assert X.shape == (190, 1024, 2) and X.dtype == np.float32 and Y.dtype == np.int64 and Z.dtype == np.float32
    assert set(np.unique(Y)) == set(range(19)) and set(np.unique(Z)) <= set(D.SNRS_DB)
    X2, Y2, _ = D.make_dataset(190, seed=1)
    assert np.array_equal(X, X2) and np.array_equal(Y, Y2)            # seeded
    mean, std = D.zscore_stats(X)
    img = D.to_vit_images(X, mean, std)
    raw = D.to_rawiq(X, mean, std)
    assert img.shape == (190, 1, 32, 64) and raw.shape == (190, 2, 1024)
    # [I ; Q] layout of V/dataloader/dataset.py:216-224
    assert np.allclose(img[0, 0].reshape(-1)[:1024], raw[0, 0]) and np.allclose(img[0, 0].reshape(-1)[1024:], raw[0, 1])
    assert D.to_vit_images(X, mean, std, 32, 32).shape == (190, 1, 32, 32)
    for name in D.CLASSES[:17]:
        c = D.constellation(name)
        assert abs(np.mean(np.abs(c) ** 2) - 1.0) < 1e-9


def test_split_indices_is_stratified_disjoint_and_deterministic():
    """the (modulation x SNR)-stratified split of V/dataloader/utils.py:58-148."""
    import numpy as np
    from vit_vs_raw_iq_amd import data as D
    rng = np.random.default_rng(0)
    mods = np.array(["BPSK", "QPSK", "8PSK", "16QAM"])
    labels = np.repeat(mods, 3 * 40)
    snrs = np.tile(np.repeat(np.array([-4.0, 6.0, 20.0]), 40), 4)
    perm = rng.permutation(len(labels))
    labels, snrs = labels[perm], snrs[perm]
    tr, va, te, lmap = D.split_indices(labels, snrs, ["BPSK", "QPSK", "16QAM"], 0.7, 0.15, 0.15, seed=49)
    assert lmap == {"BPSK": 0, "QPSK": 1, "16QAM": 2}
    allidx = np.concatenate([tr, va, te])
    assert len(np.unique(allidx)) == len(allidx) == 3 * 3 * 40          # disjoint and complete over the target classes
    assert not np.isin(labels[allidx], ["8PSK"]).any()
    for mod in ("BPSK", "QPSK", "16QAM"):
        for snr in (-4.0, 6.0, 20.0):
            cell = lambda idx: int(((labels[idx] == mod) & (snrs[idx] == snr)).sum())
            assert (cell(tr), cell(va), cell(te)) == (28, 6, 6)           # 40 -> test 6, then 34 -> valid 6 (sklearn ceil rule)
    tr2, va2, te2, _ = D.split_indices(labels, snrs, ["BPSK", "QPSK", "16QAM"], 0.7, 0.15, 0.15, seed=49)
    assert np.array_equal(tr, tr2) and np.array_equal(va, va2) and np.array_equal(te, te2)
    tr3, _, _, _ = D.split_indices(labels, snrs, ["BPSK", "QPSK", "16QAM"], 0.7, 0.15, 0.15, seed=50)
    assert not np.array_equal(tr, tr3)
    import pytest
    with pytest.raises(ValueError):
        D.split_indices(labels, snrs, ["BPSK"], 0.7, 0.2, 0.2, seed=1)


def test_normalization_stats_and_reference_preprocessing():
    import numpy as np, torch
    from vit_vs_raw_iq_amd import data as D
    X, Y, Z = D.make_dataset(300, seed=3)
    idx = np.arange(0, 300, 2)
    st = D.normalization_stats(X, idx, seed=49, n_subset=100)
    np.random.seed(49)
    pick = np.sort(np.random.choice(idx, 100, replace=False))
    sub = torch.from_numpy(X[pick])
    assert st["i_mean"] == sub[:, :, 0].mean().item() and st["q_std"] == sub[:, :, 1].flatten().std().item()
    img = D.preprocess_reference(X[:5], st, "vit")
    seq = D.preprocess_reference(X[:5], st, "rawiq")
    assert img.shape == (5, 1, 32, 64) and seq.shape == (5, 2, 1024)
    assert np.array_equal(img.reshape(5, 2, 1024), seq)                 # same numbers, two views ([I;Q] concat == (2, len))
    assert abs(float(seq[:, 0].mean())) < 0.2


def test_checkpoint_with_foreign_objects_is_refused_with_a_pointer_to_trust(tmp_path):
    """The reference's save_checkpoint stores `history` / `config` as given (V/training/utils.py:573-618): numpy
    scalars there make the file unreadable for the safe loader.  load_checkpoint must say so and name trust=True,
    never unpickle silently."""
    import torch
    from vit_vs_raw_iq_amd import checkpoint as CK
    path = tmp_path / "ref_style.pth"
    torch.save({"epoch": 3, "model_state_dict": {}, "val_loss": np.float64(0.25),
                "history": {"val_acc": [np.float32(0.5)]}, "config": {"root": tmp_path}}, path)
    with pytest.raises(RuntimeError, match="trust=True"):
        CK.load_checkpoint(path, model=None)
