"""Seeded synthetic IQ frames with the field semantics of the reference's dataset.

The reference trains on RadioML 2018.01A (X (N,1024,2) float32, Y labels, Z SNR; README.md:226-231),
which is not shipped; it contains no generator.  The closest in-repo recipe is
Transformer_Thesis/test_sps_modes.py:10-27 (seeded unit-power QPSK at 1 sample/symbol + complex AWGN).
This module generalises that recipe to the 19 class names of ViT/training/train.py:60-80 so the
accuracy comparison of BASELINE.json ("top-1 accuracy reproduced on the same synthetic IQ set") has
a shared, deterministic data source for the GPU path and the CPU oracle.

The input pipeline of SURVEY 8(f) row 4 is here too: `split_indices` (the (modulation x SNR)-stratified 70/15/15 split of
V|R/dataloader/utils.py:58-148), `normalization_stats` (the 5000-frame I/Q statistics of V/dataloader/dataset.py:116-158)
and `DeviceInputPipeline` (pinned double-buffered H2D of RAW frames + z-score / layout on the GPU through
iq_frames_preprocess -- what the reference does per frame in DataLoader worker processes).  The HDF5 reader itself
(h5py, RadioML file) is not available in this environment and stays out of scope.

Pre-processing mirrors SingleStreamImageDataset.__getitem__:
  z-score per channel with statistics from a 5000-frame subset (ViT/dataloader/dataset.py:116-158,210-213)
  ViT:    [I(1024) ; Q(1024)] -> view(1, 32, 64)           (ViT/dataloader/dataset.py:216-224)
  rawIQ:  transpose -> (2, 1024)                            (transformer_rawIQ/dataloader/dataset.py:219-222)
"""
from __future__ import annotations

import numpy as np

CLASSES = ["OOK", "4ASK", "8ASK", "BPSK", "QPSK", "8PSK", "16PSK", "32PSK", "16APSK", "32APSK", "64APSK", "128APSK",
           "16QAM", "32QAM", "64QAM", "128QAM", "256QAM", "GMSK", "OQPSK"]
SNRS_DB = (-8.0, 0.0, 8.0, 20.0)


def _psk(m):
    return np.exp(2j * np.pi * np.arange(m) / m)


def _ask(m, on_off=False):
    lv = np.arange(m, dtype=np.float64) if on_off else (2 * np.arange(m) - (m - 1)).astype(np.float64)
    return lv.astype(np.complex128)


def _qam(m):
    side = int(np.ceil(np.sqrt(m)))
    pts = np.array([complex(2 * i - (side - 1), 2 * q - (side - 1)) for i in range(side) for q in range(side)])
    if side * side > m:                      # cross constellations: drop the points of largest radius
        pts = pts[np.argsort(np.abs(pts), kind="stable")[:m]]
    return pts


def _apsk(rings):
    pts = []
    for k, (n, r) in enumerate(rings):
        pts.append(r * np.exp(2j * np.pi * (np.arange(n) + 0.5 * (k % 2)) / n))
    return np.concatenate(pts)


def constellation(name: str) -> np.ndarray:
    table = {
        "OOK": lambda: _ask(2, on_off=True), "4ASK": lambda: _ask(4), "8ASK": lambda: _ask(8),
        "BPSK": lambda: _psk(2), "QPSK": lambda: _psk(4) * np.exp(1j * np.pi / 4), "8PSK": lambda: _psk(8),
        "16PSK": lambda: _psk(16), "32PSK": lambda: _psk(32),
        "16APSK": lambda: _apsk([(4, 1.0), (12, 2.6)]), "32APSK": lambda: _apsk([(4, 1.0), (12, 2.6), (16, 4.3)]),
        "64APSK": lambda: _apsk([(4, 1.0), (12, 2.4), (20, 3.8), (28, 5.2)]),
        "128APSK": lambda: _apsk([(8, 1.0), (16, 2.2), (24, 3.4), (32, 4.6), (48, 5.8)]),
        "16QAM": lambda: _qam(16), "32QAM": lambda: _qam(32), "64QAM": lambda: _qam(64), "128QAM": lambda: _qam(128),
        "256QAM": lambda: _qam(256),
    }
    c = table[name]()
    return c / np.sqrt(np.mean(np.abs(c) ** 2))


def _frame(rng, name, n):
    if name == "GMSK":       # constant envelope, +-pi/2 phase steps smoothed over 3 symbols (sps = 1 approximation)
        bits = rng.integers(0, 2, n + 2) * 2 - 1
        step = np.convolve(bits, [0.25, 0.5, 0.25], mode="valid") * (np.pi / 2)
        return np.exp(1j * np.cumsum(step))
    if name == "OQPSK":      # I and Q change on alternate samples
        i = np.repeat(rng.integers(0, 2, n // 2 + 1) * 2 - 1, 2)[:n]
        q = np.repeat(rng.integers(0, 2, n // 2 + 1) * 2 - 1, 2)[1:n + 1]
        return (i + 1j * q) / np.sqrt(2)
    c = constellation(name)
    return c[rng.integers(0, len(c), n)]


def make_dataset(n_frames: int, seed: int = 42, classes=CLASSES, snrs_db=SNRS_DB, n_symbols: int = 1024):
    """X (N, n_symbols, 2) float32, Y (N,) int64, Z (N,) float32 -- balanced over (class, SNR)."""
    rng = np.random.default_rng(seed)
    X = np.empty((n_frames, n_symbols, 2), np.float32)
    Y = np.empty(n_frames, np.int64)
    Z = np.empty(n_frames, np.float32)
    for i in range(n_frames):
        k = i % len(classes)
        snr = snrs_db[(i // len(classes)) % len(snrs_db)]
        s = _frame(rng, classes[k], n_symbols) * np.exp(1j * rng.uniform(0, 2 * np.pi))
        s = s / np.sqrt(np.mean(np.abs(s) ** 2) + 1e-12)
        sigma = np.sqrt(0.5 * 10 ** (-snr / 10))
        s = s + sigma * (rng.standard_normal(n_symbols) + 1j * rng.standard_normal(n_symbols))
        X[i, :, 0], X[i, :, 1] = s.real, s.imag
        Y[i], Z[i] = k, snr
    perm = rng.permutation(n_frames)
    return X[perm], Y[perm], Z[perm]


def zscore_stats(X: np.ndarray, n_subset: int = 5000, seed: int = 49):
    rng = np.random.default_rng(seed)
    idx = rng.choice(len(X), size=min(n_subset, len(X)), replace=False)
    sub = X[idx]
    return sub.mean(axis=(0, 1)), sub.std(axis=(0, 1)) + 1e-8


def to_vit_images(X: np.ndarray, mean, std, h: int = 32, w: int = 64) -> np.ndarray:
    """(N,1024,2) -> (N,1,h,w): [I ; Q] concatenated then viewed as an image (reference layout 32x64);
    for h*w < 2048 the first h*w/2 I and Q samples are used (BASELINE configs[0]: 32x32)."""
    Xn = (X - mean) / std
    half = h * w // 2
    img = np.concatenate([Xn[:, :half, 0], Xn[:, :half, 1]], axis=1)
    return img.reshape(len(X), 1, h, w).astype(np.float32)


def to_rawiq(X: np.ndarray, mean, std) -> np.ndarray:
    """(N,1024,2) -> (N,2,1024)."""
    return np.ascontiguousarray(((X - mean) / std).transpose(0, 2, 1)).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------
# Input pipeline with the reference's semantics
# ----------------------------------------------------------------------------------------------------------------
def split_indices(labels, snrs, target_mods, train_ratio: float = 0.7, valid_ratio: float = 0.15, test_ratio: float = 0.15,
                  seed: int = 49):
    """(train, valid, test) index arrays + label map, stratified by (modulation, SNR) cell.

    Same procedure as `split_data` (V/dataloader/utils.py:58-148): per cell, sklearn `train_test_split` peels off the test
    share, then the validation share of the remainder, both with `random_state=seed`; the three lists are shuffled at the
    end with one `np.random.seed(seed)` stream.  `labels` holds modulation names (or ints with `target_mods` = ints)."""
    from sklearn.model_selection import train_test_split
    if not np.isclose(train_ratio + valid_ratio + test_ratio, 1.0):
        raise ValueError("Ratios must sum to 1.0")
    labels = np.asarray(labels)
    snrs = np.asarray(snrs)
    label_map = {mod: i for i, mod in enumerate(target_mods)}
    rel_valid = valid_ratio / (train_ratio + valid_ratio)
    parts = ([], [], [])
    for mod in target_mods:
        in_mod = labels == mod
        for snr in np.unique(snrs):
            cell = np.where(in_mod & (snrs == snr))[0]
            if len(cell) == 0:
                continue
            rest, test = train_test_split(cell, test_size=test_ratio, random_state=seed, shuffle=True)
            if len(rest) > 1:
                train, valid = train_test_split(rest, test_size=rel_valid, random_state=seed, shuffle=True)
            else:
                train, valid = rest, []
            for dst, src in zip(parts, (train, valid, test)):
                dst.extend(src)
    np.random.seed(seed)
    for lst in parts:
        np.random.shuffle(lst)
    return np.array(parts[0]), np.array(parts[1]), np.array(parts[2]), label_map


def normalization_stats(X: np.ndarray, indices, seed: int = 49, n_subset: int = 5000):
    """{'i_mean','i_std','q_mean','q_std'} over a seeded subset of the training indices, as `_calculate_normalization_stats`
    (V/dataloader/dataset.py:116-158): `np.random.seed(seed)` + `np.random.choice(indices, n, replace=False)`, fp32 values,
    torch's unbiased std, std floored at 1e-8."""
    import torch
    indices = np.asarray(indices)
    n = min(n_subset, len(indices))
    np.random.seed(seed)
    pick = np.sort(np.random.choice(indices, n, replace=False))
    sub = torch.from_numpy(np.ascontiguousarray(X[pick])).float()
    i_all, q_all = sub[:, :, 0].flatten(), sub[:, :, 1].flatten()
    return {"i_mean": i_all.mean().item(), "i_std": max(i_all.std().item(), 1e-8),
            "q_mean": q_all.mean().item(), "q_std": max(q_all.std().item(), 1e-8)}


def preprocess_reference(frames: np.ndarray, stats: dict, layout: str, h: int = 32, w: int = 64) -> np.ndarray:
    """CPU statement of __getitem__'s arithmetic for a batch (fp32 throughout), used by tests and the CPU oracle legs."""
    import torch
    x = torch.from_numpy(np.ascontiguousarray(frames)).float().clone()
    x[:, :, 0] = (x[:, :, 0] - stats["i_mean"]) / stats["i_std"]
    x[:, :, 1] = (x[:, :, 1] - stats["q_mean"]) / stats["q_std"]
    if layout == "rawiq":
        return x.transpose(1, 2).contiguous().numpy()
    take = h * w // 2
    return torch.cat((x[:, :take, 0], x[:, :take, 1]), dim=1).view(len(frames), 1, h, w).numpy()


class DeviceInputPipeline:
    """Raw I/Q frames -> model input on the GPU.

    The reference normalises and re-lays-out every frame on CPU worker processes and ships the result
    (`pin_memory=True`, `.to(device, non_blocking=True)`, V/training/train.py:351-364,189-190).  Here the RAW
    `(B, len, 2)` fp32 frames go into one of two pinned staging buffers, cross PCIe on a copy stream while the previous
    step computes, and `iq_frames_preprocess` produces the `(B,1,H,W)` ViT image or the `(B,2,len)` raw-IQ tensor on
    the compute stream.  No CPU fallback: construction fails without the HIP library / a GPU."""

    def __init__(self, stats: dict, layout: str, batch: int, length: int = 1024, h: int = 32, w: int = 64, device="cuda"):
        import ctypes as C
        import torch
        from . import _native as N
        if layout not in ("vit", "rawiq"):
            raise ValueError(f"unknown layout: {layout}")
        self._N, self._L, self._torch = N, N.lib(), torch
        self.layout, self.length, self.h, self.w = layout, length, h, w
        self.take = length if layout == "rawiq" else h * w // 2
        if self.take > length:
            raise ValueError(f"image {h}x{w} needs {self.take} samples per channel, frames have {length}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.IqError("DeviceInputPipeline needs a CUDA/HIP device (no CPU fallback)")
        self._stats = (C.c_float * 4)(stats["i_mean"], stats["i_std"], stats["q_mean"], stats["q_std"])
        self._host = [torch.empty(batch, length, 2, dtype=torch.float32).pin_memory() for _ in range(2)]
        self._dev = [torch.empty(batch, length, 2, dtype=torch.float32, device=self.device) for _ in range(2)]
        self._ready = [torch.cuda.Event(), torch.cuda.Event()]
        self._free = [torch.cuda.Event(), torch.cuda.Event()]
        self._copy = torch.cuda.Stream(device=self.device)
        self._slot = 0
        self._pending = None

    def submit(self, frames: np.ndarray):
        """Start moving a batch of raw frames to the GPU (returns immediately)."""
        torch = self._torch
        s = self._slot
        n = len(frames)
        self._free[s].synchronize()                       # the preprocess that last read this slot has been issued and run
        self._host[s][:n].copy_(torch.from_numpy(np.ascontiguousarray(frames, dtype=np.float32)))
        with torch.cuda.stream(self._copy):
            self._dev[s][:n].copy_(self._host[s][:n], non_blocking=True)
            self._ready[s].record(self._copy)
        self._pending = (s, n)
        self._slot ^= 1

    def get(self):
        """The submitted batch as the model's input tensor (on the current stream)."""
        torch = self._torch
        if self._pending is None:
            raise RuntimeError("get() without a submitted batch")
        s, n = self._pending
        self._pending = None
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._ready[s])
        out = torch.empty(n, 2, self.take, dtype=torch.float32, device=self.device)
        self._N.check(self._L.iq_frames_preprocess(self._dev[s].data_ptr(), out.data_ptr(), n, self.length, self.take,
                                                   self._stats, cur.cuda_stream), "iq_frames_preprocess")
        self._free[s].record(cur)
        return out.view(n, 1, self.h, self.w) if self.layout == "vit" else out

    def __call__(self, frames: np.ndarray):
        self.submit(frames)
        return self.get()


# ----------------------------------------------------------------------------------------------------------------
# The accuracy-reproduction task of SURVEY 8(d) metric (2) / BASELINE.json "top-1 accuracy is reproduced on the same
# synthetic IQ set": one definition shared by the GPU run (accuracy.py), the CPU oracle run (oracle/accuracy_oracle.py),
# bench.py's `accuracy` object and tests/test_gpu_trainer.py.
# ----------------------------------------------------------------------------------------------------------------
ACCURACY_TASKS = {
    # cfg A (BASELINE configs[0]; geometry of V/test_model.py:26-55): ViT 32x32, patch 16, 11 classes, D128/H8/L2/F512
    "vit_A": dict(kind="vit", wd=1e-3,
                  kw=dict(in_channels=1, img_size_h=32, img_size_w=32, patch_size=16, num_classes=11, d_model=128,
                          n_head=8, n_layers=2, ffn_hidden=512)),
    # the raw-IQ geometry of R/test_model.py:91-114: 2x1024, segment 64, cls token, 11 classes, D128/H8/L2/F512
    "rawiq_R": dict(kind="rawiq", wd=1e-4,
                    kw=dict(in_channels=2, seq_length=1024, num_classes=11, d_model=128, n_head=8, n_layers=2,
                            ffn_hidden=512, use_cls_token=True, embedding_type="segment", segment_size=64)),
}
ACCURACY_HYPER = dict(n_frames=1000, n_train=800, data_seed=42, init_seed=11, batch=100, steps=240, lr=1e-3, drop_prob=0.1,
                      fresh_frames=4004, fresh_seed=43)


def accuracy_task(name: str):
    """N = 1000 frames (800 train / 200 held out), seed 42, the first 11 class names, SNR in {-8, 0, 8, 20} dB,
    z-score from the frames themselves; the ViT image is the first 512 I | first 512 Q samples viewed 32x32
    (`to_vit_images`), the raw-IQ input the transposed frame.  A second held-out set of 4004 FRESH frames (seed 43, same
    statistics) shrinks the sampling error of the accuracy estimate from +-0.028 (200 frames) to +-0.006.
    Returns dict(kind, kw, wd, xtr, ytr, xte, yte, xfresh, yfresh, hyper, chance) of torch CPU tensors."""
    import torch
    t = ACCURACY_TASKS[name]
    h = ACCURACY_HYPER
    classes = CLASSES[: t["kw"]["num_classes"]]
    X, Y, _ = make_dataset(h["n_frames"], seed=h["data_seed"], classes=classes)
    mean, std = zscore_stats(X)
    XF, YF, _ = make_dataset(h["fresh_frames"], seed=h["fresh_seed"], classes=classes)
    conv = (lambda A: to_vit_images(A, mean, std, 32, 32)) if t["kind"] == "vit" else (lambda A: to_rawiq(A, mean, std))
    R, RF = torch.from_numpy(conv(X)), torch.from_numpy(conv(XF))
    Yt, YFt = torch.from_numpy(Y), torch.from_numpy(YF)
    n = h["n_train"]
    return dict(name=name, kind=t["kind"], kw=dict(t["kw"]), wd=t["wd"], xtr=R[:n], ytr=Yt[:n], xte=R[n:], yte=Yt[n:],
                xfresh=RF, yfresh=YFt, hyper=dict(h), chance=1.0 / len(classes))
