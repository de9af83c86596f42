"""Seeded synthetic IQ frames with the field semantics of the reference's dataset.

The reference trains on RadioML 2018.01A (X (N,1024,2) float32, Y labels, Z SNR; README.md:226-231),
which is not shipped; it contains no generator.  The closest in-repo recipe is
Transformer_Thesis/test_sps_modes.py:10-27 (seeded unit-power QPSK at 1 sample/symbol + complex AWGN).
This module generalises that recipe to the 19 class names of ViT/training/train.py:60-80 so the
accuracy comparison of BASELINE.json ("top-1 accuracy reproduced on the same synthetic IQ set") has
a shared, deterministic data source for the GPU path and the CPU oracle.

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
