"""pytest wiring: `gpu` marker, import paths, fixture loaders.

`-m "not gpu"` runs here (no GPU); `-m gpu` runs on an MI355X box where
/root/reference does not exist.  Nothing in tests/ reads the reference at run time.
"""
import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and not f.startswith(("param_counts", "sublayers")))


def load_golden(name):
    """-> (kind, ctor kwargs dict, npz)"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return str(z["cfg_kind"]), ast.literal_eval(str(z["cfg_json"])), z


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
