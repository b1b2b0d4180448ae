import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def report_close(name, got, ref, atol, rtol=0.0):
    """assert |got-ref| <= atol + rtol*|ref| elementwise with a useful failure message."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    assert np.isfinite(got).all(), f"{name}: non-finite values in result ({np.sum(~np.isfinite(got))})"
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    bad = err > tol
    if bad.any():
        idx = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(
            f"{name}: {bad.sum()}/{bad.size} elements out of tolerance; worst at {idx}: got {got[idx]!r} ref {ref[idx]!r} "
            f"err {err[idx]:.3e} tol {tol[idx] if np.ndim(tol) else tol:.3e}; max|ref|={np.abs(ref).max():.3e} "
            f"max err {err.max():.3e} mean err {err.mean():.3e}")
    return float(err.max())
