import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Without a GPU every ``gpu``-marked test is reported as skipped (a plain ``pytest`` run on a CPU box stays green)."""
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="needs the GPU box (MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden(name):
    """Fixture arrays by name.  Inputs stored as float32 (the config-2 cases: the reference ran on their exact
    float64 upcasts) come back as float64."""
    g = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
    return {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in g.items()}


def rank_deficient(g):
    """The reference's Wq of this fixture is numerically singular (fewer datapoints than latents): W_new is then
    defined only up to the SVD cutoff of lstsq, and the captured statistics Wq / Wp are compared instead."""
    return "Wq_rank_ratio" in g and float(g["Wq_rank_ratio"]) < 1e-13


def bsc_step_cases():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "bsc_step_*.npz")))


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session", autouse=True)
def _library_built():
    """The .so is git-ignored: on a fresh checkout build it once (hipcc cross-compiles gfx950 without a GPU),
    exactly what ``__graft_entry__.build()`` does.  A box without hipcc and without the library fails loudly in
    the tests that need it."""
    lib = os.path.join(ROOT, "prosper_amd", "libprosper_hip.so")
    script = os.path.join(ROOT, "prosper_amd", "csrc", "build.sh")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(lib) and os.path.exists(hipcc):
        import subprocess
        subprocess.run(["bash", script], check=True, capture_output=True)
    yield


@pytest.fixture(scope="session", autouse=True)
def _poisoned_empty():
    """PM_POISON_EMPTY=1: every ``torch.empty`` / ``empty_like`` on the GPU comes back filled with NaN (floats) or 0x5A bytes
    (integers) instead of whatever the caching allocator hands out -- in a long pytest session that is usually a plausible old
    result of the same shape, which hides a kernel that reads a buffer it was supposed to fill.  A diagnostic run, not the default
    (it costs a fill per allocation)."""
    if os.environ.get("PM_POISON_EMPTY") != "1" or not has_gpu():
        yield
        return
    import torch
    orig_empty, orig_like = torch.empty, torch.empty_like

    def poison(t):
        if t.is_cuda and t.numel():
            if t.dtype.is_floating_point:
                t.fill_(float("nan"))
            elif t.dtype != torch.bool:
                t.view(torch.uint8).fill_(0x5A)
        return t

    torch.empty = lambda *a, **k: poison(orig_empty(*a, **k))
    torch.empty_like = lambda *a, **k: poison(orig_like(*a, **k))
    try:
        yield
    finally:
        torch.empty, torch.empty_like = orig_empty, orig_like
