import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the GPU box shows 256 CPUs but grants a 16-CPU quota: an oversized intra-op pool gets the process throttled
    # (gan_ode_amd.limit_host_threads docstring); explicit here because importing the package no longer does it
    import gan_ode_amd
    gan_ode_amd.limit_host_threads()


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def load_sd(module, npz, prefix):
    """Load the arrays stored under '<prefix>/<state_dict key>' into module (strict)."""
    sd = {}
    for k in module.state_dict().keys():
        a = npz[f"{prefix}/{k}"]
        sd[k] = torch.from_numpy(np.asarray(a)).clone()
    module.load_state_dict(sd, strict=True)
    return module


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64).flatten()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64).flatten()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_weights_after_adam(v, w, key, frac=5e-3, min_count=4):
    """Post-Adam weights vs the oracle's: after 1-2 steps every entry has moved by ~lr = 2e-4 times a sign-like
    m/sqrt(v), so a wrong gradient shows as ~50 % of the entries off by 2*lr.  Entries whose gradient is within fp32
    rounding noise of zero may flip sign between two correct fp32 evaluations: a small share (and, for short
    vectors such as BatchNorm gamma, a handful of entries) is allowed."""
    d = (v.detach().cpu() - torch.as_tensor(w)).abs()
    flipped = int((d > 6e-5).sum())
    assert flipped <= max(min_count, frac * d.numel()), (key, flipped, d.numel(), float(d.max()))
    assert float(d.median()) < 2e-6, (key, float(d.median()))
