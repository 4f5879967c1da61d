import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_err(a, b):
    """Tensor-level relative error used for the fp32 parity bar: max|a-b| / max|b| (north_star: 1e-4 rel)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


# fp32 parity tolerance stated by BASELINE.json north_star ("fp32 embeddings within 1e-4 rel")
RTOL = 1e-4


@pytest.fixture(scope='session')
def ml100k():
    """ml-100k training pairs as internal ids (first-seen order, util/DataLoader.py:33-40) + sizes."""
    g = golden('g1_sampler.npz')
    U, I, nnz = (int(x) for x in g['sizes'])
    return dict(U=U, I=I, nnz=nnz, pairs0=g['pairs0'].copy())


@pytest.fixture(autouse=True)
def _deterministic_torch_rng():
    """Every test starts from the same torch RNG state (CPU and GPU generators): tensors drawn without an explicit generator are reproducible,
    so a tolerance that holds holds on every run."""
    import torch
    torch.manual_seed(20260)
    yield
