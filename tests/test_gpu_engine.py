"""GPU parity: the fused training engine against the golden training runs captured from the reference
(LightGCN/GMF Adam, LightGCN SGD, SimGCL forward) and against the oracle on a synthetic graph."""
import numpy as np
import pytest
import torch
from conftest import golden, rel_err, RTOL
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope='module')
def mods():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    from arlib_amd import ops, engine
    return ops, engine


def ml_graph(ops, ml100k):
    p = ml100k['pairs0']
    rowptr, col, w = O.bipartite_csr(p[:, 0], p[:, 1], ml100k['U'], ml100k['I'])
    val, _ = ops.norm_adj_values(T(rowptr.astype(np.int32)), T(col), T(w), len(rowptr) - 1)     # device normalisation
    return ops.CSRGraph(rowptr, col, val, DEV)


def run_golden(mods, ml100k, g, L, lr, opt, snaps):
    ops, engine = mods
    U, I = ml100k['U'], ml100k['I']
    A = ml_graph(ops, ml100k) if L > 0 else None
    d = g['user0'].shape[1]
    eng = engine.PropagationEngine(A, U, I, d, L, 1e-4, lr, DEV, optimizer=opt, table=T(np.concatenate([g['user0'], g['item0']])))
    off = np.concatenate([[0], np.cumsum(g['batch_sizes'])])
    for k in range(len(g['batch_sizes'])):
        sl = slice(off[k], off[k + 1])
        u, p, n = T(g['batch_u'][sl]), T(g['batch_p'][sl]), T(g['batch_n'][sl])
        if k == 0:
            lo, grad = eng.grad(u, p, n)
            gr = grad.cpu().numpy()
            assert rel_err(gr[:U], g['grad_user_step0']) < RTOL and rel_err(gr[U:], g['grad_item_step0']) < RTOL
        lo = eng.step(u, p, n).cpu().numpy()
        assert abs(lo[0] + lo[1] - g['losses'][k]) <= RTOL * abs(g['losses'][k])
        if (k + 1) in snaps:
            E = eng.E0.cpu().numpy()
            assert rel_err(E[:U], g['user_k%d' % (k + 1)]) < RTOL and rel_err(E[U:], g['item_k%d' % (k + 1)]) < RTOL
    return eng


def test_lightgcn_adam_golden_10_steps(mods, ml100k):
    g = golden('g5_lightgcn_adam.npz')
    eng = run_golden(mods, ml100k, g, 3, 0.005, 'adam', {1, 3, 10})
    U = ml100k['U']
    assert rel_err(eng.m[:U].cpu().numpy(), g['m_user']) < RTOL and rel_err(eng.v[U:].cpu().numpy(), g['v_item']) < RTOL


def test_gmf_adam_golden_25_steps(mods, ml100k):
    g = golden('g5_gmf_adam.npz')
    eng = run_golden(mods, ml100k, g, 0, 0.005, 'adam', {3, 25})
    U = ml100k['U']
    assert rel_err(eng.m[U:].cpu().numpy(), g['m_item']) < RTOL and rel_err(eng.v[:U].cpu().numpy(), g['v_user']) < RTOL


def test_lightgcn_sgd_golden(mods, ml100k):
    run_golden(mods, ml100k, golden('g5_lightgcn_sgd.npz'), 2, 0.0005, 'sgd', {3})


@pytest.mark.parametrize('L', [1, 2, 3])
def test_lightgcn_forward_golden(mods, ml100k, L):
    ops, engine = mods
    g = golden('g4_forward.npz')
    U, I = ml100k['U'], ml100k['I']
    eng = engine.PropagationEngine(ml_graph(ops, ml100k), U, I, 32, L, 1e-4, 0.005, DEV, table=T(np.concatenate([g['lgn_user0'], g['lgn_item0']])))
    out = eng.forward().cpu().numpy()
    assert rel_err(out[:U], g['lgn_L%d_user' % L]) < RTOL and rel_err(out[U:], g['lgn_L%d_item' % L]) < RTOL


def test_simgcl_forward_and_backward_golden(mods, ml100k):
    ops, engine = mods
    g = golden('g5_simgcl.npz')
    U, I = ml100k['U'], ml100k['I']
    E0 = np.concatenate([g['user0'], g['item0']])
    eng = engine.PropagationEngine(ml_graph(ops, ml100k), U, I, 16, 2, 1e-4, 0.005, DEV, skip_layer0=True, table=T(E0))
    out = eng.forward().cpu().numpy()
    assert rel_err(out[:U], g['fwd_user']) < RTOL and rel_err(out[U:], g['fwd_item']) < RTOL
    outp = eng.forward(noises=[T(g['noise'][0]), T(g['noise'][1])], eps=0.1).cpu().numpy()
    assert rel_err(outp[:U], g['fwdp_user']) < RTOL and rel_err(outp[U:], g['fwdp_item']) < RTOL
    # backward of the skip-0 mean against the oracle's Horner restatement
    rng = np.random.default_rng(0)
    G = np.zeros_like(E0); rows = rng.integers(0, U + I, 500); G[rows] = rng.standard_normal((500, 16)).astype(np.float32)
    p = ml100k['pairs0']
    rowptr, col, w = O.bipartite_csr(p[:, 0], p[:, 1], U, I)
    csr = (rowptr, col, O.norm_adj_values(rowptr, col, w))
    ref = O.lightgcn_backward(csr, G, 2, skip0=True)
    got = eng.backward_to_table(T(G)).cpu().numpy()
    assert rel_err(got, ref) < RTOL


def test_engine_vs_oracle_synthetic_long_rows(mods):
    """Synthetic power-law graph with rows far longer than the chunk size, 5 Adam steps vs the oracle."""
    ops, engine = mods
    rng = np.random.default_rng(42)
    U, I, d, L, B = 20000, 2000, 64, 3, 2048
    deg = np.clip(np.round(np.exp(rng.normal(np.log(16) - 0.5, 1.0, U))), 2, 500).astype(np.int64)
    us = np.repeat(np.arange(U), deg)
    its = np.floor(I * rng.random(len(us)) ** 2).astype(np.int64)
    key = np.unique(us * I + its)
    us, its = (key // I).astype(np.int32), (key % I).astype(np.int32)
    rowptr, col, w = O.bipartite_csr(us, its, U, I)
    val = O.norm_adj_values(rowptr, col, w)
    assert np.diff(rowptr).max() > 2000
    A = ops.CSRGraph(rowptr, col, val, DEV)
    bound = np.sqrt(6.0 / (U + d))
    E0 = ((rng.random((U + I, d)) * 2 - 1) * bound).astype(np.float32)
    st = O.TrainState(E0[:U], E0[U:], (rowptr, col, val), L, 1e-4, 0.005)
    eng = engine.PropagationEngine(A, U, I, d, L, 1e-4, 0.005, DEV, table=T(E0))
    pairs = np.stack([us, its], 1)
    for k in range(5):
        sel = rng.integers(0, len(pairs), B)
        bu, bp = pairs[sel, 0].copy(), pairs[sel, 1].copy()
        bn = rng.integers(0, I, B).astype(np.int32)
        loss = st.step(bu, bp, bn)
        lo = eng.step(T(bu), T(bp), T(bn)).cpu().numpy()
        assert abs(lo[0] + lo[1] - loss) <= RTOL * abs(loss)
    assert rel_err(eng.E0.cpu().numpy(), st.E0) < RTOL
    assert rel_err(eng.m.cpu().numpy(), st.m) < RTOL and rel_err(eng.v.cpu().numpy(), st.v) < RTOL
