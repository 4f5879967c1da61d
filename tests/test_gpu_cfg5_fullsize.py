"""Parity at BASELINE.json config 5's real size on ONE MI355X: SYN-v1 10 M users x 1 M items (~3.2e8 interactions, 6.4e8 directed edges),
NGCF d = 128, L = 3 (recommender/NGCF.py:197-212) + a DL_Attack-style masked top-50 pass (attack/White/DLAttack.py:70-115).  The oracle cannot
run this size in seconds, so the checks are (a) two independent schedules against each other on the full graph, (b) identities that hold for
any graph, (c) the CPU oracle on a few thousand sampled rows / users.  At N d 4 B = 5.6 GB per table every 32-bit byte offset in a kernel shows
up here.  Timings and peak memory go to gpurun_out/r03_cfg5.json (copied to profiles/ by hand).

  * d = 128 hop: register-blocked plan (two d = 64 column-half passes) == CSR row kernel, max-norm and row-wise, + bit-identical rerun
  * fixed point A_hat D^1/2 1 = D^1/2 1 through the blocked plan
  * 4 096 sampled output rows (incl. the longest user and item rows) against oracle.spmm on the sub-CSR of those rows, edge values recomputed
    from host degrees
  * one engine.step_ngcf (fused route) == the autograd route (NGCF_Encoder.forward_rows + backward) on table and weight gradients
  * masked top-50 of a 1 M-user slice vs dense fp32 scoring on 512 sampled users, no interacted item in ANY of the 1 M lists
"""
import json
import os
import time
from types import SimpleNamespace
import numpy as np
import pytest
import torch
from conftest import row_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
U, I, D, L, B = 10_000_000, 1_000_000, 128, 3, 2048
REPORT = {}


def _timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


@pytest.fixture(scope='module')
def cfg5():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    from arlib_amd import ops
    from arlib_amd.util import synthetic
    t0 = time.perf_counter()
    pairs = synthetic.syn_v1_pairs_native(U, I, 32.0, 2018)
    nnz = len(pairs)
    REPORT['generate_pairs_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    u = torch.from_numpy(pairs[:, 0].astype(np.int64)).to(DEV); i = torch.from_numpy(pairs[:, 1].astype(np.int64)).to(DEV)
    A = ops.bipartite_graph(u, i, U, I)                                           # CSR schedule only
    del u, i
    torch.cuda.synchronize()
    REPORT['device_graph_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    Ab = object.__new__(ops.CSRGraph); Ab.__dict__.update(A.__dict__)
    Ab.enable_blocked(split=U)                                                    # the schedule config 5 runs on: explicit, independent index arrays
    torch.cuda.synchronize()
    REPORT['blocked_plan_s'] = time.perf_counter() - t0
    assert A.blocked is None and Ab.blocked is not None and A.nnz == 2 * nnz
    REPORT.update(users=U, items=I, nnz=nnz, edges=A.nnz, d=D, layers=L,
                  plan=[dict(waves=s['n_waves'], edges=s['n_edges'], split_rows=s['n_split']) for s in Ab.blocked.sets])
    yield dict(ops=ops, pairs=pairs, nnz=nnz, A=A, Ab=Ab)
    REPORT['peak_memory_gb'] = torch.cuda.max_memory_allocated() / 1e9
    out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'r03_cfg5.json'), 'w') as f:
            json.dump(REPORT, f, indent=1)
    except OSError:
        pass


def test_cfg5_graph_size_and_digest(cfg5):
    from arlib_amd.util import synthetic as S
    assert 3.0e8 < cfg5['nnz'] < 3.4e8
    p = cfg5['pairs']
    assert int(p[:, 0].max()) == U - 1 and int(p[:, 1].max()) == I - 1 and bool((np.diff(p[:, 0].astype(np.int64)) >= 0).all())
    REPORT['graph_digest'] = '%016x' % S.graph_digest_native(p)
    # every edge is a record of the plan, every row planned
    plan = cfg5['Ab'].blocked
    assert plan.n_hub == 0 and sum(s['n_edges'] for s in plan.sets) == cfg5['A'].nnz and sum(s['n_rows'] for s in plan.sets) == U + I


def test_cfg5_blocked_hop_equals_csr_hop_d128(cfg5):
    ops, A, Ab = cfg5['ops'], cfg5['A'], cfg5['Ab']
    N = U + I
    g = torch.Generator(device=DEV).manual_seed(5)
    X = torch.randn(N, D, device=DEV, generator=g)
    yc = ops.spmm(A, X)
    yb = ops.spmm(Ab, X)
    num = (yb - yc).norm(dim=1); den = yc.norm(dim=1)
    rown = num / den.clamp_min(1e-3 * float(den.max()))
    assert float((yb - yc).abs().max() / yc.abs().max()) < 1e-5
    assert float(rown.max()) < 1e-4
    assert torch.equal(yb, ops.spmm(Ab, X))                                       # deterministic
    # epilogues of the step at this size: AXPBY with Z, fused Adam (the last backward hop of step_ngcf)
    Z = torch.randn(N, D, device=DEV, generator=g)
    zc = ops.spmm(A, X, 0.5, 0.25, Z); zb = ops.spmm(Ab, X, 0.5, 0.25, Z)
    assert float((zb - zc).abs().max() / zc.abs().max()) < 1e-5
    del zc, zb
    # (moments well away from zero: with v ~ eps^2 Adam's update g / (|g| + eps) turns rounding differences of g into whole steps)
    P = torch.randn(N, D, device=DEV, generator=g) * 0.1; M = torch.randn(N, D, device=DEV, generator=g) * 0.01
    V = torch.rand(N, D, device=DEV, generator=g) * 1e-4 + 1e-6
    Pc, Mc, Vc = P.clone(), M.clone(), V.clone()
    Pb, Mb, Vb = P, M, V
    ops.spmm_adam(A, X, 1.0, 1.0, Z, Pc, Mc, Vc, 0.005, 7)
    ops.spmm_adam(Ab, X, 1.0, 1.0, Z, Pb, Mb, Vb, 0.005, 7)
    for xb, xc in ((Pb, Pc), (Mb, Mc), (Vb, Vc)):
        assert float((xb - xc).abs().max() / xc.abs().max()) < 1e-5
    del Pc, Mc, Vc, Pb, Mb, Vb, P, M, V, Z
    REPORT['hop_blocked_ms'] = 1e3 * _timed(lambda: ops.spmm(Ab, X, out=yb))
    REPORT['hop_csr_ms'] = 1e3 * _timed(lambda: ops.spmm(A, X, out=yc))
    REPORT['hop_algorithmic_GBps'] = A.spmm_bytes(D) / (REPORT['hop_blocked_ms'] * 1e-3) / 1e9


def test_cfg5_fixed_point_through_blocked_plan(cfg5):
    ops, A, Ab = cfg5['ops'], cfg5['A'], cfg5['Ab']
    deg = (A.rowptr[1:] - A.rowptr[:-1]).float()
    assert int(deg.max()) > 200_000                                               # the hottest item row: hundreds of thousands of edges, dealt as pieces
    x = torch.sqrt(deg)[:, None].repeat(1, D).contiguous()
    y = ops.spmm(Ab, x)
    assert float((y - x).abs().max() / x.abs().max()) < 1e-4                       # all-positive sums: the worst case of a sequential fp32 chain
    assert float(((y - x).norm(dim=1) / x.norm(dim=1).clamp_min(1.0)).max()) < 1e-4


def test_cfg5_sampled_rows_against_the_oracle(cfg5):
    """4 096 output rows of one d = 128 hop against oracle.spmm (float64 accumulation) on the sub-CSR of exactly those rows; the rows' patterns
    come from the HOST pair list and their edge values 1/sqrt(deg_r deg_c) from host degrees -- nothing of the device CSR is trusted."""
    from oracle import oracle as O
    O.build()
    ops, Ab, pairs, nnz = cfg5['ops'], cfg5['Ab'], cfg5['pairs'], cfg5['nnz']
    N = U + I
    du = np.bincount(pairs[:, 0], minlength=U).astype(np.int64); di = np.bincount(pairs[:, 1], minlength=I).astype(np.int64)
    rng = np.random.default_rng(5)
    users = np.unique(np.concatenate([rng.choice(U, 2040, replace=False), np.argsort(du)[-8:]]))
    items = np.unique(np.concatenate([rng.choice(I, 2040, replace=False), np.argsort(di)[-8:]]))
    ptr_u = np.zeros(U + 1, np.int64); np.cumsum(du, out=ptr_u[1:])
    # user rows: columns = U + items of the user (pairs are user-major)
    rows, cols = [], []
    for k, u in enumerate(users):
        it = pairs[ptr_u[u]:ptr_u[u + 1], 1].astype(np.int64)
        rows.append(np.full(len(it), k, np.int64)); cols.append(U + it)
    # item rows: columns = users of the item, from one pass over the pair list
    sel = np.nonzero(np.isin(pairs[:, 1], items.astype(np.int32)))[0]
    sub = pairs[sel]
    o = np.lexsort((sub[:, 0], sub[:, 1]))
    sub = sub[o]
    rows.append(len(users) + np.searchsorted(items, sub[:, 1].astype(np.int64))); cols.append(sub[:, 0].astype(np.int64))
    rows = np.concatenate(rows); cols = np.concatenate(cols)
    node = np.concatenate([users, U + items])
    deg = np.concatenate([du, di])
    assert np.array_equal(np.bincount(rows, minlength=len(node)), deg[node])       # host pattern has every edge of the sampled rows
    val = (1.0 / np.sqrt(deg[node][rows].astype(np.float64) * deg[cols].astype(np.float64))).astype(np.float32)
    ucols, inv = np.unique(cols, return_inverse=True)
    rp = np.zeros(len(node) + 1, np.int64); np.cumsum(np.bincount(rows, minlength=len(node)), out=rp[1:])
    g = torch.Generator(device=DEV).manual_seed(9)
    X = torch.randn(N, D, device=DEV, generator=g)
    Y = ops.spmm(Ab, X)
    Xs = X[torch.from_numpy(ucols).to(DEV)].cpu().numpy()
    ref = O.spmm((rp, inv.astype(np.int32), val), Xs)
    got = Y[torch.from_numpy(node).to(DEV)].cpu().numpy()
    assert float(np.abs(got - ref).max() / np.abs(ref).max()) < 1e-5
    assert row_err(got, ref) < 1e-4
    # the device CSR's values on those rows equal the host-degree values
    Av, Ar = cfg5['A'].val, cfg5['A'].rowptr
    for k in (0, len(users) - 1, len(users), len(node) - 1):
        r = int(node[k]); b, e = int(Ar[r]), int(Ar[r + 1])
        assert e - b == rp[k + 1] - rp[k]
        assert np.allclose(np.sort(Av[b:e].cpu().numpy()), np.sort(val[rp[k]:rp[k + 1]]), rtol=1e-6, atol=0)        # (dinv_r w) dinv_c in fp32 on the device vs one rounding of the float64 value
    REPORT['oracle_rows'] = dict(rows=int(len(node)), edges=int(len(rows)), longest_row=int(deg[node].max()))


def _encoder(ops, A, table, weights):
    """NGCF_Encoder over an existing device graph / packed table (no DataLoader: the reference's dict-based loader cannot exist at this size)."""
    from arlib_amd.recommender._base import SparseNormAdj
    from arlib_amd.recommender.NGCF import NGCF_Encoder
    enc = NGCF_Encoder.__new__(NGCF_Encoder)
    torch.nn.Module.__init__(enc)
    enc.data = SimpleNamespace(user_num=U, item_num=I)
    enc.latent_size = enc.emb_size = D
    enc.layers = enc.n_prop_layers = L
    enc._eng = None
    enc.embedding_dict = torch.nn.ParameterDict({'user_emb': torch.nn.Parameter(table[:U]), 'item_emb': torch.nn.Parameter(table[U:])})
    enc.W = torch.nn.ParameterDict({n + str(k): torch.nn.Parameter(weights[k][j].clone()) for k in range(L) for j, n in enumerate(('w1_', 'w2_'))})
    adj = SparseNormAdj.__new__(SparseNormAdj)
    adj.shape, adj._indptr, adj._indices, adj.values, adj._graph = (U + I, U + I), None, None, A.val, A
    enc.sparse_norm_adj = adj
    return enc


def test_cfg5_fused_ngcf_step_equals_autograd_route(cfg5):
    """engine.step_ngcf (what NGCF.train() runs) against the autograd route on the same batch at 10 M x 1 M, d = 128, L = 3: loss, table gradient
    (max-norm and row-wise over the rows that carry gradient), weight gradients; then the fused Adam update against torch.optim.Adam."""
    from arlib_amd import engine
    from arlib_amd.util.loss import bpr_loss, l2_reg_loss
    ops, Ab, pairs, nnz = cfg5['ops'], cfg5['Ab'], cfg5['pairs'], cfg5['nnz']
    torch.manual_seed(2018)
    table = torch.empty(U + I, D, device=DEV).uniform_(-0.05, 0.05)
    weights = [tuple(torch.nn.init.xavier_uniform_(torch.empty(D, D)).to(DEV) for _ in range(2)) for _ in range(L)]
    gen = torch.Generator().manual_seed(1)
    sel = torch.randint(0, nnz, (B,), generator=gen).numpy()
    bu = torch.from_numpy(pairs[sel, 0].astype(np.int32)).to(DEV); bp = torch.from_numpy(pairs[sel, 1].astype(np.int32)).to(DEV)
    bn = torch.randint(0, I, (B,), generator=gen).to(torch.int32).to(DEV)
    rows = torch.cat([bu, bp + U, bn + U])
    # autograd route
    enc = _encoder(ops, Ab, table.clone(), weights)
    opt = torch.optim.Adam(enc.parameters(), lr=0.005)

    def auto_step():
        o = enc.forward_rows(rows)
        loss = bpr_loss(o[:B], o[B:2 * B], o[2 * B:]) + l2_reg_loss(1e-4, o[:B], o[B:2 * B])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        return loss
    loss_a = float(auto_step().detach())
    g_auto = torch.cat([enc.embedding_dict['user_emb'].grad, enc.embedding_dict['item_emb'].grad], 0)
    gw_auto = [torch.cat([enc.W['w1_%d' % k].grad, enc.W['w2_%d' % k].grad], 0) for k in range(L)]
    opt.step()
    after_auto = torch.cat([enc.embedding_dict['user_emb'].data, enc.embedding_dict['item_emb'].data], 0)
    # fused route
    eng = engine.PropagationEngine(Ab, U, I, D, L, 1e-4, 0.005, DEV, table=table.clone())
    Wf = [(a.clone(), b.clone()) for a, b in weights]
    eng.init_ngcf(Wf)
    cap = {}
    lo = eng.step_ngcf(bu, bp, bn, capture=cap).cpu().numpy()
    assert abs(float(lo[0] + lo[1]) - loss_a) <= 1e-5 * abs(loss_a)
    gf = cap['table']
    assert float((gf - g_auto).abs().max() / g_auto.abs().max()) < 1e-4
    nrm = g_auto.norm(dim=1)
    rown = (gf - g_auto).norm(dim=1) / nrm.clamp_min(1e-3 * float(nrm.max()))
    assert float(rown.max()) < 1e-4
    for k in range(L):
        assert float((cap['W'][k] - gw_auto[k]).abs().max() / gw_auto[k].abs().max()) < 1e-4
    # Adam: the bulk (|g| well above eps) must move exactly alike; near |g| ~ eps the update is lr * g / (|g| + eps): bounded by lr
    d = (eng.E0 - after_auto).abs()
    well = g_auto.abs() > 1e-6
    assert float(d[well].max()) < 1e-4 * 0.005 * 4 and float(d.max()) <= 0.005 * 1.001
    assert float((Wf[0][0] - enc.W['w1_0'].data).abs().max()) < 1e-4 * 0.005 * 4
    REPORT['grad_nonzero_rows'] = int((nrm > 0).sum())
    del gf, g_auto, after_auto, d, well, cap
    torch.cuda.empty_cache()
    t = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.step_ngcf(bu, bp, bn)
        torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    REPORT['ngcf_fused_step_ms'] = 1e3 * float(np.median(t[1:]))
    t = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        auto_step(); opt.step()
        torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    REPORT['ngcf_autograd_step_ms'] = 1e3 * float(np.median(t[1:]))
    a = eng.E0.clone()                                                             # run-to-run: the step is bit-reproducible at this size too
    e2 = engine.PropagationEngine(Ab, U, I, D, L, 1e-4, 0.005, DEV, table=table.clone())
    e2.init_ngcf([(x.clone(), y.clone()) for x, y in weights])
    for _ in range(5):
        e2.step_ngcf(bu, bp, bn)
    assert torch.equal(e2.E0, a)


def test_cfg5_masked_top50_slice_against_dense_rows(cfg5):
    ops, A, nnz = cfg5['ops'], cfg5['A'], cfg5['nnz']
    n_slice = 1_000_000
    torch.manual_seed(3)
    X = torch.empty(U + I, D, device=DEV).uniform_(-0.05, 0.05)
    X = ops.spmm(cfg5['Ab'], X)                                                    # propagated tables: popularity-skewed scores
    Pu, Pi = X[:n_slice].contiguous(), X[U:].contiguous()
    del X
    rp = A.rowptr[:n_slice + 1].contiguous()
    mc = (A.col[:int(rp[-1])] - U).contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, val = ops.score_mask_topk(Pu, Pi, 50, rp, mc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    REPORT['masked_top50_slice'] = dict(users=n_slice, items=I, seconds=dt, tflops_fp32_equivalent=2.0 * n_slice * I * D / dt / 1e12)
    assert bool((val[:, :-1] >= val[:, 1:]).all())
    assert int(idx.min()) >= 0 and int(idx.max()) < I
    keys = (torch.arange(n_slice, device=DEV, dtype=torch.int64)[:, None] * I + idx.long()).flatten()
    inter = torch.repeat_interleave(torch.arange(n_slice, device=DEV, dtype=torch.int64), (rp[1:] - rp[:-1]).long()) * I + mc.long()
    pos = torch.searchsorted(inter, keys).clamp_(max=inter.numel() - 1)
    assert not bool((inter[pos] == keys).any())                                   # no interacted item in any of the 1 M lists
    del keys, inter, pos
    sample = torch.from_numpy(np.random.default_rng(0).choice(n_slice, 512, replace=False)).to(DEV)
    sc = Pu[sample].double() @ Pi.double().T                                       # dense reference in float64 (a library fp32 GEMM may split k)
    for r, u in enumerate(sample.tolist()):
        sc[r, mc[rp[u]:rp[u + 1]].long()] = -10e8
    rv, ri = torch.topk(sc, 50)
    assert (ri == idx[sample].long()).float().mean().item() > 0.999               # ties / last-ulp orderings aside
    assert torch.allclose(rv.float(), val[sample], rtol=1e-5, atol=1e-7)
