"""GPU parity tests: every HIP kernel (through the C ABI / ctypes) against the CPU oracle on seeded inputs and
against the golden fixtures captured from the reference.  Bar: fp32 within 1e-4 rel (north_star), ints bit-exact."""
import numpy as np
import pytest
import torch
from conftest import golden, rel_err, row_err, close, RTOL
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU (run with -m gpu on the MI355X box)')
    from arlib_amd import ops as _ops
    return _ops


DEV = 'cuda:0'


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def random_graph(rng, U, I, avg_deg, hot_items=0, hot_deg=0, empty_users=()):
    """Bipartite interaction list with power-law-ish item popularity, optional very hot items (long rows)
    and isolated users (empty rows)."""
    deg = np.clip(rng.poisson(avg_deg, U), 1, I)
    us = np.repeat(np.arange(U), deg)
    its = np.floor(I * rng.random(len(us)) ** 2).astype(np.int64)
    for h in range(hot_items):
        extra = rng.choice(U, size=min(hot_deg, U), replace=False)
        us = np.concatenate([us, extra]); its = np.concatenate([its, np.full(len(extra), h)])
    keep = ~np.isin(us, np.array(list(empty_users), dtype=np.int64))
    us, its = us[keep], its[keep]
    key = np.unique(us * I + its)
    return (key // I).astype(np.int32), (key % I).astype(np.int32)


def make_csr(u, i, U, I, w=None):
    rowptr, col, ww = O.bipartite_csr(u, i, U, I, w)
    val = O.norm_adj_values(rowptr, col, ww)
    return rowptr, col, ww, val


@pytest.mark.parametrize('d', [16, 32, 64, 128, 24])
def test_spmm_all_epilogues(ops, d):
    rng = np.random.default_rng(d)
    U, I = 3000, 700
    u, i = random_graph(rng, U, I, 12, hot_items=3, hot_deg=2500, empty_users=(5, 77))
    rowptr, col, w, val = make_csr(u, i, U, I)
    N = U + I
    assert np.diff(rowptr).max() > 4 * 512 and np.diff(rowptr).min() == 0      # long rows and empty rows present
    A = ops.CSRGraph(rowptr, col, val, DEV, chunk=512)
    assert A.n_long >= 3
    X = rng.standard_normal((N, d)).astype(np.float32)
    Z = rng.standard_normal((N, d)).astype(np.float32)
    csr = (rowptr, col, val)
    ref = O.spmm(csr, X)
    assert rel_err(ops.spmm(A, T(X)).cpu().numpy(), ref) < RTOL
    assert rel_err(ops.spmm(A, T(X), 0.25, 0.25, T(Z)).cpu().numpy(), O.spmm(csr, X, 0.25, 0.25, Z)) < RTOL
    # no-plan variant (every row whole) must agree too
    A0 = ops.CSRGraph(rowptr, col, val, DEV, chunk=0)
    assert A0.n_chunks == 0
    assert rel_err(ops.spmm(A0, T(X)).cpu().numpy(), ref) < RTOL
    # layer-sum epilogue
    S = T(Z.copy()); Y = torch.empty_like(S)
    ops.spmm_layersum(A, T(X), S, S, Y)
    assert rel_err(Y.cpu().numpy(), ref) < RTOL and rel_err(S.cpu().numpy(), Z + ref) < RTOL
    # fused Adam epilogue == oracle spmm + oracle adam
    P = rng.standard_normal((N, d)).astype(np.float32) * 0.1
    M = rng.standard_normal((N, d)).astype(np.float32) * 0.01
    V = (rng.random((N, d)).astype(np.float32)) * 1e-4
    g = O.spmm(csr, X, 0.25, 0.25, Z)
    Pr, Mr, Vr = P.copy(), M.copy(), V.copy()
    O.adam_step(Pr, g, Mr, Vr, 0.005, 7)
    Pt, Mt, Vt = T(P), T(M), T(V)
    ops.spmm_adam(A, T(X), 0.25, 0.25, T(Z), Pt, Mt, Vt, 0.005, 7)
    assert rel_err(Pt.cpu().numpy(), Pr) < RTOL and rel_err(Mt.cpu().numpy(), Mr) < RTOL and rel_err(Vt.cpu().numpy(), Vr) < RTOL


@pytest.mark.parametrize('rpw,split,hub,d,split_hubs', [(32, True, 64, 64, True), (32, True, 64, 64, False), (16, True, 24, 64, True), (32, False, 100000, 64, True),
                                                        (32, True, 64, 128, True), (16, False, 300, 128, False), (16, False, 300, 128, True)])
def test_spmm_blocked_schedule_all_epilogues(ops, rpw, split, hub, d, split_hubs):
    """Register-blocked schedule (arl_spmm_blocked_*) against the oracle: every epilogue, empty rows, rows above the hub threshold (dealt as
    strided pieces + combine pass with split_hubs, through the chunked kernel without), ragged last wave, and edge values replaced through
    with_values."""
    rng = np.random.default_rng(rpw + hub + d)
    U, I = 3001, 703
    u, i = random_graph(rng, U, I, 12, hot_items=3, hot_deg=2500, empty_users=(5, 77))
    rowptr, col, w, val = make_csr(u, i, U, I)
    N = U + I
    A = ops.CSRGraph(rowptr, col, val, DEV, chunk=512).enable_blocked(split=U if split else None, rows_per_wave=rpw, hub=hub, col_block=256, split_hubs=split_hubs)
    bp = A.blocked
    assert len(bp.sets) == (2 if split else 1) and (bp.n_hub > 0) == (hub < 2500 and not split_hubs)
    assert (sum(s['n_split'] for s in bp.sets) > 0) == (hub < 2500 and split_hubs)
    assert sum(s['n_rows'] for s in bp.sets) + bp.n_hub == N
    X = rng.standard_normal((N, d)).astype(np.float32)
    Z = rng.standard_normal((N, d)).astype(np.float32)
    csr = (rowptr, col, val)
    ref = O.spmm(csr, X)
    y = ops.spmm(A, T(X))
    assert rel_err(y.cpu().numpy(), ref) < RTOL
    assert torch.equal(y, ops.spmm(A, T(X)))                                    # deterministic
    assert rel_err(ops.spmm(A, T(X), 0.25, 0.25, T(Z)).cpu().numpy(), O.spmm(csr, X, 0.25, 0.25, Z)) < RTOL
    zf = (rng.random(N) < 0.3).astype(np.uint8)
    Zs = Z * zf[:, None]
    assert rel_err(ops.spmm_flagged(A, T(X), None, 0.5, 2.0, T(Zs), T(zf)).cpu().numpy(), O.spmm(csr, X, 0.5, 2.0, Zs)) < RTOL
    S = T(Z.copy()); Y = torch.empty_like(S)
    ops.spmm_layersum(A, T(X), S, S, Y)
    assert rel_err(Y.cpu().numpy(), ref) < RTOL and rel_err(S.cpu().numpy(), Z + ref) < RTOL
    P = rng.standard_normal((N, d)).astype(np.float32) * 0.1
    M = rng.standard_normal((N, d)).astype(np.float32) * 0.01
    V = (rng.random((N, d)).astype(np.float32)) * 1e-4
    g = O.spmm(csr, X, 0.25, 0.25, Zs)
    Pr, Mr, Vr = P.copy(), M.copy(), V.copy()
    O.adam_step(Pr, g, Mr, Vr, 0.005, 7)
    Pt, Mt, Vt = T(P), T(M), T(V)
    ops.spmm_adam(A, T(X), 0.25, 0.25, T(Zs), Pt, Mt, Vt, 0.005, 7, zflags=T(zf))
    assert rel_err(Pt.cpu().numpy(), Pr) < RTOL and rel_err(Mt.cpu().numpy(), Mr) < RTOL and rel_err(Vt.cpu().numpy(), Vr) < RTOL
    # other widths keep the CSR kernel on the same graph object
    X32 = rng.standard_normal((N, 32)).astype(np.float32)
    assert rel_err(ops.spmm(A, T(X32)).cpu().numpy(), O.spmm(csr, X32)) < RTOL
    # same pattern, new edge values
    val2 = (val * rng.random(len(val))).astype(np.float32)
    A2 = A.with_values(T(val2))
    assert A2.blocked is not None and A2.blocked is not A.blocked
    assert rel_err(ops.spmm(A2, T(X)).cpu().numpy(), O.spmm((rowptr, col, val2), X)) < RTOL
    assert rel_err(ops.spmm(A, T(X)).cpu().numpy(), ref) < RTOL


def test_spmm_blocked_nonfinite_operand_row_stays_on_its_neighbours(ops):
    """A non-finite operand row reaches exactly the rows adjacent to it, as with the CSR kernel (padding records repeat the wave's last real
    record with value 0 instead of pointing at row 0)."""
    rng = np.random.default_rng(5)
    U, I, d = 2000, 300, 64
    u, i = random_graph(rng, U, I, 9, hot_items=2, hot_deg=1500)
    rowptr, col, w, val = make_csr(u, i, U, I)
    N = U + I
    X = rng.standard_normal((N, d)).astype(np.float32)
    X[0] = np.inf                                                               # user 0
    Ac = ops.CSRGraph(rowptr, col, val, DEV)
    Ab = ops.CSRGraph(rowptr, col, val, DEV).enable_blocked(split=U, hub=200)
    yc, yb = ops.spmm(Ac, T(X)), ops.spmm(Ab, T(X))
    bad_c, bad_b = ~torch.isfinite(yc).all(1), ~torch.isfinite(yb).all(1)
    nbrs = torch.zeros(N, dtype=torch.bool, device=DEV); nbrs[torch.from_numpy(col[rowptr[0]:rowptr[1]].astype(np.int64)).to(DEV)] = True
    assert torch.equal(bad_c, nbrs) and torch.equal(bad_b, nbrs)


def test_spmm_blocked_small_row_set_stays_with_csr_kernel(ops):
    """A row set that would fill fewer than min_waves waves is left to the chunked CSR kernel as a whole (BlockedPlan)."""
    rng = np.random.default_rng(11)
    U, I, d = 3001, 703, 64
    u, i = random_graph(rng, U, I, 12, hot_items=2, hot_deg=2000, empty_users=(9,))
    rowptr, col, w, val = make_csr(u, i, U, I)
    A = ops.CSRGraph(rowptr, col, val, DEV).enable_blocked(split=U, min_waves=50)
    assert len(A.blocked.sets) == 1 and A.blocked.sets[0]['n_rows'] == U and A.blocked.n_hub == I
    X = rng.standard_normal((U + I, d)).astype(np.float32)
    Z = rng.standard_normal((U + I, d)).astype(np.float32)
    assert rel_err(ops.spmm(A, T(X), 0.5, -1.0, T(Z)).cpu().numpy(), O.spmm((rowptr, col, val), X, 0.5, -1.0, Z)) < RTOL


def test_spmm_blocked_graph_without_edges(ops):
    """All rows empty: the blocked plan has waves but no records; epilogues still run on every row."""
    n, d = 100, 64
    A = ops.CSRGraph(np.zeros(n + 1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32), DEV).enable_blocked(split=60)
    X = torch.randn(n, d, device=DEV); Z = torch.randn(n, d, device=DEV)
    assert torch.equal(ops.spmm(A, X, 2.0, -0.5, Z), -0.5 * Z)
    assert float(ops.spmm(A, X).abs().max()) == 0.0


def test_spmm_blocked_rejects_bad_plans(ops):
    import ctypes as C
    from arlib_amd import _lib
    rng = np.random.default_rng(0)
    u, i = random_graph(rng, 200, 50, 5)
    rowptr, col, w, val = make_csr(u, i, 200, 50)
    A = ops.CSRGraph(rowptr, col, val, DEV)
    with pytest.raises(ValueError):
        A.enable_blocked(rows_per_wave=64)
    A.enable_blocked(split=200)
    st = A.blocked.struct(0, 64)
    X = torch.randn(250, 32, device=DEV); Y = torch.empty_like(X)
    L = _lib.lib()
    assert L.arl_spmm_blocked_f32(C.byref(st), X.data_ptr(), 32, 1.0, 0.0, None, None, Y.data_ptr(), None) == -2
    X = torch.randn(250, 64, device=DEV)
    assert L.arl_spmm_blocked_f32(C.byref(st), X.data_ptr(), 64, 1.0, 0.0, None, None, X.data_ptr(), None) == -4
    bad = _lib.arl_blocked(st.n_waves, 48, 16, st.wave_ptr, st.wave_rows, st.rec_col, st.rec_val, 0, None, None, None, None, 0)
    Y = torch.empty_like(X)
    assert L.arl_spmm_blocked_f32(C.byref(bad), X.data_ptr(), 64, 1.0, 0.0, None, None, Y.data_ptr(), None) == -4


def test_batch_rows_set_and_clear(ops):
    """Fused per-batch updates of the sparse-batch step (gradient rows += with duplicates, byte flags, bitmap) against torch."""
    rng = np.random.default_rng(12)
    N, d, n = 1000, 48, 400
    idx = rng.integers(0, N, n).astype(np.int32); idx[:50] = 7; idx[50:60] = N - 1
    src = rng.standard_normal((n, d)).astype(np.float32)
    G = torch.zeros(N, d, device=DEV); G[3] = 1.0
    flags = torch.zeros(N, dtype=torch.uint8, device=DEV); bits = torch.zeros((N + 31) // 32, dtype=torch.int32, device=DEV)
    ops.batch_rows_set_(G, flags, bits, T(idx), T(src), 0.5)
    ref = np.zeros((N, d), np.float64); ref[3] = 1.0
    np.add.at(ref, idx.astype(np.int64), 0.5 * src.astype(np.float64))
    assert rel_err(G.cpu().numpy(), ref.astype(np.float32)) < 1e-5
    want = np.zeros(N, np.uint8); want[idx] = 1
    assert np.array_equal(flags.cpu().numpy(), want)
    bn = bits.cpu().numpy().view(np.uint32)
    assert np.array_equal(((bn[np.arange(N) >> 5] >> (np.arange(N) & 31)) & 1).astype(np.uint8), want)
    ops.batch_rows_clear_(G, flags, bits, T(idx))
    assert int(flags.max()) == 0 and int(bits.abs().max()) == 0
    g = G.cpu().numpy()
    assert np.all(g[3] == 1.0) and float(np.abs(np.delete(g, 3, axis=0)).max()) == 0.0
    with pytest.raises(IndexError):
        ops.batch_rows_set_(G, flags, bits, T(np.array([N], np.int32)), T(src[:1]))


@pytest.mark.parametrize('blocked', [False, True])
def test_spmm_row_scale_epilogue(ops, blocked):
    """out = alpha * diag(row_scale) (A X) + beta * Z on both schedules (PGA's factored operator)."""
    rng = np.random.default_rng(31)
    U, I, d = 2001, 503, 64
    u, i = random_graph(rng, U, I, 10, hot_items=2, hot_deg=1500, empty_users=(3,))
    rowptr, col, w, val = make_csr(u, i, U, I)
    N = U + I
    A = ops.CSRGraph(rowptr, col, val, DEV, chunk=512)
    if blocked:
        A.enable_blocked(split=U, hub=100)
    X = rng.standard_normal((N, d)).astype(np.float32); Z = rng.standard_normal((N, d)).astype(np.float32)
    rs = rng.random(N).astype(np.float32); rs[7] = 0
    ref = O.spmm((rowptr, col, val), X)
    got = ops.spmm(A, T(X), 0.5, -1.5, T(Z), row_scale=T(rs))
    assert rel_err(got.cpu().numpy(), 0.5 * rs[:, None] * ref - 1.5 * Z) < RTOL
    assert rel_err(ops.spmm(A, T(X), row_scale=T(rs)).cpu().numpy(), rs[:, None] * ref) < RTOL


def test_spmm_deterministic_and_linear(ops):
    rng = np.random.default_rng(3)
    U, I, d = 5000, 900, 64
    u, i = random_graph(rng, U, I, 20, hot_items=2, hot_deg=4000)
    rowptr, col, w, val = make_csr(u, i, U, I)
    A = ops.CSRGraph(rowptr, col, val, DEV)
    X1 = torch.randn(U + I, d, device=DEV); X2 = torch.randn(U + I, d, device=DEV)
    y1 = ops.spmm(A, X1); y1b = ops.spmm(A, X1)
    assert torch.equal(y1, y1b)                                   # chunk partials are combined in fixed order
    lin = ops.spmm(A, X1 + 2 * X2)
    assert rel_err(lin.cpu().numpy(), (y1 + 2 * ops.spmm(A, X2)).cpu().numpy()) < RTOL
    # symmetry of the normalised adjacency: <y, A x> == <A y, x>
    a = (X2 * y1).sum().item(); b = (ops.spmm(A, X2) * X1).sum().item()
    assert abs(a - b) <= 1e-3 * max(abs(a), 1.0)


def test_norm_adj_values_golden_and_weighted(ops, ml100k):
    g = golden('g3_adj.npz')
    p = ml100k['pairs0']
    rowptr, col, w = O.bipartite_csr(p[:, 0], p[:, 1], ml100k['U'], ml100k['I'])
    val, dinv = ops.norm_adj_values(T(rowptr.astype(np.int32)), T(col), T(w), len(rowptr) - 1)
    assert rel_err(val.cpu().numpy(), g['norm_data']) < 1e-6
    U, I = (int(x) for x in g['w_shape'])
    rowptr, col, w = O.bipartite_csr(g['w_R_row'], g['w_R_col'], U, I, g['w_R_val'])
    val, dinv = ops.norm_adj_values(T(rowptr.astype(np.int32)), T(col), T(w), U + I)
    assert np.allclose(val.cpu().numpy(), g['w_norm_val'], rtol=1e-5, atol=0)
    assert dinv[7].item() == 0.0                                  # isolated user: empty row, no inf/NaN


@pytest.mark.parametrize('tag', ['n', 'sat'])
def test_bpr_l2_golden(ops, tag):
    g = golden('g2_losses.npz')
    u, p, n = g[tag + '_u'], g[tag + '_p'], g[tag + '_n']
    B = len(u)
    emb = T(np.concatenate([u, p, n], 0))
    G = torch.zeros_like(emb)
    ar = torch.arange(B, dtype=torch.int32, device=DEV)
    out = ops.bpr_l2_fwd_bwd(emb, B, ar, ar, ar + B, 1e-4, G).cpu().numpy()
    assert abs(out[0] - g[tag + '_bpr'][0]) <= RTOL * abs(g[tag + '_bpr'][0])
    assert abs(out[1] - g[tag + '_reg'][0]) <= RTOL * abs(g[tag + '_reg'][0])
    Gn = G.cpu().numpy()
    assert rel_err(Gn[:B], g[tag + '_du']) < RTOL and rel_err(Gn[B:2 * B], g[tag + '_dp']) < RTOL
    assert rel_err(Gn[2 * B:], g[tag + '_dn']) < RTOL


def test_bpr_l2_duplicates_and_ragged(ops):
    g = golden('g2_losses.npz')
    emb = T(g['dup_T'])
    G = torch.zeros_like(emb)
    out = ops.bpr_l2_fwd_bwd(emb, 0, T(g['dup_ui']), T(g['dup_pi']), T(g['dup_ni']), 1e-4, G).cpu().numpy()
    assert abs(out[0] + out[1] - g['dup_loss'][0]) <= RTOL * abs(g['dup_loss'][0])
    assert rel_err(G.cpu().numpy(), g['dup_dT']) < RTOL
    # ragged batch (B=1, odd d) against the oracle
    rng = np.random.default_rng(0)
    e = (0.3 * rng.standard_normal((50, 20))).astype(np.float32)      # un-saturated: s*(1-s) is ill-conditioned in fp32 when s -> 1
    for B in (1, 3, 257):
        ui = rng.integers(0, 20, B).astype(np.int32); pi = rng.integers(0, 30, B).astype(np.int32); ni = rng.integers(0, 30, B).astype(np.int32)
        lb, lr_, Gr = O.bpr_l2(e, 20, ui, pi, ni, 1e-3)
        G = torch.zeros(50, 20, device=DEV)
        out = ops.bpr_l2_fwd_bwd(T(e), 20, T(ui), T(pi), T(ni), 1e-3, G).cpu().numpy()
        assert abs(out[0] - lb) <= RTOL * abs(lb) and abs(out[1] - lr_) <= RTOL * abs(lr_)
        assert rel_err(G.cpu().numpy(), Gr) < RTOL
    with pytest.raises(IndexError):
        ops.bpr_l2_fwd_bwd(T(e), 20, T(np.array([3], np.int32)), T(np.array([30], np.int32)), T(np.array([0], np.int32)), 1e-3)   # item 30 + offset 20 = row 50: out of range


def test_adam_sgd_dense(ops):
    rng = np.random.default_rng(1)
    for n in (4096, 4099, 3):
        p = rng.standard_normal(n).astype(np.float32); g = rng.standard_normal(n).astype(np.float32)
        m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        pt, mt, vt = T(p), T(m), T(v)
        for t in (1, 2, 3):
            O.adam_step(p, g, m, v, 0.005, t)
            ops.adam_dense(pt, T(g), mt, vt, 0.005, t)
        assert rel_err(pt.cpu().numpy(), p) < 1e-6 and rel_err(mt.cpu().numpy(), m) < 1e-6 and rel_err(vt.cpu().numpy(), v) < 1e-6
        ps = p.copy(); O.sgd_step(ps, g, 0.0005)
        pt2 = T(p); ops.sgd_dense(pt2, T(g), 0.0005)
        assert rel_err(pt2.cpu().numpy(), ps) < 1e-6


def test_gather_scatter_rows(ops):
    rng = np.random.default_rng(2)
    src = rng.standard_normal((100, 64)).astype(np.float32)
    idx = rng.integers(0, 100, 333).astype(np.int32)
    assert np.array_equal(ops.gather_rows(T(src), T(idx)).cpu().numpy(), src[idx])
    add = rng.standard_normal((333, 64)).astype(np.float32)
    ref = np.zeros((100, 64), np.float64); np.add.at(ref, idx, 0.5 * add.astype(np.float64))
    dst = torch.zeros(100, 64, device=DEV)
    ops.scatter_add_rows(dst, T(idx), T(add), 0.5)
    assert rel_err(dst.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize('nA,nV,d', [(2048, 20000, 64), (77, 333, 64), (16, 64, 64), (1000, 4097, 64), (500, 3000, 16), (300, 1412, 32), (260, 2000, 128)])
def test_nce_allrows_against_float64(ops, nA, nV, d):
    """All-rows InfoNCE (recommender/NCL.py:96-115): log-sum-exp over the whole table and both gradient sums against float64 torch; ragged
    sizes (rows past the last full 16 / 64 tile on both sides), temperature 0.05 (logits up to 20), two runs bit-identical."""
    g = torch.Generator().manual_seed(nA + nV)
    A = torch.nn.functional.normalize(torch.randn(nA, d, generator=g), dim=1).to(DEV)
    V = torch.nn.functional.normalize(torch.randn(nV, d, generator=g) + 0.3, dim=1).to(DEV)
    tau = 0.05
    lse, dA, dV = ops.nce_allrows(A, V, tau)
    S = (A.double() @ V.double().T) / tau
    lse_ref = torch.logsumexp(S, dim=1)
    P = torch.exp(S - lse_ref[:, None])
    assert float((lse.double() - lse_ref).abs().max()) < 1e-5
    assert rel_err(dA.cpu().numpy(), (P @ V.double()).cpu().numpy()) < 1e-5
    assert rel_err(dV.cpu().numpy(), (P.T @ A.double()).cpu().numpy()) < 1e-5
    lse2, dA2, dV2 = ops.nce_allrows(A, V, tau)
    assert torch.equal(lse, lse2) and torch.equal(dA, dA2) and torch.equal(dV, dV2)
    assert torch.equal(ops.nce_allrows(A, V, tau, want_grad=False), lse)
    # the three-pass form (log-sum-exp given): same sums with the normalisation inside the exponent instead of after the sum
    _, dA3, dV3 = ops.nce_allrows(A, V, tau, lse=lse, want_dV=True)
    assert rel_err(dA3.cpu().numpy(), (P @ V.double()).cpu().numpy()) < 1e-5 and torch.equal(dV3, dV)
    with pytest.raises(ValueError):
        ops.nce_allrows(A[:, :8].contiguous(), V[:, :8].contiguous(), tau)
    # the kernels shift by the constant 1 (no running maximum): at the smallest temperature they accept, a batch row whose best cosine is NEGATIVE
    # still has a finite log-sum-exp and finite gradients; anything smaller is refused (callers take the panel form with a running maximum)
    Aneg = torch.nn.functional.normalize(-V[:nA if nA <= nV else nV].mean(0, keepdim=True).repeat(min(nA, 16), 1) + 0.01 * torch.randn(min(nA, 16), d, generator=g).to(DEV), dim=1).contiguous()
    l4, dA4, dV4 = ops.nce_allrows(Aneg, V, ops.NCE_ALLROWS_MIN_TAU)
    S4 = (Aneg.double() @ V.double().T) / ops.NCE_ALLROWS_MIN_TAU
    assert bool(torch.isfinite(l4).all()) and bool(torch.isfinite(dA4).all()) and bool(torch.isfinite(dV4).all())
    assert float((l4.double() - torch.logsumexp(S4, dim=1)).abs().max()) < 1e-4
    with pytest.raises(ValueError):
        ops.nce_allrows(A, V, 0.01)


@pytest.mark.parametrize('n,d', [(1000, 64), (37, 16), (513, 128), (90, 100), (5, 256)])
def test_normalize_rows_forward_and_autograd(ops, n, d):
    """arl_normalize_rows_f32 / _bwd_f32 against F.normalize and torch's autograd of it (recommender/NCL.py:98-99), incl. a zero row (eps clamp)."""
    g = torch.Generator().manual_seed(n * d)
    X = (torch.randn(n, d, generator=g) * 3).to(DEV)
    X[n // 2] = 0
    dY = torch.randn(n, d, generator=g).to(DEV)
    Xr = X.clone().requires_grad_(True)
    Yr = torch.nn.functional.normalize(Xr, dim=1)
    Yr.backward(dY * 0.7)
    Y, nrm = ops.normalize_rows(X)
    assert rel_err(Y.cpu().numpy(), Yr.detach().cpu().numpy()) < 1e-6
    sc = torch.tensor([0.35], device=DEV)
    dX = ops.normalize_rows_bwd(Y, nrm, dY, 2.0, scale_dev=sc)
    live = torch.ones(n, dtype=torch.bool); live[n // 2] = False
    assert rel_err(dX[live].cpu().numpy(), Xr.grad[live].cpu().numpy()) < 1e-5
    assert torch.equal(dX[n // 2], dY[n // 2] * 0.7 / 1e-12) or torch.allclose(dX[n // 2], dY[n // 2] * 0.7 / 1e-12, rtol=1e-6)
    dY2 = dY.clone()
    assert ops.normalize_rows_bwd(Y, nrm, dY2, 0.7, out=dY2) is dY2 and rel_err(dY2[live].cpu().numpy(), Xr.grad[live].cpu().numpy()) < 1e-5


def test_all_rows_nce_fused_equals_panel_form():
    """recommender/NCL.py:96-115: the fused route (normalisation kernels + arl_nce_allrows_*) against the panel route (F.normalize + library GEMMs)
    on raw rows: loss and both gradients."""
    from arlib_amd.recommender import NCL as M
    g = torch.Generator().manual_seed(7)
    Xv0 = torch.randn(3000, 64, generator=g).to(DEV)
    idx = torch.randint(0, 3000, (500,), generator=g).to(DEV)
    Xc0 = (Xv0 + 2.0 * torch.randn(3000, 64, generator=g).to(DEV))          # contexts at an angle to their own rows: lse - pos does not cancel
    out = []
    for fused in (True, False):
        M._AllRowsNCE.FUSED = fused
        try:
            Xv = Xv0.clone().requires_grad_(True)
            Xc = Xc0.clone().requires_grad_(True)
            loss = 1e-3 * M.all_rows_nce(Xc[idx], Xv, idx, 0.05)
            loss.backward()
            out.append((loss.item(), Xc.grad.cpu().numpy(), Xv.grad.cpu().numpy()))
        finally:
            M._AllRowsNCE.FUSED = True
    # float64 autograd of the reference's expression as the arbiter
    Xv = Xv0.double().cpu().requires_grad_(True); Xc = Xc0.double().cpu().requires_grad_(True)
    a, v = torch.nn.functional.normalize(Xc[idx.cpu()]), torch.nn.functional.normalize(Xv)
    l64 = 1e-3 * -torch.log(torch.exp((a * v[idx.cpu()]).sum(1) / 0.05) / torch.exp(a @ v.T / 0.05).sum(1)).sum()
    l64.backward()
    for name, (l, gc, gv) in zip(('fused', 'panel'), out):
        errs = (abs(l - l64.item()) / abs(l64.item()), rel_err(gc, Xc.grad.numpy()), rel_err(gv, Xv.grad.numpy()))
        print('all_rows_nce %s vs float64: loss %.2e, context grad %.2e, table grad %.2e' % ((name,) + errs))
        assert errs[0] < 1e-5 and errs[1] < 2e-5 and errs[2] < 2e-5


def test_fused_adam_optimizer_equals_torch_adam():
    """util/optim.Adam (arl_adam_dense_f32 behind torch.optim.Adam's interface) against the stock class: 4 steps on a table and an odd-sized
    vector, parameters and both moments; a non-default flag falls through to the stock step."""
    from arlib_amd.util.optim import Adam
    g = torch.Generator().manual_seed(3)
    base = [torch.randn(1000, 64, generator=g).to(DEV), torch.randn(37, generator=g).to(DEV)]
    grads = [[torch.randn(b.shape, generator=g).to(DEV) * (0.1 ** s) for b in base] for s in range(4)]
    runs = []
    for cls, kw in ((torch.optim.Adam, {}), (Adam, {}), (Adam, {'amsgrad': True}), (torch.optim.Adam, {'amsgrad': True})):
        ps = [torch.nn.Parameter(b.clone()) for b in base]
        opt = cls(ps, lr=0.005, **kw)
        for s in range(4):
            for p, gr in zip(ps, grads[s]):
                p.grad = gr.clone()
            opt.step()
        runs.append(([p.detach().cpu().numpy() for p in ps], [opt.state[p]['exp_avg'].cpu().numpy() for p in ps], [opt.state[p]['exp_avg_sq'].cpu().numpy() for p in ps],
                     [float(opt.state[p]['step']) for p in ps]))
    for a, b in ((runs[0], runs[1]), (runs[3], runs[2])):
        for x, y in zip(a[:3], b[:3]):
            for u, v in zip(x, y):
                assert rel_err(v, u) < 1e-6
        assert a[3] == b[3] == [4.0, 4.0]


def _seq_add(dst, idx, src, scale):
    """CPU index_put_(accumulate) association: contributions added one after the other in index order, fp32, product rounded first."""
    out = dst.copy()
    c = (np.float32(scale) * src).astype(np.float32)
    np.add.at(out, idx.astype(np.int64), c)           # unbuffered, in order, in the array's own precision
    return out


@pytest.mark.parametrize('n,d', [(333, 64), (6144, 64), (40000, 16), (700, 300)])
def test_ordered_accumulation_is_sequential_and_bit_exact(ops, n, d):
    """scatter_add_rows / batch_rows_set_ add duplicate rows IN INDEX ORDER (no float atomics): the result equals a sequential fp32 sum
    bit for bit -- heavy duplicates, a list longer than one launch window (16 384), a width above one 256-column pass -- and two runs agree."""
    rng = np.random.default_rng(n + d)
    N = 500
    idx = rng.integers(0, N, n).astype(np.int32); idx[: n // 8] = 7; idx[-5:] = 7; idx[n // 2] = N - 1
    src = (rng.standard_normal((n, d)) * 10.0 ** rng.integers(-6, 3, (n, 1))).astype(np.float32)     # wide dynamic range: order matters
    base = rng.standard_normal((N, d)).astype(np.float32)
    want = _seq_add(base, idx, src, 0.37)
    outs = []
    for _ in range(2):
        dst = T(base.copy())
        ops.scatter_add_rows(dst, T(idx), T(src), 0.37)
        outs.append(dst.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], want)
    perm = rng.permutation(n)                                                                         # another order gives other bits (the test has teeth)
    assert not np.array_equal(_seq_add(base, idx[perm], src[perm], 0.37), want)
    G = T(base.copy()); flags = torch.zeros(N, dtype=torch.uint8, device=DEV); bits = torch.zeros((N + 31) // 32, dtype=torch.int32, device=DEV)
    ops.batch_rows_set_(G, flags, bits, T(idx), T(src), 0.37)
    assert np.array_equal(G.cpu().numpy(), want)
    wantf = np.zeros(N, np.uint8); wantf[idx] = 1
    assert np.array_equal(flags.cpu().numpy(), wantf)
    # with the duplicate bitmap (only rows named more than once take the ordered scan): same bits, and the bitmap holds exactly those rows
    G2 = T(base.copy()); flags2 = torch.zeros_like(flags); bits2 = torch.zeros_like(bits); dup = torch.zeros_like(bits)
    ops.batch_rows_set_(G2, flags2, bits2, T(idx), T(src), 0.37, dup_bits=dup)
    assert np.array_equal(G2.cpu().numpy(), want) and torch.equal(flags2, flags) and torch.equal(bits2, bits)
    cnt = np.bincount(idx, minlength=N)
    dn = dup.cpu().numpy().view(np.uint32)
    assert np.array_equal(((dn[np.arange(N) >> 5] >> (np.arange(N) & 31)) & 1).astype(bool), cnt > 1)
    ops.batch_rows_clear_(G2, flags2, bits2, T(idx), dup_bits=dup)
    assert int(flags2.max()) == 0 and int(bits2.abs().max()) == 0 and int(dup.abs().max()) == 0


def test_batch_rows_set_zero_factor_entries_are_absent(ops):
    """The user-sharded step keeps static shapes by giving every foreign sample factor 0 on a clamped local row (arl_shard_batch_prep_i32): such entries
    are ABSENT -- they add nothing, mark nothing, are no duplicates -- so ~7/8 of a batch's user entries at 8 ranks do not pile up on one row."""
    rng = np.random.default_rng(5)
    N, n, d = 300, 4096, 64
    idx = rng.integers(1, N - 1, n).astype(np.int32)
    rs = np.ones(n, np.float32)
    foreign = rng.random(n) < 0.875
    idx[foreign] = np.where(rng.random(int(foreign.sum())) < 0.5, 0, N - 1)         # clamped rows 0 and N - 1, named by foreign entries only
    rs[foreign] = 0.0
    src = rng.standard_normal((n, d)).astype(np.float32)
    src[foreign] = np.nan                                                           # an absent entry's payload is never read into the sum
    base = rng.standard_normal((N, d)).astype(np.float32)
    own = ~foreign
    want = _seq_add(base, idx[own], src[own], 0.5)
    for with_dup in (False, True):
        G = T(base.copy()); flags = torch.zeros(N, dtype=torch.uint8, device=DEV); bits = torch.zeros((N + 31) // 32, dtype=torch.int32, device=DEV)
        dup = torch.zeros_like(bits) if with_dup else None
        ops.batch_rows_set_(G, flags, bits, T(idx), T(src), 0.5, row_scale=T(rs), dup_bits=dup)
        assert np.array_equal(G.cpu().numpy(), want)
        wantf = np.zeros(N, np.uint8); wantf[idx[own]] = 1
        assert np.array_equal(flags.cpu().numpy(), wantf) and wantf[0] == 0 and wantf[N - 1] == 0
        bn = bits.cpu().numpy().view(np.uint32)
        assert np.array_equal(((bn[np.arange(N) >> 5] >> (np.arange(N) & 31)) & 1).astype(np.uint8), wantf)
        if with_dup:
            dn = dup.cpu().numpy().view(np.uint32)
            assert np.array_equal(((dn[np.arange(N) >> 5] >> (np.arange(N) & 31)) & 1).astype(bool), np.bincount(idx[own], minlength=N) > 1)
        ops.batch_rows_clear_(G, flags, bits, T(idx), dup_bits=dup)
        assert int(flags.max()) == 0 and int(bits.abs().max()) == 0


@pytest.mark.parametrize('U,F,I,d,k,nT', [(3000, 7, 900, 64, 50, 5), (500, 0, 70, 16, 8, 3), (1200, 3, 5000, 128, 20, 1), (257, 1, 300, 100, 64, 64)])
def test_cw_topk_term_against_float64_and_deterministic(ops, U, F, I, d, k, nT):
    """arl_cw_topk_term_f32 (attack/White/CLeaR.py:83-95): loss, gradient on every row of the packed table and the SFA row multiplicities against a
    float64 restatement of the pair lists (negative = successive .pop()s from the tail of the top-k list), the oracle's cw_loss_grad, and two runs
    bit for bit -- with the negatives concentrated on a handful of items (thousands of addends per item row: the order-free fixed-point sums)."""
    from arlib_amd.attack._common import cw_pairs
    rng = np.random.default_rng(U + I + d)
    Up = U + F
    X = (rng.standard_normal((Up + I, d)) * 10.0 ** rng.integers(-3, 1, (Up + I, 1))).astype(np.float32)
    top = np.stack([rng.choice(I, size=k, replace=False) for _ in range(Up)]).astype(np.int32)
    hot = rng.choice(I, size=3, replace=False)
    top[: U // 2, k - 1] = hot[0]; top[U // 4: U, max(k - 2, 0)] = hot[1]                     # heavy rows (a distinct-list violation does not matter to the term)
    targets = rng.choice(I, size=nT, replace=False).astype(np.int64)
    c = 1.0 / (U * nT)
    Xd = X.astype(np.float64)
    neg = top[:U][:, [k - 1 - t for t in range(nT)]].astype(np.int64)                          # [U, nT]
    loss_ref = c * sum(((Xd[:U] * Xd[Up + neg[:, t]]).sum() - (Xd[:U] * Xd[Up + targets[t]]).sum()) for t in range(nT))
    G_ref = np.zeros_like(Xd)
    for t in range(nT):
        G_ref[:U] += c * (Xd[Up + neg[:, t]] - Xd[Up + targets[t]])
        np.add.at(G_ref, Up + neg[:, t], c * Xd[:U])
        G_ref[Up + targets[t]] -= c * Xd[:U].sum(0)
    w_ref = np.zeros(Up + I); w_ref[:U] = nT
    np.add.at(w_ref, Up + neg.reshape(-1), 1.0); np.add.at(w_ref, Up + targets, float(U))
    outs = []
    for _ in range(2):
        loss, G, w = ops.cw_topk_term(T(X), Up, U, T(top), torch.from_numpy(targets).to(DEV))
        outs.append((loss.clone(), G.clone(), w.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    loss, G, w = (t.cpu().numpy() for t in outs[0])
    assert abs(loss[0] - loss_ref) <= 1e-5 * max(abs(loss_ref), np.abs(G_ref).max())
    assert rel_err(G, G_ref) < 5e-6 and row_err(G, G_ref) < 2e-5
    assert np.array_equal(w, w_ref.astype(np.float32)) and float(np.abs(G[U:Up]).max() if F else 0.0) == 0.0
    users, pos, ng = cw_pairs(T(top).long(), U, [int(t) for t in targets], pop=True)
    lo, Go = O.cw_loss_grad(X, Up, users.cpu().numpy(), pos.cpu().numpy(), ng.cpu().numpy())
    assert abs(loss[0] - lo) <= 1e-4 * max(abs(lo), np.abs(Go).max()) and close(G, Go)
    with pytest.raises(IndexError):
        bad = top.copy(); bad[0, k - 1] = I
        ops.cw_topk_term(T(X), Up, U, T(bad), torch.from_numpy(targets).to(DEV))


def test_bpr_backward_ordered(ops):
    """BPR + L2 backward with repeated users / items in the batch: bit-identical run to run, equal to the oracle within fp32 rounding; a batch
    longer than one launch window too."""
    rng = np.random.default_rng(77)
    U, I, d = 40, 30, 64
    emb = (rng.standard_normal((U + I, d)) * 0.1).astype(np.float32)
    for B in (512, 7000):
        u = rng.integers(0, U, B).astype(np.int32); p = rng.integers(0, I, B).astype(np.int32); n = rng.integers(0, I, B).astype(np.int32)
        res = []
        for _ in range(2):
            G = torch.zeros(U + I, d, device=DEV)
            lo = ops.bpr_l2_fwd_bwd(T(emb), U, T(u), T(p), T(n), 1e-3, G)
            res.append((G.cpu().numpy(), lo.cpu().numpy()))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        lb, lr_, Gref = O.bpr_l2(emb, U, u, p, n, 1e-3)
        assert rel_err(res[0][0], Gref) < 1e-5 and abs(float(res[0][1][0]) - lb) <= 1e-5 * abs(lb)


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_infonce_golden(ops, tag):
    g = golden('g6_infonce.npz')
    loss, d1, d2 = ops.infonce_fwd_bwd(T(g[tag + '_v1']), T(g[tag + '_v2']), 0.2)
    assert abs(loss.item() - g[tag + '_loss'][0]) <= RTOL * abs(g[tag + '_loss'][0])
    assert rel_err(d1.cpu().numpy(), g[tag + '_dv1']) < RTOL and rel_err(d2.cpu().numpy(), g[tag + '_dv2']) < RTOL


def test_simgcl_perturb(ops):
    rng = np.random.default_rng(4)
    E = rng.standard_normal((500, 64)).astype(np.float32); E[3, 5] = 0.0
    noise = rng.random((500, 64)).astype(np.float32)
    out = ops.simgcl_perturb_(T(E), T(noise), 0.1).cpu().numpy()
    assert rel_err(out, O.simgcl_perturb(E, noise, 0.1)) < 1e-6


@pytest.mark.parametrize('n,d', [(1, 4), (777, 16), (5000, 64), (3001, 100), (9000, 256)])
def test_sfa_l1_weighted_rows(ops, n, d):
    """CLeaR's SFA term on H = table rows with multiplicities (kernel never builds H) vs the literal restatement on H."""
    rng = np.random.default_rng(n + d)
    X = (rng.normal(size=(n, d)) * 0.2).astype(np.float32)
    w = rng.integers(0, 4, n).astype(np.float32)
    w[0] = 2.0                                                   # at least one row in H
    if n > 10:
        w[5] = 1000.0                                            # a target item's row: once per real user
    r0 = rng.normal(size=d).astype(np.float32)
    rows = np.repeat(np.arange(n), w.astype(np.int64))
    loss_ref, gH = O.sfa_l1_loss_grad(X[rows], r0)
    G_ref = np.zeros((n, d)); np.add.at(G_ref, rows, gH)
    loss, G = ops.sfa_l1(T(X), T(w), T(r0), len(rows) * d)
    assert abs(loss.item() - loss_ref) <= RTOL * abs(loss_ref)
    assert rel_err(G.cpu().numpy(), G_ref) < RTOL
    assert not G.cpu().numpy()[w == 0].any()                     # rows outside H get exact zeros
    base = torch.full_like(G, 0.5)
    _, G2 = ops.sfa_l1(T(X), T(w), T(r0), len(rows) * d, out=base, scale=-2.0, accumulate=True)
    assert rel_err(G2.cpu().numpy(), 0.5 - 2.0 * G_ref) < RTOL
    loss_only, none = ops.sfa_l1(T(X), T(w), T(r0), len(rows) * d, want_grad=False)
    assert none is None and loss_only.item() == loss.item()      # deterministic reductions
    with pytest.raises(ValueError):
        ops.sfa_l1(T(X), T(w[:-1]) if n > 1 else T(np.zeros(2, np.float32)), T(r0), len(rows) * d)


def test_sddmm_rows_dense_and_pga_update(ops):
    rng = np.random.default_rng(5)
    N, d, I, off = 900, 64, 333, 500
    dY = rng.standard_normal((N, d)).astype(np.float32); X = rng.standard_normal((N, d)).astype(np.float32)
    rows = np.array([497, 498, 499, 3, 3], np.int32)
    ref = O.sddmm_rows_dense(dY, X, rows, off, I)
    out = ops.sddmm_rows_dense(T(dY), T(X), T(rows), off, I)
    assert rel_err(out.cpu().numpy(), ref) < RTOL
    out2 = ops.sddmm_rows_dense(T(dY), T(X), T(rows), off, I, out=out)        # accumulates
    assert rel_err(out2.cpu().numpy(), 2 * ref) < RTOL
    S = rng.random((5, I)).astype(np.float32); gr = (rng.standard_normal((5, I)) * 3).astype(np.float32)
    S[1, :40] = 0.0                                                   # entries outside the sparse pattern: gradient ignored
    dr = rng.random(5).astype(np.float32); dc = rng.random(I).astype(np.float32)
    assert rel_err(ops.pga_update_(T(S), T(gr)).cpu().numpy(), O.pga_update(S, gr)) < 1e-6
    got = ops.pga_update_(T(S), T(gr), T(dr), T(dc)).cpu().numpy()
    assert rel_err(got, O.pga_update(S, gr, dr, dc)) < 1e-6
    assert got.min() >= 9.9e-8 and np.all(got[1, :40] == np.float32(10e-8))


@pytest.mark.parametrize('exact', [False, True])
@pytest.mark.parametrize('U,I,d,k,masked', [(100, 1412, 64, 50, True), (37, 300, 16, 5, False), (70, 5000, 32, 128, True), (17, 60, 64, 50, True),
                                            (300, 2000, 128, 64, True), (130, 700, 64, 64, False), (129, 65, 64, 1, True), (5, 4000, 48, 20, False)])
def test_score_mask_topk(ops, U, I, d, k, masked, exact):
    rng = np.random.default_rng(U + I)
    Pu = rng.standard_normal((U, d)).astype(np.float32); Pi = rng.standard_normal((I, d)).astype(np.float32)
    mask = None
    if masked:
        deg = rng.integers(0, min(40, I - 1), U)
        if I <= 64:
            deg[:] = I - 10            # fewer than k unmasked items: masked (-10e8) entries must appear at the tail
        cols = [np.sort(rng.choice(I, size=dg, replace=False)) for dg in deg]
        rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        mask = (rp, np.concatenate(cols).astype(np.int32) if rp[-1] else np.zeros(0, np.int32))
    ridx, rval = O.score_mask_topk(Pu, Pi, k, mask)
    if mask is None:
        idx, val = ops.score_mask_topk(T(Pu), T(Pi), k, exact=exact)
    else:
        mc = mask[1] if len(mask[1]) else np.zeros(1, np.int32)
        idx, val = ops.score_mask_topk(T(Pu), T(Pi), k, T(mask[0].astype(np.int32)), T(mc), exact=exact)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    assert rel_err(val, rval) < RTOL
    # indices: identical except where two scores tie within fp32 rounding of the different summation orders
    same = idx == ridx
    if not same.all():
        bad = np.argwhere(~same)
        for r, c in bad:
            assert abs(rval[r, c] - val[r, c]) <= 1e-5 * max(1.0, abs(rval[r, c]))
            assert set(idx[r]) == set(ridx[r]) or abs(rval[r, -1] - val[r, -1]) <= 1e-5 * max(1.0, abs(rval[r, -1]))
    assert same.mean() > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize('d,k', [(64, 50), (128, 20), (64, 64), (32, 10)])
def test_score_mask_topk_bootstrap_threshold(ops, d, k):
    """Cold calls with I >= 32 K open with a bootstrap pass over the first 4 096 items whose (k + m)-th best sample score pre-sets every
    user's threshold (m = the user's interacted items inside the sample).  The bound must stay valid when the mask takes out exactly the
    sample's best items: users here interact with their top-scoring items of the sample range (few, more than 64 - k, and none)."""
    rng = np.random.default_rng(1000 * d + k)
    U, I = 200, 40000
    Pu = (rng.normal(size=(U, d)) * 0.1).astype(np.float32)
    Pi = (rng.normal(size=(I, d)) * 0.1).astype(np.float32)
    Pi[:4096] *= 1.5                                     # the sample range holds most of every user's best items
    scores = Pu.astype(np.float64) @ Pi.astype(np.float64).T
    lens = np.concatenate([np.zeros(40, np.int64), rng.integers(1, 14, 80), rng.integers(15, 120, 80)])
    cols = []
    for u in range(U):
        best = np.argsort(-scores[u, :4096])[:lens[u]]   # interacted = the user's best items of the sample ...
        extra = rng.choice(np.arange(4096, I), size=int(lens[u]) // 2, replace=False)       # ... plus some outside it
        cols.append(np.unique(np.concatenate([best, extra])).astype(np.int32))
    rp = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32)
    mc = np.concatenate(cols + [np.zeros(1, np.int32)])[:max(int(rp[-1]), 1)]
    for masked in (False, True):
        sc = scores.copy()
        if masked:
            for u in range(U):
                sc[u, cols[u]] = -10e8
        ridx = np.argsort(-sc, axis=1, kind='stable')[:, :k]
        rval = np.take_along_axis(sc, ridx, 1)
        if masked:
            idx, val = ops.score_mask_topk(T(Pu), T(Pi), k, T(rp), T(mc))
        else:
            idx, val = ops.score_mask_topk(T(Pu), T(Pi), k)
        idx, val = idx.cpu().numpy(), val.cpu().numpy()
        assert rel_err(val, rval.astype(np.float32)) < 1e-5
        assert (val[:, :-1] >= val[:, 1:]).all()
        same = idx == ridx
        assert same.mean() > 0.999
        for r, c in np.argwhere(~same):                  # differences are near-ties of the split-bf16 contraction
            assert abs(rval[r, c] - val[r, c]) <= 1e-5 * max(1.0, abs(rval[r, c]))
        ex_i, ex_v = ops.score_mask_topk(T(Pu), T(Pi), k, T(rp) if masked else None, T(mc) if masked else None, exact=True)
        assert rel_err(ex_v.cpu().numpy(), rval.astype(np.float32)) < 1e-5 and (ex_i.cpu().numpy() == ridx).mean() > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize('d,k,masked', [(64, 50, False), (64, 50, True), (128, 50, False)])
def test_score_mask_topk_high_piece_bound_worst_case(ops, d, k, masked):
    """The fp16 matrix path streams scores of the HIGH pieces only and filters against threshold - E (E = 1.05 * 2^-10 |a| |b|), then scores
    every queued candidate with all three products (arl_kernels.hip, REFINE).  Worst case for that bound: every element sits just under the
    midpoint between two fp16 values (its high piece rounds down by almost half an ulp) and all signs agree, so the high-piece score of every
    pair is LOW by almost 2^-10 of the score, coherently -- an item that beats a user's k-th best by less than that has a high-piece score
    below the threshold.  A bound that is too small drops such items; the lists must still be those of the float64 scores."""
    rng = np.random.default_rng(77 * d + k + masked)
    U, I = 300, 40000

    def worst(shape, lo_exp):
        j = rng.integers(0, 1024, size=shape).astype(np.float64)
        e = rng.integers(lo_exp, 1, size=shape).astype(np.float64)
        return ((1.0 + (j + 0.4995) / 1024.0) * 2.0 ** e).astype(np.float32)       # mantissa just below a rounding midpoint of the 11-bit grid

    Pu, Pi = worst((U, d), -2), worst((I, d), -3)
    Pi *= (2.0 ** rng.integers(-2, 1, size=(I, 1))).astype(np.float32)             # a spread of norms (powers of two keep the mantissas)
    scores = Pu.astype(np.float64) @ Pi.astype(np.float64).T
    rp = mc = None
    if masked:
        cols = [np.unique(np.concatenate([np.argsort(-scores[u])[:int(rng.integers(0, 30))], rng.choice(I, size=20, replace=False)])).astype(np.int32) for u in range(U)]
        for u in range(U):
            scores[u, cols[u]] = -10e8
        rp = T(np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32))
        mc = T(np.concatenate(cols))
    ridx = np.argsort(-scores, axis=1, kind='stable')[:, :k]
    rval = np.take_along_axis(scores, ridx, 1)
    first = None
    for order in ('norm', None):
        idx, val = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, item_order=order)
        if first is None:
            first = (idx, val)
        else:
            assert torch.equal(idx, first[0]) and torch.equal(val, first[1])
        idx, val = idx.cpu().numpy(), val.cpu().numpy()
        assert rel_err(val, rval.astype(np.float32)) < 1e-5
        same = idx == ridx
        assert same.mean() > 0.995, same.mean()                                     # (dense scores: many last-ulp near-ties)
        for r, c in np.argwhere(~same):
            assert abs(rval[r, c] - val[r, c]) <= 2e-6 * abs(rval[r, c])            # a mismatch is a near-tie, never a dropped item
        assert np.abs(val[:, -1] - rval[:, -1]).max() <= 2e-6 * np.abs(rval[:, -1]).max()      # every user's k-th best score: nothing above it was lost
    wi, wv = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, warm_idx=first[0])
    assert torch.equal(wi, first[0]) and torch.equal(wv, first[1])


@pytest.mark.parametrize('d,k,masked', [(64, 50, False), (64, 50, True), (128, 20, True), (64, 64, True)])
def test_score_mask_topk_item_stream_order_is_result_neutral(ops, d, k, masked):
    """item_order only changes the ORDER in which the items are scored (default on long streams: descending row norm, so thresholds rise
    early); masks, ids, values and tie order are those of the table order.  Same ids and the same value bits as the table-order call for
    the norm order, a random permutation and its combination with a warm start; users whose interacted items sit inside the permuted
    bootstrap sample keep a valid bound (the sample's membership goes through the inverse permutation)."""
    rng = np.random.default_rng(31 * d + k)
    U, I = 300, 40000
    Pu = (rng.normal(size=(U, d)) * 0.1).astype(np.float32)
    Pi = (rng.normal(size=(I, d)) * 0.1 * rng.uniform(0.5, 2.0, size=(I, 1))).astype(np.float32)      # a spread of norms
    Pi[7] = Pi[11]                                        # two identical items: an exact tie in every user's scores
    norm_order = np.argsort(-np.linalg.norm(Pi, axis=1), kind='stable')
    rp = mc = None
    if masked:
        cols = []
        for u in range(U):
            n_in = int(rng.integers(0, 40))               # interacted items: many of them among the largest-norm items = the permuted sample
            c = np.concatenate([rng.choice(norm_order[:4096], size=n_in, replace=False), rng.choice(I, size=int(rng.integers(0, 30)), replace=False)])
            cols.append(np.unique(c).astype(np.int32))
        rp = T(np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32))
        mc = T(np.concatenate(cols + [np.zeros(1, np.int32)])[:max(sum(len(c) for c in cols), 1)])
    base_i, base_v = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, item_order=None)
    for order in ('norm', T(rng.permutation(I).astype(np.int32))):
        i2, v2 = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, item_order=order)
        assert torch.equal(i2, base_i) and torch.equal(v2, base_v)
        i3, v3 = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, item_order=order, warm_idx=base_i)
        assert torch.equal(i3, base_i) and torch.equal(v3, base_v)
    with pytest.raises(ValueError):
        ops.score_mask_topk(T(Pu), T(Pi), k, item_order=T(np.arange(5, dtype=np.int32)))


@pytest.mark.parametrize('d,k,masked', [(64, 50, False), (64, 50, True), (128, 20, False)])
def test_score_mask_topk_early_exit_is_exact_and_one_user_keeps_the_stream_alive(ops, d, k, masked):
    """Exact early exit of the norm-ordered item stream (arl_kernels.hip: EXIT): a workgroup stops once |a_u| * (largest later item norm) is below
    every one of its users' k-th best scores (Cauchy-Schwarz).  Items here fall steeply in norm and point along +e1, users along +e1: every
    list is complete after the first few stages and most of the stream is skipped.  ONE user of the first workgroup points along -e1: its best
    items are the 100 SMALLEST-norm items at the very end of the stream, so that workgroup must stream to the end -- and the user's list must
    hold exactly those items.  Lists and values equal the table-order call's (no exit there) bit for bit and float64's."""
    rng = np.random.default_rng(4100 + d + k + masked)
    U, I = 700, 66000
    e1 = np.zeros(d); e1[0] = 1.0
    norms = np.concatenate([np.geomspace(8.0, 0.05, I - 100), np.full(100, 1e-3)])
    sign = np.concatenate([np.ones(I - 100), -np.ones(100)])
    Pi = (norms[:, None] * (sign[:, None] * e1[None, :] + 0.05 * rng.standard_normal((I, d)))).astype(np.float32)
    Pi = Pi[rng.permutation(I)]                                  # table order is not norm order
    Pu = (e1[None, :] + 0.05 * rng.standard_normal((U, d))).astype(np.float32)
    lone = 37
    Pu[lone] = (-e1 + 0.01 * rng.standard_normal(d)).astype(np.float32)
    scores = Pu.astype(np.float64) @ Pi.astype(np.float64).T
    rp = mc = None
    if masked:
        cols = [np.unique(np.concatenate([np.argsort(-scores[u])[:int(rng.integers(0, 20))], rng.choice(I, size=10, replace=False)])).astype(np.int32) for u in range(U)]
        for u in range(U):
            scores[u, cols[u]] = -10e8
        rp = T(np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32)); mc = T(np.concatenate(cols))
    ridx = np.argsort(-scores, axis=1, kind='stable')[:, :k]
    rval = np.take_along_axis(scores, ridx, 1)
    small = set(np.nonzero(np.linalg.norm(Pi, axis=1) < 5e-3)[0].tolist())
    assert set(ridx[lone].tolist()) <= small                      # the lone user's list = items of the stream's last two stages
    ops.TOPK_STATS['record_exit'] = True; ops.TOPK_STATS['exit'] = []
    ops.reset_exit_probe()
    try:
        base_i, base_v = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, item_order=None)        # table order: suffix maxima stay high, nothing is skipped
        i2, v2 = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, item_order='norm')
        i3, v3 = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, item_order='norm', warm_idx=base_i)
        Pu2 = Pu.copy(); Pu2[lone] = Pu[lone + 1]
        ops.score_mask_topk(T(Pu2), T(Pi), k, rp, mc, item_order='norm')                       # the same call without the lone user
        frac = ops.topk_exit_fractions()
    finally:
        ops.TOPK_STATS['record_exit'] = False
    assert torch.equal(i2, base_i) and torch.equal(v2, base_v) and torch.equal(i3, base_i) and torch.equal(v3, base_v)
    idx, val = i2.cpu().numpy(), v2.cpu().numpy()
    assert set(idx[lone].tolist()) == set(ridx[lone].tolist())
    same = idx == ridx
    assert same.mean() > 0.999
    for r, c in np.argwhere(~same):
        assert abs(rval[r, c] - val[r, c]) <= 2e-6 * max(abs(rval[r, c]), 1e-3)
    assert np.abs(val[:, -1] - rval[:, -1]).max() <= 2e-6 * np.abs(rval[:, -1]).max()
    n_wg = -(-U // (512 if d == 64 else 192))                      # users per workgroup: the second form at d = 64 (512), the first at d = 128
    assert frac[0] < 0.01                                          # table order: large items until the last stages, nothing to skip
    # norm order with the lone user: its workgroup consumes every stage, the others leave early; without it all of them do
    assert frac[1] > 0.5 * (n_wg - 1) / n_wg and frac[1] < (n_wg - 1) / n_wg + 1e-9, frac
    assert frac[3] > frac[1] + 0.5 / n_wg and frac[3] > 0.8, frac


@pytest.mark.parametrize('I', [5000, 40000])
@pytest.mark.parametrize('k', [1, 50, 64])
@pytest.mark.parametrize('masked', [False, True])
def test_score_mask_topk_second_form_equals_first_form(ops, I, k, masked):
    """The two forms of the fp16 split stream at d = 64 (arl_kernels.hip: score_mask_topk_mfma16_kernel / topk2_main_kernel; the second needs the user
    workspace): 32 users per wave on 32 x 32 tiles with the items as rows, bit-mask pre-filter, counter ring of three tiles, lists kept in the outputs,
    bootstrap by group maxima, pipelined merges -- same candidates' exact three-product scores, same keys: indices AND values bit-identical, cold and
    warm-started (also from deliberately stale candidates), masked or not, with and without the bootstrap phase (I >= 32 768), a ragged last workgroup
    (1 300 users = 2 x 512 + 276), and popularity-skewed norms so that thresholds differ wildly between users."""
    rng = np.random.default_rng(9100 + I + k + masked)
    U, d = 1300, 64
    Pu = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Pi = (rng.standard_normal((I, d)) * 0.1 * (rng.pareto(2.0, I) + 0.1)[:, None]).astype(np.float32)
    rp = mc = None
    if masked:
        sc = Pu @ Pi.T
        cols = [np.unique(np.concatenate([np.argsort(-sc[u])[:int(rng.integers(0, 40))], rng.choice(I, size=int(rng.integers(0, 30)), replace=False)])).astype(np.int32) for u in range(U)]
        rp = T(np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32)); mc = T(np.concatenate(cols) if sum(len(c) for c in cols) else np.zeros(0, np.int32))
    out = {}
    for form2 in (False, True):
        ops.TOPK_FORM2 = form2
        ops.reset_exit_probe()
        try:
            i_c, v_c = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc)
            i_w, v_w = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, warm_idx=i_c)
            stale = torch.from_numpy(np.stack([rng.choice(I, size=k, replace=False) for _ in range(U)]).astype(np.int32)).cuda()      # random candidates: valid but useless thresholds
            i_s, v_s = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, warm_idx=stale)
            i_t, v_t = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, item_order=None)
        finally:
            ops.TOPK_FORM2 = True
        out[form2] = (i_c, v_c)
        for i_x, v_x in ((i_w, v_w), (i_s, v_s), (i_t, v_t)):
            assert torch.equal(i_x, i_c) and torch.equal(v_x, v_c)
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    ref = torch.topk((T(Pu).double() @ T(Pi).double().T) if not masked else _masked_scores(Pu, Pi, rp, mc), k, dim=1)
    got_v = out[True][1].double()
    assert (got_v - ref.values).abs().max().item() <= 2e-6 * max(1e-3, ref.values.abs().max().item())      # and float64's values (at least k unmasked items everywhere)


def _masked_scores(Pu, Pi, rp, mc):
    sc = T(Pu).double() @ T(Pi).double().T
    rows = torch.repeat_interleave(torch.arange(sc.shape[0], device=sc.device), (rp[1:] - rp[:-1]).long())
    sc[rows, mc.long()] = -10e8
    return sc


@pytest.mark.parametrize('F,I,d', [(64, 100000, 64), (5, 777, 32), (130, 301, 128), (64, 1000, 16), (1, 4, 4), (7, 12345, 256)])
def test_fake_block_products(ops, F, I, d):
    """The F x I fake-user block of the poisoned adjacency as two dense products (attack/White/PGA.py:118-134) against float64."""
    g = torch.Generator().manual_seed(F * 1000 + I)
    S = torch.rand(F, I, generator=g)
    S[S < 0.3] = 0.0
    X = torch.randn(I, d, generator=g); Xf = torch.randn(F, d, generator=g)
    rs_f = torch.rand(F, generator=g) + 0.1; rs_i = torch.rand(I, generator=g) + 0.1
    Y0f = torch.randn(F, d, generator=g); Y0i = torch.randn(I, d, generator=g)
    ref_rows = Y0f.double() + 0.7 * rs_f.double()[:, None] * (S.double() @ X.double())
    ref_cols = Y0i.double() - 1.3 * rs_i.double()[:, None] * (S.double().t() @ Xf.double())
    Yf, Yi = Y0f.cuda(), Y0i.cuda()
    ops.fake_block_rows_(S.cuda(), X.cuda(), Yf, rscale=rs_f.cuda(), alpha=0.7)
    ops.fake_block_cols_(S.cuda(), Xf.cuda(), Yi, rscale=rs_i.cuda(), alpha=-1.3)
    assert rel_err(Yf.cpu().numpy(), ref_rows.numpy()) < 2e-6
    assert rel_err(Yi.cpu().numpy(), ref_cols.numpy()) < 2e-6
    # no scale vector, deterministic (two calls give the same bits)
    A = torch.zeros(F, d, device='cuda'); B = torch.zeros(F, d, device='cuda')
    ops.fake_block_rows_(S.cuda(), X.cuda(), A); ops.fake_block_rows_(S.cuda(), X.cuda(), B)
    assert torch.equal(A, B) and rel_err(A.cpu().numpy(), (S.double() @ X.double()).numpy()) < 2e-6
    with pytest.raises(ValueError):
        ops.fake_block_rows_(S.cuda(), X.cuda()[:-1].contiguous(), Yf)


def test_tables_sum(ops):
    """alpha * (t0 + ... + tk) in one pass (the layer mean of LightGCN.py:236-240), also in place."""
    g = torch.Generator().manual_seed(3)
    ts = [torch.randn(1001, 64, generator=g) for _ in range(4)]
    ref = 0.25 * (((ts[0] + ts[1]) + ts[2]) + ts[3])
    dts = [t.cuda() for t in ts]
    assert torch.equal(ops.tables_sum(dts, 0.25).cpu(), ref)
    assert torch.equal(ops.tables_sum(dts[:1], 2.0).cpu(), 2.0 * ts[0])
    out = ops.tables_sum(dts, 0.25, out=dts[0])
    assert out is dts[0] and torch.equal(out.cpu(), ref)
    with pytest.raises(ValueError):
        ops.tables_sum([dts[1], dts[2][:-1].contiguous()])


@pytest.mark.parametrize('d', [64, 16, 128, 200])
def test_simgcl_perturb_rng(ops, d):
    """SimGCL.py:203-205 with the noise drawn in the kernel: sign-aligned, every row moved by exactly eps, uniform-looking directions,
    reproducible per (seed, stream), and a compact slice receives the noise of the same rows of a full-table call."""
    g = torch.Generator().manual_seed(d)
    n, eps = 5000, 0.1
    src = torch.randn(n, d, generator=g).cuda()
    src[7] = 0.0                                             # sign(0) = 0: the row stays put
    a = ops.simgcl_perturb_rng(src, eps, 1234, 5)
    delta = (a - src).double()
    assert bool((delta * torch.sign(src).double() >= 0).all())
    nrm = delta.norm(dim=1)
    keep = torch.ones(n, dtype=torch.bool, device='cuda'); keep[7] = False
    assert float((nrm[keep] - eps).abs().max()) < 1e-6 and float(nrm[7]) == 0.0
    u = (delta.abs() / eps)[keep]                            # = u / ||u||, u ~ U[0,1)^d: mean component about sqrt(3)/2 / sqrt(d)
    assert abs(float(u.mean()) * d ** 0.5 - 0.866) < 0.02
    assert torch.equal(a, ops.simgcl_perturb_rng(src, eps, 1234, 5))                       # reproducible
    assert not torch.equal(a, ops.simgcl_perturb_rng(src, eps, 1234, 6)) and not torch.equal(a, ops.simgcl_perturb_rng(src, eps, 1235, 5))
    sel = torch.tensor([3, 4999, 17, 17, 0], dtype=torch.int32, device='cuda')
    part = ops.simgcl_perturb_rng(src[sel.long()].contiguous(), eps, 1234, 5, row_ids=sel)
    assert torch.equal(part, a[sel.long()])
    b = src.clone(); ops.simgcl_perturb_rng(b, eps, 1234, 5, out=b)                      # in place
    assert torch.equal(a, b)


def test_topn_project_rows(ops):
    rng = np.random.default_rng(6)
    M = rng.random((7, 1412)).astype(np.float32)
    M[2, :] = 1e-7; M[2, 10] = 1.0                                   # massive ties (PGA's clamp floor): lower index first
    for n in (0, 1, 46):
        ro, ri = O.topn_project_rows(M, n) if n else (np.zeros_like(M), np.zeros((7, 0), np.int32))
        out, idx = ops.topn_project_rows(T(M), n)
        assert np.array_equal(out.cpu().numpy(), ro) and np.array_equal(idx.cpu().numpy(), ri)


def test_missing_gpu_tensor_fails_loudly(ops):
    from arlib_amd._lib import ArlError
    with pytest.raises(ArlError):
        ops.sgd_dense(torch.zeros(4), torch.zeros(4), 0.1)


@pytest.mark.parametrize('d,masked', [(64, True), (64, False), (32, True), (128, False)])
def test_score_mask_topk_warm_start_is_result_neutral(ops, d, masked):
    """A warm start only pre-sets thresholds: same lists as the cold call, whether the candidates are last step's result, a
    poor guess, or invalid (a masked candidate -> underflow -> automatic cold repeat)."""
    rng = np.random.default_rng(d + masked)
    U, I, k = 700, 3000, 50
    Pu = (rng.normal(size=(U, d)) * 0.1).astype(np.float32)
    Pi = (rng.normal(size=(I, d)) * 0.1).astype(np.float32)
    rp = mc = None
    if masked:
        lens = rng.integers(0, 60, U)
        cols = [np.sort(rng.choice(I, n, replace=False)).astype(np.int32) for n in lens]
        rp = T(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)); mc = T(np.concatenate(cols + [np.zeros(1, np.int32)])[:max(int(lens.sum()), 1)])
    cold_i, cold_v = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc)
    # (a) previous step's lists after a small move of the tables
    Pu2 = Pu + (rng.normal(size=Pu.shape) * 1e-3).astype(np.float32)
    ref_i, ref_v = ops.score_mask_topk(T(Pu2), T(Pi), k, rp, mc)
    w_i, w_v = ops.score_mask_topk(T(Pu2), T(Pi), k, rp, mc, warm_idx=cold_i)
    assert torch.equal(w_i, ref_i) and torch.equal(w_v, ref_v)
    # (b) a poor guess: arbitrary distinct items (lower thresholds, still valid)
    guess = T(np.stack([rng.permutation(I)[:k] for _ in range(U)]).astype(np.int32))
    if masked:                                   # keep the guess unmasked: drop it to the cold lists where it collides
        guess = cold_i.clone()
        guess[:, ::2] = cold_i.flip(1)[:, ::2]   # same sets, other order
    g_i, g_v = ops.score_mask_topk(T(Pu2), T(Pi), k, rp, mc, warm_idx=guess)
    assert torch.equal(g_i, ref_i) and torch.equal(g_v, ref_v)
    # (c) invalid candidates (a repeated item): the bound over-excludes, the call repeats itself cold (second launch, gated on the flag on the device)
    bad = cold_i.clone(); bad[:, 1:] = bad[:, :1]
    b_i, b_v = ops.score_mask_topk(T(Pu2), T(Pi), k, rp, mc, warm_idx=bad)
    assert torch.equal(b_i, ref_i) and torch.equal(b_v, ref_v)


def test_bipartite_graph_device_build_matches_golden_and_oracle(ops, ml100k):
    """ops.bipartite_graph (what every attack's _init_uiAdj goes through, and the array-native DataLoader's norm_adj): pattern and values
    against the reference's normalize_graph_mat on ml-100k (g3_adj) and against its `_init_uiAdj` on a weighted graph with an isolated
    node, plus the oracle's own build on a random weighted graph."""
    g = golden('g3_adj.npz')
    p = ml100k['pairs0']
    o = np.lexsort((p[:, 1], p[:, 0]))
    U, I = ml100k['U'], ml100k['I']
    gr = ops.bipartite_graph(T(p[o, 0].astype(np.int64)), T(p[o, 1].astype(np.int64)), U, I)
    assert np.array_equal(gr.rowptr.cpu().numpy(), g['norm_indptr']) and np.array_equal(gr.col.cpu().numpy(), g['norm_indices'])
    assert rel_err(gr.val.cpu().numpy(), g['norm_data']) < 1e-6
    Uw, Iw = (int(x) for x in g['w_shape'])
    o = np.lexsort((g['w_R_col'], g['w_R_row']))
    gw = ops.bipartite_graph(T(g['w_R_row'][o].astype(np.int64)), T(g['w_R_col'][o].astype(np.int64)), Uw, Iw, weights=T(g['w_R_val'][o]))
    rows = np.repeat(np.arange(Uw + Iw), np.diff(gw.rowptr.cpu().numpy()))
    ref = sp_coo(g['w_norm_val'], g['w_norm_row'], g['w_norm_col'], Uw + Iw)
    got = sp_coo(gw.val.cpu().numpy(), rows, gw.col.cpu().numpy(), Uw + Iw)
    assert abs(got - ref).max() <= 1e-5 * abs(ref).max() and (got != 0).nnz == (ref != 0).nnz
    assert gw.dinv[7].item() == 0.0
    rng = np.random.default_rng(8)
    u, i = random_graph(rng, 500, 90, 7, hot_items=1, hot_deg=300, empty_users=(3,))
    w = (rng.random(len(u)) + 0.25).astype(np.float32)
    rowptr, col, ww = O.bipartite_csr(u, i, 500, 90, w)
    gr = ops.bipartite_graph(T(u.astype(np.int64)), T(i.astype(np.int64)), 500, 90, weights=T(w))
    assert np.array_equal(gr.rowptr.cpu().numpy(), rowptr) and np.array_equal(gr.col.cpu().numpy(), col)
    assert rel_err(gr.val.cpu().numpy(), O.norm_adj_values(rowptr, col, ww)) < 1e-6


def sp_coo(v, r, c, n):
    import scipy.sparse as sp
    return sp.csr_matrix((np.asarray(v, np.float64), (np.asarray(r), np.asarray(c))), shape=(n, n))


@pytest.mark.parametrize('planned', [False, True])
def test_incremental_bipartite_update_equals_fresh_build(ops, planned, monkeypatch):
    """ops.IncrementalBipartite (real block fixed, fake users' rows replaced per attack epoch; the CSR merged and the blocked plan patched on
    the device) against a fresh ops.bipartite_graph of the stacked interactions: identical pattern, bit-identical values, and -- with the
    patched plan -- the same products as the CSR kernel, for several successive fake blocks (binary, weighted, empty, a fake user on a hot item)."""
    if planned:
        monkeypatch.setattr(ops, 'BLOCKED_MIN_NNZ', 0); monkeypatch.setattr(ops, 'BLOCKED_MIN_WAVES', 0)
    rng = np.random.default_rng(21)
    U, F, I, d = 3000, 7, 400, 64
    u, i = random_graph(rng, U, I, 10, hot_items=2, hot_deg=2500, empty_users=(11,))
    inc = ops.IncrementalBipartite(T(u.astype(np.int64)), T(i.astype(np.int64)), U, F, I, DEV, emb_size=d)
    assert (inc.base is not None) == planned
    Up, N = U + F, U + F + I
    X = torch.randn(N, d, generator=torch.Generator().manual_seed(3)).to(DEV)
    for trial in range(4):
        if trial == 2:
            fu = np.zeros(0, np.int64); fi = np.zeros(0, np.int64); fw = None
        else:
            n_f = [30, 5, 0, 120][trial]
            fu = np.repeat(np.arange(F), n_f); fi = np.concatenate([np.sort(rng.choice(I, n_f, replace=False)) for _ in range(F)])
            if trial == 3:
                fi[0] = 0 if fi[0] != 0 and 0 not in fi[:n_f] else fi[0]                 # a fake edge on the hottest (split) item row
                fi[:n_f] = np.sort(fi[:n_f])
            fw = None if trial != 1 else (rng.random(len(fu)) + 0.5).astype(np.float32)
        g = inc.update(T(fu), T(fi), None if fw is None else T(fw))
        au = np.concatenate([u.astype(np.int64), U + fu]); ai = np.concatenate([i.astype(np.int64), fi])
        aw = None if fw is None else np.concatenate([np.ones(len(u), np.float32), fw])
        o = np.lexsort((ai, au))
        ref = ops.bipartite_graph(T(au[o]), T(ai[o]), Up, I, weights=None if aw is None else T(aw[o]))
        assert torch.equal(g.rowptr, ref.rowptr) and torch.equal(g.col, ref.col) and torch.equal(g.val, ref.val) and torch.equal(g.dinv, ref.dinv)
        assert (g.blocked is not None) == planned
        y_ref = ops.spmm(ref, X)
        assert rel_err(ops.spmm(g, X).cpu().numpy(), y_ref.cpu().numpy()) < 1e-5          # blocked vs CSR schedule: different summation order
        if planned:
            assert sum(s['n_edges'] for s in g.blocked.sets) == g.nnz
            Z = torch.randn(N, d, generator=torch.Generator().manual_seed(4)).to(DEV)
            assert rel_err(ops.spmm(g, X, 0.5, -2.0, Z).cpu().numpy(), ops.spmm(ref, X, 0.5, -2.0, Z).cpu().numpy()) < 1e-5


@pytest.mark.parametrize('d,n', [(16, 1000), (32, 517), (64, 4099), (128, 2050), (64, 7)])
def test_ngcf_dense_layer_on_mfma_matches_float64(ops, d, n):
    """arl_ngcf_dense_{fwd,dgrad,wgrad}_f32 (fp32 MFMA, [P + E | P * E] formed in registers) against the layer written out in float64 torch with
    autograd: output, gP, gE and both weight gradients; ragged last row tile; leaky-relu at both signs."""
    g = torch.Generator().manual_seed(d + n)
    P = (torch.randn(n, d, generator=g) * 0.5).to(DEV); E = (torch.randn(n, d, generator=g) * 0.5).to(DEV)
    W = (torch.randn(2 * d, d, generator=g) * 0.2).to(DEV); gOut = torch.randn(n, d, generator=g).to(DEV)
    out = ops.ngcf_dense_fwd(P, E, W, 0.01)
    Pd, Ed, Wd = (t.double().requires_grad_(True) for t in (P, E, W))
    ref = torch.nn.functional.leaky_relu(torch.cat([Pd + Ed, Pd * Ed], 1) @ Wd, 0.01)
    assert rel_err(out.cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5
    assert int((out < 0).sum()) > 0 and int((out > 0).sum()) > 0
    ref.backward(gOut.double())
    gP, gE, gW = ops.ngcf_dense_bwd(gOut, out, P, E, W, 0.01)
    assert rel_err(gP.cpu().numpy(), Pd.grad.cpu().numpy()) < 1e-5 and rel_err(gE.cpu().numpy(), Ed.grad.cpu().numpy()) < 1e-5
    assert rel_err(gW.cpu().numpy(), Wd.grad.cpu().numpy()) < 1e-5
    assert torch.equal(gW, ops.ngcf_dense_bwd(gOut, out, P, E, W, 0.01)[2])            # deterministic partial sums


def test_masked_hop_with_rows_taken_in_length_order(ops):
    """ops.spmm_flagged on a graph with enable_masked_order() (what the engine does on large graphs: row tasks sorted by edge count, one
    (row, begin, end) record per task) -- same numbers as the index-order launch, long rows and empty rows included, also after with_values()."""
    rng = np.random.default_rng(33)
    U, I, d = 5000, 300, 64
    u, i = random_graph(rng, U, I, 9, hot_items=2, hot_deg=1800, empty_users=(4, 5, 4098))
    rowptr, col, w, val = make_csr(u, i, U, I)
    N = U + I
    A1, A2 = ops.CSRGraph(rowptr, col, val, DEV, chunk=128), ops.CSRGraph(rowptr, col, val, DEV, chunk=128).enable_masked_order()
    assert A2._row_tasks is not None and A2._row_tasks.shape == (N, 4) and A2.n_chunks > 0
    rows = torch.from_numpy(rng.choice(N, 500, replace=False).astype(np.int32)).to(DEV)
    G = torch.zeros(N, d, device=DEV); flags = torch.zeros(N, dtype=torch.uint8, device=DEV); bits = torch.zeros((N + 31) // 32, dtype=torch.int32, device=DEV)
    ops.batch_rows_set_(G, flags, bits, rows, torch.randn(500, d, device=DEV))
    y1 = ops.spmm_flagged(A1, G, bits, 0.5, 2.0, G, flags)
    y2 = ops.spmm_flagged(A2, G, bits, 0.5, 2.0, G, flags)
    assert torch.equal(y1, y2)                                                 # same per-row arithmetic, only the task order differs
    assert rel_err(y2.cpu().numpy(), O.spmm((rowptr, col, val), G.cpu().numpy(), 0.5, 2.0, G.cpu().numpy())) < RTOL
    B2 = A2.with_values(T((val * 0.5).astype(np.float32)))
    assert B2._row_tasks is A2._row_tasks and torch.equal(ops.spmm_flagged(B2, G, bits), 0.5 * ops.spmm_flagged(A1, G, bits))
    assert torch.equal(ops.spmm(A2, G), ops.spmm(A1, G))                       # other kernels ignore the table
