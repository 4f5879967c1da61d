"""Pins the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import hashlib
import random
import numpy as np
import pytest
from conftest import golden, rel_err, RTOL
from oracle import oracle as O


def test_mt_matches_cpython_random():
    for seed in (2018, 0, 7, 2**32 + 5, 12345678901234567890):
        st = O.mt_seed(seed)
        random.seed(seed)
        assert tuple(int(x) for x in st) == random.getstate()[1]
        for n in (1, 2, 3, 1000, 44212, 2**31 + 3):
            assert O.mt_randbelow(st, n) == random.randrange(n)
        assert O.mt_random(st) == random.random()


def _epoch(st, pairs, bs, I, memb):
    us, ps, ns, h, nb = [], [], [], hashlib.sha256(), 0
    for u, p, n in O.next_batch_pairwise(st, pairs, bs, I, memb):
        h.update(np.stack([u, p, n]).astype(np.int32).tobytes())
        us.append(u); ps.append(p); ns.append(n); nb += 1
    return np.concatenate(us), np.concatenate(ps), np.concatenate(ns), h.hexdigest(), nb, len(us[-1])


def test_sampler_two_epochs_bit_exact(ml100k):
    g = golden('g1_sampler.npz')
    pairs = ml100k['pairs0'].copy()
    memb = O.build_membership(pairs, ml100k['U'])
    st = O.mt_seed(2018)
    for ep in range(2):
        u, p, n, sha, nb, last = _epoch(st, pairs, 2048, ml100k['I'], memb)
        assert np.array_equal(u, g['ep%d_u' % ep]) and np.array_equal(p, g['ep%d_p' % ep])
        assert np.array_equal(n, g['ep%d_n' % ep])
        assert sha == bytes(g['ep%d_sha' % ep]).decode()
        assert [nb, last] == list(g['ep%d_nb' % ep])
    # SURVEY 8c anchors
    assert bytes(g['ep0_sha']).decode() == '507829bf43d45af1cc7d028d9d591f03c49f16207ef787ea817e64317f9978db'
    assert list(g['ep0_u'][:8]) == [821, 457, 932, 114, 204, 568, 404, 5]
    assert O.mt_random(st) == float(g['next_random'][0])      # after TWO epochs (SURVEY's 0.6449990 was after one)
    # ragged batch size, other seed (continues from the twice-shuffled order: in-place carry-over, Q7)
    st = O.mt_seed(7)
    u, p, n, _, _, _ = _epoch(st, pairs, 1000, ml100k['I'], memb)
    assert np.array_equal(u, g['b1000_u']) and np.array_equal(p, g['b1000_p']) and np.array_equal(n, g['b1000_n'])
    assert O.mt_random(st) == float(g['b1000_next_random'][0])


def test_sampler_heavy_rejection_toy():
    g = golden('g1_sampler_toy.npz')
    U, I, nnz = (int(x) for x in g['sizes'])
    pairs = g['pairs0'].copy()
    memb = O.build_membership(pairs, U)
    st = O.mt_seed(12345678901234567890)
    us, ps, ns = [], [], []
    for ep in range(5):
        for u, p, n in O.next_batch_pairwise(st, pairs, 7, I, memb):
            us.append(u); ps.append(p); ns.append(n)
    assert np.array_equal(np.concatenate(us), g['u']) and np.array_equal(np.concatenate(ps), g['p'])
    assert np.array_equal(np.concatenate(ns), g['n'])
    assert O.mt_random(st) == float(g['next_random'][0])


def test_norm_adj_ml100k(ml100k):
    g = golden('g3_adj.npz')
    p = ml100k['pairs0']
    rowptr, col, w = O.bipartite_csr(p[:, 0], p[:, 1], ml100k['U'], ml100k['I'])
    assert np.array_equal(rowptr, g['norm_indptr']) and np.array_equal(col, g['norm_indices'])
    val = O.norm_adj_values(rowptr, col, w)
    assert rel_err(val, g['norm_data']) < 1e-6
    assert list(col[:5]) == [942, 947, 948, 959, 972] and abs(val[0] - 0.021995295) < 1e-8   # SURVEY 8c anchor


def test_init_uiadj_weighted_with_isolated_node():
    g = golden('g3_adj.npz')
    U, I = (int(x) for x in g['w_shape'])
    rowptr, col, w = O.bipartite_csr(g['w_R_row'], g['w_R_col'], U, I, g['w_R_val'])
    val = O.norm_adj_values(rowptr, col, w)
    rows = np.repeat(np.arange(U + I), np.diff(rowptr))
    assert np.array_equal(rows, g['w_norm_row']) and np.array_equal(col, g['w_norm_col'])
    assert rowptr[8] == rowptr[7]                       # isolated user 7 has no entries (and no NaN)
    assert np.allclose(val, g['w_norm_val'], rtol=1e-5, atol=0)


@pytest.mark.parametrize('tag', ['n', 'sat'])
def test_bpr_l2_fwd_bwd(tag):
    g = golden('g2_losses.npz')
    u, p, n = g[tag + '_u'], g[tag + '_p'], g[tag + '_n']
    B = len(u)
    emb = np.concatenate([u, p, n], 0)
    lb, lr_, G = O.bpr_l2(emb, B, np.arange(B), np.arange(B), np.arange(B, 2 * B), 1e-4)
    assert abs(lb - g[tag + '_bpr'][0]) <= RTOL * abs(g[tag + '_bpr'][0])
    assert abs(lr_ - g[tag + '_reg'][0]) <= RTOL * abs(g[tag + '_reg'][0])
    assert rel_err(G[:B], g[tag + '_du']) < RTOL
    assert rel_err(G[B:2 * B], g[tag + '_dp']) < RTOL
    assert rel_err(G[2 * B:], g[tag + '_dn']) < RTOL


def test_bpr_l2_duplicate_indices_scatter_add():
    g = golden('g2_losses.npz')
    lb, lr_, G = O.bpr_l2(g['dup_T'], 0, g['dup_ui'], g['dup_pi'], g['dup_ni'], 1e-4)
    assert abs(lb + lr_ - g['dup_loss'][0]) <= RTOL * abs(g['dup_loss'][0])
    assert rel_err(G, g['dup_dT']) < RTOL


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_infonce(tag):
    g = golden('g6_infonce.npz')
    loss, d1, d2 = O.infonce(g[tag + '_v1'], g[tag + '_v2'], 0.2)
    assert abs(loss - g[tag + '_loss'][0]) <= RTOL * abs(g[tag + '_loss'][0])
    assert rel_err(d1, g[tag + '_dv1']) < RTOL and rel_err(d2, g[tag + '_dv2']) < RTOL


def _ml100k_csr(ml100k):
    p = ml100k['pairs0']
    rowptr, col, w = O.bipartite_csr(p[:, 0], p[:, 1], ml100k['U'], ml100k['I'])
    return rowptr, col, O.norm_adj_values(rowptr, col, w)


@pytest.mark.parametrize('L', [1, 2, 3])
def test_lightgcn_forward(ml100k, L):
    g = golden('g4_forward.npz')
    csr = _ml100k_csr(ml100k)
    out = O.lightgcn_forward(csr, np.concatenate([g['lgn_user0'], g['lgn_item0']]), L)
    U = ml100k['U']
    assert rel_err(out[:U], g['lgn_L%d_user' % L]) < RTOL and rel_err(out[U:], g['lgn_L%d_item' % L]) < RTOL


def _run_steps(ml100k, g, L, lr, opt, snaps, with_csr=True):
    csr = _ml100k_csr(ml100k) if with_csr else None
    st = O.TrainState(g['user0'], g['item0'], csr, L, 1e-4, lr, optimizer=opt)
    U = ml100k['U']
    off = np.concatenate([[0], np.cumsum(g['batch_sizes'])])
    for k in range(len(g['batch_sizes'])):
        sl = slice(off[k], off[k + 1])
        if k == 0:
            loss, grad = st.grad(g['batch_u'][sl], g['batch_p'][sl], g['batch_n'][sl])
            assert rel_err(grad[:U], g['grad_user_step0']) < RTOL and rel_err(grad[U:], g['grad_item_step0']) < RTOL
        loss = st.step(g['batch_u'][sl], g['batch_p'][sl], g['batch_n'][sl])
        assert abs(loss - g['losses'][k]) <= RTOL * abs(g['losses'][k])
        if (k + 1) in snaps:
            assert rel_err(st.E0[:U], g['user_k%d' % (k + 1)]) < RTOL
            assert rel_err(st.E0[U:], g['item_k%d' % (k + 1)]) < RTOL
    return st


def test_lightgcn_adam_10_steps(ml100k):
    g = golden('g5_lightgcn_adam.npz')
    st = _run_steps(ml100k, g, 3, 0.005, 'adam', {1, 3, 10})
    U = ml100k['U']
    assert rel_err(st.m[:U], g['m_user']) < RTOL and rel_err(st.v[U:], g['v_item']) < RTOL
    assert abs(g['losses'][0] - 0.6937738) < 1e-5      # SURVEY 8c anchor


def test_gmf_adam_25_steps_crosses_epoch(ml100k):
    g = golden('g5_gmf_adam.npz')
    st = _run_steps(ml100k, g, 0, 0.005, 'adam', {3, 25}, with_csr=False)
    U = ml100k['U']
    assert rel_err(st.m[U:], g['m_item']) < RTOL and rel_err(st.v[:U], g['v_user']) < RTOL


def test_lightgcn_sgd_3_steps(ml100k):
    g = golden('g5_lightgcn_sgd.npz')
    _run_steps(ml100k, g, 2, 0.0005, 'sgd', {3})


def test_simgcl_forward_and_step(ml100k):
    g = golden('g5_simgcl.npz')
    csr = _ml100k_csr(ml100k)
    U = ml100k['U']
    E0 = np.concatenate([g['user0'], g['item0']])
    out = O.lightgcn_forward(csr, E0, 2, skip0=True)
    assert rel_err(out[:U], g['fwd_user']) < RTOL and rel_err(out[U:], g['fwd_item']) < RTOL
    outp = O.lightgcn_forward(csr, E0, 2, skip0=True, noises=g['noise'][:2], eps=0.1)
    assert rel_err(outp[:U], g['fwdp_user']) < RTOL and rel_err(outp[U:], g['fwdp_item']) < RTOL
    # full step: rec loss + 0.2*(InfoNCE users + InfoNCE items) (recommender/SimGCL.py:51-63,212-219)
    bu, bp, bn = g['batch_u'], g['batch_p'], g['batch_n']
    lb, lr_, G = O.bpr_l2(out, U, bu, bp, bn, 1e-4)
    assert abs(lb - g['rec_loss'][0]) <= RTOL * abs(g['rec_loss'][0])
    grad = O.lightgcn_backward(csr, G, 2, skip0=True)
    v1 = O.lightgcn_forward(csr, E0, 2, skip0=True, noises=g['noise'][0:2])
    v2 = O.lightgcn_forward(csr, E0, 2, skip0=True, noises=g['noise'][2:4])
    uu = np.unique(bu); ii = np.unique(bp) + U
    cl = 0.0
    Gv1 = np.zeros_like(E0); Gv2 = np.zeros_like(E0)
    for idx in (uu, ii):
        l, d1, d2 = O.infonce(v1[idx], v2[idx], 0.2)
        cl += l
        Gv1[idx] += 0.2 * d1; Gv2[idx] += 0.2 * d2
    assert abs(0.2 * cl - g['cl_loss'][0]) <= RTOL * abs(g['cl_loss'][0])
    grad = grad + O.lightgcn_backward(csr, Gv1, 2, skip0=True) + O.lightgcn_backward(csr, Gv2, 2, skip0=True)
    assert rel_err(grad[:U], g['grad_user']) < RTOL and rel_err(grad[U:], g['grad_item']) < RTOL
    m = np.zeros_like(E0); v = np.zeros_like(E0); E = E0.copy()
    O.adam_step(E, grad, m, v, 0.005, 1)
    assert rel_err(E[:U], g['user_k1']) < RTOL and rel_err(E[U:], g['item_k1']) < RTOL


def test_ngcf_forward(ml100k):
    g = golden('g9_ngcf.npz')
    csr = _ml100k_csr(ml100k)
    out = O.ngcf_forward(csr, np.concatenate([g['user0'], g['item0']]), [g['w1_0'], g['w1_1']], [g['w2_0'], g['w2_1']])
    U = ml100k['U']
    assert rel_err(out[:U], g['fwd_user']) < RTOL and rel_err(out[U:], g['fwd_item']) < RTOL


def test_xsimgcl_forward_and_step(ml100k):
    """XSimGCL restated with the oracle's pieces against the reference's own iteration (g11, injected noise)."""
    g = golden('g11_xsimgcl.npz')
    csr = _ml100k_csr(ml100k)
    U = ml100k['U']
    L, lc, cl_rate, eps, temp = int(g['hyper'][0]), int(g['hyper'][1]), float(g['hyper'][2]), float(g['hyper'][3]), float(g['hyper'][4])
    E0 = np.concatenate([g['user0'], g['item0']])
    out = O.lightgcn_forward(csr, E0, L, skip0=True)
    assert rel_err(out[:U], g['fwd_user']) < RTOL and rel_err(out[U:], g['fwd_item']) < RTOL
    mean, layers = O.lightgcn_forward(csr, E0, L, skip0=True, noises=g['noise'], eps=eps, return_layers=True)
    cl = layers[lc]
    assert rel_err(mean[:U], g['fwdp_user']) < RTOL and rel_err(mean[U:], g['fwdp_item']) < RTOL
    assert rel_err(cl[:U], g['cl_user']) < RTOL and rel_err(cl[U:], g['cl_item']) < RTOL
    bu, bp, bn = g['batch_u'], g['batch_p'], g['batch_n']
    lb, lr_, G = O.bpr_l2(mean, U, bu, bp, bn, 1e-4)                          # BPR + L2 read the PERTURBED mean
    assert abs(lb - g['rec_loss'][0]) <= RTOL * abs(g['rec_loss'][0])
    G = G.astype(np.float64); Gcl = np.zeros_like(G)
    closs = 0.0
    for idx in (np.unique(bu), np.unique(bp) + U):
        l, d1, d2 = O.infonce(mean[idx], cl[idx], temp)
        closs += l
        G[idx] += cl_rate * d1; Gcl[idx] += cl_rate * d2
    assert abs(cl_rate * closs - g['cl_loss'][0]) <= RTOL * abs(g['cl_loss'][0])
    # acc_L = c_L, acc_k = c_k + A acc_{k+1}, dE0 = A acc_1 with c_k = G/L + [k == layer_cl] G_cl
    acc = None
    for k in range(L, 0, -1):
        c = (G / L + (Gcl if k == lc else 0.0)).astype(np.float32)
        acc = c if acc is None else O.spmm(csr, acc, 1.0, 1.0, c)
    grad = O.spmm(csr, acc)
    assert rel_err(grad[:U], g['grad_user']) < RTOL and rel_err(grad[U:], g['grad_item']) < RTOL
    m = np.zeros_like(E0); v = np.zeros_like(E0); E = E0.copy()
    O.adam_step(E, grad, m, v, 0.005, 1)
    assert rel_err(E[:U], g['user_k1']) < RTOL and rel_err(E[U:], g['item_k1']) < RTOL


def _sgl_views(ml100k, seed=2018, drop=0.1):
    """The two edge-dropped graphs of one SGL epoch, restated (recommender/SGL.py:211-229,288-299): the edges in CSR order of the
    interaction matrix, `random.sample(range(E), int(E * (1 - drop)))` kept, bipartite Laplacian with the isolated-node guard."""
    import random
    p = ml100k['pairs0']
    U, I = ml100k['U'], ml100k['I']
    lin = np.unique(p[:, 0].astype(np.int64) * I + p[:, 1])                 # sp_adj.nonzero(): row-major, sorted columns
    random.seed(seed)
    views = []
    for _ in range(2):
        keep = np.sort(lin[np.asarray(random.sample(range(len(lin)), int(len(lin) * (1 - drop))))])
        rowptr, col, w = O.bipartite_csr(keep // I, keep % I, U, I)
        views.append((keep, (rowptr, col, O.norm_adj_values(rowptr, col, w))))
    return views, random.random()


def test_sgl_views_and_step(ml100k):
    g = golden('g12_sgl.npz')
    U, I = ml100k['U'], ml100k['I']
    L, cl_rate, drop, temp = int(g['hyper'][0]), float(g['hyper'][1]), float(g['hyper'][2]), float(g['hyper'][3])
    views, nxt = _sgl_views(ml100k, 2018, drop)
    assert nxt == float(g['next_random'][0])
    for (keep, csr_k), tag in zip(views, ('1', '2')):
        assert np.array_equal(keep.astype(np.int32), g['keep' + tag])
        rows = np.repeat(np.arange(U + I), np.diff(csr_k[0]))
        sel = rows < U
        assert rel_err(csr_k[2][sel], g['val' + tag]) < 1e-6                 # normalised weights of the user rows (CSR order)
    csr = _ml100k_csr(ml100k)
    E0 = np.concatenate([g['user0'], g['item0']])
    out = O.lightgcn_forward(csr, E0, L)
    assert rel_err(out[:U], g['fwd_user']) < RTOL and rel_err(out[U:], g['fwd_item']) < RTOL
    bu, bp, bn = g['batch_u'], g['batch_p'], g['batch_n']
    lb, lr_, G = O.bpr_l2(out, U, bu, bp, bn, 1e-4)
    assert abs(lb - g['rec_loss'][0]) <= RTOL * abs(g['rec_loss'][0])
    v1 = O.lightgcn_forward(views[0][1], E0, L)
    v2 = O.lightgcn_forward(views[1][1], E0, L)
    idx = np.concatenate([np.unique(bu), np.unique(bp) + U])                 # ONE InfoNCE over the concatenated user and item rows
    l, d1, d2 = O.infonce(v1[idx], v2[idx], temp)
    assert abs(cl_rate * l - g['cl_loss'][0]) <= RTOL * abs(g['cl_loss'][0])
    G1 = np.zeros_like(E0); G2 = np.zeros_like(E0)
    G1[idx] = cl_rate * d1; G2[idx] = cl_rate * d2
    grad = O.lightgcn_backward(csr, G, L) + O.lightgcn_backward(views[0][1], G1, L) + O.lightgcn_backward(views[1][1], G2, L)
    assert rel_err(grad[:U], g['grad_user']) < RTOL and rel_err(grad[U:], g['grad_item']) < RTOL
    m = np.zeros_like(E0); v = np.zeros_like(E0); E = E0.copy()
    O.adam_step(E, grad, m, v, 0.005, 1)
    assert rel_err(E[:U], g['user_k1']) < RTOL and rel_err(E[U:], g['item_k1']) < RTOL
    w1 = O.lightgcn_forward(views[0][1], np.concatenate([g['user_k1'], g['item_k1']]), L)
    assert rel_err(w1[:U], g['view1_user']) < RTOL and rel_err(w1[U:], g['view1_item']) < RTOL


def test_ngcf_forward_at_cfg5_width(ml100k):
    """NGCF at BASELINE cfg5's width (d = 128, L = 3) against the reference class (g9_ngcf128)."""
    g = golden('g9_ngcf128.npz')
    csr = _ml100k_csr(ml100k)
    out = O.ngcf_forward(csr, np.concatenate([g['user0'], g['item0']]), [g['w1_%d' % k] for k in range(3)], [g['w2_%d' % k] for k in range(3)])
    U = ml100k['U']
    assert g['user0'].shape[1] == 128
    assert rel_err(out[:U], g['fwd_user']) < RTOL and rel_err(out[U:], g['fwd_item']) < RTOL
