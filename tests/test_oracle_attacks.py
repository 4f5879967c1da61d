"""Pins the oracle's attack compositions (PGA gradient step, masked top-k, top-n projection) against intermediates
traced from the UNMODIFIED reference attacks (tests/golden/g7_attacks.npz).  CPU only."""
import numpy as np
from conftest import golden, rel_err, RTOL
from oracle import oracle as O


def test_pga_gradient_steps_match_reference_trace():
    g = golden('g7_attacks.npz')
    U, I, F, L, d = (int(x) for x in g['pga_sizes'])
    E0 = np.concatenate([g['pga_user_tab'], g['pga_item_tab']])
    users, pos, neg = O.cw_pairs(g['pga_top50'], U, g['pga_targets'], pop=True)
    for s in range(g['pga_grad'].shape[0]):
        grad, S_next, loss = O.pga_step(g['pga_real_indptr'], g['pga_real_indices'], U, F, I, g['pga_S'][s], E0, L, users, pos, neg)
        assert rel_err(grad, g['pga_grad'][s]) < RTOL, s                 # D^-1/2 (dL/dA) D^-1/2 on the fake block
        assert rel_err(S_next, g['pga_S'][s + 1]) < 1e-6, s              # after -0.2 tanh + clamp
    # first step: only pattern entries (targets + popular fillers) carry gradient, everything else lands on the 1e-7 floor
    assert (g['pga_S'][0] > 0).sum() < (g['pga_S'][1] > 0).sum() == F * I
    assert np.count_nonzero(g['pga_grad'][0][g['pga_S'][0] == 0]) == 0


def test_top50_of_reference_forward_matches_oracle_topk():
    g = golden('g7_attacks.npz')
    U, I, F, L, d = (int(x) for x in g['pga_sizes'])
    E0 = np.concatenate([g['pga_user_tab'], g['pga_item_tab']])
    csr, _ = O.pga_weighted_graph(g['pga_real_indptr'], g['pga_real_indices'], U, F, I, g['pga_S'][0])
    out = O.lightgcn_forward(csr, E0, L)
    idx, _ = O.score_mask_topk(out[:U], out[U + F:], 50)
    assert (idx == g['pga_top50']).mean() > 0.999                        # ties/near-ties aside, same ranking as torch.topk


def test_dlattack_masked_topk_and_project():
    g = golden('g7_attacks.npz')
    k = int(g['dl_k'][0])
    idx, val = O.score_mask_topk(g['dl_Pu'], g['dl_Pi'], k, (g['dl_mask_indptr'], g['dl_mask_indices']))
    assert (idx == g['dl_topk']).mean() > 0.999
    assert (np.sort(idx, 1) == np.sort(g['dl_topk'], 1)).mean() > 0.9999
    for r in range(len(g['dl_proj_n'])):
        out, ind = O.topn_project_rows(g['dl_proj_in'][r][None, :], int(g['dl_proj_n'][r]))
        assert np.array_equal(out[0], g['dl_proj_out'][r]) and np.array_equal(ind[0], g['dl_proj_idx'][r])
    assert list(g['dl_result_fake_rowsums']) == [5.0, 46.0]              # quirk Q6: the first fake user keeps only its targets


def test_clear_surrogate_step_matches_reference_trace():
    """CW + SFA loss and the parameter gradients of one CLeaR surrogate step (reference autograd, injected r0)."""
    g = golden('g7_attacks.npz')
    U, I, F, topk = (int(x) for x in g['cl_sizes'])
    Up, L = U + F, 2
    E0 = np.concatenate([g['cl_user_tab'], g['cl_item_tab']])
    rows = np.repeat(np.arange(Up), np.diff(g['cl_ui_indptr']))
    rowptr, col, w = O.bipartite_csr(rows, g['cl_ui_indices'], Up, I, g['cl_ui_data'])
    csr = (rowptr, col, O.norm_adj_values(rowptr, col, w))
    out = O.lightgcn_forward(csr, E0, L)
    idx, _ = O.score_mask_topk(out[:Up], out[Up:], topk, (g['cl_ui_indptr'], g['cl_ui_indices']))
    users, pos, neg = O.cw_pairs(idx, U, g['cl_targets'], pop=True)
    cw, sfa, G = O.clear_loss_grad(out, Up, users, pos, neg, g['cl_r0'])
    assert abs(cw + sfa - g['cl_loss'][0]) <= RTOL * abs(g['cl_loss'][0])
    dE0 = O.lightgcn_backward(csr, G, L)
    grads = {a.shape[0]: a for a in (g['cl_grad_user'], g['cl_grad_item'])}      # keyed by row count (945 users / 1412 items)
    assert rel_err(dE0[:Up], grads[Up]) < RTOL
    assert rel_err(dE0[Up:], grads[I]) < RTOL


def test_sfa_closed_form_equals_literal_restatement():
    """The weighted closed form the HIP kernel uses (loss = S A / (numel Q), rows with multiplicities) against the literal
    reverse pass on the materialised H."""
    rng = np.random.default_rng(3)
    n, d = 40, 12
    X = rng.normal(size=(n, d)).astype(np.float32) * 0.3
    w = rng.integers(0, 4, n).astype(np.float64)
    r0 = rng.normal(size=d).astype(np.float32)
    rows = np.repeat(np.arange(n), w.astype(np.int64))
    loss, gH = O.sfa_l1_loss_grad(X[rows], r0)
    G = np.zeros((n, d)); np.add.at(G, rows, gH)
    Xd = X.astype(np.float64)
    q = Xd @ r0; r = Xd.T @ (w * q); s = Xd @ r
    S, A, Q, numel = (w * np.abs(s)).sum(), np.abs(r).sum(), r @ r, w.sum() * d
    a = Xd.T @ (w * np.sign(s))
    g_r = ((A / Q) * a + (S / Q) * np.sign(r) - (2 * S * A / Q ** 2) * r) / numel
    G2 = w[:, None] * ((A / (numel * Q)) * np.sign(s)[:, None] * r[None, :] + q[:, None] * g_r[None, :] + (Xd @ g_r)[:, None] * r0[None, :])
    assert abs(S * A / (numel * Q) - loss) < 1e-12 * abs(loss)
    assert rel_err(G2, G) < 1e-10


def test_clear_surrogate_step_on_a_simgcl_victim_matches_reference_trace():
    """BASELINE config 4 (SimGCL + CLeaR): the surrogate is the victim's encoder -- mean of layers 1..L, layer 0 skipped -- so the CW + SFA
    loss and the table gradients follow the skip0 forward/backward (g19, reference autograd with injected r0)."""
    g = golden('g19_victims.npz')
    U, I, F, topk = (int(x) for x in g['simgcl_cl_sizes'])
    Up, L = U + F, 2
    E0 = np.concatenate([g['simgcl_cl_user_tab'], g['simgcl_cl_item_tab']])
    rows = np.repeat(np.arange(Up), np.diff(g['simgcl_cl_ui_indptr']))
    rowptr, col, w = O.bipartite_csr(rows, g['simgcl_cl_ui_indices'], Up, I, g['simgcl_cl_ui_data'])
    csr = (rowptr, col, O.norm_adj_values(rowptr, col, w))
    out = O.lightgcn_forward(csr, E0, L, skip0=True)
    idx, _ = O.score_mask_topk(out[:Up], out[Up:], topk, (g['simgcl_cl_ui_indptr'], g['simgcl_cl_ui_indices']))
    users, pos, neg = O.cw_pairs(idx, U, g['simgcl_cl_targets'], pop=True)
    cw, sfa, G = O.clear_loss_grad(out, Up, users, pos, neg, g['simgcl_cl_r0'])
    assert abs(cw + sfa - g['simgcl_cl_loss'][0]) <= 2 * RTOL * abs(g['simgcl_cl_loss'][0])
    dE0 = O.lightgcn_backward(csr, G, L, skip0=True)
    assert rel_err(dE0[:Up], g['simgcl_cl_grad_user']) < RTOL and rel_err(dE0[Up:], g['simgcl_cl_grad_item']) < RTOL


def test_attacks_on_ngcf_and_simgcl_victims_keep_the_reference_structure():
    """The reference runs recorded in g19: DLAttack's fake rows [5, 46] (quirk Q6) and CLeaR's 51 = 46 fillers + 5 targets hold for NGCF and
    SimGCL victims alike; NoneAttack returns the clean matrix and draws nothing beyond the target selection."""
    g = golden('g19_victims.npz')
    for tag in ('ngcf', 'simgcl'):
        assert list(g[tag + '_dl_result_fake_rowsums']) == [5.0, 46.0] and list(g[tag + '_dl_shape']) == [944, 1412]
        assert list(g[tag + '_cl_result_fake_rowsums']) == [51.0, 51.0, 51.0]
    assert int(g['none_identity'][0]) == 0 and list(g['none_flags']) == [0, 0]
