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
