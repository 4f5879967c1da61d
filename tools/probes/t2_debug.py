import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arlib_amd import ops
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
I, k, masked = 5000, 1, True
rng = np.random.default_rng(9100 + I + k + masked)
U, d = 1300, 64
Pu = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
Pi = (rng.standard_normal((I, d)) * 0.1 * (rng.pareto(2.0, I) + 0.1)[:, None]).astype(np.float32)
sc = Pu @ Pi.T
cols = [np.unique(np.concatenate([np.argsort(-sc[u])[:int(rng.integers(0, 40))], rng.choice(I, size=int(rng.integers(0, 30)), replace=False)])).astype(np.int32) for u in range(U)]
rp = T(np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32)); mc = T(np.concatenate(cols))
scm = sc.astype(np.float64).copy()
for u in range(U): scm[u, cols[u]] = -10e8
ref = np.argsort(-scm, axis=1, kind='stable')[:, :k]
for form2 in (False, True):
    ops.TOPK_FORM2 = form2
    ops.reset_exit_probe()
    i_c, v_c = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc)
    i_w, v_w = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, warm_idx=i_c)
    stale = torch.from_numpy(np.stack([rng.choice(I, size=k, replace=False) for _ in range(U)]).astype(np.int32)).cuda()
    i_s, v_s = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, warm_idx=stale)
    i_t, v_t = ops.score_mask_topk(T(Pu), T(Pi), k, rp, mc, item_order=None)
    print('form2', form2, 'cold vs ref rows differing', int((i_c.cpu().numpy() != ref).any(1).sum()))
    for name, (a, b) in dict(warm=(i_w, v_w), stale=(i_s, v_s), table=(i_t, v_t)).items():
        bad = ((a != i_c) | (b != v_c)).any(1).nonzero().flatten().tolist()
        print('   ', name, 'rows differing from cold:', len(bad), bad[:8])
        for r in bad[:3]:
            print('       row', r, 'cold', i_c[r].tolist(), v_c[r].tolist(), 'this', a[r].tolist(), b[r].tolist(), 'ref', ref[r].tolist(), 'stale cand', stale[r].tolist(), 'masked?', int(stale[r, 0]) in set(cols[r].tolist()))
