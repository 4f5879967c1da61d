"""Randomised cross-check of score_mask_topk (all paths: exact-f32 / split-fp16 / VALU fallback, masked or not, warm or cold)
against dense torch scoring + topk on many small random shapes.   python3 tools/topk_fuzz.py [n_cases]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(12345)
dev = 'cuda'
bad = 0
for case in range(n_cases):
    d = int(rng.choice([4, 8, 16, 20, 32, 48, 64, 128, 256]))
    U = int(rng.integers(1, 700)); I = int(rng.integers(1, 3000))
    if case % 8 == 7:                           # long item streams: the bootstrap-threshold pass runs from 32 K items on (MFMA path: d in 16/32/64/128, k <= 64)
        U = int(rng.integers(1, 300)); I = int(rng.integers(32768, 70000)); d = int(rng.choice([16, 32, 64, 128]))
    k = int(rng.integers(1, min(I, 128) + 1))
    if case % 8 == 7 and rng.random() < 0.8:
        k = int(rng.integers(1, 65))
    scale = float(rng.choice([1e-3, 0.1, 1.0, 30.0]))
    Pu = torch.from_numpy((rng.standard_normal((U, d)) * scale).astype(np.float32)).to(dev)
    Pi = torch.from_numpy((rng.standard_normal((I, d)) * scale).astype(np.float32)).to(dev)
    if rng.random() < 0.3:                      # popularity-skewed norms
        Pi *= torch.from_numpy((rng.pareto(2.0, I) + 0.1).astype(np.float32)).to(dev)[:, None]
    masked = rng.random() < 0.6
    rp = mc = None
    sc = (Pu.double() @ Pi.double().T)
    if masked:
        lens = rng.integers(0, min(I, 80) + 1, U)
        if case % 8 == 7 and rng.random() < 0.5:  # interacted items piled up inside the bootstrap's sample range, where they score best
            Pi[:4096] *= 2.0
            sc = (Pu.double() @ Pi.double().T)
        if rng.random() < 0.2 and I < 3000:
            lens[:] = max(I - max(k // 2, 1), 0)                                  # fewer than k unmasked items
        if case % 8 == 7:
            top = torch.topk(sc[:, :4096], 80, dim=1)[1].cpu().numpy()
            cols = [np.unique(np.concatenate([top[u, :int(n)], rng.choice(I, int(n) // 3, replace=False)])).astype(np.int32) for u, n in enumerate(lens)]
            lens = np.array([len(c) for c in cols])
        else:
            cols = [np.sort(rng.choice(I, int(n), replace=False)).astype(np.int32) for n in lens]
        rp = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
        flat = np.concatenate(cols) if lens.sum() else np.zeros(1, np.int32)
        mc = torch.from_numpy(flat.astype(np.int32)).to(dev)
        rows = torch.from_numpy(np.repeat(np.arange(U), lens)).to(dev)
        if lens.sum():
            sc[rows, mc[:int(lens.sum())].long()] = -10e8
    ref_v, ref_i = torch.topk(sc, k)
    for exact in (False, True):
        idx, val = ops.score_mask_topk(Pu, Pi, k, rp, mc, exact=exact)
        warm_i, warm_v = ops.score_mask_topk(Pu, Pi, k, rp, mc, exact=exact, warm_idx=idx)
        checks = {}
        checks['warm==cold'] = torch.equal(warm_i, idx) and torch.equal(warm_v, val)
        tol = 2e-5 * max(scale * scale * d ** 0.5, 1e-12) * (Pi.norm(dim=1).max().item() / (scale * d ** 0.5) + 1.0) + 1e-6 * ref_v.abs()
        checks['values'] = bool(((val.double() - ref_v).abs() <= tol).all())
        # every returned item's true score must be >= the reference k-th score (up to tol), and rows hold distinct items
        got_true = torch.gather(sc, 1, idx.long())
        checks['members'] = bool((got_true >= ref_v[:, -1:] - tol[:, -1:]).all())
        checks['distinct'] = bool((torch.sort(idx, 1)[0][:, 1:] != torch.sort(idx, 1)[0][:, :-1]).all()) if k > 1 else True
        checks['sorted'] = bool((val[:, :-1] >= val[:, 1:]).all()) if k > 1 else True
        ok = all(checks.values())
        if not ok:
            bad += 1
            print('MISMATCH case %d: U=%d I=%d d=%d k=%d masked=%s exact=%s scale=%g failed=%s' % (case, U, I, d, k, masked, exact, scale, [c for c, v in checks.items() if not v]), flush=True)
            if not checks['values']:
                e = ((val.double() - ref_v).abs() - tol)
                r, c = divmod(int(e.argmax()), k)
                print('   worst: row %d rank %d got %.9g (item %d) ref %.9g (item %d) tol %.3g' % (r, c, val[r, c].item(), idx[r, c].item(), ref_v[r, c].item(), ref_i[r, c].item(), tol[r, c].item()), flush=True)
print('%d cases x 2 paths: %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)
