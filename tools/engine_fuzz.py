"""Sparse-batch training step vs the dense 2L-hop step on random graphs / depths / embedding sizes / batch sizes (incl. heavy
duplicates): tables, Adam moments and losses must agree and the sparse state must be left clean.   python3 tools/engine_fuzz.py [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops, engine
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(99)
dev = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
bad = 0
for case in range(n_cases):
    d = 64 if rng.random() < 0.5 else int(rng.choice([4, 8, 32, 64, 128, 256])); L = int(rng.integers(1, 5))
    U, I = int(rng.integers(5, 3000)), int(rng.integers(3, 600)); B = int(rng.integers(1, 1500))
    deg = np.clip(rng.poisson(rng.choice([2, 8, 40]), U), 1, I)
    us = np.repeat(np.arange(U), deg); its = np.floor(I * rng.random(len(us)) ** 2).astype(np.int64)
    key = np.unique(us * I + its); us, its = key // I, key % I
    g = ops.bipartite_graph(T(us), T(its), U, I)
    g2 = ops.bipartite_graph(T(us), T(its), U, I)            # sparse step on the register-blocked hop schedule (d = 64), dense step on the CSR kernels
    if rng.random() < 0.7:
        g2.enable_blocked(split=U, rows_per_wave=int(rng.choice([16, 32])), hub=int(rng.choice([5, 50, 100000])), col_block=int(rng.choice([16, 1024])))
    E0 = T(((rng.random((U + I, d)) * 2 - 1) * 0.1).astype(np.float32))
    ea = engine.PropagationEngine(g2, U, I, d, L, 1e-4, 0.005, dev, table=E0.clone(), schedule='csr')
    eb = engine.PropagationEngine(g, U, I, d, L, 1e-4, 0.005, dev, table=E0.clone(), schedule='csr')
    ok = True
    for k in range(3):
        sel = rng.integers(0, len(us), B)
        bu, bp, bn = T(us[sel].astype(np.int32)), T(its[sel].astype(np.int32)), T(rng.integers(0, I, B).astype(np.int32))
        if k == 1 and B > 4:
            bu[:B // 2] = bu[0]; bn[:B // 3] = bp[0]
        la = ea.step(bu, bp, bn).cpu(); lb = eb.step_dense(bu, bp, bn).cpu()
        ok &= bool(torch.allclose(la, lb, rtol=1e-4, atol=0))
    errs = {'E': rel(ea.E0, eb.E0), 'm': rel(ea.m, eb.m), 'v': rel(ea.v, eb.v)}
    clean = float(ea.G.abs().max()) == 0.0 and int(ea.flags.max()) == 0 and int(ea.bits.abs().max()) == 0
    if not (ok and max(errs.values()) < 1e-4 and clean):
        bad += 1
        print('MISMATCH case %d: U=%d I=%d d=%d L=%d B=%d nnz=%d loss_ok=%s clean=%s %s' % (case, U, I, d, L, B, len(us), ok, clean, {k_: '%.2e' % v for k_, v in errs.items()}), flush=True)
print('%d cases: %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)
