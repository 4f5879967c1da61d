"""Randomised cross-check of the SpMM family (plain/AXPBY, layer-sum, Adam epilogue, flag-masked, row-subset; CSR and register-blocked
schedules with random plan parameters; square and
rectangular CSR; empty rows; rows far longer than the chunk size; d = 4..256) against float64 torch on many small random cases.
    python3 tools/spmm_fuzz.py [n_cases]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(777)
dev = 'cuda'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
bad = 0
for case in range(n_cases):
    d = int(rng.choice([64, 128])) if rng.random() < 0.5 else int(rng.choice([4, 8, 16, 24, 32, 64, 100, 128, 256]))
    n_rows = int(rng.integers(1, 3000)); n_cols = n_rows if rng.random() < 0.5 else int(rng.integers(1, 3000))
    chunk = int(rng.choice([32, 64, 512]))
    deg = rng.poisson(rng.choice([0.5, 4, 30]), n_rows)
    for h in range(int(rng.integers(0, 3))):
        deg[rng.integers(0, n_rows)] = int(rng.integers(chunk + 1, 6 * chunk))          # long rows (chunk plan + fixup)
    deg = np.minimum(deg, n_cols * 3)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    nnz = int(rowptr[-1])
    col = rng.integers(0, n_cols, max(nnz, 1)).astype(np.int32)[:nnz] if nnz else np.zeros(0, np.int32)      # duplicates allowed
    val = rng.standard_normal(nnz).astype(np.float32)
    A = ops.CSRGraph(rowptr, col if nnz else np.zeros(0, np.int32), val if nnz else np.zeros(0, np.float32), dev, chunk=chunk, n_cols=n_cols)
    sched = 'csr'
    if rng.random() < 0.6:           # register-blocked hop schedule (only taken at d = 64 and 128) with random plan parameters
        kw = dict(split=int(rng.integers(0, n_rows + 1)) if rng.random() < 0.5 else None, rows_per_wave=int(rng.choice([16, 32])), hub=int(rng.choice([3, 40, 100000])),
                  col_block=int(rng.choice([1, 64, 4096])), min_waves=int(rng.choice([0, 0, 8])), unroll=[None, 16, 32][int(rng.integers(0, 3))])
        A.enable_blocked(**kw)
        sched = 'blocked %s' % kw
    Ad = torch.zeros(n_rows, n_cols, dtype=torch.float64, device=dev)
    if nnz:
        Ad.index_put_((T(np.repeat(np.arange(n_rows), deg)), T(col.astype(np.int64))), T(val.astype(np.float64)), accumulate=True)
    X = T(rng.standard_normal((n_cols, d)).astype(np.float32)); Z = T(rng.standard_normal((n_rows, d)).astype(np.float32))
    ref = Ad @ X.double()
    errs = {}
    errs['plain'] = rel(ops.spmm(A, X), ref)
    a, b = float(rng.normal()), float(rng.normal())
    errs['axpby'] = rel(ops.spmm(A, X, a, b, Z), a * ref + b * Z.double())
    if n_rows == n_cols:
        S_in = T(rng.standard_normal((n_rows, d)).astype(np.float32)); S = torch.empty_like(S_in)
        Y = torch.empty_like(S_in)
        ops.spmm_layersum(A, X, S_in, S, Y)
        errs['layersum'] = max(rel(S, S_in.double() + ref), rel(Y, ref))
    # Adam epilogue
    P = T(rng.standard_normal((n_rows, d)).astype(np.float32)); M = T((rng.standard_normal((n_rows, d)) * 0.1).astype(np.float32))
    V = T((rng.random((n_rows, d)) * 0.01).astype(np.float32))
    g = a * ref + b * Z.double()
    t = int(rng.integers(1, 50)); lr, b1, b2, eps = 0.005, 0.9, 0.999, 1e-8
    m2 = b1 * M.double() + (1 - b1) * g; v2 = b2 * V.double() + (1 - b2) * g * g
    p2 = P.double() - (lr / (1 - b1 ** t)) * m2 / (v2.sqrt() / (1 - b2 ** t) ** 0.5 + eps)
    ops.spmm_adam(A, X, a, b, Z, P, M, V, lr, t)
    errs['adam'] = max(rel(P, p2), rel(M, m2), rel(V, v2))
    # flag-masked: X zero outside flagged rows, Z read only on flagged output rows
    fx = rng.random(n_cols) < 0.1; fz = rng.random(n_rows) < 0.2
    Xs = X.clone(); Xs[T(~fx)] = 0
    Zs = Z.clone(); Zs[T(~fz)] = 0
    bits = torch.zeros((n_cols + 31) // 32, dtype=torch.int32, device=dev)
    if fx.any():
        ops.mark_bits_(bits, T(np.nonzero(fx)[0].astype(np.int32)), True, n_cols)
    errs['flagged'] = rel(ops.spmm_flagged(A, Xs, bits, a, b, Zs, T(fz.astype(np.uint8))), a * (Ad @ Xs.double()) + b * Zs.double())
    # row subset with duplicates + earlier layers
    if n_rows == n_cols:
        rows = T(rng.integers(0, n_rows, int(rng.integers(1, 200))).astype(np.int32))
        L1 = T(rng.standard_normal((n_rows, d)).astype(np.float32))
        got = ops.spmm_rows(A, X, rows, [L1], 0.25, nsplit=int(rng.choice([1, 4, 32])))
        errs['rows'] = rel(got, 0.25 * (ref[rows.long()] + L1.double()[rows.long()]))
    worst = max(errs.values())
    if not worst < 2e-5:
        bad += 1
        print('MISMATCH case %d: rows=%d cols=%d d=%d chunk=%d nnz=%d %s -> %s' % (case, n_rows, n_cols, d, chunk, nnz, sched, {k: '%.2e' % v for k, v in errs.items()}), flush=True)
print('%d cases: %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)
