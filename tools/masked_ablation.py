"""Masked-hop ablation at cfg2: time ops.spmm_flagged with a real batch bitmap (library built with -DARL_MASK_ABL=k skips stages)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
from arlib_amd.util import synthetic
U, I, d = 1_000_000, 100_000, 64
data = synthetic.syn_v1(U, I)
rowptr, col = data.adjacency_pattern()
dev = 'cuda:0'
N = U + I
val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(dev), torch.from_numpy(col).to(dev), torch.ones(len(col), device=dev), N)
A = ops.CSRGraph(rowptr, col, val, dev)
rng = np.random.default_rng(0)
sel = rng.integers(0, data.nnz, 2048)
rows = np.concatenate([data.pairs0[sel, 0], U + data.pairs0[sel, 1], U + rng.integers(0, I, 2048)]).astype(np.int32)
G = torch.zeros(N, d, device=dev); flags = torch.zeros(N, dtype=torch.uint8, device=dev); bits = torch.zeros((N + 31) // 32, dtype=torch.int32, device=dev)
r = torch.from_numpy(rows).to(dev)
ops.batch_rows_set_(G, flags, bits, r, torch.randn(len(rows), d, device=dev))
out = torch.empty(N, d, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('masked hop (x flagged, +G through zflags): %.3f ms' % t(lambda: ops.spmm_flagged(A, G, bits, 1.0, 1.0, G, flags, out=out)))
print('masked hop, no Z term:                     %.3f ms' % t(lambda: ops.spmm_flagged(A, G, bits, 1.0, 0.0, None, None, out=out)))
eu = int(rowptr[U])
valn = val.cpu().numpy()
Au = ops.CSRGraph(rowptr[:U + 1], col[:eu], valn[:eu], dev, n_cols=N)
Ai = ops.CSRGraph(rowptr[U:] - eu, col[eu:], valn[eu:], dev, n_cols=N)
Yu = torch.empty(U, d, device=dev); Yi = torch.empty(I, d, device=dev)
print('user rows only: %.3f ms' % t(lambda: ops.spmm_flagged(Au, G, bits, 1.0, 0.0, None, None, out=Yu)))
print('item rows only: %.3f ms (%d chunk tasks, %d long rows)' % (t(lambda: ops.spmm_flagged(Ai, G, bits, 1.0, 0.0, None, None, out=Yi)), Ai.n_chunks, Ai.n_long))
for ch in (2048, 100000):
    Ai2 = ops.CSRGraph(rowptr[U:] - eu, col[eu:], valn[eu:], dev, n_cols=N, chunk=ch)
    print('item rows only, chunk=%d: %.3f ms (%d chunk tasks)' % (ch, t(lambda: ops.spmm_flagged(Ai2, G, bits, 1.0, 0.0, None, None, out=Yi)), Ai2.n_chunks))
