"""Experiment: time the user-row half and the item-row half of the cfg2 SpMM separately (rectangular CSR blocks)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
from arlib_amd.util import synthetic
U, I = 1000000, 100000
data = synthetic.syn_v1(U, I)
rowptr, col = data.adjacency_pattern()
N = U + I
dev = 'cuda:0'
val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(dev), torch.from_numpy(col).to(dev), torch.ones(len(col), device=dev), N)
valn = val.cpu().numpy()
eu = int(rowptr[U])
Au = ops.CSRGraph(rowptr[:U + 1], col[:eu], valn[:eu], dev, n_cols=N)
Ai = ops.CSRGraph(rowptr[U:] - eu, col[eu:], valn[eu:], dev, n_cols=N)
A = ops.CSRGraph(rowptr, col, valn, dev)
X = torch.randn(N, 64, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
Yu = torch.empty(U, 64, device=dev); Yi = torch.empty(I, 64, device=dev); Y = torch.empty(N, 64, device=dev)
print('user rows (gather 25.6MB item table): %.3f ms, %d edges' % (t(lambda: ops.spmm(Au, X, out=Yu)), eu))
print('item rows (gather 256MB user table):  %.3f ms, %d edges' % (t(lambda: ops.spmm(Ai, X, out=Yi)), len(col) - eu))
print('full: %.3f ms' % t(lambda: ops.spmm(A, X, out=Y)))
for ch in (128, 256, 1024, 4096):
    Ai2 = ops.CSRGraph(rowptr[U:] - eu, col[eu:], valn[eu:], dev, n_cols=N, chunk=ch)
    print('item rows chunk=%d: %.3f ms (n_chunks %d)' % (ch, t(lambda: ops.spmm(Ai2, X, out=Yi)), Ai2.n_chunks))
