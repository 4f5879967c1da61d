"""Experiment: how fast is the SpMM gather when the gathered table is L2-sized?  Same CSR row structure as the cfg2 item half
(100K rows, 32.1M edges) but with the column ids folded into a table of T rows (T*256 B)."""
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
eu = int(rowptr[U])
rp_i = rowptr[U:] - eu
col_i = col[eu:].astype(np.int64)
val = np.ones(len(col_i), np.float32)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
Y = torch.empty(I, 64, device=dev)
for T in (1000000, 262144, 65536, 16384, 8192, 2048):
    c = (col_i % T).astype(np.int32)
    A = ops.CSRGraph(rp_i, c, val, dev, n_cols=T)
    X = torch.randn(T, 64, device=dev)
    ms = t(lambda: ops.spmm(A, X, out=Y))
    print('table %7d rows (%6.1f MB): %.3f ms  -> %.2f TB/s of row gathers' % (T, T * 256 / 1e6, ms, len(c) * 256 / ms / 1e9), flush=True)
