"""How fast is one full-graph hop when every GPU holds ALL rows but only d/N columns (feature-sharded LightGCN: no collective in the
propagation at all)?  Times the cfg2 hop at d = 64, 32, 16, 8 on one GPU = the per-rank hop of a 1-, 2-, 4-, 8-GPU feature-sharded run.
    python3 tools/dslice_bench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
from arlib_amd.util import synthetic
U, I = 1_000_000, 100_000
data = synthetic.syn_v1(U, I)
rowptr, col = data.adjacency_pattern()
val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).cuda(), torch.from_numpy(col).cuda(), torch.ones(len(col), device='cuda'), U + I)
A = ops.CSRGraph(rowptr, col, val, 'cuda')
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
base = None
for d in (64, 32, 16, 8, 4):
    X = torch.randn(U + I, d, device='cuda'); Y = torch.empty_like(X)
    ms = t(lambda: ops.spmm(A, X, out=Y))
    base = base or ms
    print('d = %2d (N = %d ranks): CSR hop %.3f ms  -> hop speed-up over d = 64: %.2fx' % (d, 64 // d, ms, base / ms))
