"""Experiment: L2-blocked SpMM vs the row-per-wave kernel at cfg2 (parity + time), sweeping cap / column block."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
from arlib_amd.util import synthetic
U, I = int(os.environ.get('U', 1000000)), int(os.environ.get('I', 100000))
data = synthetic.syn_v1(U, I)
rowptr, col = data.adjacency_pattern()
N = U + I
dev = 'cuda:0'
col_d = torch.from_numpy(col).to(dev)
val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(dev), col_d, torch.ones(len(col), device=dev), N)
A = ops.CSRGraph(rowptr, col_d, val, dev)
X = torch.randn(N, 64, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ref = ops.spmm(A, X)
print('row-per-wave: %.3f ms' % t(lambda: ops.spmm(A, X, out=ref)), flush=True)
for cap, cb, ns, hub in [(384, 16384, 256, 1024), (384, 8192, 256, 1024), (384, 4096, 256, 1024), (384, 32768, 256, 1024), (192, 8192, 512, 1024)]:
    t0 = time.perf_counter()
    P = ops.TiledPlan(A, [(0, U), (U, N)], cap=cap, col_block=cb, n_slots=ns, hub_threshold=hub)
    torch.cuda.synchronize(); tb = time.perf_counter() - t0
    Y = ops.spmm_tiled(P, X)
    err = ((Y - ref).abs().max() / ref.abs().max()).item()
    print('tiled cap=%d col_block=%d slots=%d hub>%d (%d hub rows): %.3f ms  (sweeps %d, build %.1f s, rel err %.1e)' % (cap, cb, ns, hub, P.hub_rows.numel(), t(lambda: ops.spmm_tiled(P, X, out=Y)), P.n_sweeps, tb, err), flush=True)
    del P
