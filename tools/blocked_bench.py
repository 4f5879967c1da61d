"""Register-blocked SpMM schedule (ops.BlockedPlan, arl_spmm_blocked_f32) against the row-per-group CSR kernel on the cfg2 graph:
one full hop at d = 64, and the pieces of the blocked hop (user rows, item rows, hub rows).
    python3 tools/blocked_bench.py      env: U, I, RPW (16|32), HUB, UB (columns per block)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
from arlib_amd.util import synthetic

U, I, d = int(os.environ.get('U', 1_000_000)), int(os.environ.get('I', 100_000)), int(os.environ.get('D', 64))
RPW, HUB, UB = int(os.environ.get('RPW', 32)), int(os.environ.get('HUB', 1024)), int(os.environ.get('UB', 1024))
UNR = int(os.environ['UNR']) if os.environ.get('UNR') else None
dev = 'cuda:0'
data = synthetic.syn_v1(U, I)
import numpy as np
rowptr, col = data.adjacency_pattern()
val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(dev), torch.from_numpy(col).to(dev), torch.ones(len(col), device=dev), U + I)
A = ops.CSRGraph(rowptr, col, val, dev)
N = U + I
X = torch.randn(N, d, device=dev)
Yr, Yb = torch.empty(N, d, device=dev), torch.zeros(N, d, device=dev)


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print('CSR hop (row-per-group + chunked long rows): %.3f ms' % t(lambda: ops.spmm(A, X, out=Yr)))
torch.cuda.synchronize(); t0 = time.perf_counter()
if os.environ.get('ISPLIT'):                 # experiment: the item rows in k separate launches (each launch's waves resident together)
    k = int(os.environ['ISPLIT'])
    cuts = [U + (I * j) // k for j in range(k + 1)]
    ku = int(os.environ.get('USPLIT', 1))
    ucuts = [(U * j) // ku for j in range(ku + 1)]
    A.blocked = ops.BlockedPlan(A, [(ucuts[j], ucuts[j + 1]) for j in range(ku)] + [(cuts[j], cuts[j + 1]) for j in range(k)], RPW, HUB, UB, 0, UNR)
else:
    A.enable_blocked(split=U, rows_per_wave=RPW, hub=HUB, col_block=UB, unroll=UNR)
torch.cuda.synchronize()
bp = A.blocked
print('plan built in %.2f s: %s; %d hub rows' % (time.perf_counter() - t0, ', '.join('%d rows / %d waves / %d edges' % (s['n_rows'], s['n_waves'], s['n_edges']) for s in bp.sets),
                                                 bp.n_hub))
ops.spmm(A, X, out=Yb); torch.cuda.synchronize()
print('rel err blocked vs CSR: %.2e' % ((Yb - Yr).norm() / Yr.norm()).item())
print('blocked hop (RPW=%d HUB=%d UB=%d): %.3f ms' % (RPW, HUB, UB, t(lambda: ops.spmm(A, X, out=Yb))))
import ctypes as C
from arlib_amd import _lib
st = ops._stream()
for k, s in enumerate([bp.struct(j, d) for j in range(len(bp.sets))]):
    print('  row set %d: %.3f ms' % (k, t(lambda: _lib.check(_lib.lib().arl_spmm_blocked_f32(C.byref(s), X.data_ptr(), d, 1.0, 0.0, None, None, Yb.data_ptr(), st), 'blocked'))))
if bp.hub is not None:
    print('  hub rows (chunked CSR kernel): %.3f ms' % t(lambda: _lib.check(_lib.lib().arl_spmm_csr_f32(C.byref(bp.hub._struct(d)), X.data_ptr(), d, 1.0, 0.0, None, Yb.data_ptr(), st), 'csr')))
