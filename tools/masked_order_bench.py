"""Masked hop at cfg2 under different orders of its row-task table: index order (no table), globally by descending length, and by descending
length inside windows of W consecutive rows (keeps the column stream / output rows of a wave's neighbours close in memory).
python3 tools/masked_order_bench.py"""
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
ops.batch_rows_set_(G, flags, bits, torch.from_numpy(rows).to(dev), torch.randn(len(rows), d, device=dev))
out = torch.empty(N, d, device=dev)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rp = A.rowptr.long(); deg = rp[1:] - rp[:-1]
ref = None
for W in (None, 0, 256, 2048, 16384, 131072):
    if W is None:
        A._row_tasks = None; A._rows_disabled = False
        tasks = None
    else:
        key = -deg if W == 0 else (torch.arange(N, device=dev) // W) * (1 << 20) - deg
        order = torch.sort(key, stable=True)[1]
        tasks = torch.stack([order, rp[order], rp[order + 1], torch.zeros_like(order)], 1).to(torch.int32).contiguous()
    A._row_tasks = tasks
    ms = t(lambda: ops.spmm_flagged(A, G, bits, 1.0, 1.0, G, flags, out=out))
    if ref is None: ref = out.clone()
    print('window %s: %.3f ms  (max diff vs index order %.2e)' % (W, ms, float((out - ref).abs().max())))
