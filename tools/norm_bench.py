"""Timing + fp64 check of the adjacency normalisation (row form and edge-parallel form) on the cfg2 graph.   python3 tools/norm_bench.py"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from arlib_amd import ops
from arlib_amd.util import synthetic
U, I = 1_000_000, 100_000
data = synthetic.syn_v1(U, I)
rowptr, col = data.adjacency_pattern()
rp = torch.from_numpy(rowptr.astype(np.int32)).cuda(); c = torch.from_numpy(col).cuda(); w = torch.rand(len(col), device='cuda') + 0.5
val, dinv = ops.norm_adj_values(rp, c, w, U + I)
# reference in float64 torch
erow = torch.repeat_interleave(torch.arange(U + I, device='cuda'), (rp[1:] - rp[:-1]).long())
s = torch.zeros(U + I, dtype=torch.float64, device='cuda').index_add_(0, erow, w.double())
di = torch.where(s > 0, s.rsqrt(), torch.zeros_like(s))
ref = di[erow] * w.double() * di[c.long()]
print('rel err val %.2e dinv %.2e' % (((val.double() - ref).norm() / ref.norm()).item(), ((dinv.double() - di).norm() / di.norm()).item()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.norm_adj_values(rp, c, w, U + I)
e1.record(); torch.cuda.synchronize()
print('norm_adj_values (64.2M edges): %.3f ms' % (e0.elapsed_time(e1) / 10))
er32 = erow.to(torch.int32)
v2, d2 = ops.norm_adj_values(rp, c, w, U + I, erow=er32)
print('edge-parallel form equal:', torch.equal(v2, val), torch.equal(d2, dinv))
e0.record()
for _ in range(10): ops.norm_adj_values(rp, c, w, U + I, erow=er32)
e1.record(); torch.cuda.synchronize()
print('norm_adj_values with erow: %.3f ms' % (e0.elapsed_time(e1) / 10))
