"""SGL training step at cfg2 scale (1M x 100K, d=64, L=2): fused sparse-batch step over the clean graph and two edge-dropped views.
    python3 tools/sgl_bench.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops, engine
from arlib_amd.util import synthetic
U, I, d, L, B = 1_000_000, 100_000, 64, 2, 2048
dev = torch.device('cuda', 0)
data = synthetic.syn_v1(U, I, 32.0, 2018)
pairs = data.pairs0
t0 = time.perf_counter()
g = ops.bipartite_graph(torch.from_numpy(pairs[:, 0].astype(np.int64)).to(dev), torch.from_numpy(pairs[:, 1].astype(np.int64)).to(dev), U, I)
torch.cuda.synchronize(); print('device graph build (32.1M interactions): %.2f s' % (time.perf_counter() - t0))
rng = np.random.default_rng(0)
views = []
for _ in range(2):
    t0 = time.perf_counter()
    keep = np.sort(rng.choice(len(pairs), int(0.9 * len(pairs)), replace=False))       # (timing harness: numpy draw; the class API draws CPython's)
    views.append(ops.bipartite_graph(torch.from_numpy(pairs[keep, 0].astype(np.int64)).to(dev), torch.from_numpy(pairs[keep, 1].astype(np.int64)).to(dev), U, I))
    torch.cuda.synchronize(); print('view build: %.2f s' % (time.perf_counter() - t0))
torch.manual_seed(2018)
E0 = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, d)), torch.nn.init.xavier_uniform_(torch.empty(I, d))], 0).to(dev)
eng = engine.PropagationEngine(g, U, I, d, L, 1e-4, 0.005, dev, table=E0)
sel = rng.integers(0, len(pairs), (12, B))
for k in range(12):
    if k == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    u = torch.from_numpy(pairs[sel[k], 0].astype(np.int32)).to(dev); p = torch.from_numpy(pairs[sel[k], 1].astype(np.int32)).to(dev)
    n = torch.from_numpy(rng.integers(0, I, B).astype(np.int32)).to(dev)
    lo, cl = eng.step_sgl(u, p, n, views[0], views[1])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print('SGL fused step: %.2f ms = %.0f interactions/s (loss %.5f cl %.5f)' % (1e3 * dt, B / dt, float(lo[0] + lo[1]), float(cl)))
