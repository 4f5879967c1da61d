"""Time the fused SimGCL step (L=2) at cfg2 scale against the autograd route (3 forwards + 3 backward passes)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops, engine
from arlib_amd.util import synthetic
from arlib_amd.util.sampler import MTState
U, I, d, L, B = 1000000, 100000, 64, 2, 2048
data = synthetic.syn_v1(U, I)
rowptr, col = data.adjacency_pattern()
N = U + I
dev = 'cuda:0'
col_d = torch.from_numpy(col).to(dev)
val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(dev), col_d, torch.ones(len(col), device=dev), N)
A = ops.CSRGraph(rowptr, col_d, val, dev)
torch.manual_seed(2018)
E0 = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, d)), torch.nn.init.xavier_uniform_(torch.empty(I, d))], 0).to(dev)
eng = engine.PropagationEngine(A, U, I, d, L, 1e-4, 0.005, dev, skip_layer0=True, table=E0)
mt = MTState.from_seed(2018); s = data.pair_sampler; s.shuffle(mt)
bs = [torch.from_numpy(s.batch(mt, k * B, B)).to(dev) for k in range(12)]
for k in range(2):
    eng.step_simgcl(bs[k][0], bs[k][1], bs[k][2])
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(2, 12):
    lo, cl = eng.step_simgcl(bs[k][0], bs[k][1], bs[k][2])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print('fused SimGCL step (cfg2, L=2, B=2048): %.2f ms/step = %.0f interactions/s; rec %.4f cl %.4f' % (dt * 1e3, B / dt, float(lo[0]), float(cl)))
