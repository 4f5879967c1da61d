"""Sanity: many consecutive sparse-batch steps at cfg2 (loss goes down, sparse state stays clean, no NaN)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops, engine
from arlib_amd.util import synthetic
from arlib_amd.util.sampler import MTState
U, I, d, L, B = 1000000, 100000, 64, 3, 2048
steps = int(os.environ.get('STEPS', 400))
data = synthetic.syn_v1(U, I)
rowptr, col = data.adjacency_pattern()
N = U + I
dev = 'cuda:0'
col_d = torch.from_numpy(col).to(dev)
val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(dev), col_d, torch.ones(len(col), device=dev), N)
A = ops.CSRGraph(rowptr, col_d, val, dev)
torch.manual_seed(2018)
E0 = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, d)), torch.nn.init.xavier_uniform_(torch.empty(I, d))], 0).to(dev)
eng = engine.PropagationEngine(A, U, I, d, L, 1e-4, 0.005, dev, table=E0)
mt = MTState.from_seed(2018); s = data.pair_sampler; s.shuffle(mt)
hb = torch.empty(steps, 3, B, dtype=torch.int32).pin_memory()
t0 = time.perf_counter()
for k in range(steps):
    s.batch(mt, k * B, B, out=hb[k].numpy())
print('host sampler: %.1f us per batch of %d' % (1e6 * (time.perf_counter() - t0) / steps, B))
db = hb.to(dev)
losses = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(steps):
    lo = eng.step(db[k, 0], db[k, 1], db[k, 2])
    if k % 50 == 0 or k == steps - 1:
        losses.append((k, float(lo[0]), float(lo[1])))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('%d steps: %.2f ms/step' % (steps, 1e3 * dt / steps))
print('loss (step, bpr, reg):', losses)
print('G max |.|: %g  flags: %d  bits: %d  table finite: %s' % (float(eng.G.abs().max()), int(eng.flags.max()), int(eng.bits.abs().max()), bool(torch.isfinite(eng.E0).all())))
