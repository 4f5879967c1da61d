"""NGCF (a9) training step at cfg2 scale on one GPU: SpMM hops in HIP, d x d products in rocBLAS, autograd + torch Adam.
    python3 tools/ngcf_bench.py            (env: U, I, D, L)"""
import os, sys, time
from types import SimpleNamespace
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
from arlib_amd.util import synthetic
from arlib_amd.util.loss import bpr_loss, l2_reg_loss
from arlib_amd.recommender._base import SparseNormAdj
from arlib_amd.recommender.NGCF import NGCF_Encoder

U, I, d, L, B = int(os.environ.get('U', 1_000_000)), int(os.environ.get('I', 100_000)), int(os.environ.get('D', 64)), int(os.environ.get('L', 3)), 2048
dev = torch.device('cuda', 0)
data = synthetic.syn_v1(U, I, 32.0, 2018)
nnz = data.training_size()[2]
rowptr, col = data.adjacency_pattern()
col_d = torch.from_numpy(col).to(dev)
val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(dev), col_d, torch.ones(2 * nnz, device=dev), U + I)
A = ops.auto_blocked(ops.CSRGraph(rowptr, col_d, val, dev), d, split=U)
torch.manual_seed(2018)
enc = NGCF_Encoder.__new__(NGCF_Encoder)
torch.nn.Module.__init__(enc)
enc.data = SimpleNamespace(user_num=U, item_num=I)
enc.latent_size = enc.emb_size = d
enc.layers = enc.n_prop_layers = L
enc._eng = None
packed = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, d)), torch.nn.init.xavier_uniform_(torch.empty(I, d))], 0).to(dev)
enc.embedding_dict = torch.nn.ParameterDict({'user_emb': torch.nn.Parameter(packed[:U]), 'item_emb': torch.nn.Parameter(packed[U:])})
enc.W = torch.nn.ParameterDict({n + str(i): torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(d, d)).to(dev)) for i in range(L) for n in ('w1_', 'w2_')})
adj = SparseNormAdj.__new__(SparseNormAdj)
adj.shape, adj.indptr, adj.indices, adj.values, adj._graph = (U + I, U + I), None, None, A.val, A
enc.sparse_norm_adj = adj
opt = torch.optim.Adam(enc.parameters(), lr=0.005)
g = torch.Generator().manual_seed(1)
u = torch.randint(0, U, (B,), generator=g).to(dev); p = torch.randint(0, I, (B,), generator=g).to(dev); n = torch.randint(0, I, (B,), generator=g).to(dev)


ROWS = not os.environ.get('FULL')            # the training loop's form: last layer on the batch rows only
rows = torch.cat([u, p + U, n + U]).to(torch.int32)


def step():
    if ROWS:
        o = enc.forward_rows(rows)
        loss = bpr_loss(o[:B], o[B:2 * B], o[2 * B:]) + l2_reg_loss(1e-4, o[:B], o[B:2 * B])
    else:
        ue, ie = enc()
        loss = bpr_loss(ue[u], ie[p], ie[n]) + l2_reg_loss(1e-4, ue[u], ie[p])
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(('rows-form ' if ROWS else 'full-table ') + 'NGCF d=%d L=%d on %dx%d (nnz %d): %.1f ms/step = %.0f interactions/s, loss %.5f, peak mem %.1f GB'
      % (d, L, U, I, nnz, 1e3 * dt, B / dt, float(loss.detach()), torch.cuda.max_memory_allocated() / 1e9))

# the fused route (engine.step_ngcf: what NGCF(args, data).train() runs with its default optimizer): no autograd, no torch optimizer
from arlib_amd.engine import PropagationEngine
torch.cuda.empty_cache()
eng = PropagationEngine(A, U, I, d, L, 1e-4, 0.005, dev, table=packed.clone())
eng.init_ngcf([(enc.W['w1_%d' % k].detach().clone(), enc.W['w2_%d' % k].detach().clone()) for k in range(L)])
u32, p32, n32 = u.to(torch.int32), p.to(torch.int32), n.to(torch.int32)
for _ in range(2):
    eng.step_ngcf(u32, p32, n32, rows=rows)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    lo = eng.step_ngcf(u32, p32, n32, rows=rows)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('fused engine.step_ngcf d=%d L=%d: %.1f ms/step = %.0f interactions/s, loss %.5f' % (d, L, 1e3 * dt, B / dt, float(lo[0] + lo[1])))
