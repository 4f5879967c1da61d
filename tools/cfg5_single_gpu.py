"""BASELINE config 5's sizes on ONE MI355X (288 GB): SYN-v1 10 M users x 1 M items (~3.2e8 interactions), NGCF d = 128 L = 3 training step through
the encoder's sparse-batch route, and a DL_Attack-style masked top-50 scoring pass on a 1 M-user slice against all 1 M items.  Capacity check +
timings; the 8-GPU run of this config shards the same structures (dist_engine.step_ngcf / score_topk).
    python3 tools/cfg5_single_gpu.py            env: U, I, D, L"""
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

U, I, d, L, B = int(os.environ.get('U', 10_000_000)), int(os.environ.get('I', 1_000_000)), int(os.environ.get('D', 128)), int(os.environ.get('L', 3)), 2048
dev = torch.device('cuda', 0)
t0 = time.perf_counter()
pairs = synthetic.syn_v1_pairs_native(U, I, 32.0, 2018)
nnz = len(pairs)
print('SYN-v1 %d x %d: %d interactions generated natively in %.1f s (digest %016x)' % (U, I, nnz, time.perf_counter() - t0, synthetic.graph_digest_native(pairs)), flush=True)
t0 = time.perf_counter()
u = torch.from_numpy(pairs[:, 0].astype(np.int64)).to(dev); i = torch.from_numpy(pairs[:, 1].astype(np.int64)).to(dev)
A = ops.bipartite_graph(u, i, U, I)
del u, i
torch.cuda.synchronize()
print('normalised adjacency on the device: %d nodes, %d edges, %.1f s' % (A.n_rows, A.nnz, time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
ops.auto_blocked(A, d, split=U)
torch.cuda.synchronize()
print('register-blocked hop plan (d = %d): %s in %.1f s; memory %.1f GB' % (d, ', '.join('%d waves / %d edges / %d split rows' % (s['n_waves'], s['n_edges'], s['n_split']) for s in A.blocked.sets),
                                                                         time.perf_counter() - t0, torch.cuda.memory_allocated() / 1e9), flush=True)
torch.manual_seed(2018)
enc = NGCF_Encoder.__new__(NGCF_Encoder)
torch.nn.Module.__init__(enc)
enc.data = SimpleNamespace(user_num=U, item_num=I)
enc.latent_size = enc.emb_size = d
enc.layers = enc.n_prop_layers = L
enc._eng = None
packed = torch.empty(U + I, d, device=dev).uniform_(-0.01, 0.01)
enc.embedding_dict = torch.nn.ParameterDict({'user_emb': torch.nn.Parameter(packed[:U]), 'item_emb': torch.nn.Parameter(packed[U:])})
enc.W = torch.nn.ParameterDict({n + str(k): torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(d, d)).to(dev)) for k in range(L) for n in ('w1_', 'w2_')})
adj = SparseNormAdj.__new__(SparseNormAdj)
adj.shape, adj.indptr, adj.indices, adj.values, adj._graph = (U + I, U + I), None, None, A.val, A
enc.sparse_norm_adj = adj
opt = torch.optim.Adam(enc.parameters(), lr=0.005)
g = torch.Generator().manual_seed(1)
sel = torch.randint(0, nnz, (B,), generator=g).numpy()
bu = torch.from_numpy(pairs[sel, 0].astype(np.int64)).to(dev); bp = torch.from_numpy(pairs[sel, 1].astype(np.int64)).to(dev); bn = torch.randint(0, I, (B,), generator=g).to(dev)
rows = torch.cat([bu, bp + U, bn + U]).to(torch.int32)


def step():
    o = enc.forward_rows(rows)
    loss = bpr_loss(o[:B], o[B:2 * B], o[2 * B:]) + l2_reg_loss(1e-4, o[:B], o[B:2 * B])
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


step(); torch.cuda.synchronize()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('NGCF d=%d L=%d training step (B = %d): %.1f ms = %.0f interactions/s, loss %.5f, peak memory %.1f GB' % (d, L, B, 1e3 * dt, B / dt, float(loss.detach()), torch.cuda.max_memory_allocated() / 1e9), flush=True)
with torch.no_grad():
    hop = torch.empty_like(packed)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        ops.spmm(A, packed, out=hop)
    torch.cuda.synchronize()
    print('one full-graph hop at d = %d over %d edges: %.2f ms' % (d, A.nnz, 1e3 * (time.perf_counter() - t0) / 3), flush=True)
    n_slice = min(U, 1_000_000)
    rp = A.rowptr[:n_slice + 1].contiguous()
    mc = (A.col[:int(rp[-1])] - U).contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, val = ops.score_mask_topk(hop[:n_slice].contiguous(), hop[U:].contiguous(), 50, rp, mc)
    torch.cuda.synchronize()
    dts = time.perf_counter() - t0
    print('masked top-50 of %d users x %d items (d = %d): %.2f s = %.0f TFLOP/s fp32-equivalent; all %d users: %.0f s on one GPU, /8 when user-sharded'
          % (n_slice, I, d, dts, 2.0 * n_slice * I * d / dts / 1e12, U, dts * U / n_slice), flush=True)
