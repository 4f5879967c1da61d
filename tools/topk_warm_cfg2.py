"""Cold vs warm-started masked top-50 at cfg2 size (1M x 100K, SYN-v1 interaction mask): the candidates are the lists of the
same tables before a +-move step on every element (what consecutive surrogate steps look like).   python3 tools/topk_warm_cfg2.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
from arlib_amd.util import synthetic
U, I, d, k = 1_000_000, 100_000, 64, 50
dev = torch.device('cuda', 0)
data = synthetic.syn_v1(U, I, 32.0, 2018)
nnz = data.training_size()[2]
rowptr, col = data.adjacency_pattern()
mask = (torch.from_numpy(rowptr[:U + 1].astype(np.int32)).to(dev), torch.from_numpy((col[:nnz] - U).astype(np.int32)).to(dev))
torch.manual_seed(1)
X = torch.randn(U + I, d, device=dev) * 0.1
prev, _ = ops.score_mask_topk(X[:U].contiguous(), X[U:].contiguous(), k, *mask)
for move in (5e-4, 5e-3):
    X2 = X + torch.sign(torch.randn_like(X)) * move
    Pu, Pi = X2[:U].contiguous(), X2[U:].contiguous()
    for name, w in (('cold', None), ('warm', prev)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, _ = ops.score_mask_topk(Pu, Pi, k, *mask, warm_idx=w)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('move %g %s: %.1f ms' % (move, name, dt * 1e3), flush=True)
