"""Does the ORDER of the item stream matter for score_mask_topk?  Same tables, items streamed in table order vs by descending row norm
(a proxy of "likely in many users' lists").  Tables: (a) random, (b) one LightGCN propagation of xavier tables on the cfg2 graph (what the bench's
attack legs rank), (c) tables with a popularity-shaped norm spread (trained-recommender-like).   python3 tools/topk_order_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
from arlib_amd.util import synthetic
U, I, d, k = 1_000_000, 100_000, 64, 50
dev = torch.device('cuda', 0)


def t(Pu, Pi):
    ops.score_mask_topk(Pu[:4096].contiguous(), Pi, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.score_mask_topk(Pu, Pi, k)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


def run(name, Pu, Pi):
    order = torch.argsort(Pi.norm(dim=1), descending=True)
    print('%-28s table order %.1f ms   norm-descending %.1f ms   (norm max/median %.2f)' % (name, t(Pu, Pi), t(Pu, Pi[order].contiguous()),
          float(Pi.norm(dim=1).max() / Pi.norm(dim=1).median())), flush=True)


torch.manual_seed(0)
run('random N(0, 0.1)', torch.randn(U, d, device=dev) * 0.1, torch.randn(I, d, device=dev) * 0.1)
data = synthetic.syn_v1(U, I, 32.0, 2018)
p = data.pairs0
A = ops.bipartite_graph(torch.from_numpy(p[:, 0].astype(np.int64)).to(dev), torch.from_numpy(p[:, 1].astype(np.int64)).to(dev), U, I)
E0 = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, d)), torch.nn.init.xavier_uniform_(torch.empty(I, d))], 0).to(dev)
out = E0.clone(); E = E0
for _ in range(3):
    E = ops.spmm(A, E); out += E
out /= 4
run('LightGCN-propagated (bench)', out[:U].contiguous(), out[U:].contiguous())
deg = torch.from_numpy(np.bincount(p[:, 1], minlength=I).astype(np.float32)).to(dev)
Pi = torch.randn(I, d, device=dev) * 0.1 * (deg / deg.median()).pow(0.25)[:, None]
run('popularity-scaled norms', torch.randn(U, d, device=dev) * 0.1, Pi)
