"""Per-epoch graph rebuild of the attack loops at cfg2: encoder._init_uiAdj_from_interactions on the poisoned U' x I matrix,
full device build vs the incremental merge (ops.IncrementalBipartite + patched hop plan), and the first hop afterwards.
    python3 tools/incremental_graph_bench.py      env: U, I, F, FILL"""
import os, sys, time
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
from arlib_amd import ops
from arlib_amd.util import synthetic
from arlib_amd.recommender._base import GraphEncoder

U, I, F, FILL, d = int(os.environ.get('U', 1_000_000)), int(os.environ.get('I', 100_000)), int(os.environ.get('F', 64)), int(os.environ.get('FILL', 32)), 64
pairs = synthetic.syn_v1_pairs(U, I)
real = sp.csr_matrix((np.ones(len(pairs), np.float32), (pairs[:, 0], pairs[:, 1])), shape=(U, I))
rng = np.random.default_rng(0)


def poisoned():
    rows = np.repeat(np.arange(F), FILL); cols = np.concatenate([np.sort(rng.choice(I, FILL, replace=False)) for _ in range(F)])
    return sp.vstack([real, sp.csr_matrix((np.ones(F * FILL, np.float32), (rows, cols)), shape=(F, I))], format='csr', dtype=np.float32)


enc = GraphEncoder.__new__(GraphEncoder)
torch.nn.Module.__init__(enc)
enc.data = SimpleNamespace(user_num=U + F, item_num=I)
enc.latent_size = enc.emb_size = d
enc.n_prop_layers = 3
enc._eng = None
X = torch.randn(U + F + I, d, device='cuda')
for mode, n_real in (('full build', None), ('incremental', U)):
    for k in range(4):
        m = poisoned()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        enc._init_uiAdj_from_interactions(m, n_real=n_real)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        g = ops.auto_blocked(enc.sparse_norm_adj.graph(), d, split=U + F)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        y = ops.spmm(g, X)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print('%s, call %d: _init_uiAdj %.1f ms, hop plan %.1f ms, first hop %.2f ms (blocked plan: %s)' % (mode, k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), g.blocked is not None), flush=True)
    if mode == 'full build':
        y_full = y
print('last products agree to %.1e (different fake rows: structure check only)' % float((y[:U] - y_full[:U]).abs().max() / y_full.abs().max()))
