"""score_mask_topk at cfg2 size (1 M x 100 K, d = 64, k = 50) on tables with and without norm structure: time per pass (cold and warm-started) and
the share of the item stream the exact early exit skipped.  A/B builds: ARLIB_AMD_LIB=<variant .so> (make variant ...).
  random      -- i.i.d. normal tables: item norms within a few percent of each other, nothing to skip
  propagated  -- xavier tables after ONE normalised-adjacency hop on the SYN-v1 graph (what an attack's surrogate looks like early on):
                 item norms follow popularity
  skewed      -- random directions, item norms log-normal (sigma 1): the norm-ordered stream's tail is far below every threshold
    python3 tools/topk_exit_bench.py        env: U, I, KINDS=random,propagated,skewed, ORDER_USERS=1 (users sorted by warm threshold / norm)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
from arlib_amd.util import synthetic

U, I, d, k = int(os.environ.get('U', 1_000_000)), int(os.environ.get('I', 100_000)), 64, 50
kinds = os.environ.get('KINDS', 'random,propagated,skewed').split(',')
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)


def tables(kind):
    if kind == 'random':
        return torch.randn(U, d, device=dev, generator=g) * 0.1, torch.randn(I, d, device=dev, generator=g) * 0.1
    if kind == 'skewed':
        Pi = torch.randn(I, d, device=dev, generator=g) * 0.1
        Pi *= torch.exp(torch.randn(I, 1, device=dev, generator=g))
        return torch.randn(U, d, device=dev, generator=g) * 0.1, Pi
    data = synthetic.syn_v1(U, I, 32.0, 2018)
    nnz = data.training_size()[2]
    rowptr, col = data.adjacency_pattern()
    col_d = torch.from_numpy(col).to(dev)
    val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(dev), col_d, torch.ones(2 * nnz, device=dev), U + I)
    A = ops.CSRGraph(rowptr, col_d, val, dev, validate=False)
    torch.manual_seed(3)
    X = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, d)), torch.nn.init.xavier_uniform_(torch.empty(I, d))], 0).to(dev)
    X = ops.spmm(A, X)
    return X[:U].contiguous(), X[U:].contiguous()


def timed(fn, n=3):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0))
    return sorted(out)[len(out) // 2], r


print('lib', os.environ.get('ARLIB_AMD_LIB', 'default'), ' U=%d I=%d d=%d k=%d' % (U, I, d, k))
ops.score_mask_topk(torch.randn(256, d, device=dev), torch.randn(I, d, device=dev), k); torch.cuda.synchronize()
for kind in kinds:
    Pu, Pi = tables(kind)
    ops.reset_exit_probe()
    nrm = torch.linalg.vector_norm(Pi, dim=1)
    ops.TOPK_STATS['record_exit'], ops.TOPK_STATS['exit'] = True, []
    cold, (idx, val) = timed(lambda: ops.score_mask_topk(Pu, Pi, k))
    warm, _ = timed(lambda: ops.score_mask_topk(Pu, Pi, k, warm_idx=idx))
    fr = ops.topk_exit_fractions()
    digest = '%d/%.6e' % (int(idx.long().sum()), float(val.double().sum()))       # same lists and values across builds
    line = '%-10s item-norm min/median/max %.3g/%.3g/%.3g: cold %.1f ms (skipped %.3f), warm %.1f ms (skipped %.3f)' % (
        kind, nrm.min().item(), nrm.median().item(), nrm.max().item(), cold, fr[1], warm, fr[4])
    if os.environ.get('ORDER_USERS') == '1':
        # users sorted by (k-th best score) / |a_u| from the previous result: workgroups of 256 users with similar exit points
        ratio = val[:, -1] / torch.linalg.vector_norm(Pu, dim=1).clamp_min(1e-30)
        perm = torch.argsort(ratio, descending=True)
        Pu2, idx2 = Pu[perm].contiguous(), idx[perm].contiguous()
        c2, (i2, v2) = timed(lambda: ops.score_mask_topk(Pu2, Pi, k))
        w2, _ = timed(lambda: ops.score_mask_topk(Pu2, Pi, k, warm_idx=idx2))
        fr2 = ops.topk_exit_fractions()
        assert torch.equal(i2, idx2) and torch.equal(v2, val[perm])
        line += ' | users ordered: cold %.1f (skipped %.3f), warm %.1f (skipped %.3f)' % (c2, fr2[1], w2, fr2[4])
    print(line + '  [digest ' + digest + ']', flush=True)
    ops.TOPK_STATS['record_exit'] = False
    del Pu, Pi
    torch.cuda.empty_cache()
