"""Wall-clock of the reference-shaped class API on ml-100k (fixture data): X(args, data).train(Epoch=3) incl. sampler,
per-epoch evaluation and best-epoch keep -- the thing a user of the reference actually runs (config 1 scale)."""
import sys, os, time, io, contextlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from types import SimpleNamespace
from test_host_api import make_data
from arlib_amd.util.tool import seedSet
from arlib_amd.recommender.GMF import GMF
from arlib_amd.recommender.LightGCN import LightGCN
from arlib_amd.recommender.SimGCL import SimGCL
from arlib_amd.recommender.NGCF import NGCF

def args(**kw):
    a = dict(dataset='ml-100k', model_name='X', maxEpoch=30, batch_size=2048, emb_size=64, n_layers=3, reg=1e-4, lRate=0.005, seed=2018, topK='50')
    a.update(kw); return SimpleNamespace(**a)

for name, cls, kw in (('GMF', GMF, {}), ('LightGCN L=3', LightGCN, {}), ('SimGCL', SimGCL, {}), ('NGCF L=2', NGCF, dict(n_layers=2))):
    seedSet(2018)
    data = make_data()
    rec = cls(args(**kw), data)
    with contextlib.redirect_stdout(io.StringIO()):
        rec.train(Epoch=1, evalNum=1)                  # warm-up (kernel load, allocator)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rec.train(Epoch=3, evalNum=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = 3 * 22
    print('%-13s train(Epoch=3): %.3f s  (%.2f ms per step incl. sampler + 3 evaluations; %d interactions/s)  best %s' %
          (name, dt, 1e3 * dt / steps, 3 * 44212 / dt, {k: round(v, 4) for k, v in rec.bestPerformance[1].items()}), flush=True)
