"""Where does the wall clock of a whole posionDataAttack() go beyond toy size?  Class API end to end on a SYN-v1 instance
(default 50K users x 10K items, ~1.6M interactions), cProfile of the host side.   ATTACK=CLeaR|DLAttack|PGA python3 tools/attack_scale.py"""
import os, sys, time, io, contextlib, cProfile, pstats, importlib
from types import SimpleNamespace
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from arlib_amd.util import synthetic
from arlib_amd.util.DataLoader import DataLoader
from arlib_amd.util.tool import seedSet
from arlib_amd.recommender.LightGCN import LightGCN

U, I = int(os.environ.get('U', 50_000)), int(os.environ.get('I', 10_000))
name = os.environ.get('ATTACK', 'CLeaR')
os.chdir(os.environ.get('TMPDIR', '/tmp'))
t0 = time.perf_counter()
pairs = synthetic.syn_v1_pairs(U, I, mean_deg=32.0, seed=2018)
rng = np.random.default_rng(0)
test_sel = rng.random(len(pairs)) < 0.02
tr, te = pairs[~test_sel], pairs[test_sel]
one = lambda p: (p[:, 0], p[:, 1], np.ones(len(p)))
seedSet(2018)
data = DataLoader.from_arrays(one(tr), one(te[:1000]), one(te), dataName='synM')
print('DataLoader over %d interactions: %.1f s' % (len(tr), time.perf_counter() - t0), flush=True)
rec_args = SimpleNamespace(dataset='synM', model_name='LightGCN', maxEpoch=1, batch_size=2048, emb_size=64, n_layers=2, reg=1e-4, lRate=0.005, seed=2018, topK='50')
atk_args = SimpleNamespace(maliciousUserSize=16, maliciousFeedbackSize=0, Epoch=1, innerEpoch=1, outerEpoch=2, attackTargetChooseWay='unpopular', targetSize=5,
                           dataset='synM', attackModelName=name)
t0 = time.perf_counter()
rec = LightGCN(rec_args, data)
with contextlib.redirect_stdout(io.StringIO()):
    rec.train(Epoch=1, evalNum=1)
torch.cuda.synchronize()
print('LightGCN build + 1 epoch (%d steps) + evaluation: %.1f s' % ((len(tr) + 2047) // 2048, time.perf_counter() - t0), flush=True)
mod = importlib.import_module('arlib_amd.attack.White.' + name)
atk = getattr(mod, name)(atk_args, data)
pr = cProfile.Profile()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    res = atk.posionDataAttack(rec)
    torch.cuda.synchronize()
    pr.disable()
print('%s.posionDataAttack: %.1f s -> %s, fake rows sum %s' % (name, time.perf_counter() - t0, res.shape, np.asarray(res[U:].sum(1)).ravel()[:4]), flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print('\n'.join(l[:150] for l in s.getvalue().splitlines()[4:44]))
