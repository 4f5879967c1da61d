"""cProfile of LightGCN.train(Epoch=3) on ml-100k through the class API: where does the host time go?"""
import sys, os, io, contextlib, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from types import SimpleNamespace
from test_host_api import make_data
from arlib_amd.util.tool import seedSet
from arlib_amd.recommender.LightGCN import LightGCN
seedSet(2018)
data = make_data()
rec = LightGCN(SimpleNamespace(dataset='ml-100k', model_name='LightGCN', maxEpoch=30, batch_size=2048, emb_size=64, n_layers=3, reg=1e-4, lRate=0.005, seed=2018, topK='50'), data)
with contextlib.redirect_stdout(io.StringIO()):
    rec.train(Epoch=1, evalNum=1)
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable(); rec.train(Epoch=3, evalNum=1); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
