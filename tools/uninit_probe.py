"""Debug aid: every torch.empty / empty_like / new_empty float buffer is filled with NaN, then one NGCF epoch runs through the autograd route and
through the fused route; the first arlib_amd.ops call whose float OUTPUT contains a NaN while its float inputs do not is reported (an op that
reads memory nobody wrote).   python3 tools/uninit_probe.py"""
import sys, io, contextlib, functools, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
_empty, _empty_like = torch.empty, torch.empty_like
def _nan(t):
    if t.is_floating_point() and t.numel():
        t.fill_(float('nan'))
    elif t.dtype in (torch.int32, torch.int64) and t.numel():
        t.fill_(-7)
    return t
torch.empty = lambda *a, **k: _nan(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _nan(_empty_like(*a, **k))
import test_gpu_api as T
from arlib_amd import ops
from arlib_amd.util.tool import seedSet
from arlib_amd.recommender.NGCF import NGCF
reported = set()
def has_nan(x):
    return isinstance(x, torch.Tensor) and x.is_floating_point() and x.numel() and bool(torch.isnan(x).any())
def flat(args):
    for a in args:
        if isinstance(a, (list, tuple)): yield from flat(a)
        else: yield a
def wrap(name, fn):
    @functools.wraps(fn)
    def w(*a, **k):
        ins = [x for x in flat(list(a) + list(k.values())) if isinstance(x, torch.Tensor)]
        nan_in = any(has_nan(x) for x in ins if k.get('out') is not x)
        r = fn(*a, **k)
        outs = [x for x in flat([r])] + ([k['out']] if 'out' in k and k['out'] is not None else [])
        if any(has_nan(x) for x in outs) and not nan_in and name not in reported:
            reported.add(name)
            print('UNINITIALISED READ? op %s produced NaN from NaN-free inputs; shapes %s' % (name, [tuple(x.shape) for x in ins]), flush=True)
        return r
    return w
for name in dir(ops):
    f = getattr(ops, name)
    if callable(f) and not name.startswith('_') and getattr(f, '__module__', '') == 'arlib_amd.ops' and not isinstance(f, type):
        setattr(ops, name, wrap(name, f))
for fused in (False, True):
    seedSet(2018)
    rec = NGCF(T.rec_args(emb_size=32, n_layers=2, model_name='NGCF'), T.make_data())
    with contextlib.redirect_stdout(io.StringIO()):
        rec.train(Epoch=1, evalNum=1, requires_embgrad=not fused)
    tabs = [rec.model.embedding_dict[k].detach() for k in ('user_emb', 'item_emb')]
    print('route fused=%s: NaN in tables: %s' % (fused, [bool(torch.isnan(t).any()) for t in tabs]), flush=True)
print('done; ops flagged:', sorted(reported))
