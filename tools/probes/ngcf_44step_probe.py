"""How far do 44 NGCF steps (2 epochs of ml-100k, d = 32, L = 2) drift from the reference's own run (g22: tables after train(requires_adjgrad=True),
whose updates are those of a plain train())?  Routes: fused engine, autograd with row-subset last layer, autograd full forward."""
import io, contextlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from conftest import golden, rel_err, row_err
from test_host_api import make_data
from test_gpu_api import rec_args
from arlib_amd.util.tool import seedSet
from arlib_amd.recommender.NGCF import NGCF
g = golden('g22_adjgrad_ngcf.npz')
base = None
for route in ('fused', 'fused_perturbed', 'autograd_rows', 'autograd_full', 'adjgrad'):
    seedSet(2018)
    data = make_data()
    rec = NGCF(rec_args(emb_size=32, n_layers=2, model_name='NGCF'), data)
    model = rec.model.cuda()
    with torch.no_grad():
        model.embedding_dict['user_emb'][:] = torch.from_numpy(g['user0']).cuda(); model.embedding_dict['item_emb'][:] = torch.from_numpy(g['item0']).cuda()
        for k in ('w1_0', 'w1_1', 'w2_0', 'w2_1'):
            model.W[k][:] = torch.from_numpy(g['init_W__' + k]).cuda()
        if route == 'fused_perturbed':                          # one rounding error's worth of noise on the initial tables: how much of the drift is the trajectory's own sensitivity?
            gen = torch.Generator().manual_seed(1)
            for t in (model.embedding_dict['user_emb'], model.embedding_dict['item_emb']):
                t.mul_(1.0 + 1e-7 * torch.randn(t.shape, generator=gen).cuda())
    with contextlib.redirect_stdout(io.StringIO()):
        if route in ('fused', 'fused_perturbed'):
            rec.train(Epoch=2, evalNum=1)
        elif route == 'adjgrad':
            rec.train(requires_adjgrad=True, Epoch=2, evalNum=1)
        else:
            rec.rows_forward = route == 'autograd_rows'
            rec._fusable = lambda opt: None
            rec.train(Epoch=2, evalNum=1)
    u, i = rec.model.embedding_dict['user_emb'].detach().cpu().numpy(), rec.model.embedding_dict['item_emb'].detach().cpu().numpy()
    if route == 'fused':
        base = (u.copy(), i.copy())
    if route == 'fused_perturbed':
        print('fused vs fused with 1e-7 relative noise on the initial tables, after 44 steps: max-norm %.2e / %.2e' % (rel_err(u, base[0]), rel_err(i, base[1])), flush=True)
    print('%-14s tables vs reference after 44 steps: max-norm %.2e / %.2e, row-wise %.2e / %.2e' % (route, rel_err(u, g['user']), rel_err(i, g['item']), row_err(u, g['user']), row_err(i, g['item'])), flush=True)
