"""Diagnostic (GPU): NGCF d = 128, L = 3 on the reference-captured g9_ngcf128 -- where do the fused route (engine.step_ngcf) and the autograd
route (forward_rows + backward) differ from the reference gradient, entry by entry?  Prints per-route error statistics conditioned on |g|."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_api import rec_args                     # noqa: E402
from test_host_api import make_data                   # noqa: E402
from conftest import golden                            # noqa: E402


def stats(name, got, ref):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    err = np.abs(got - ref); mx = np.abs(ref).max()
    print('%-28s max|d|/max|ref| %.2e   rms|d|/max %.2e' % (name, err.max() / mx, np.sqrt((err ** 2).mean()) / mx))
    a = np.abs(ref)
    for lo, hi in ((0, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1)):
        sel = (a >= lo) & (a < hi)
        if sel.any():
            print('    |ref| in [%.0e, %.0e): n = %7d   max|d| %.2e   median|d| %.2e' % (lo, hi, sel.sum(), err[sel].max(), np.median(err[sel])))


def main():
    from arlib_amd.recommender.NGCF import NGCF
    from arlib_amd.util.loss import bpr_loss, l2_reg_loss
    g = golden('g9_ngcf128.npz')
    data = make_data()
    emb, L = 128, 3

    def fresh():
        rec = NGCF(rec_args(emb_size=emb, n_layers=L, model_name='NGCF'), data)
        model = rec.model.cuda()
        with torch.no_grad():
            model.embedding_dict['user_emb'][:] = torch.from_numpy(g['user0']).cuda(); model.embedding_dict['item_emb'][:] = torch.from_numpy(g['item0']).cuda()
            for k in range(L):
                model.W['w1_%d' % k][:] = torch.from_numpy(g['w1_%d' % k]).cuda(); model.W['w2_%d' % k][:] = torch.from_numpy(g['w2_%d' % k]).cuda()
        return rec, model
    U = data.user_num
    bu, bp, bn = (torch.from_numpy(g[x][0].astype(np.int32)).cuda() for x in ('batch_u', 'batch_p', 'batch_n'))
    B = bu.numel()
    # full autograd forward (model()) -- the form the golden comparison of grads uses
    rec0, m0 = fresh()
    ue, ie = m0()
    loss = bpr_loss(ue[bu.long()], ie[bp.long()], ie[bn.long()]) + l2_reg_loss(1e-4, ue[bu.long()], ie[bp.long()])
    loss.backward()
    stats('full-forward autograd: user', m0.embedding_dict['user_emb'].grad.cpu().numpy(), g['grad_user'])
    stats('full-forward autograd: w1_0', m0.W['w1_0'].grad.cpu().numpy(), g['grad_w1_0'])
    stats('full-forward autograd: w2_2', m0.W['w2_2'].grad.cpu().numpy(), g['grad_w2_2'])
    # rows route
    rec1, m1 = fresh()
    out_r = m1.forward_rows(torch.cat([bu, bp + U, bn + U]))
    loss = bpr_loss(out_r[:B], out_r[B:2 * B], out_r[2 * B:]) + l2_reg_loss(1e-4, out_r[:B], out_r[B:2 * B])
    loss.backward()
    stats('rows autograd: user', m1.embedding_dict['user_emb'].grad.cpu().numpy(), g['grad_user'])
    stats('rows autograd: w1_0', m1.W['w1_0'].grad.cpu().numpy(), g['grad_w1_0'])
    stats('rows autograd: w2_2', m1.W['w2_2'].grad.cpu().numpy(), g['grad_w2_2'])
    # fused route
    rec2, m2 = fresh()
    opt = torch.optim.Adam(m2.parameters(), lr=0.005)
    eng = m2._engine(1e-4, 0.005, 'adam'); eng.reg = 1e-4
    rec2._bind_optimizer_state(eng, opt, 'adam')
    cap = {}
    eng.step_ngcf(bu, bp, bn, capture=cap)
    stats('fused: user', cap['table'][:U].cpu().numpy(), g['grad_user'])
    stats('fused: w1_0', cap['W'][0][:emb].cpu().numpy(), g['grad_w1_0'])
    stats('fused: w2_2', cap['W'][2][emb:].cpu().numpy(), g['grad_w2_2'])
    stats('fused vs rows-autograd: user', cap['table'][:U].cpu().numpy(), m1.embedding_dict['user_emb'].grad.cpu().numpy())
    stats('fused vs rows-autograd: item', cap['table'][U:].cpu().numpy(), m1.embedding_dict['item_emb'].grad.cpu().numpy())
    # three fused steps vs golden tables
    rec3, m3 = fresh()
    opt3 = torch.optim.Adam(m3.parameters(), lr=0.005)
    eng3 = m3._engine(1e-4, 0.005, 'adam'); eng3.reg = 1e-4
    rec3._bind_optimizer_state(eng3, opt3, 'adam')
    for k in range(3):
        b = [torch.from_numpy(g[x][k].astype(np.int32)).cuda() for x in ('batch_u', 'batch_p', 'batch_n')]
        eng3.step_ngcf(*b)
    for nm, got, ref in (('user_k3', m3.embedding_dict['user_emb'], g['user_k3']), ('item_k3', m3.embedding_dict['item_emb'], g['item_k3']), ('w1_0_k3', m3.W['w1_0'], g['w1_0_k3'])):
        got = got.detach().cpu().numpy()
        err = np.abs(got - ref)
        print('fused 3 steps %-8s max|d| %.2e (max|ref| %.2e) frac>1e-5 %.4f frac>1e-4 %.4f' % (nm, err.max(), np.abs(ref).max(), (err > 1e-5).mean(), (err > 1e-4).mean()))


if __name__ == '__main__':
    main()
