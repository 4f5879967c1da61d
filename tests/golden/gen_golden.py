#!/usr/bin/env python3
"""Golden-vector generator: runs the *reference itself* (CoderWZW/ARLib, mounted read-only at
/root/reference) on CPU in this container and stores inputs + expected outputs as small .npz
fixtures next to this script.  The reference never travels to the GPU box; these fixtures do.

Only data is written (inputs and expected outputs); no reference source is copied.

Harness-side shims (the reference hard-codes .cuda() and imports numba):
  * sys.modules['numba'] = stub with an identity `jit`
  * torch.Tensor.cuda / nn.Module.cuda = identity
  * SimGCL noise: torch.rand_like is patched *during the reference call* to hand out
    pre-generated tensors so the product can be fed the same noise.

Usage:  python tests/golden/gen_golden.py            (writes tests/golden/*.npz)
"""
import os, sys, types, random, hashlib, tempfile, copy
from types import SimpleNamespace

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

numba_stub = types.ModuleType('numba')
numba_stub.jit = lambda *a, **k: (lambda f: f)
sys.modules['numba'] = numba_stub

import numpy as np
import torch
import scipy.sparse as sp

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
torch.set_num_threads(4)

# the reference writes ./log, ./modelsaved, ./data/... relative to cwd
SCRATCH = tempfile.mkdtemp(prefix='arl_golden_')
os.chdir(SCRATCH)
os.makedirs('data/clean', exist_ok=True)
os.symlink(os.path.join(REF, 'data/clean/ml-100k'), 'data/clean/ml-100k-ro')

from util.tool import seedSet                      # noqa: E402
from util.DataLoader import DataLoader             # noqa: E402
from util import sampler as ref_sampler            # noqa: E402
from util import loss as ref_loss                  # noqa: E402
from recommender.GMF import GMF                    # noqa: E402
from recommender.LightGCN import LightGCN          # noqa: E402
from recommender.SimGCL import SimGCL              # noqa: E402


def rec_args(**kw):
    a = dict(dataset='ml-100k', data_path=REF + '/data/clean/', training_data='/train.txt',
             val_data='/val.txt', test_data='/test.txt', model_name='LightGCN', maxEpoch=30,
             batch_size=2048, emb_size=64, n_layers=3, reg=1e-4, lRate=0.005, dropout=True,
             dropout_rate=0.3, cuda=True, gpu_id='0', seed=2018, topK='50', load=False, save=False,
             save_dir='./modelsaved/')
    a.update(kw)
    return SimpleNamespace(**a)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print('wrote %s (%.1f KB)' % (name, os.path.getsize(path) / 1024))


def sha_batches(batches):
    h = hashlib.sha256()
    for u, p, n in batches:
        h.update(np.asarray([u, p, n], dtype=np.int32).tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------- dataset (data, not code)
def gen_dataset():
    def raw(fn):
        rows = [l.split() for l in open(REF + '/data/clean/ml-100k/' + fn)]
        return (np.array([int(r[0]) for r in rows], np.int32), np.array([int(r[1]) for r in rows], np.int32),
                np.array([float(r[2]) for r in rows], np.float32))
    tu, ti, tr = raw('train.txt')
    vu, vi, vr = raw('val.txt')
    su, si, sr = raw('test.txt')
    save('ml100k_data.npz', train_u=tu, train_i=ti, train_r=tr, val_u=vu, val_i=vi, val_r=vr,
         test_u=su, test_i=si, test_r=sr)


# --------------------------------------------------------------------------- G1 sampler
def gen_sampler():
    args = rec_args()
    seedSet(2018)
    data = DataLoader(args)
    out = {}
    U, I, nnz = data.training_size()
    out['sizes'] = np.array([U, I, nnz], np.int64)
    # internal ids of the file-order training pairs (pre-shuffle)
    out['pairs0'] = np.array([[data.user[r[0]], data.item[r[1]]] for r in data.training_data], np.int32)
    random.seed(2018)
    for ep in range(2):                          # second epoch pins the in-place shuffle carry-over (Q7)
        bs = list(ref_sampler.next_batch_pairwise(data, 2048))
        out['ep%d_u' % ep] = np.concatenate([np.asarray(b[0], np.int32) for b in bs])
        out['ep%d_p' % ep] = np.concatenate([np.asarray(b[1], np.int32) for b in bs])
        out['ep%d_n' % ep] = np.concatenate([np.asarray(b[2], np.int32) for b in bs])
        out['ep%d_sha' % ep] = np.frombuffer(sha_batches(bs).encode(), np.uint8)
        out['ep%d_nb' % ep] = np.array([len(bs), len(bs[-1][0])], np.int64)
    out['next_random'] = np.array([random.random()], np.float64)
    st = random.getstate()
    out['mt_state_after'] = np.array(st[1], np.uint32)
    # ragged batch size + different seed
    random.seed(7)
    bs = list(ref_sampler.next_batch_pairwise(data, 1000))
    out['b1000_u'] = np.concatenate([np.asarray(b[0], np.int32) for b in bs])
    out['b1000_p'] = np.concatenate([np.asarray(b[1], np.int32) for b in bs])
    out['b1000_n'] = np.concatenate([np.asarray(b[2], np.int32) for b in bs])
    out['b1000_next_random'] = np.array([random.random()], np.float64)
    save('g1_sampler.npz', **out)

    # heavy-rejection toy: 6 users x 5 items, most users hold 4 of 5 items
    toy = SimpleNamespace()
    pairs = [(u, i) for u in range(6) for i in range(5) if (u + i) % 5 != 0 or u == 5]
    pairs = [p for p in pairs if not (p[0] == 5 and p[1] >= 2)]
    toy.user, toy.item = {}, {}
    toy.training_data = []
    from collections import defaultdict
    toy.training_set_u = defaultdict(dict)
    for u, i in pairs:
        us, it = 'u%d' % u, 'i%d' % i
        if us not in toy.user: toy.user[us] = len(toy.user)
        if it not in toy.item: toy.item[it] = len(toy.item)
        toy.training_data.append([us, it, 1.0])
        toy.training_set_u[us][it] = 1.0
    o = {'pairs0': np.array([[toy.user[r[0]], toy.item[r[1]]] for r in toy.training_data], np.int32),
         'sizes': np.array([len(toy.user), len(toy.item), len(toy.training_data)], np.int64)}
    random.seed(12345678901234567890)             # > 64-bit seed exercises init_by_array with 3 key words
    us, ps, ns = [], [], []
    for ep in range(5):
        for b in ref_sampler.next_batch_pairwise(toy, 7):
            us += b[0]; ps += b[1]; ns += b[2]
    o['u'], o['p'], o['n'] = (np.array(x, np.int32) for x in (us, ps, ns))
    o['next_random'] = np.array([random.random()], np.float64)
    save('g1_sampler_toy.npz', **o)
    return data


# --------------------------------------------------------------------------- G2 / G6 losses
def gen_losses():
    g = torch.Generator().manual_seed(11)
    out = {}
    B, d = 2048, 64
    for tag, scale, B in (('n', 0.1, 1024), ('sat', 6.0, 256)):   # 'sat' drives |x| > 30: sigmoid saturation + 1e-7 eps
        u = (torch.randn(B, d, generator=g) * scale).requires_grad_()
        p = (torch.randn(B, d, generator=g) * scale).requires_grad_()
        n = (torch.randn(B, d, generator=g) * scale).requires_grad_()
        l = ref_loss.bpr_loss(u, p, n)
        r = ref_loss.l2_reg_loss(1e-4, u, p)
        (l + r).backward()
        out.update({tag + '_u': u.detach().numpy(), tag + '_p': p.detach().numpy(), tag + '_n': n.detach().numpy(),
                    tag + '_bpr': np.array([l.item()], np.float32), tag + '_reg': np.array([r.item()], np.float32),
                    tag + '_du': u.grad.numpy(), tag + '_dp': p.grad.numpy(), tag + '_dn': n.grad.numpy()})
    # gather form with duplicate indices (scatter-add accumulates duplicates)
    T = torch.randn(300, d, generator=g) * 0.1
    T.requires_grad_()
    B = 2048
    ui = torch.randint(0, 100, (B,), generator=g)
    pi = torch.randint(100, 300, (B,), generator=g)
    ni = torch.randint(100, 300, (B,), generator=g)
    l = ref_loss.bpr_loss(T[ui], T[pi], T[ni]) + ref_loss.l2_reg_loss(1e-4, T[ui], T[pi])
    l.backward()
    out.update(dup_T=T.detach().numpy(), dup_ui=ui.numpy().astype(np.int32), dup_pi=pi.numpy().astype(np.int32),
               dup_ni=ni.numpy().astype(np.int32), dup_loss=np.array([l.item()], np.float32), dup_dT=T.grad.numpy())
    save('g2_losses.npz', **out)

    out = {}
    for tag, n_, d_ in (('a', 700, 64), ('b', 2048, 16), ('c', 33, 16)):
        v1 = torch.randn(n_, d_, generator=g).requires_grad_()
        v2 = torch.randn(n_, d_, generator=g).requires_grad_()
        l = ref_loss.InfoNCE(v1, v2, 0.2)
        l.backward()
        out.update({tag + '_v1': v1.detach().numpy(), tag + '_v2': v2.detach().numpy(),
                    tag + '_loss': np.array([l.item()], np.float32),
                    tag + '_dv1': v1.grad.numpy(), tag + '_dv2': v2.grad.numpy()})
    save('g6_infonce.npz', **out)


# --------------------------------------------------------------------------- G3 adjacency
def gen_adj(data):
    na = data.norm_adj.tocsr()
    na.sort_indices()
    out = dict(norm_indptr=na.indptr.astype(np.int64), norm_indices=na.indices.astype(np.int32),
               norm_data=na.data.astype(np.float32))
    # _init_uiAdj on a weighted symmetric adjacency with fractional weights and one isolated node
    rng = np.random.RandomState(5)
    U_, I_ = 50, 40
    R = sp.random(U_, I_, density=0.15, random_state=rng, format='lil', dtype=np.float32)
    R[7, :] = 0                                    # isolated user -> 1/sqrt(0)=inf, no guard (Q15)
    R = R.tocsr(); R.eliminate_zeros()
    A = sp.bmat([[None, R], [R.T, None]], format='csr', dtype=np.float32)
    args = rec_args(emb_size=8, n_layers=2)
    m = LightGCN(args, data).model
    with np.errstate(divide='ignore'):
        m._init_uiAdj(A)
    t = m.sparse_norm_adj.coalesce()
    Rc = R.tocoo()
    out.update(w_R_row=Rc.row.astype(np.int32), w_R_col=Rc.col.astype(np.int32), w_R_val=Rc.data.astype(np.float32),
               w_shape=np.array([U_, I_], np.int64),
               w_norm_row=t.indices()[0].numpy().astype(np.int32), w_norm_col=t.indices()[1].numpy().astype(np.int32),
               w_norm_val=t.values().numpy())
    save('g3_adj.npz', **out)


# --------------------------------------------------------------------------- G4 forward / G5 steps
def gen_forward_and_steps(data):
    out = {}
    seedSet(2018)
    for L in (1, 2, 3):
        args = rec_args(emb_size=32, n_layers=L)
        rec = LightGCN(args, data)
        if L == 1:
            out['lgn_user0'] = rec.model.embedding_dict['user_emb'].detach().numpy().copy()
            out['lgn_item0'] = rec.model.embedding_dict['item_emb'].detach().numpy().copy()
        with torch.no_grad():       # same tables for every L
            rec.model.embedding_dict['user_emb'][:] = torch.from_numpy(out['lgn_user0'])
            rec.model.embedding_dict['item_emb'][:] = torch.from_numpy(out['lgn_item0'])
            ue, ie = rec.model()
        out['lgn_L%d_user' % L] = ue.numpy().copy()
        out['lgn_L%d_item' % L] = ie.numpy().copy()
    save('g4_forward.npz', **out)

    # --- G5: k training steps, exactly the reference loop body (LightGCN.py:47-64 / GMF.py:39-54)
    def run_steps(cls, args, k_snap, opt='adam'):
        seedSet(2018)
        rec = cls(args, data)
        model = rec.model
        if opt == 'adam':
            optim = torch.optim.Adam(model.parameters(), lr=args.lRate)
        else:
            optim = torch.optim.SGD(model.parameters(), lr=args.lRate / 10)     # PGA.py:59
        o = {'user0': model.embedding_dict['user_emb'].detach().numpy().copy(),
             'item0': model.embedding_dict['item_emb'].detach().numpy().copy()}
        losses, bu, bp, bn = [], [], [], []
        random.seed(2018)
        d2 = copy.copy(data); d2.training_data = [list(r) for r in data_training0]
        step = 0
        done = False
        while not done:
            for batch in ref_sampler.next_batch_pairwise(d2, args.batch_size):
                user_idx, pos_idx, neg_idx = batch
                rec_user_emb, rec_item_emb = model()
                user_emb, pos_item_emb, neg_item_emb = rec_user_emb[user_idx], rec_item_emb[pos_idx], rec_item_emb[neg_idx]
                batch_loss = ref_loss.bpr_loss(user_emb, pos_item_emb, neg_item_emb) + \
                    ref_loss.l2_reg_loss(args.reg, user_emb, pos_item_emb)
                optim.zero_grad()
                batch_loss.backward()
                if step == 0:
                    o['grad_user_step0'] = model.embedding_dict['user_emb'].grad.numpy().copy()
                    o['grad_item_step0'] = model.embedding_dict['item_emb'].grad.numpy().copy()
                optim.step()
                losses.append(batch_loss.item())
                bu.append(np.asarray(user_idx, np.int32)); bp.append(np.asarray(pos_idx, np.int32)); bn.append(np.asarray(neg_idx, np.int32))
                step += 1
                if step in k_snap:
                    o['user_k%d' % step] = model.embedding_dict['user_emb'].detach().numpy().copy()
                    o['item_k%d' % step] = model.embedding_dict['item_emb'].detach().numpy().copy()
                    if opt == 'adam' and step == max(k_snap):
                        st = optim.state[model.embedding_dict['user_emb']]
                        o['m_user'] = st['exp_avg'].numpy().copy(); o['v_user'] = st['exp_avg_sq'].numpy().copy()
                        st = optim.state[model.embedding_dict['item_emb']]
                        o['m_item'] = st['exp_avg'].numpy().copy(); o['v_item'] = st['exp_avg_sq'].numpy().copy()
                if step >= max(k_snap):
                    done = True
                    break
        o['losses'] = np.array(losses, np.float32)
        o['batch_u'] = np.concatenate(bu); o['batch_p'] = np.concatenate(bp); o['batch_n'] = np.concatenate(bn)
        o['batch_sizes'] = np.array([len(x) for x in bu], np.int64)
        return o

    o = run_steps(LightGCN, rec_args(emb_size=64, n_layers=3), {1, 3, 10})
    assert abs(o['losses'][0] - 0.6937738) < 1e-5, o['losses'][0]      # SURVEY 8c anchor
    save('g5_lightgcn_adam.npz', **o)
    o = run_steps(GMF, rec_args(emb_size=64, model_name='GMF'), {3, 25})   # 25 > one epoch (22 batches)
    save('g5_gmf_adam.npz', **o)
    o = run_steps(LightGCN, rec_args(emb_size=16, n_layers=2), {3}, opt='sgd')
    save('g5_lightgcn_sgd.npz', **o)


# --------------------------------------------------------------------------- SimGCL with injected noise
def gen_simgcl(data):
    args = rec_args(emb_size=16, n_layers=2, model_name='SimGCL')
    seedSet(2018)
    rec = SimGCL(args, data)
    model = rec.model
    N = data.user_num + data.item_num
    g = torch.Generator().manual_seed(77)
    noises = [torch.rand(N, 16, generator=g) for _ in range(4)]       # 2 views x 2 hops, in call order
    o = {'user0': model.embedding_dict['user_emb'].detach().numpy().copy(),
         'item0': model.embedding_dict['item_emb'].detach().numpy().copy(),
         'noise': torch.stack(noises).numpy()}
    random.seed(2018)
    d2 = copy.copy(data); d2.training_data = [list(r) for r in data_training0]
    batch = next(iter(ref_sampler.next_batch_pairwise(d2, 2048)))
    user_idx, pos_idx, neg_idx = batch
    orig = torch.rand_like
    it = iter(noises)
    torch.rand_like = lambda x, *a, **k: next(it)
    try:
        optim = torch.optim.Adam(model.parameters(), lr=args.lRate)
        rec_user_emb, rec_item_emb = model()
        user_emb, pos_item_emb, neg_item_emb = rec_user_emb[user_idx], rec_item_emb[pos_idx], rec_item_emb[neg_idx]
        rec_loss = ref_loss.bpr_loss(user_emb, pos_item_emb, neg_item_emb)
        cl_loss = rec.cl_rate * model.cal_cl_loss([user_idx, pos_idx])
        batch_loss = rec_loss + ref_loss.l2_reg_loss(args.reg, user_emb, pos_item_emb) + cl_loss
        optim.zero_grad()
        batch_loss.backward()
        o['grad_user'] = model.embedding_dict['user_emb'].grad.numpy().copy()
        o['grad_item'] = model.embedding_dict['item_emb'].grad.numpy().copy()
        optim.step()
    finally:
        torch.rand_like = orig
    o['rec_loss'] = np.array([rec_loss.item()], np.float32)
    o['cl_loss'] = np.array([cl_loss.item()], np.float32)
    o['user_k1'] = model.embedding_dict['user_emb'].detach().numpy().copy()
    o['item_k1'] = model.embedding_dict['item_emb'].detach().numpy().copy()
    o['batch_u'] = np.asarray(user_idx, np.int32); o['batch_p'] = np.asarray(pos_idx, np.int32); o['batch_n'] = np.asarray(neg_idx, np.int32)
    # forward-only outputs (unperturbed + perturbed with noises[0:2]) from the *initial* tables
    seedSet(2018)
    rec2 = SimGCL(args, data)
    with torch.no_grad():
        u0, i0 = rec2.model()
        it2 = iter(noises[:2])
        torch.rand_like = lambda x, *a, **k: next(it2)
        try:
            u1, i1 = rec2.model(perturbed=True)
        finally:
            torch.rand_like = orig
    o['fwd_user'] = u0.numpy().copy(); o['fwd_item'] = i0.numpy().copy()
    o['fwdp_user'] = u1.numpy().copy(); o['fwdp_item'] = i1.numpy().copy()
    save('g5_simgcl.npz', **o)


# --------------------------------------------------------------------------- G10: whole train() through the class API
def gen_train_api():
    """Reference `X(args,data).train(Epoch=2)` end to end (sampler + steps + per-epoch evaluation + best-epoch keep)."""
    import io, contextlib
    out = {}
    for name, cls, kw in (('gmf', GMF, dict(emb_size=64, model_name='GMF')), ('lgn', LightGCN, dict(emb_size=64, n_layers=2))):
        args = rec_args(**kw)
        seedSet(2018)
        data = DataLoader(args)
        rec = cls(args, data)
        with contextlib.redirect_stdout(io.StringIO()):
            rec.train(Epoch=2, evalNum=1)
            rec_list, measure = rec.test()
        out[name + '_user'] = rec.model.embedding_dict['user_emb'].detach().numpy().copy()
        out[name + '_item'] = rec.model.embedding_dict['item_emb'].detach().numpy().copy()
        out[name + '_best_user'] = rec.best_user_emb.detach().numpy().copy()
        out[name + '_best_epoch'] = np.array([rec.bestPerformance[0]], np.int64)
        out[name + '_best_perf'] = np.array([rec.bestPerformance[1][k] for k in ('Hit Ratio', 'Precision', 'Recall', 'NDCG')], np.float64)
        out[name + '_measure'] = np.array([float(m.strip().split(':')[1]) for m in measure[1:]], np.float64)
        out[name + '_next_random'] = np.array([random.random()], np.float64)
        out[name + '_predict0'] = rec.predict(data.id2user[0]).astype(np.float32)
    save('g10_train_api.npz', **out)


# --------------------------------------------------------------------------- G7-G9: white-box attacks, traced
# The attacks are run UNMODIFIED; intermediates are captured by wrapping functions they call (no attack line is re-typed):
#   LGCN_Encoder._init_uiAdj -> the weighted adjacency handed over + the model's tables at that moment
#   LGCN_Encoder.forward     -> last propagated tables
#   torch.topk / torch.tanh / torch.randn / Tensor.backward / Adam.step -> inputs / outputs of interest
def _attack_args(**kw):
    a = dict(attackCategory='White', attackModelName='PGA', times=1, poisonDatasetOutPath='data/poison/', poisondataSaveFlag=False,
             maliciousUserSize=3, maliciousFeedbackSize=0, Epoch=1, innerEpoch=1, outerEpoch=1, gradMaxLimitation=1,
             gradNumLimitation=60, gradIterationNum=10, attackTargetChooseWay='unpopular', targetSize=5)
    a.update(kw)
    return SimpleNamespace(**a)


def _scipy_torch_index_shim():
    """Quirk Q13: under scipy >= 1.8 a torch tensor used as a scipy index fails (PGA.py:56,73); convert in the harness."""
    import scipy.sparse._index as spi
    orig = spi.IndexMixin._validate_indices

    def patched(self, key, *a, **k):
        conv = lambda x: x.numpy().copy() if isinstance(x, torch.Tensor) else x
        if isinstance(key, tuple):
            key = tuple(conv(x) for x in key)
        else:
            key = conv(key)
        return orig(self, key, *a, **k)
    spi.IndexMixin._validate_indices = patched
    return lambda: setattr(spi.IndexMixin, '_validate_indices', orig)


def gen_attacks():
    import io, contextlib
    from copy import deepcopy
    import recommender.LightGCN as RL
    from attack.White.PGA import PGA
    from attack.White.DLAttack import DLAttack
    from attack.White.CLeaR import CLeaR
    os.makedirs('data/clean/ml-100k', exist_ok=True)
    undo_shim = _scipy_torch_index_shim()
    rargs = rec_args(emb_size=16, n_layers=2, maxEpoch=1)

    def fresh():
        seedSet(2018)
        data = DataLoader(rargs)
        rec = LightGCN(rargs, data)
        with contextlib.redirect_stdout(io.StringIO()):
            rec.train(Epoch=1, evalNum=5)
        return data, rec

    trace = {}
    orig_init, orig_fwd = RL.LGCN_Encoder._init_uiAdj, RL.LGCN_Encoder.forward
    orig_topk, orig_tanh, orig_randn = torch.topk, torch.tanh, torch.randn
    orig_backward, orig_adam_step = torch.Tensor.backward, torch.optim.Adam.step

    def init_wrap(self, ui_adj):
        trace.setdefault('init', []).append((sp.csr_matrix(ui_adj).copy(), self.embedding_dict['user_emb'].detach().numpy().copy(),
                                             self.embedding_dict['item_emb'].detach().numpy().copy()))
        return orig_init(self, ui_adj)

    def fwd_wrap(self, *a, **k):
        out = orig_fwd(self, *a, **k)
        trace['last_fwd'] = (out[0].detach().numpy().copy(), out[1].detach().numpy().copy())
        return out

    def topk_wrap(inp, k, *a, **kw):
        out = orig_topk(inp, k, *a, **kw)
        if inp.dim() == 2 and inp.shape[0] > 100:
            trace.setdefault('topk', []).append((k, out[1].numpy().copy(), trace.get('last_fwd'), len(trace.get('init', []))))
        return out

    def tanh_wrap(x):
        trace.setdefault('tanh', []).append(x.detach().numpy().copy())
        return orig_tanh(x)

    RL.LGCN_Encoder._init_uiAdj, RL.LGCN_Encoder.forward = init_wrap, fwd_wrap
    torch.topk, torch.tanh = topk_wrap, tanh_wrap
    out = {}
    try:
        # ------------------------------------------------------------------ PGA (a16)
        data, rec = fresh()
        aargs = _attack_args(attackModelName='PGA')
        atk = PGA(aargs, data)
        trace.clear()
        with contextlib.redirect_stdout(io.StringIO()):
            res = atk.posionDataAttack(deepcopy(rec))
        U, I, F = atk.userNum, atk.itemNum, atk.fakeUserNum
        out['pga_sizes'] = np.array([U, I, F, rargs.n_layers, rargs.emb_size], np.int64)
        out['pga_targets'] = np.array(atk.targetItem, np.int32)
        # init[0] = outer-loop _init_uiAdj; init[1..] = per-gradient-step adjacency (PGA.py:93-97)
        steps = trace['init'][1:]
        assert len(steps) == len(trace['tanh'])
        real = steps[0][0][:U, U + F:]
        out['pga_real_indptr'], out['pga_real_indices'] = real.indptr.astype(np.int64), real.indices.astype(np.int32)
        out['pga_user_tab'], out['pga_item_tab'] = steps[0][1], steps[0][2]
        K = 4
        out['pga_S'] = np.stack([np.asarray(steps[s][0][U:U + F, U + F:].todense(), np.float32) for s in range(K + 1)])   # fake block before step s
        out['pga_grad'] = np.stack(trace['tanh'][:K]).astype(np.float32)                                                 # D^-1/2 g D^-1/2 block of step s
        k50 = [t for t in trace['topk'] if t[0] == 50][0]
        out['pga_top50'] = k50[1][:U].astype(np.int32)
        out['pga_result_fake_rows'] = np.asarray(res[U:U + F, :].todense(), np.float32)

        # ------------------------------------------------------------------ DLAttack (a18 scoring+mask+top-k, a17 project)
        data, rec = fresh()
        atk = DLAttack(_attack_args(attackModelName='DLAttack', maliciousUserSize=2), data)
        proj = []
        orig_project = DLAttack.project

        def project_wrap(self, mat, n):
            m, ind = orig_project(self, mat, n)
            proj.append((np.asarray(mat.detach().numpy(), np.float32).copy(), int(n), m.numpy().copy(), ind.numpy().copy()))
            return m, ind
        DLAttack.project = project_wrap
        trace.clear()
        with contextlib.redirect_stdout(io.StringIO()):
            res = atk.posionDataAttack(deepcopy(rec))
        DLAttack.project = orig_project
        tk = trace['topk'][0]
        adj = trace['init'][tk[3] - 1][0]
        Un = tk[2][0].shape[0]
        mask = adj[:Un, Un:]
        out['dl_Pu'], out['dl_Pi'] = tk[2]
        out['dl_mask_indptr'], out['dl_mask_indices'] = mask.indptr.astype(np.int64), mask.indices.astype(np.int32)
        out['dl_topk'] = tk[1].astype(np.int32)
        out['dl_k'] = np.array([tk[0]], np.int64)
        out['dl_proj_in'] = np.stack([p[0] for p in proj]); out['dl_proj_n'] = np.array([p[1] for p in proj], np.int64)
        out['dl_proj_out'] = np.stack([p[2] for p in proj]); out['dl_proj_idx'] = np.stack([p[3] for p in proj]).astype(np.int32)
        out['dl_result_fake_rowsums'] = np.asarray(res[atk.userNum:, :].sum(1)).ravel().astype(np.float32)

        # ------------------------------------------------------------------ CLeaR (a19: CW + SFA loss and parameter gradients)
        data, rec = fresh()
        atk = CLeaR(_attack_args(attackModelName='CLeaR', maliciousUserSize=3), data)
        r_fixed = torch.Generator().manual_seed(99)
        r0 = torch.randn(rargs.emb_size, generator=r_fixed)
        cap = {}

        def randn_wrap(*a, **k):
            if len(a) == 1 and a[0] == rargs.emb_size:
                return r0.clone()
            return orig_randn(*a, **k)

        def backward_wrap(self, *a, **k):
            if 'loss' not in cap:
                cap['loss'] = float(self.item())
            return orig_backward(self, *a, **k)

        def adam_step_wrap(self, *a, **k):
            if 'grads' not in cap and 'loss' in cap:
                ps = self.param_groups[0]['params']
                cap['grads'] = [p.grad.detach().numpy().copy() for p in ps]
                cap['n_init'] = len(trace.get('init', []))
            return orig_adam_step(self, *a, **k)
        torch.randn, torch.Tensor.backward, torch.optim.Adam.step = randn_wrap, backward_wrap, adam_step_wrap
        trace.clear()
        random.seed(4242)
        with contextlib.redirect_stdout(io.StringIO()):
            res = atk.posionDataAttack(deepcopy(rec))
        torch.randn, torch.Tensor.backward, torch.optim.Adam.step = orig_randn, orig_backward, orig_adam_step
        adj, utab, itab = trace['init'][cap['n_init'] - 1]
        Un = utab.shape[0]
        blk = adj[:Un, Un:]
        out['cl_user_tab'], out['cl_item_tab'] = utab, itab
        out['cl_ui_indptr'], out['cl_ui_indices'], out['cl_ui_data'] = blk.indptr.astype(np.int64), blk.indices.astype(np.int32), blk.data.astype(np.float32)
        out['cl_r0'] = r0.numpy()
        out['cl_loss'] = np.array([cap['loss']], np.float32)
        out['cl_grad_user'], out['cl_grad_item'] = cap['grads']
        out['cl_targets'] = np.array(atk.targetItem, np.int32)
        out['cl_sizes'] = np.array([atk.userNum, atk.itemNum, atk.fakeUserNum, min(rec.topN)], np.int64)
        out['cl_result_fake_rowsums'] = np.asarray(res[atk.userNum:, :].sum(1)).ravel().astype(np.float32)
    finally:
        RL.LGCN_Encoder._init_uiAdj, RL.LGCN_Encoder.forward = orig_init, orig_fwd
        torch.topk, torch.tanh, torch.randn = orig_topk, orig_tanh, orig_randn
        torch.Tensor.backward, torch.optim.Adam.step = orig_backward, orig_adam_step
        undo_shim()
    save('g7_attacks.npz', **out)

# --------------------------------------------------------------------------- BiLevelAttackBatch: CW step, relaxProject
def gen_fake_rows():
    """The FINAL fake-user rows of unmodified end-to-end DLAttack / CLeaR runs of the reference (same protocol as the product's end-to-end test:
    seedSet(2018), LightGCN d = 16 L = 2 trained one epoch, posionDataAttack(deepcopy(rec)); every random draw from the reference's own
    generators, nothing patched) -- also on an NGCF d = 128 victim for DLAttack (BASELINE config 5's pairing)."""
    import io, contextlib
    from copy import deepcopy
    from attack.White.DLAttack import DLAttack
    from attack.White.CLeaR import CLeaR
    from recommender.NGCF import NGCF
    os.makedirs('data/clean/ml-100k', exist_ok=True)
    undo_shim = _scipy_torch_index_shim()
    out = {}
    try:
        for tag, cls, F, rec_cls, rkw in (('dl', DLAttack, 2, LightGCN, dict(emb_size=16, n_layers=2)), ('cl', CLeaR, 3, LightGCN, dict(emb_size=16, n_layers=2)),
                                          ('dl_ngcf128', DLAttack, 2, NGCF, dict(emb_size=128, n_layers=3, model_name='NGCF'))):
            rargs = rec_args(maxEpoch=1, **rkw)
            seedSet(2018)
            data = DataLoader(rargs)
            rec = rec_cls(rargs, data)
            with contextlib.redirect_stdout(io.StringIO()):
                rec.train(Epoch=1, evalNum=5)
            atk = cls(_attack_args(attackModelName=cls.__name__, maliciousUserSize=F), data)
            with contextlib.redirect_stdout(io.StringIO()):
                res = sp.csr_matrix(atk.posionDataAttack(deepcopy(rec)))
            U = atk.userNum
            out[tag + '_targets'] = np.array(atk.targetItem, np.int32)
            out[tag + '_fake_rows'] = np.asarray(res[U:, :].todense(), np.float32)
            print(tag, 'fake row sums', out[tag + '_fake_rows'].sum(1))
    finally:
        undo_shim()
    save('g20_fake_rows.npz', **out)


def gen_bilevel():
    import io, contextlib
    from copy import deepcopy
    import recommender.LightGCN as RL
    from attack.White.BiLevelAttackBatch import BiLevelAttackBatch
    os.makedirs('data/clean/ml-100k', exist_ok=True)
    undo_shim = _scipy_torch_index_shim()
    rargs = rec_args(emb_size=16, n_layers=2, maxEpoch=1)
    seedSet(2018)
    data = DataLoader(rargs)
    rec = LightGCN(rargs, data)
    with contextlib.redirect_stdout(io.StringIO()):
        rec.train(Epoch=1, evalNum=5)
    atk = BiLevelAttackBatch(_attack_args(attackModelName='BiLevelAttackBatch', maliciousUserSize=3, Epoch=2, outerEpoch=2, innerEpoch=1), data)
    trace, cap, relax = {}, {}, []
    orig_init = RL.LGCN_Encoder._init_uiAdj
    orig_backward, orig_adam_step = torch.Tensor.backward, torch.optim.Adam.step
    orig_relax = BiLevelAttackBatch.relaxProject

    def init_wrap(self, ui_adj):
        trace.setdefault('init', []).append((sp.csr_matrix(ui_adj).copy(), self.embedding_dict['user_emb'].detach().numpy().copy(),
                                             self.embedding_dict['item_emb'].detach().numpy().copy()))
        return orig_init(self, ui_adj)

    def backward_wrap(self, *a, **k):
        if 'loss' not in cap and self.dim() == 0 and len(trace.get('init', [])) > 0:
            cap['loss'] = float(self.item())
        return orig_backward(self, *a, **k)

    def adam_step_wrap(self, *a, **k):
        if 'grads' not in cap and 'loss' in cap:
            ps = self.param_groups[0]['params']
            cap['grads'] = [p.grad.detach().numpy().copy() for p in ps]
            cap['n_init'] = len(trace['init'])
        return orig_adam_step(self, *a, **k)

    def relax_wrap(self, mat, n):
        st = random.getstate()
        inp = np.asarray(mat[:, :].todense(), np.float32).copy()
        m, ind = orig_relax(self, mat, n)
        relax.append((inp, int(n), np.array(st[1], np.int64), np.asarray(m[:, :].todense(), np.float32).copy(), ind.numpy().copy(),
                      np.array(random.getstate()[1], np.int64)))
        return m, ind
    RL.LGCN_Encoder._init_uiAdj = init_wrap
    torch.Tensor.backward, torch.optim.Adam.step = backward_wrap, adam_step_wrap
    BiLevelAttackBatch.relaxProject = relax_wrap
    out = {}
    try:
        random.seed(777)
        with contextlib.redirect_stdout(io.StringIO()):
            res = atk.posionDataAttack(deepcopy(rec))
    finally:
        RL.LGCN_Encoder._init_uiAdj = orig_init
        torch.Tensor.backward, torch.optim.Adam.step = orig_backward, orig_adam_step
        BiLevelAttackBatch.relaxProject = orig_relax
        undo_shim()
    U, I, F = atk.userNum, atk.itemNum, atk.fakeUserNum
    adj, utab, itab = trace['init'][cap['n_init'] - 1]
    blk = adj[:U + F, U + F:]
    out['bl_sizes'] = np.array([U, I, F, rargs.n_layers, rargs.emb_size, atk.maliciousFeedbackNum, atk.Epoch], np.int64)
    out['bl_targets'] = np.array(atk.targetItem, np.int32)
    out['bl_user_tab'], out['bl_item_tab'] = utab, itab
    out['bl_ui_indptr'], out['bl_ui_indices'], out['bl_ui_data'] = blk.indptr.astype(np.int64), blk.indices.astype(np.int32), blk.data.astype(np.float32)
    out['bl_loss'] = np.array([cap['loss']], np.float32)
    for gr in cap['grads']:                                            # parameter order of the deep-copied ParameterDict: tell by shape
        out['bl_grad_user' if gr.shape[0] == U + F else 'bl_grad_item'] = gr
    for k, (inp, n, st0, m, ind, st1) in enumerate(relax):
        out['bl_relax%d_in' % k], out['bl_relax%d_n' % k], out['bl_relax%d_state' % k] = inp, np.array([n], np.int64), st0
        out['bl_relax%d_out' % k], out['bl_relax%d_ind' % k], out['bl_relax%d_state_after' % k] = m, ind.astype(np.float32), st1
    out['bl_result_fake_rows'] = np.asarray(res[U:U + F, :].todense(), np.float32)
    save('g13_bilevel.npz', **out)


# --------------------------------------------------------------------------- InfoAttack: a*CW + b*Info step, relaxProject
def gen_infoattack():
    import io, contextlib
    from copy import deepcopy
    import recommender.LightGCN as RL
    from attack.White.InfoAttack import InfoAttack
    os.makedirs('data/clean/ml-100k', exist_ok=True)
    undo_shim = _scipy_torch_index_shim()
    rargs = rec_args(emb_size=16, n_layers=2, maxEpoch=1)
    seedSet(2018)
    data = DataLoader(rargs)
    rec = LightGCN(rargs, data)
    with contextlib.redirect_stdout(io.StringIO()):
        rec.train(Epoch=1, evalNum=5)
    atk = InfoAttack(_attack_args(attackModelName='InfoAttack', maliciousUserSize=3, Epoch=1, outerEpoch=2, innerEpoch=1), data)
    trace, cap, relax = {}, {}, []
    orig_init, orig_fwd = RL.LGCN_Encoder._init_uiAdj, RL.LGCN_Encoder.forward
    orig_backward, orig_adam_step = torch.Tensor.backward, torch.optim.Adam.step
    orig_relax = InfoAttack.relaxProject

    def init_wrap(self, ui_adj):
        trace.setdefault('init', []).append((sp.csr_matrix(ui_adj).copy(), self.embedding_dict['user_emb'].detach().numpy().copy(),
                                             self.embedding_dict['item_emb'].detach().numpy().copy()))
        return orig_init(self, ui_adj)

    def fwd_wrap(self, *a, **k):
        out = orig_fwd(self, *a, **k)
        if 'view1' not in cap:
            cap['view1'] = out[1].detach().numpy().copy()          # first forward of posionDataAttack: the fixed view
        return out

    def backward_wrap(self, *a, **k):
        if 'loss' not in cap and self.dim() == 0 and len(trace.get('init', [])) > 0:
            cap['loss'] = float(self.item()); cap['a'] = float(atk.a); cap['b'] = float(atk.b)
        return orig_backward(self, *a, **k)

    def adam_step_wrap(self, *a, **k):
        if 'grads' not in cap and 'loss' in cap:
            cap['grads'] = [p.grad.detach().numpy().copy() for p in self.param_groups[0]['params']]
            cap['n_init'] = len(trace['init'])
        return orig_adam_step(self, *a, **k)

    def relax_wrap(self, mat, n):
        st = random.getstate()
        inp = np.asarray(mat[:, :].todense(), np.float32).copy()
        m, ind = orig_relax(self, mat, n)
        relax.append((inp, int(n), np.array(st[1], np.int64), np.asarray(m[:, :].todense(), np.float32).copy(), ind.numpy().copy(),
                      np.array(random.getstate()[1], np.int64)))
        return m, ind
    RL.LGCN_Encoder._init_uiAdj, RL.LGCN_Encoder.forward = init_wrap, fwd_wrap
    torch.Tensor.backward, torch.optim.Adam.step = backward_wrap, adam_step_wrap
    InfoAttack.relaxProject = relax_wrap
    out = {}
    try:
        random.seed(31337)
        with contextlib.redirect_stdout(io.StringIO()):
            res = atk.posionDataAttack(deepcopy(rec))
    finally:
        RL.LGCN_Encoder._init_uiAdj, RL.LGCN_Encoder.forward = orig_init, orig_fwd
        torch.Tensor.backward, torch.optim.Adam.step = orig_backward, orig_adam_step
        InfoAttack.relaxProject = orig_relax
        undo_shim()
    U, I, F = atk.userNum, atk.itemNum, atk.fakeUserNum
    adj, utab, itab = trace['init'][cap['n_init'] - 1]
    blk = adj[:U + F, U + F:]
    out['ia_sizes'] = np.array([U, I, F, rargs.n_layers, rargs.emb_size, atk.maliciousFeedbackNum, min(rec.topN)], np.int64)
    out['ia_targets'] = np.array(atk.targetItem, np.int32)
    out['ia_user_tab'], out['ia_item_tab'], out['ia_view1'] = utab, itab, cap['view1']
    out['ia_ui_indptr'], out['ia_ui_indices'], out['ia_ui_data'] = blk.indptr.astype(np.int64), blk.indices.astype(np.int32), blk.data.astype(np.float32)
    out['ia_loss'] = np.array([cap['loss'], cap['a'], cap['b']], np.float32)
    for gr in cap['grads']:
        out['ia_grad_user' if gr.shape[0] == U + F else 'ia_grad_item'] = gr
    inp, n, st0, m, ind, st1 = relax[0]
    out['ia_relax_in'], out['ia_relax_n'], out['ia_relax_state'] = inp, np.array([n], np.int64), st0
    out['ia_relax_out'], out['ia_relax_ind'], out['ia_relax_state_after'] = m, ind.astype(np.float32), st1
    out['ia_result_fake_rowsums'] = np.asarray(res[U:U + F, :].sum(1)).ravel().astype(np.float32)
    save('g14_infoattack.npz', **out)


# --------------------------------------------------------------------------- PipAttack: RNG use of the constructor, first step
def gen_pipattack():
    import io, contextlib
    from copy import deepcopy
    import recommender.LightGCN as RL
    from attack.White.PipAttack import PipAttack
    os.makedirs('data/clean/ml-100k', exist_ok=True)
    undo_shim = _scipy_torch_index_shim()
    rargs = rec_args(emb_size=16, n_layers=2, maxEpoch=1)
    seedSet(2018)
    data = DataLoader(rargs)
    rec = LightGCN(rargs, data)
    with contextlib.redirect_stdout(io.StringIO()):
        rec.train(Epoch=1, evalNum=5)
    torch.manual_seed(4321)
    with contextlib.redirect_stdout(io.StringIO()):
        atk = PipAttack(_attack_args(attackModelName='PipAttack', maliciousUserSize=3, Epoch=1, outerEpoch=2, innerEpoch=1), data)
    out = {'pip_rng_probe': torch.rand(4).numpy()}                     # the global torch stream right after the constructor
    out['pip_mlp_w0'] = atk.popularity_model.layers[0].weight.detach().numpy()[:8].copy()
    out['pip_mlp_b2'] = atk.popularity_model.layers[4].bias.detach().numpy().copy()
    trace, cap = {}, {}
    orig_init = RL.LGCN_Encoder._init_uiAdj
    orig_backward, orig_adam_step = torch.Tensor.backward, torch.optim.Adam.step

    def init_wrap(self, ui_adj):
        trace.setdefault('init', []).append((sp.csr_matrix(ui_adj).copy(), self.embedding_dict['user_emb'].detach().numpy().copy(),
                                             self.embedding_dict['item_emb'].detach().numpy().copy()))
        return orig_init(self, ui_adj)

    def backward_wrap(self, *a, **k):
        if 'loss' not in cap and self.dim() == 0 and len(trace.get('init', [])) > 0:
            cap['loss'] = float(self.item())
        return orig_backward(self, *a, **k)

    def adam_step_wrap(self, *a, **k):
        if 'grads' not in cap and 'loss' in cap:
            cap['grads'] = [p.grad.detach().numpy().copy() for p in self.param_groups[0]['params']]
            cap['n_init'] = len(trace['init'])
        return orig_adam_step(self, *a, **k)
    RL.LGCN_Encoder._init_uiAdj = init_wrap
    torch.Tensor.backward, torch.optim.Adam.step = backward_wrap, adam_step_wrap
    try:
        random.seed(99)
        with contextlib.redirect_stdout(io.StringIO()):
            res = atk.posionDataAttack(deepcopy(rec))
    finally:
        RL.LGCN_Encoder._init_uiAdj = orig_init
        torch.Tensor.backward, torch.optim.Adam.step = orig_backward, orig_adam_step
        undo_shim()
    U, I, F = atk.userNum, atk.itemNum, atk.fakeUserNum
    adj, utab, itab = trace['init'][cap['n_init'] - 1]
    blk = adj[:U + F, U + F:]
    out['pip_sizes'] = np.array([U, I, F, rargs.n_layers, rargs.emb_size, atk.maliciousFeedbackNum], np.int64)
    out['pip_targets'] = np.array(atk.targetItem, np.int32)
    out['pip_user_tab'], out['pip_item_tab'] = utab, itab
    out['pip_ui_indptr'], out['pip_ui_indices'], out['pip_ui_data'] = blk.indptr.astype(np.int64), blk.indices.astype(np.int32), blk.data.astype(np.float32)
    out['pip_loss'] = np.array([cap['loss']], np.float32)
    for gr in cap['grads']:
        out['pip_grad_user' if gr.shape[0] == U + F else 'pip_grad_item'] = gr
    out['pip_result_fake_rowsums'] = np.asarray(res[U:U + F, :].sum(1)).ravel().astype(np.float32)
    save('g15_pipattack.npz', **out)


# --------------------------------------------------------------------------- gray-box siblings: A_ra loss step, result structure
def gen_gray():
    import io, contextlib
    from copy import deepcopy
    import recommender.LightGCN as RL
    from attack.Gray.A_ra import A_ra
    from attack.Gray.FedRecAttack import FedRecAttack
    os.makedirs('data/clean/ml-100k', exist_ok=True)
    undo_shim = _scipy_torch_index_shim()
    rargs = rec_args(emb_size=16, n_layers=2, maxEpoch=1)
    out = {}
    for cls, tag in ((A_ra, 'ara'), (FedRecAttack, 'fed')):
        seedSet(2018)
        data = DataLoader(rargs)
        rec = LightGCN(rargs, data)
        with contextlib.redirect_stdout(io.StringIO()):
            rec.train(Epoch=1, evalNum=5)
        atk = cls(_attack_args(attackCategory='Gray', attackModelName=cls.__name__, maliciousUserSize=3, Epoch=1, outerEpoch=1, innerEpoch=1), data)
        cap = {}
        orig_fwd, orig_randn = RL.LGCN_Encoder.forward, torch.randn
        orig_backward, orig_adam_step = torch.Tensor.backward, torch.optim.Adam.step
        a_fixed = orig_randn(100, rargs.emb_size, generator=torch.Generator().manual_seed(7))

        def randn_wrap(*a, **k):
            if len(a) == 1 and isinstance(a[0], tuple) and a[0] == (100, rargs.emb_size):
                cap['armed'] = True                      # the attack loss of this outer step follows
                return a_fixed.clone()
            return orig_randn(*a, **k)

        def fwd_wrap(self, *a, **k):
            o = orig_fwd(self, *a, **k)
            cap['last_tabs'] = (self.embedding_dict['user_emb'].detach().numpy().copy(), self.embedding_dict['item_emb'].detach().numpy().copy(),
                                o[1].detach().numpy().copy())
            return o

        def backward_wrap(self, *a, **k):
            if cap.get('armed') and 'loss' not in cap:
                cap['loss'] = float(self.item()); cap['tabs'] = cap['last_tabs']
            return orig_backward(self, *a, **k)

        def adam_step_wrap(self, *a, **k):
            if 'loss' in cap and 'grads' not in cap:
                cap['grads'] = [p.grad.detach().numpy().copy() for p in self.param_groups[0]['params']]
            return orig_adam_step(self, *a, **k)
        if tag == 'ara':
            RL.LGCN_Encoder.forward, torch.randn = fwd_wrap, randn_wrap
            torch.Tensor.backward, torch.optim.Adam.step = backward_wrap, adam_step_wrap
        try:
            random.seed(11)
            with contextlib.redirect_stdout(io.StringIO()):
                res = atk.posionDataAttack(deepcopy(rec))
        finally:
            RL.LGCN_Encoder.forward, torch.randn = orig_fwd, orig_randn
            torch.Tensor.backward, torch.optim.Adam.step = orig_backward, orig_adam_step
        U, F = atk.userNum, atk.fakeUserNum
        out[tag + '_result_fake_rowsums'] = np.asarray(res[U:U + F, :].sum(1)).ravel().astype(np.float32)
        out[tag + '_targets'] = np.array(atk.targetItem, np.int32)
        if tag == 'ara':
            out['ara_a'] = a_fixed.numpy()
            out['ara_item_prop'] = cap['tabs'][2]                     # propagated item table the loss was computed on
            out['ara_loss'] = np.array([cap['loss']], np.float32)
            out['ara_sizes'] = np.array([U, atk.itemNum, F, atk.n], np.int64)
    undo_shim()
    save('g17_gray.npz', **out)


# --------------------------------------------------------------------------- GTA: proxyLG's first training step, result structure
def gen_gta():
    import io, contextlib
    from copy import deepcopy
    import attack.Black.GTA as G
    os.makedirs('data/clean/ml-100k', exist_ok=True)
    undo_shim = _scipy_torch_index_shim()
    rargs = rec_args(emb_size=16, n_layers=2, maxEpoch=1)
    out = {}
    # (1) one proxyLG training step from fresh tables: loss and parameter gradients of the first batch
    seedSet(2018)
    data = DataLoader(rargs)
    with contextlib.redirect_stdout(io.StringIO()):
        proxy = G.proxyLG(rargs, data, [3, 77, 500, 1000, 1411])
    out['gta_user0'] = proxy.model.embedding_dict['user_emb'].detach().numpy().copy()
    out['gta_item0'] = proxy.model.embedding_dict['item_emb'].detach().numpy().copy()
    cap = {}
    orig_backward, orig_adam_step, orig_nbp = torch.Tensor.backward, torch.optim.Adam.step, G.next_batch_pairwise

    def nbp_wrap(d, bs):
        for b in orig_nbp(d, bs):
            if 'batch' not in cap:
                cap['batch'] = [np.asarray(x, np.int32) for x in b]
            yield b

    def backward_wrap(self, *a, **k):
        if 'loss' not in cap:
            cap['loss'] = float(self.item())
        return orig_backward(self, *a, **k)

    def adam_step_wrap(self, *a, **k):
        if 'grads' not in cap:
            cap['grads'] = [p.grad.detach().numpy().copy() for p in self.param_groups[0]['params']]
        return orig_adam_step(self, *a, **k)
    G.next_batch_pairwise = nbp_wrap
    torch.Tensor.backward, torch.optim.Adam.step = backward_wrap, adam_step_wrap
    try:
        random.seed(2018)
        with contextlib.redirect_stdout(io.StringIO()):
            proxy.train(Epoch=1, evalNum=5)
    finally:
        G.next_batch_pairwise = orig_nbp
        torch.Tensor.backward, torch.optim.Adam.step = orig_backward, orig_adam_step
    out['gta_targets0'] = np.array([3, 77, 500, 1000, 1411], np.int32)
    out['gta_batch_u'], out['gta_batch_p'], out['gta_batch_n'] = cap['batch']
    out['gta_loss'] = np.array([cap['loss']], np.float32)
    for gr in cap['grads']:
        out['gta_grad_user' if gr.shape[0] == data.user_num else 'gta_grad_item'] = gr
    # (2) the whole attack: structure of the result
    seedSet(2018)
    data = DataLoader(rargs)
    rec = LightGCN(rargs, data)
    with contextlib.redirect_stdout(io.StringIO()):
        rec.train(Epoch=1, evalNum=5)
    atk = G.GTA(_attack_args(attackCategory='Black', attackModelName='GTA', maliciousUserSize=3, Epoch=2, outerEpoch=1, innerEpoch=1), data)
    random.seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        res = atk.posionDataAttack(rec)
    U, F = atk.userNum, atk.fakeUserNum
    out['gta_result_fake_rowsums'] = np.asarray(res[U:U + F, :].sum(1)).ravel().astype(np.float32)
    out['gta_targets'] = np.array(atk.targetItem, np.int32)
    undo_shim()
    save('g18_gta.npz', **out)


# --------------------------------------------------------------------------- NGCF (a9): forward + 3 Adam steps
def gen_ngcf():
    from recommender.NGCF import NGCF
    args = rec_args(emb_size=32, n_layers=2, model_name='NGCF')
    seedSet(2018)
    data = DataLoader(args)
    rec = NGCF(args, data)
    model = rec.model
    o = {'user0': model.embedding_dict['user_emb'].detach().numpy().copy(), 'item0': model.embedding_dict['item_emb'].detach().numpy().copy()}
    for k in range(2):
        o['w1_%d' % k] = model.W['w1_%d' % k].detach().numpy().copy(); o['w2_%d' % k] = model.W['w2_%d' % k].detach().numpy().copy()
    with torch.no_grad():
        u, i = model()
    o['fwd_user'], o['fwd_item'] = u.numpy().copy(), i.numpy().copy()
    optim = torch.optim.Adam(model.parameters(), lr=args.lRate)
    random.seed(2018)
    losses, bu, bp, bn = [], [], [], []
    for step, batch in enumerate(ref_sampler.next_batch_pairwise(data, args.batch_size)):
        if step == 3:
            break
        user_idx, pos_idx, neg_idx = batch
        ue, ie = model()
        loss = ref_loss.bpr_loss(ue[user_idx], ie[pos_idx], ie[neg_idx]) + ref_loss.l2_reg_loss(args.reg, ue[user_idx], ie[pos_idx])
        optim.zero_grad(); loss.backward()
        if step == 0:
            o['grad_user'] = model.embedding_dict['user_emb'].grad.numpy().copy()
            o['grad_w1_0'] = model.W['w1_0'].grad.numpy().copy(); o['grad_w2_1'] = model.W['w2_1'].grad.numpy().copy()
        optim.step()
        losses.append(loss.item()); bu.append(np.asarray(user_idx, np.int32)); bp.append(np.asarray(pos_idx, np.int32)); bn.append(np.asarray(neg_idx, np.int32))
    o['losses'] = np.array(losses, np.float32)
    o['batch_u'], o['batch_p'], o['batch_n'] = np.stack(bu), np.stack(bp), np.stack(bn)
    o['user_k3'] = model.embedding_dict['user_emb'].detach().numpy().copy(); o['item_k3'] = model.embedding_dict['item_emb'].detach().numpy().copy()
    o['w1_0_k3'] = model.W['w1_0'].detach().numpy().copy()
    save('g9_ngcf.npz', **o)



# --------------------------------------------------------------------------- NGCF at cfg5's width (d = 128, L = 3): forward + 3 Adam steps
def gen_ngcf128():
    from recommender.NGCF import NGCF
    args = rec_args(emb_size=128, n_layers=3, model_name='NGCF')
    seedSet(2018)
    data = DataLoader(args)
    rec = NGCF(args, data)
    model = rec.model
    o = {'user0': model.embedding_dict['user_emb'].detach().numpy().copy(), 'item0': model.embedding_dict['item_emb'].detach().numpy().copy()}
    for k in range(3):
        o['w1_%d' % k] = model.W['w1_%d' % k].detach().numpy().copy(); o['w2_%d' % k] = model.W['w2_%d' % k].detach().numpy().copy()
    with torch.no_grad():
        u, i = model()
    o['fwd_user'], o['fwd_item'] = u.numpy().copy(), i.numpy().copy()
    optim = torch.optim.Adam(model.parameters(), lr=args.lRate)
    random.seed(2018)
    losses, bu, bp, bn = [], [], [], []
    for step, batch in enumerate(ref_sampler.next_batch_pairwise(data, args.batch_size)):
        if step == 3:
            break
        user_idx, pos_idx, neg_idx = batch
        ue, ie = model()
        loss = ref_loss.bpr_loss(ue[user_idx], ie[pos_idx], ie[neg_idx]) + ref_loss.l2_reg_loss(args.reg, ue[user_idx], ie[pos_idx])
        optim.zero_grad(); loss.backward()
        if step == 0:
            o['grad_user'] = model.embedding_dict['user_emb'].grad.numpy().copy()
            o['grad_w1_0'] = model.W['w1_0'].grad.numpy().copy(); o['grad_w2_2'] = model.W['w2_2'].grad.numpy().copy()
        optim.step()
        losses.append(loss.item()); bu.append(np.asarray(user_idx, np.int32)); bp.append(np.asarray(pos_idx, np.int32)); bn.append(np.asarray(neg_idx, np.int32))
    o['losses'] = np.array(losses, np.float32)
    o['batch_u'], o['batch_p'], o['batch_n'] = np.stack(bu), np.stack(bp), np.stack(bn)
    o['user_k3'] = model.embedding_dict['user_emb'].detach().numpy().copy(); o['item_k3'] = model.embedding_dict['item_emb'].detach().numpy().copy()
    o['w1_0_k3'] = model.W['w1_0'].detach().numpy().copy()
    save('g9_ngcf128.npz', **o)


# --------------------------------------------------------------------------- G19: DLAttack / CLeaR on NGCF and SimGCL victims; NoneAttack protocol
def gen_victims():
    """BASELINE configs 4 and 5 name SimGCL + CLeaR and NGCF + DL_Attack.  The unmodified reference attacks are run end to end on both
    victims (structure of the returned matrix), and CLeaR's first surrogate step on each victim's encoder is captured (tables, dense
    weights, poisoned interactions, injected r0 -> loss and parameter gradients).  NoneAttack: the identity protocol of config 1."""
    import io, contextlib
    from copy import deepcopy
    import recommender.NGCF as RN
    import recommender.SimGCL as RS
    from attack.White.DLAttack import DLAttack
    from attack.White.CLeaR import CLeaR
    from attack.Black.NoneAttack import NoneAttack
    os.makedirs('data/clean/ml-100k', exist_ok=True)
    undo_shim = _scipy_torch_index_shim()
    out = {}
    orig_randn, orig_backward, orig_adam_step = torch.randn, torch.Tensor.backward, torch.optim.Adam.step
    try:
        for tag, mod, cls, enc, kw in (('ngcf', RN, RN.NGCF, RN.NGCF_Encoder, dict(emb_size=16, n_layers=2, model_name='NGCF')),
                                       ('simgcl', RS, RS.SimGCL, RS.SimGCL_Encoder, dict(emb_size=16, n_layers=2, model_name='SimGCL'))):
            rargs = rec_args(maxEpoch=1, **kw)

            def fresh():
                seedSet(2018)
                data = DataLoader(rargs)
                rec = cls(rargs, data)
                with contextlib.redirect_stdout(io.StringIO()):
                    rec.train(Epoch=1, evalNum=5)
                return data, rec
            trace = []
            orig_init = enc._init_uiAdj

            def init_wrap(self, ui_adj, _orig=orig_init):
                snap = {'adj': sp.csr_matrix(ui_adj).copy(), 'user': self.embedding_dict['user_emb'].detach().numpy().copy(),
                        'item': self.embedding_dict['item_emb'].detach().numpy().copy()}
                if hasattr(self, 'W'):
                    snap['W'] = {k: v.detach().numpy().copy() for k, v in self.W.items()}
                trace.append(snap)
                return _orig(self, ui_adj)
            enc._init_uiAdj = init_wrap
            # ---- DLAttack end to end
            data, rec = fresh()
            atk = DLAttack(_attack_args(attackModelName='DLAttack', maliciousUserSize=2), data)
            out[tag + '_dl_targets'] = np.array(atk.targetItem, np.int32)
            with contextlib.redirect_stdout(io.StringIO()):
                res = atk.posionDataAttack(deepcopy(rec))
            res = sp.csr_matrix(res)
            out[tag + '_dl_result_fake_rowsums'] = np.asarray(res[atk.userNum:, :].sum(1)).ravel().astype(np.float32)
            out[tag + '_dl_shape'] = np.array(res.shape, np.int64)
            # ---- CLeaR end to end, first surrogate step captured
            data, rec = fresh()
            atk = CLeaR(_attack_args(attackModelName='CLeaR', maliciousUserSize=3), data)
            r0 = torch.randn(rargs.emb_size, generator=torch.Generator().manual_seed(99))
            cap = {}

            def randn_wrap(*a, **k):
                if len(a) == 1 and a[0] == rargs.emb_size:
                    return r0.clone()
                return orig_randn(*a, **k)

            def backward_wrap(self, *a, **k):
                if 'loss' not in cap:
                    cap['loss'] = float(self.item())
                return orig_backward(self, *a, **k)

            def adam_step_wrap(self, *a, **k):
                if 'grads' not in cap and 'loss' in cap:
                    cap['grads'] = {id(p): p.grad.detach().numpy().copy() for p in self.param_groups[0]['params'] if p.grad is not None}
                    cap['params'] = list(self.param_groups[0]['params'])
                    cap['values'] = {id(p): p.detach().numpy().copy() for p in cap['params']}        # before the step: still the snapshot's values
                    cap['n_init'] = len(trace)
                return orig_adam_step(self, *a, **k)
            del trace[:]
            torch.randn, torch.Tensor.backward, torch.optim.Adam.step = randn_wrap, backward_wrap, adam_step_wrap
            random.seed(4242)
            holder = {}
            orig_deepcopy_target = CLeaR.posionDataAttack
            with contextlib.redirect_stdout(io.StringIO()):
                res = atk.posionDataAttack(deepcopy(rec))
            torch.randn, torch.Tensor.backward, torch.optim.Adam.step = orig_randn, orig_backward, orig_adam_step
            snap = trace[cap['n_init'] - 1]
            Un = snap['user'].shape[0]
            blk = snap['adj'][:Un, Un:]
            out[tag + '_cl_user_tab'], out[tag + '_cl_item_tab'] = snap['user'], snap['item']
            for k, v in snap.get('W', {}).items():
                out[tag + '_cl_' + k] = v
            out[tag + '_cl_ui_indptr'], out[tag + '_cl_ui_indices'], out[tag + '_cl_ui_data'] = blk.indptr.astype(np.int64), blk.indices.astype(np.int32), blk.data.astype(np.float32)
            out[tag + '_cl_r0'] = r0.numpy()
            out[tag + '_cl_loss'] = np.array([cap['loss']], np.float32)
            # gradients keyed by parameter shape (user table / item table / the d x d weights in creation order)
            gl = [cap['grads'][id(p)] for p in cap['params'] if id(p) in cap['grads']]
            for g_ in gl:
                if g_.shape[0] == Un:
                    out[tag + '_cl_grad_user'] = g_
                elif g_.shape[0] == atk.itemNum:
                    out[tag + '_cl_grad_item'] = g_
            if tag == 'ngcf':
                # name each d x d gradient by matching its parameter's pre-step value with the snapshot of W taken at _init_uiAdj
                for p_ in cap['params']:
                    if tuple(p_.shape) == (rargs.emb_size, rargs.emb_size) and id(p_) in cap['grads']:
                        hit = [k for k, v in snap['W'].items() if np.array_equal(v, cap['values'][id(p_)])]
                        assert len(hit) == 1, hit
                        out[tag + '_cl_grad_' + hit[0]] = cap['grads'][id(p_)]
            out[tag + '_cl_targets'] = np.array(atk.targetItem, np.int32)
            out[tag + '_cl_sizes'] = np.array([atk.userNum, atk.itemNum, atk.fakeUserNum, min(rec.topN)], np.int64)
            out[tag + '_cl_result_fake_rowsums'] = np.asarray(sp.csr_matrix(res)[atk.userNum:, :].sum(1)).ravel().astype(np.float32)
            enc._init_uiAdj = orig_init
        # ---- NoneAttack (config 1's attack leg): constructor contract + identity
        seedSet(2018)
        rargs = rec_args(emb_size=16, model_name='GMF', maxEpoch=1)
        data = DataLoader(rargs)
        tf = './data/clean/' + data.dataName + '/targetItem_unpopular_5.txt'        # written by the attacks above: NoneAttack must draw its own
        if os.path.exists(tf):
            os.remove(tf)
        atk = NoneAttack(_attack_args(attackCategory='Black', attackModelName='NoneAttack'), data)
        res = sp.csr_matrix(atk.posionDataAttack())
        out['none_targets'] = np.array(atk.targetItem, np.int32)
        out['none_sizes'] = np.array([atk.userNum, atk.itemNum, atk.fakeUserNum, atk.maliciousFeedbackNum, res.nnz], np.int64)
        out['none_flags'] = np.array([atk.recommenderGradientRequired, atk.recommenderModelRequired], np.int8)
        out['none_identity'] = np.array([(res != sp.csr_matrix(data.matrix())).nnz], np.int64)
        out['none_next_random'] = np.array([random.random()])
    finally:
        torch.randn, torch.Tensor.backward, torch.optim.Adam.step = orig_randn, orig_backward, orig_adam_step
        undo_shim()
    save('g19_victims.npz', **out)


# --------------------------------------------------------------------------- G11: XSimGCL (LightGCN + one extra term)
def gen_xsimgcl(data):
    """One reference XSimGCL iteration (recommender/XSimGCL.py:62-75,205-223) with injected noise: ONE perturbed forward whose
    mean is used for BPR and, against its own layer-1 output, for the InfoNCE terms (temperature 0.1)."""
    from recommender.XSimGCL import XSimGCL
    args = rec_args(emb_size=16, n_layers=2, model_name='XSimGCL')
    seedSet(2018)
    rec = XSimGCL(args, data)
    model = rec.model
    N = data.user_num + data.item_num
    g = torch.Generator().manual_seed(78)
    noises = [torch.rand(N, 16, generator=g) for _ in range(2)]       # one per hop
    o = {'user0': model.embedding_dict['user_emb'].detach().numpy().copy(),
         'item0': model.embedding_dict['item_emb'].detach().numpy().copy(),
         'noise': torch.stack(noises).numpy(),
         'hyper': np.array([rec.n_layers, rec.layer_cl, rec.cl_rate, rec.eps, rec.temp], np.float64)}
    random.seed(2018)
    d2 = copy.copy(data); d2.training_data = [list(r) for r in data_training0]
    user_idx, pos_idx, neg_idx = next(iter(ref_sampler.next_batch_pairwise(d2, 2048)))
    with torch.no_grad():
        u0, i0 = model()
    o['fwd_user'] = u0.numpy().copy(); o['fwd_item'] = i0.numpy().copy()
    orig = torch.rand_like
    it = iter(noises)
    torch.rand_like = lambda x, *a, **k: next(it)
    try:
        optim = torch.optim.Adam(model.parameters(), lr=args.lRate)
        ru, ri, cu, ci = model(True)
        o['fwdp_user'] = ru.detach().numpy().copy(); o['fwdp_item'] = ri.detach().numpy().copy()
        o['cl_user'] = cu.detach().numpy().copy(); o['cl_item'] = ci.detach().numpy().copy()
        user_emb, pos_item_emb, neg_item_emb = ru[user_idx], ri[pos_idx], ri[neg_idx]
        rec_loss = ref_loss.bpr_loss(user_emb, pos_item_emb, neg_item_emb)
        cl_loss = rec.cl_rate * rec.cal_cl_loss([user_idx, pos_idx], ru, cu, ri, ci)
        batch_loss = rec_loss + ref_loss.l2_reg_loss(args.reg, user_emb, pos_item_emb) + cl_loss
        optim.zero_grad()
        batch_loss.backward()
        o['grad_user'] = model.embedding_dict['user_emb'].grad.numpy().copy()
        o['grad_item'] = model.embedding_dict['item_emb'].grad.numpy().copy()
        optim.step()
    finally:
        torch.rand_like = orig
    o['rec_loss'] = np.array([rec_loss.item()], np.float32)
    o['cl_loss'] = np.array([cl_loss.item()], np.float32)
    o['user_k1'] = model.embedding_dict['user_emb'].detach().numpy().copy()
    o['item_k1'] = model.embedding_dict['item_emb'].detach().numpy().copy()
    o['batch_u'] = np.asarray(user_idx, np.int32); o['batch_p'] = np.asarray(pos_idx, np.int32); o['batch_n'] = np.asarray(neg_idx, np.int32)
    save('g11_xsimgcl.npz', **o)


# --------------------------------------------------------------------------- G16: NCL (structure + prototype contrast)
def gen_ncl(data):
    """One reference NCL iteration of the prototype phase (recommender/NCL.py:131-166): BPR + L2/batch_size + ssl_layer_loss (2-hop
    context against ALL initial rows) + ProtoNCE_loss (k-means centroids of e_step, k lowered from 2000 to 30: ml-100k has 942 users)."""
    import io, contextlib
    from recommender.NCL import NCL, TorchGraphInterface
    args = rec_args(emb_size=16, n_layers=2, model_name='NCL')
    seedSet(2018)
    with contextlib.redirect_stdout(io.StringIO()):
        rec = NCL(args, data)
    rec.k = 30
    model = rec.model
    o = {'user0': model.embedding_dict['user_emb'].detach().numpy().copy(), 'item0': model.embedding_dict['item_emb'].detach().numpy().copy(),
         'hyper': np.array([rec.n_layers, rec.hyper_layers, rec.ssl_temp, rec.ssl_reg, rec.alpha, rec.proto_reg, rec.k, rec.batch_size], np.float64)}
    np.random.seed(515)
    rec.e_step()
    o['user_centroids'], o['user_2cluster'] = rec.user_centroids.numpy().copy(), rec.user_2cluster.numpy().astype(np.int32)
    o['item_centroids'], o['item_2cluster'] = rec.item_centroids.numpy().copy(), rec.item_2cluster.numpy().astype(np.int32)
    random.seed(2018)
    d2 = copy.copy(data); d2.training_data = [list(r) for r in data_training0]
    user_idx, pos_idx, neg_idx = next(iter(ref_sampler.next_batch_pairwise(d2, 2048)))
    optim = torch.optim.Adam(model.parameters(), lr=args.lRate)
    ru, ri = model()
    A = TorchGraphInterface.convert_sparse_mat_to_tensor(data.norm_adj)
    ego = torch.cat([model.embedding_dict['user_emb'], model.embedding_dict['item_emb']], 0)
    emb_list = [ego]
    for k in range(rec.n_layers):
        ego = torch.sparse.mm(A, ego)
        emb_list.append(ego)
    user_emb, pos_item_emb, neg_item_emb = ru[user_idx], ri[pos_idx], ri[neg_idx]
    rec_loss = ref_loss.bpr_loss(user_emb, pos_item_emb, neg_item_emb)
    ssl = rec.ssl_layer_loss(emb_list[rec.hyper_layers * 2], emb_list[0], user_idx, pos_idx)
    proto = rec.ProtoNCE_loss(emb_list[0], user_idx, pos_idx)
    ps = [model.embedding_dict['user_emb'], model.embedding_dict['item_emb']]
    gs = torch.autograd.grad(ssl, ps, retain_graph=True)
    gp = torch.autograd.grad(proto, ps, retain_graph=True)
    o['ssl_grad_user'], o['ssl_grad_item'] = gs[0].numpy().copy(), gs[1].numpy().copy()
    o['proto_grad_user'], o['proto_grad_item'] = gp[0].numpy().copy(), gp[1].numpy().copy()
    batch_loss = rec_loss + ref_loss.l2_reg_loss(args.reg, user_emb, pos_item_emb, neg_item_emb) / rec.batch_size + ssl + proto
    optim.zero_grad()
    batch_loss.backward()
    o['grad_user'] = model.embedding_dict['user_emb'].grad.numpy().copy(); o['grad_item'] = model.embedding_dict['item_emb'].grad.numpy().copy()
    optim.step()
    o['losses'] = np.array([rec_loss.item(), ssl.item(), proto.item(), batch_loss.item()], np.float64)
    o['user_k1'] = model.embedding_dict['user_emb'].detach().numpy().copy(); o['item_k1'] = model.embedding_dict['item_emb'].detach().numpy().copy()
    o['batch_u'] = np.asarray(user_idx, np.int32); o['batch_p'] = np.asarray(pos_idx, np.int32); o['batch_n'] = np.asarray(neg_idx, np.int32)
    save('g16_ncl.npz', **o)


# --------------------------------------------------------------------------- G12: SGL (edge-dropout views)
def gen_sgl(data):
    """Reference SGL (recommender/SGL.py): the two edge-dropped graphs of an epoch (Python `random.sample` over the edges) and one
    training iteration on them (clean forward for BPR, InfoNCE between the two views' concatenated user/item rows, temp 0.2)."""
    from recommender.SGL import SGL
    args = rec_args(emb_size=16, n_layers=2, model_name='SGL')
    seedSet(2018)
    rec = SGL(args, data)
    model = rec.model
    U, I = data.user_num, data.item_num
    o = {'user0': model.embedding_dict['user_emb'].detach().numpy().copy(),
         'item0': model.embedding_dict['item_emb'].detach().numpy().copy(),
         'hyper': np.array([rec.n_layers, rec.cl_rate, rec.drop_rate, rec.temp, rec.aug_type], np.float64)}
    random.seed(2018)
    adj1 = model.graph_reconstruction()
    adj2 = model.graph_reconstruction()
    o['next_random'] = np.array([random.random()], np.float64)
    for tag, adj in (('1', adj1), ('2', adj2)):
        a = adj.coalesce()
        r, c, v = a.indices()[0].numpy(), a.indices()[1].numpy(), a.values().numpy()
        sel = r < U
        order = np.lexsort((c[sel], r[sel]))
        o['keep' + tag] = (r[sel][order].astype(np.int64) * I + (c[sel][order] - U)).astype(np.int32)      # kept (user, item) pairs, sorted
        o['val' + tag] = v[sel][order].astype(np.float32)                                                   # their normalised weights
    random.seed(2018)
    d2 = copy.copy(data); d2.training_data = [list(r) for r in data_training0]
    user_idx, pos_idx, neg_idx = next(iter(ref_sampler.next_batch_pairwise(d2, 2048)))
    optim = torch.optim.Adam(model.parameters(), lr=args.lRate)
    rec_user_emb, rec_item_emb = model()
    o['fwd_user'] = rec_user_emb.detach().numpy().copy(); o['fwd_item'] = rec_item_emb.detach().numpy().copy()
    user_emb, pos_item_emb, neg_item_emb = rec_user_emb[user_idx], rec_item_emb[pos_idx], rec_item_emb[neg_idx]
    rec_loss = ref_loss.bpr_loss(user_emb, pos_item_emb, neg_item_emb)
    cl_loss = rec.cl_rate * model.cal_cl_loss([user_idx, pos_idx], adj1, adj2)
    batch_loss = rec_loss + ref_loss.l2_reg_loss(args.reg, user_emb, pos_item_emb) + cl_loss
    optim.zero_grad()
    batch_loss.backward()
    o['grad_user'] = model.embedding_dict['user_emb'].grad.numpy().copy()
    o['grad_item'] = model.embedding_dict['item_emb'].grad.numpy().copy()
    optim.step()
    o['rec_loss'] = np.array([rec_loss.item()], np.float32)
    o['cl_loss'] = np.array([cl_loss.item()], np.float32)
    o['user_k1'] = model.embedding_dict['user_emb'].detach().numpy().copy()
    o['item_k1'] = model.embedding_dict['item_emb'].detach().numpy().copy()
    o['batch_u'] = np.asarray(user_idx, np.int32); o['batch_p'] = np.asarray(pos_idx, np.int32); o['batch_n'] = np.asarray(neg_idx, np.int32)
    with torch.no_grad():
        v1u, v1i = model(adj1)
    o['view1_user'] = v1u.numpy().copy(); o['view1_item'] = v1i.numpy().copy()      # (after the step: tables user_k1/item_k1)
    save('g12_sgl.npz', **o)


def gen_adjgrad():
    """Reference `X.train(requires_adjgrad=True)` (recommender/LightGCN.py:29-80, NGCF.py:31-79): the gradient of the batch losses with respect to the
    normalised adjacency's stored entries, accumulated the reference's way -- `sparse_norm_adj.grad` is never zeroed (it is not an optimizer
    parameter), and `Matgrad += sparse_norm_adj.grad` adds that running sum after every step -- and returned as (Matgrad + Matgrad.T)[:U, U:].
    NOTE: the NGCF run is not reproducible from one generation to the next (tables after the 44 steps: two outcomes 3.8e-3 / 7.4e-3 apart; the first
    step agrees to 1e-7); tests/test_gpu_api.py::test_ngcf_train_requires_adjgrad_matches_reference_run holds the 44-step figures to bars that cover both."""
    import io, contextlib
    from recommender.NGCF import NGCF
    for fname, cls, kw in (('g21_adjgrad.npz', LightGCN, dict(emb_size=16, n_layers=2)), ('g22_adjgrad_ngcf.npz', NGCF, dict(emb_size=32, n_layers=2, model_name='NGCF'))):
        args = rec_args(**kw)
        seedSet(2018)
        data = DataLoader(args)
        rec = cls(args, data)
        init = {k: v.detach().numpy().copy() for k, v in rec.model.state_dict().items()}
        u0 = rec.model.embedding_dict['user_emb'].detach().numpy().copy(); i0 = rec.model.embedding_dict['item_emb'].detach().numpy().copy()
        # the first step alone (a 44-step trajectory amplifies rounding differences, NGCF's by a lot): sparse_norm_adj.grad after the first backward
        first = {}
        orig_backward = torch.Tensor.backward

        def backward_and_capture(self, *a, **k):
            out = orig_backward(self, *a, **k)
            if not first:
                gr = rec.model.sparse_norm_adj.grad.coalesce()
                first['idx'], first['val'] = gr.indices().numpy().copy(), gr.values().numpy().copy()
            return out
        torch.Tensor.backward = backward_and_capture
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                block = rec.train(requires_adjgrad=True, Epoch=2, gradIterationNum=10, evalNum=1)
        finally:
            torch.Tensor.backward = orig_backward
        U = data.user_num
        N = U + data.item_num
        M1 = sp.coo_matrix((first['val'], (first['idx'][0], first['idx'][1])), shape=(N, N)).tocsr()
        B1 = (M1 + M1.T).tocsr()[:U, U:].tocoo()
        block = block.detach().numpy()
        r, c = np.nonzero(block)
        extra = {'init_' + k.replace('.', '__'): v for k, v in init.items() if 'embedding_dict' not in k}
        save(fname, user0=u0, item0=i0, block_row=r.astype(np.int32), block_col=c.astype(np.int32), block_val=block[r, c].astype(np.float32),
             first_row=B1.row.astype(np.int32), first_col=B1.col.astype(np.int32), first_val=B1.data.astype(np.float32),
             block_shape=np.array(block.shape, np.int64), user=rec.model.embedding_dict['user_emb'].detach().numpy().copy(),
             item=rec.model.embedding_dict['item_emb'].detach().numpy().copy(), next_random=np.array([random.random()], np.float64), **extra)


def gen_adjgrad_simgcl(which='simgcl'):
    """(which = 'xsimgcl': the same capture for `XSimGCL.train(requires_adjgrad=True)`, recommender/XSimGCL.py:46-85 -- ONE perturbed forward per step, two noise
    draws; written to g24_adjgrad_xsimgcl.npz.)
    Reference `SimGCL.train(requires_adjgrad=True)` (recommender/SimGCL.py:36-85): `sparse_norm_adj` takes gradient from all THREE forwards of a step
    (the clean one and the two perturbed views of cal_cl_loss, :212-219; the perturbation itself carries none).  The views' noise is injected:
    the k-th torch.rand_like call of the run returns torch.rand(N, d, generator=manual_seed(5000 + k)) -- the product's test regenerates the same
    sequence instead of storing 88 tables.  Captured: the first step's gradient, the returned block after one epoch (22 steps, running-sum quirk as
    in g21), the tables, the RNG stream."""
    import io, contextlib
    if which == 'xsimgcl':
        from recommender.XSimGCL import XSimGCL as Cls
    elif which == 'ncl':
        # NCL (recommender/NCL.py:114-186), one warm-up epoch (no k-means before epoch 5): sparse_norm_adj takes gradient from the main forward only -- the
        # structure term propagates over a FRESH tensor built per step (:135), which carries none; no noise draws; g25_adjgrad_ncl.npz
        from recommender.NCL import NCL as Cls
    else:
        Cls = SimGCL
    args = rec_args(emb_size=16, n_layers=2, model_name=Cls.__name__)
    seedSet(2018)
    data = DataLoader(args)
    with contextlib.redirect_stdout(io.StringIO()):
        rec = Cls(args, data)
    u0 = rec.model.embedding_dict['user_emb'].detach().numpy().copy(); i0 = rec.model.embedding_dict['item_emb'].detach().numpy().copy()
    first, calls = {}, [0]
    orig_backward, orig_rand_like = torch.Tensor.backward, torch.rand_like

    def rand_like(x, *a, **k):
        g = torch.Generator().manual_seed(5000 + calls[0])
        calls[0] += 1
        return torch.rand(x.shape, generator=g)

    def backward_and_capture(self, *a, **k):
        out = orig_backward(self, *a, **k)
        if not first:
            gr = rec.model.sparse_norm_adj.grad.coalesce()
            first['idx'], first['val'] = gr.indices().numpy().copy(), gr.values().numpy().copy()
        return out
    torch.Tensor.backward, torch.rand_like = backward_and_capture, rand_like
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            block = rec.train(requires_adjgrad=True, Epoch=1, gradIterationNum=10, evalNum=1)
    finally:
        torch.Tensor.backward, torch.rand_like = orig_backward, orig_rand_like
    U = data.user_num
    N = U + data.item_num
    M1 = sp.coo_matrix((first['val'], (first['idx'][0], first['idx'][1])), shape=(N, N)).tocsr()
    B1 = (M1 + M1.T).tocsr()[:U, U:].tocoo()
    block = block.detach().numpy()
    r, c = np.nonzero(block)
    save({'xsimgcl': 'g24_adjgrad_xsimgcl.npz', 'ncl': 'g25_adjgrad_ncl.npz'}.get(which, 'g23_adjgrad_simgcl.npz'), user0=u0, item0=i0, block_row=r.astype(np.int32), block_col=c.astype(np.int32), block_val=block[r, c].astype(np.float32),
         first_row=B1.row.astype(np.int32), first_col=B1.col.astype(np.int32), first_val=B1.data.astype(np.float32),
         block_shape=np.array(block.shape, np.int64), user=rec.model.embedding_dict['user_emb'].detach().numpy().copy(),
         item=rec.model.embedding_dict['item_emb'].detach().numpy().copy(), next_random=np.array([random.random()], np.float64),
         noise_calls=np.array([calls[0]], np.int64), noise_seed0=np.array([5000], np.int64))


SGL_SGD_LR = 20.0


def gen_adjgrad_sgl():
    """Reference `SGL.train(requires_adjgrad=True)` / `(requires_embgrad=True)` (recommender/SGL.py:39-95): unlike the other models the gradient is taken
    w.r.t. the two DROPPED graphs of the epoch (`dropped_adj*.requires_grad`), summed per step into grad_mat* -- `.grad` is never zeroed inside the epoch, so
    grad_mat adds the running sum -- and folded into gradAll[U, I] = the upper-right block of grad_mat1 + grad_mat2 at the END of the epoch; with
    requires_embgrad the tables' `.grad` of the epoch's LAST step is added at the end of the epoch (:82-84).  One epoch each; the first step's gradient of view 1
    is captured through the adjacency handed to cal_cl_loss."""
    import io, contextlib
    from recommender.SGL import SGL
    args = rec_args(emb_size=16, n_layers=2, model_name='SGL')
    out = {}
    for mode in ('adj', 'emb', 'sgd'):
        seedSet(2018)
        data = DataLoader(args)
        rec = SGL(args, data)
        U, I = data.user_num, data.item_num
        if mode == 'sgd':
            # the same adjacency-gradient run under plain SGD (train()'s `optimizer` argument): Adam's g / sqrt(v) amplifies rounding-level differences in
            # small gradients to +-lr per step (tools/sgl_adjgrad_conditioning.py), SGD does not, so this trajectory pins the 22-step accumulation at 1e-4
            with contextlib.redirect_stdout(io.StringIO()):
                res = rec.train(requires_adjgrad=True, Epoch=1, gradIterationNum=10, evalNum=1, optimizer=torch.optim.SGD(rec.model.parameters(), lr=SGL_SGD_LR))
            block = res.detach().numpy()
            r, c = np.nonzero(block)
            moved = (rec.model.embedding_dict['item_emb'].detach() - torch.from_numpy(out['item0'])).norm() / torch.from_numpy(out['item0']).norm()
            print('  SGL under SGD(lr=%g): item table moved by %.3f of its norm in one epoch' % (SGL_SGD_LR, float(moved)))
            out.update(sgd_block_row=r.astype(np.int32), sgd_block_col=c.astype(np.int32), sgd_block_val=block[r, c].astype(np.float32), sgd_lr=np.array([SGL_SGD_LR]),
                       sgd_user=rec.model.embedding_dict['user_emb'].detach().numpy().copy(), sgd_item=rec.model.embedding_dict['item_emb'].detach().numpy().copy())
            continue
        if mode == 'adj':
            out['user0'] = rec.model.embedding_dict['user_emb'].detach().numpy().copy(); out['item0'] = rec.model.embedding_dict['item_emb'].detach().numpy().copy()
        first, seen = {}, {}
        orig_cl, orig_backward = rec.model.cal_cl_loss, torch.Tensor.backward

        def cl_wrap(idx, a1, a2):
            seen['a1'] = a1
            return orig_cl(idx, a1, a2)

        def backward_and_capture(self, *a, **k):
            res = orig_backward(self, *a, **k)
            if mode == 'adj' and not first and 'a1' in seen and seen['a1'].grad is not None:
                gr = seen['a1'].grad.coalesce()
                first['idx'], first['val'] = gr.indices().numpy().copy(), gr.values().numpy().copy()
            return res
        rec.model.cal_cl_loss = cl_wrap
        torch.Tensor.backward = backward_and_capture
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                res = rec.train(requires_adjgrad=(mode == 'adj'), requires_embgrad=(mode == 'emb'), Epoch=1, gradIterationNum=10, evalNum=1)
        finally:
            torch.Tensor.backward = orig_backward
        if mode == 'adj':
            block = res.detach().numpy()
            r, c = np.nonzero(block)
            sel = first['idx'][0] < U
            out.update(block_row=r.astype(np.int32), block_col=c.astype(np.int32), block_val=block[r, c].astype(np.float32), block_shape=np.array(block.shape, np.int64),
                       first_row=first['idx'][0][sel].astype(np.int32), first_col=(first['idx'][1][sel] - U).astype(np.int32), first_val=first['val'][sel].astype(np.float32),
                       user=rec.model.embedding_dict['user_emb'].detach().numpy().copy(), item=rec.model.embedding_dict['item_emb'].detach().numpy().copy(),
                       next_random=np.array([random.random()], np.float64))
        else:
            out.update(emb_usergrad=res[2].detach().numpy().copy(), emb_itemgrad=res[3].detach().numpy().copy(), emb_next_random=np.array([random.random()], np.float64))
    save('g26_adjgrad_sgl.npz', **out)


if __name__ == '__main__':
    only = set(sys.argv[1:])                          # e.g. `gen_golden.py xsimgcl` regenerates that fixture alone
    if only:
        data = DataLoader(rec_args())
        data_training0 = [list(r) for r in data.training_data]
        if 'xsimgcl' in only:
            gen_xsimgcl(data)
        if 'sgl' in only:
            gen_sgl(data)
        if 'ncl' in only:
            gen_ncl(data)
        if 'bilevel' in only:
            gen_bilevel()
        if 'infoattack' in only:
            gen_infoattack()
        if 'pipattack' in only:
            gen_pipattack()
        if 'gray' in only:
            gen_gray()
        if 'gta' in only:
            gen_gta()
        if 'ngcf128' in only:
            gen_ngcf128()
        if 'victims' in only:
            gen_victims()
        if 'fake_rows' in only:
            gen_fake_rows()
        if 'adjgrad' in only:
            gen_adjgrad()
        if 'adjgrad_simgcl' in only:
            gen_adjgrad_simgcl()
        if 'adjgrad_xsimgcl' in only:
            gen_adjgrad_simgcl('xsimgcl')
        if 'adjgrad_ncl' in only:
            gen_adjgrad_simgcl('ncl')
        if 'adjgrad_sgl' in only:
            gen_adjgrad_sgl()
        sys.exit(0)
    gen_dataset()
    data = gen_sampler()
    # the sampler shuffled data.training_data in place; keep a pristine file-order copy for the step fixtures
    args0 = rec_args()
    data = DataLoader(args0)
    data_training0 = [list(r) for r in data.training_data]
    gen_losses()
    gen_adj(data)
    gen_forward_and_steps(data)
    gen_simgcl(data)
    gen_train_api()
    gen_attacks()
    gen_ngcf()
    gen_xsimgcl(data)
    gen_sgl(data)
    gen_ncl(data)
    gen_bilevel()
    gen_infoattack()
    gen_pipattack()
    gen_gray()
    gen_gta()
    gen_ngcf128()
    gen_victims()
    gen_fake_rows()
    gen_adjgrad()
    gen_adjgrad_simgcl()
    gen_adjgrad_simgcl('xsimgcl')
    gen_adjgrad_simgcl('ncl')
    gen_adjgrad_sgl()
    print('done; scratch dir', SCRATCH)
