"""CLeaR -- mirror of the reference's attack/White/CLeaR.py (posionDataAttack :56-159, project :161-175,
fakeUserInject :177-210) on the MI355X kernels: bi-level attack, surrogate step = CW loss over (real user x target)
pairs + spectral-feature-augmentation L1 loss, fake rows := top-n of the surrogate's scores, keep the best poisoned
graph by target hit-rate.

Mechanics: streaming score+mask+top-k instead of the host U x I buffer (CLeaR.py:75-82); index tensors instead of Python
list building of U*T triples (:83-88); propagation forward/backward through the SpMM kernels; AttackMetric through the
top-k kernel; the CW + SFA losses and their gradients (:89-126) without materialising the [3UT, d] matrix (_CwSfaLoss).
"""
import random
from copy import deepcopy

import numpy as np
import scipy.sparse as sp
import torch

from ... import ops
from ...util.metrics import AttackMetric
from .._common import AttackBase, DEVICE, symmetric_adjacency, init_graph, rebuild_interaction_matrix, reinit_with_tables, cw_pairs, with_fake_rows, append_rows
from ...util.optim import Adam        # torch.optim.Adam, stepped by arl_adam_dense_f32
from .DLAttack import masked_topk, device_mask


def _packed(Pu, Pi):
    """[U'+I, d] table holding both outputs: the encoder hands out two views of one buffer, otherwise concatenate."""
    U, d = Pu.shape
    if (Pu.is_contiguous() and Pi.is_contiguous() and Pi.data_ptr() == Pu.data_ptr() + U * d * Pu.element_size()
            and Pu.untyped_storage().data_ptr() == Pi.untyped_storage().data_ptr()):
        return torch.as_strided(Pu, (U + Pi.shape[0], d), (d, 1))
    return torch.cat([Pu, Pi], 0).contiguous()


class _CwSfaLoss(torch.autograd.Function):
    """(CWloss, sfaloss) of CLeaR.py:89-126 from the propagated tables and the users' top-k lists.

    H = cat(Pu[users], Pi[pos], Pi[neg]) has 3*U*T rows but only U + I distinct ones, so neither it nor the U*T index lists
    are gathered: the CW term is bilinear in the packed table (ops.cw_topk_term: user rows gather, item rows sum in 64-bit fixed point) and the SFA
    term is three weighted passes over the table (ops.sfa_l1, row weight = multiplicity in H)."""

    @staticmethod
    def forward(ctx, Pu, Pi, top_idx, n_real, targets, r0):
        X = _packed(Pu, Pi)
        Up, I, d, T = Pu.shape[0], Pi.shape[0], Pu.shape[1], len(targets)
        tg = targets.to(X.device, torch.int64) if isinstance(targets, torch.Tensor) else torch.as_tensor(targets, device=X.device, dtype=torch.int64)
        # one hand-written kernel group (arl_cw_topk_term_f32): the loss, its gradient on every row and the SFA term's row multiplicities straight from the
        # top-k lists -- no operator build (sort, searchsorted, scatters), no ATen launches between the scoring pass and the SFA kernels, deterministic
        cw1, G_cw, w = ops.cw_topk_term(X.contiguous(), Up, n_real, top_idx.contiguous(), tg, check_range=False)
        cw = cw1[0]
        sfa, G_sfa = ops.sfa_l1(X, w, r0.to(X.device, torch.float32).contiguous(), 3 * n_real * T * d)
        ctx.save_for_backward(G_cw, G_sfa)
        ctx.Up = Up
        return cw, sfa[0]

    @staticmethod
    def backward(ctx, g_cw, g_sfa):
        G_cw, G_sfa = ctx.saved_tensors
        G = torch.addcmul(g_cw * G_cw, G_sfa, g_sfa)              # g_cw G_cw + g_sfa G_sfa in two passes
        return G[:ctx.Up], G[ctx.Up:], None, None, None, None


class CLeaR(AttackBase):
    def __init__(self, arg, data):
        super().__init__(arg, data)
        self.batchSize = 2048

    # The surrogate step must not make the host wait for the device: a blocking copy in the middle of a step drains the stream, exposes the host's
    # wake-up latency (tens of ms on some boxes) and stops the host from queueing ahead.  Small host data therefore travels through pinned
    # staging slots with asynchronous copies, and the target ids live on the device.
    _RING = 8

    def _targets_on(self, device):
        t = getattr(self, '_tg_dev', None)
        if t is None or t.device != device or t.numel() != len(self.targetItem):
            t = self._tg_dev = torch.as_tensor(self.targetItem, device=device, dtype=torch.int64)
        return t

    def _upload(self, host, device):
        """Device copy of a small CPU tensor without a stream synchronisation (pinned slot + asynchronous copy; a slot is reused after
        _RING uploads, once the copy that read it has completed)."""
        if device.type != 'cuda':
            return host.to(device)
        ring = getattr(self, '_up_ring', None)
        if ring is None or ring['pin'].shape[1] != host.numel() or ring['pin'].dtype != host.dtype or ring['dev'].device != device:
            ring = self._up_ring = {'pin': torch.empty(self._RING, host.numel(), dtype=host.dtype).pin_memory(),
                                    'dev': torch.empty(self._RING, host.numel(), dtype=host.dtype, device=device), 'ev': [None] * self._RING, 'n': 0}
        i = ring['n'] % self._RING
        ring['n'] += 1
        if ring['ev'][i] is not None:
            ring['ev'][i].synchronize()
        ring['pin'][i].copy_(host.reshape(-1))
        ring['dev'][i].copy_(ring['pin'][i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ring['ev'][i] = ev
        return ring['dev'][i].view(host.shape)

    def surrogate_loss(self, model, uiAdj2, topk, r0=None, warm=None):
        """One evaluation of lossall = CWloss + sfaloss (CLeaR.py:74-126); returns (lossall, Pu, Pi, cw, sfa).
        `uiAdj2`: the poisoned U' x I interactions (scipy) or their device_mask()."""
        Pu, Pi = model()
        with torch.no_grad():
            top_idx, _ = masked_topk(Pu.detach(), Pi.detach(), uiAdj2, min(topk, self.itemNum), warm=warm)
            self.last_top_idx = top_idx                                    # next step's warm start (same users, same mask)
        if r0 is None:
            r0 = self._upload(torch.randn(Pu.size(1)), Pu.device)          # CLeaR.py:100-103: CPU generator, then moved
        cw, sfa = _CwSfaLoss.apply(Pu, Pi, top_idx, self.userNum, self._targets_on(Pu.device), r0)
        return cw + sfa, Pu, Pi, cw, sfa

    def posionDataAttack(self, recommender):
        self.fakeUserInject(recommender)
        uiAdj = sp.csr_matrix(recommender.data.matrix())              # CSR throughout (the reference keeps a lil matrix)
        optimizer = torch.optim.Adam(recommender.model.parameters(), lr=recommender.args.lRate / 10)
        topk = min(recommender.topN)
        bestTargetHitRate, bestAdj = -1, None
        Up = self.userNum + self.fakeUserNum
        for epoch in range(self.Epoch):
            tmpRecommender = deepcopy(recommender)
            uiAdj2 = uiAdj.copy()
            init_graph(tmpRecommender.model, uiAdj2, Up, self.itemNum, n_real=self.userNum)
            optimizer_attack = Adam(tmpRecommender.model.parameters(), lr=recommender.args.lRate)
            Pu = Pi = None
            mask = device_mask(uiAdj2)          # the poisoned pattern is fixed while the surrogate is trained
            warm = None
            for _ in range(self.outerEpoch):
                lossall, Pu, Pi, _, _ = self.surrogate_loss(tmpRecommender.model, mask, topk, warm=warm)
                warm = self.last_top_idx
                optimizer_attack.zero_grad()
                lossall.backward()
                optimizer_attack.step()
            # fake rows := top-n of the scores from the last forward (computed before the last step, as in CLeaR.py:130-135)
            with torch.no_grad():
                fake = torch.as_tensor(self.fakeUser, device=Pu.device)
                scores = (Pu[fake] @ Pi.T).contiguous()
            proj, _ = ops.topn_project_rows(scores, int(self.maliciousFeedbackNum))
            proj[:, self.targetItem] = 1
            uiAdj2 = with_fake_rows(uiAdj2, self.userNum, proj.cpu().numpy())
            uiAdj = uiAdj2.copy()
            init_graph(recommender.model, uiAdj, Up, self.itemNum, n_real=self.userNum)
            recommender.train(Epoch=self.innerEpoch, optimizer=optimizer, evalNum=5)
            targetHitRate = AttackMetric(recommender, self.targetItem, [topk]).hitRate()[0]
            print(targetHitRate)
            if targetHitRate > bestTargetHitRate:
                bestAdj = uiAdj.copy()
                bestTargetHitRate = targetHitRate
            uiAdj = bestAdj.copy()
            print('BiLevel epoch {} is over\n'.format(epoch + 1))
        self.interact = bestAdj
        return self.interact

    def project(self, mat, n):
        """Per-row top-n -> ({0,1} matrix, indices) (CLeaR.py:161-175)."""
        M = torch.as_tensor(np.asarray(mat.todense() if hasattr(mat, 'todense') else mat), dtype=torch.float32, device=DEVICE).contiguous()
        out, idx = ops.topn_project_rows(M, int(n))
        return out.cpu(), idx.long().cpu()

    def fakeUserInject(self, recommender):
        """F fake users with `maliciousFeedbackNum` random filler items each, re-init, keep the old tables (CLeaR.py:177-210)."""
        Pu, Pi = recommender.model()
        data = recommender.data
        data.user_num += self.fakeUserNum
        for i in range(self.fakeUserNum):
            data.user['fakeuser{}'.format(i)] = len(data.user)
            data.id2user[len(data.user) - 1] = 'fakeuser{}'.format(i)
        self.fakeUser = list(range(self.userNum, self.userNum + self.fakeUserNum))
        for u in self.fakeUser:
            # random.sample(set(range(I)), n) in the reference (CLeaR.py:187): CPython samples from tuple(set)
            append_rows(data, [(data.id2user[u], data.id2item[i]) for i in random.sample(tuple(set(range(self.itemNum))), int(self.maliciousFeedbackNum))])
        _, _, data.interaction_mat = rebuild_interaction_matrix(data)
        reinit_with_tables(recommender, Pu, Pi)
