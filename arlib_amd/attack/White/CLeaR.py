"""CLeaR -- mirror of the reference's attack/White/CLeaR.py (posionDataAttack :56-159, project :161-175,
fakeUserInject :177-210) on the MI355X kernels: bi-level attack, surrogate step = CW loss over (real user x target)
pairs + spectral-feature-augmentation L1 loss, fake rows := top-n of the surrogate's scores, keep the best poisoned
graph by target hit-rate.

Mechanics: streaming score+mask+top-k instead of the host U x I buffer (CLeaR.py:75-82); index tensors instead of Python
list building of U*T triples (:83-88); propagation forward/backward through the SpMM kernels; AttackMetric through the
top-k kernel.  The small dense SFA algebra on the gathered [3UT, d] matrix (:98-125) uses ATen matmuls (plumbing).
"""
import random
from copy import deepcopy

import numpy as np
import scipy.sparse as sp
import torch

from ... import ops
from ...util.metrics import AttackMetric
from .._common import AttackBase, DEVICE, symmetric_adjacency, rebuild_interaction_matrix, reinit_with_tables, cw_pairs
from .DLAttack import masked_topk


def spectral_feature_augmentation_loss(H, r0):
    """F.l1_loss(SFA(H, 1), H) with r(0) = r0 (CLeaR.py:98-125): r = H^T H r0; H_aug = H - H r r^T / ||r||^2."""
    r = H.T @ (H @ r0)
    H_aug = H - (H @ torch.outer(r, r)) / torch.norm(r) ** 2
    return torch.nn.functional.l1_loss(H_aug, H)


class CLeaR(AttackBase):
    def __init__(self, arg, data):
        super().__init__(arg, data)
        self.batchSize = 2048

    def surrogate_loss(self, model, uiAdj2, topk, r0=None):
        """One evaluation of lossall = CWloss + sfaloss (CLeaR.py:74-126); returns (lossall, Pu, Pi, cw, sfa)."""
        Pu, Pi = model()
        with torch.no_grad():
            top_idx, _ = masked_topk(Pu.detach(), Pi.detach(), uiAdj2, min(topk, self.itemNum))
            users, pos, neg = cw_pairs(top_idx, self.userNum, self.targetItem, pop=True)
        user_emb, pos_items_emb, neg_items_emb = Pu[users], Pi[pos], Pi[neg]
        cw = ((user_emb * neg_items_emb).sum(1) - (user_emb * pos_items_emb).sum(1)).mean()
        emb_cat = torch.cat((user_emb, pos_items_emb, neg_items_emb), dim=0)
        if r0 is None:
            r0 = torch.randn(emb_cat.size(1)).to(emb_cat.device)          # CLeaR.py:100-103: CPU generator, then moved
        sfa = spectral_feature_augmentation_loss(emb_cat, r0)
        return cw + sfa, Pu, Pi, cw, sfa

    def posionDataAttack(self, recommender):
        self.fakeUserInject(recommender)
        uiAdj = recommender.data.matrix().tolil()
        optimizer = torch.optim.Adam(recommender.model.parameters(), lr=recommender.args.lRate / 10)
        topk = min(recommender.topN)
        bestTargetHitRate, bestAdj = -1, None
        Up = self.userNum + self.fakeUserNum
        for epoch in range(self.Epoch):
            tmpRecommender = deepcopy(recommender)
            uiAdj2 = uiAdj.copy()
            tmpRecommender.model._init_uiAdj(symmetric_adjacency(uiAdj2, Up, self.itemNum))
            optimizer_attack = torch.optim.Adam(tmpRecommender.model.parameters(), lr=recommender.args.lRate)
            Pu = Pi = None
            for _ in range(self.outerEpoch):
                lossall, Pu, Pi, _, _ = self.surrogate_loss(tmpRecommender.model, uiAdj2, topk)
                optimizer_attack.zero_grad()
                lossall.backward()
                optimizer_attack.step()
            # fake rows := top-n of the scores from the last forward (computed before the last step, as in CLeaR.py:130-135)
            with torch.no_grad():
                fake = torch.as_tensor(self.fakeUser, device=Pu.device)
                scores = (Pu[fake] @ Pi.T).contiguous()
            proj, _ = ops.topn_project_rows(scores, int(self.maliciousFeedbackNum))
            proj[:, self.targetItem] = 1
            uiAdj2[self.fakeUser, :] = proj.cpu().numpy()
            uiAdj = uiAdj2.copy()
            recommender.model._init_uiAdj(symmetric_adjacency(uiAdj, Up, self.itemNum))
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
            for i in random.sample(tuple(set(range(self.itemNum))), int(self.maliciousFeedbackNum)):
                data.training_data.append((data.id2user[u], data.id2item[i]))
        _, _, data.interaction_mat = rebuild_interaction_matrix(data)
        reinit_with_tables(recommender, Pu, Pi)
