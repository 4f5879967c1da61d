"""GTA -- mirror of the reference's attack/Black/GTA.py (GTA.posionDataAttack :57-98, fakeUserInject :116-145, proxyLG :147-236) on the
MI355X kernels.  Black-box: the attacker trains its own proxy recommender (`proxyLG`, a LightGCN whose every training batch adds
0.01 x CW loss over (user, target) pairs) on the observed data plus the fake users, and takes the fake profiles from the proxy's scores.

proxyLG.train (:155-236) per batch: forward, masked U x I scores and their top-k (the reference fills a host matrix in 1024-row slabs),
negatives = successive pops from the tail of each user's list, CW = mean over pairs of mean over the d coordinates of
(u * neg - u * target)  [`.mean(dim=1)`, i.e. the CW loss of CLeaR divided by d], batch loss = 0.01 CW + BPR + L2.  Here: the streaming
score+mask+top-k kernel and the bilinear operator form of the CW term (one SpMM), on the outputs of the step's own forward.

As executed by the reference:
* `getPopularItemId(m, n)` is `np.argsort(m.sum(0))[-n:]` on a 1 x I np.matrix: the slice takes the only ROW, so the "popular" pool
  is every item (in ascending popularity order) and the seed items are `random.sample`d from all of them (:66);
* the proxy shares the victim's DataLoader: fake users and their filler interactions are appended to it;
* fakeUserInject re-creates the proxy with fresh tables (nothing is copied back) and trains it for 30 epochs (:139-145);
* the seed items' scores are set to 0 (not masked) before the top-n/2 projection, then seeds and targets are forced to 1 (:87-93);
* the graph returned is the best EVALUATED one: the profile built in the last epoch is never evaluated.
"""
import random

import numpy as np
import scipy.sparse as sp
import torch

from ... import ops
from ...recommender.LightGCN import LightGCN
from ...util.metrics import AttackMetric
from .._common import AttackBase, init_graph, rebuild_interaction_matrix, with_fake_rows, append_rows
from ..White.BiLevelAttackByBatchInject import _CwLoss
from ..White.DLAttack import device_mask, masked_topk


class proxyLG(LightGCN):
    has_extra_loss = True
    fused_extra_loss = False
    extra_loss_takes_outputs = True
    print_every = 1000

    def __init__(self, args, data, targetItem):
        super().__init__(args, data)
        self.userNum, self.itemNum = data.user_num, data.item_num
        self.targetItem = targetItem
        self.batchSize = 1024
        self._mask = None

    def cw_term(self, Pu, Pi):
        m = self.data.matrix()
        if self._mask is None or self._mask[0] is not m:
            self._mask = (m, device_mask(m))
        rp, mc = self._mask[1]
        n = self.userNum
        with torch.no_grad():
            top_idx, _ = masked_topk(Pu.detach()[:n], Pi.detach(), (rp[:n + 1], mc), min(min(self.topN), self.itemNum))
        return _CwLoss.apply(Pu, Pi, top_idx, n, self.targetItem) / Pu.shape[1]

    def _extra_loss(self, model, user_idx, pos_idx, rec_user_emb, rec_item_emb):
        return 0.01 * self.cw_term(rec_user_emb, rec_item_emb)


class GTA(AttackBase):
    def __init__(self, arg, data):
        super().__init__(arg, data)
        self.batchSize = 128

    def posionDataAttack(self, recommend):
        recommender = proxyLG(recommend.args, recommend.data, self.targetItem)
        self.fakeUserInject(recommender)
        uiAdj = sp.csr_matrix(recommender.data.matrix())
        optimizer = torch.optim.Adam(recommender.model.parameters(), lr=recommender.args.lRate)
        recommender.train(Epoch=self.innerEpoch, optimizer=optimizer, evalNum=5)
        topk = min(recommender.topN)
        bestTargetHitRate, bestAdj = -1, None
        order = np.argsort(recommend.data.matrix()[:, :].sum(0))[-(self.itemNum // 5):]        # 1 x I np.matrix: the slice keeps its only row
        seedItem = random.sample(order.tolist()[0], self.maliciousFeedbackNum // 2)
        Up = self.userNum + self.fakeUserNum
        for epoch in range(self.Epoch):
            init_graph(recommender.model, uiAdj, Up, self.itemNum, n_real=self.userNum)
            recommender.train(Epoch=self.innerEpoch, optimizer=optimizer, evalNum=5)
            targetHitRate = AttackMetric(recommender, self.targetItem, [topk]).hitRate()[0]
            print(targetHitRate)
            if targetHitRate > bestTargetHitRate:
                bestAdj = uiAdj.copy()
                bestTargetHitRate = targetHitRate
            uiAdj = bestAdj.copy()
            with torch.no_grad():
                Pu, Pi = recommender.model()
                fake = torch.as_tensor(self.fakeUser, device=Pu.device)
                scores = (Pu[fake] @ Pi.T).contiguous()
                scores[:, seedItem] = 0
            rows, _ = ops.topn_project_rows(scores, int(self.maliciousFeedbackNum // 2))
            rows[:, self.targetItem + seedItem] = 1
            uiAdj = with_fake_rows(uiAdj, self.userNum, rows.cpu().numpy())
            print('BiLevel epoch {} is over\n'.format(epoch + 1))
        self.interact = bestAdj
        return self.interact

    def fakeUserInject(self, recommender):
        """GTA.py:116-145: fake users with random fillers, proxy re-created (fresh tables) on the extended data and trained 30 epochs."""
        recommender.model = recommender.model.cuda()
        data = recommender.data
        data.user_num += self.fakeUserNum
        for i in range(self.fakeUserNum):
            data.user['fakeuser{}'.format(i)] = len(data.user)
            data.id2user[len(data.user) - 1] = 'fakeuser{}'.format(i)
        self.fakeUser = list(range(self.userNum, self.userNum + self.fakeUserNum))
        for u in self.fakeUser:
            append_rows(data, [(data.id2user[u], data.id2item[i]) for i in random.sample(tuple(set(range(self.itemNum))), int(self.maliciousFeedbackNum))])
        _, _, data.interaction_mat = rebuild_interaction_matrix(data)
        recommender.__init__(recommender.args, data, self.targetItem)
        init_graph(recommender.model, sp.csr_matrix(data.matrix()), self.userNum + self.fakeUserNum, self.itemNum, n_real=self.userNum)
        recommender.train(Epoch=30)
