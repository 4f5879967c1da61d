"""Shared body of the two bi-level batch attacks, BiLevelAttackBatch.py and BiLevelAttackByBatchInject.py of the reference
(attack/White/): the CLeaR scaffold (fake-user injection, deep-copied surrogate, best poisoned graph by target hit rate) with a
different surrogate loss and with the filler budget spread over the outer epochs: epoch e adds n_e = m // E (+1 for the last
m % E epochs) items per fake user, chosen among the items not chosen before.
"""
from copy import deepcopy

import scipy.sparse as sp
import torch

from ...util.metrics import AttackMetric
from .._common import DEVICE, init_graph, with_fake_rows
from .CLeaR import CLeaR
from .DLAttack import device_mask


class ScheduledBiLevel(CLeaR):
    """Subclasses provide surrogate_loss(model, mask, topk, warm) -> (loss, Pu, Pi) and select(scores, n) -> ({0,1} rows, indices)."""

    def budget(self, epoch):
        m, E = int(self.maliciousFeedbackNum), int(self.Epoch)
        return ([m // E] * (E - m % E) + [m // E + 1] * (m % E))[epoch]

    def posionDataAttack(self, recommender):
        self.fakeUserInject(recommender)
        uiAdj = sp.csr_matrix(recommender.data.matrix())
        optimizer = torch.optim.Adam(recommender.model.parameters(), lr=recommender.args.lRate / 10)
        topk = min(recommender.topN)
        bestTargetHitRate, bestAdj = -1, None
        Up = self.userNum + self.fakeUserNum
        ind = None
        for epoch in range(self.Epoch):
            tmpRecommender = deepcopy(recommender)
            uiAdj2 = uiAdj.copy()
            init_graph(tmpRecommender.model, uiAdj2, Up, self.itemNum, n_real=self.userNum)
            optimizer_attack = torch.optim.Adam(tmpRecommender.model.parameters(), lr=recommender.args.lRate)
            mask = device_mask(uiAdj2)
            Pu = Pi = None
            self.last_top_idx = None
            for _ in range(self.outerEpoch):
                loss, Pu, Pi = self.outer_loss(tmpRecommender.model, mask, topk)
                self.last_outer_loss = loss.detach()
                optimizer_attack.zero_grad()
                loss.backward()
                optimizer_attack.step()
            # fake rows: scores of the last forward (taken before the last step, as in the reference)
            with torch.no_grad():
                fake = torch.as_tensor(self.fakeUser, device=Pu.device)
                scores = (Pu[fake] @ Pi.T).contiguous()
            n_e = self.budget(epoch)
            if ind is None:
                rows, ind = self.select(scores, n_e)
            else:
                scores.scatter_(1, ind.to(scores.device), -10e9)          # items picked in earlier epochs cannot be picked again ...
                rows, cur = self.select(scores, n_e)
                rows.scatter_(1, ind.to(rows.device), 1.0)                # ... and stay in the profile
                ind = torch.cat((ind, cur), dim=1)
            rows[:, self.targetItem] = 1
            uiAdj2 = with_fake_rows(uiAdj2, self.userNum, rows.cpu().numpy())
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
