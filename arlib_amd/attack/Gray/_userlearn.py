"""Shared body of the two gray-box bi-level attacks, attack/Gray/FedRecAttack.py and attack/Gray/A_ra.py of the reference: the
attacker does not know the user table, so every outer step first re-learns it (5 epochs of the surrogate's own training with an Adam over
`user_emb` alone), then takes one step of the attack loss on the surrogate's parameters.

As executed by the reference (FedRecAttack.py:55-127, A_ra.py:57-120):
* the user-only optimiser is bound to the NAME `optimizer`, which the inner optimisation `recommender.train(..., optimizer=optimizer)`
  uses afterwards: from the first outer step on, the victim's "retraining" is driven by an optimiser that owns none of its parameters
  and moves nothing (it still consumes the sampler's random stream and evaluates) -- Recommender._train_loop's `inert` path;
* the whole filler budget is projected every epoch (plain top-n), targets are forced to 1, the best poisoned graph by target hit rate
  is kept.
"""
from copy import deepcopy

import scipy.sparse as sp
import torch

from ... import ops
from ...util.metrics import AttackMetric
from .._common import init_graph, with_fake_rows
from ..White.CLeaR import CLeaR
from ..White.DLAttack import device_mask
from ...util.optim import Adam        # torch.optim.Adam, stepped by arl_adam_dense_f32


class UserLearningBiLevel(CLeaR):
    """Subclasses provide outer_loss(model, mask, topk) -> (loss, Pu, Pi) and `fresh_forward_for_rows`."""
    fresh_forward_for_rows = False
    relearn_epochs = 5

    def __init__(self, arg, data):
        super().__init__(arg, data)
        self.batchSize = 128

    def posionDataAttack(self, recommender):
        self.fakeUserInject(recommender)
        uiAdj = sp.csr_matrix(recommender.data.matrix())
        optimizer = torch.optim.Adam(recommender.model.parameters(), lr=recommender.args.lRate / 10)
        topk = min(recommender.topN)
        bestTargetHitRate, bestAdj = -1, None
        Up = self.userNum + self.fakeUserNum
        for epoch in range(self.Epoch):
            tmpRecommender = deepcopy(recommender)
            uiAdj2 = uiAdj.copy()
            init_graph(tmpRecommender.model, uiAdj2, Up, self.itemNum, n_real=self.userNum)
            optimizer_attack = Adam(tmpRecommender.model.parameters(), lr=recommender.args.lRate)
            mask = device_mask(uiAdj2)
            Pu = Pi = None
            self.last_top_idx = None
            for _ in range(self.outerEpoch):
                optimizer = torch.optim.Adam([tmpRecommender.model.embedding_dict['user_emb']], lr=recommender.args.lRate)
                tmpRecommender.train(Epoch=self.relearn_epochs, optimizer=optimizer, evalNum=5)
                loss, Pu, Pi = self.outer_loss(tmpRecommender.model, mask, topk)
                self.last_outer_loss = loss.detach()
                optimizer_attack.zero_grad()
                loss.backward()
                optimizer_attack.step()
            with torch.no_grad():
                if self.fresh_forward_for_rows or Pu is None:
                    Pu, Pi = tmpRecommender.model()
                fake = torch.as_tensor(self.fakeUser, device=Pu.device)
                scores = (Pu[fake] @ Pi.T).contiguous()
            rows, _ = ops.topn_project_rows(scores, int(self.maliciousFeedbackNum))
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
