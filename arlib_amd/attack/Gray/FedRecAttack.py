"""FedRecAttack -- mirror of the reference's attack/Gray/FedRecAttack.py (posionDataAttack :55-127) on the MI355X kernels: every outer
step re-learns the user table (5 epochs, Adam over `user_emb` only), then takes one step of the CW loss
mean(<Pu[u], Pi[neg]> - <Pu[u], Pi[t]>) over (real user, target), neg = successive pops from the tail of the user's masked top-k
(streaming score+mask+top-k kernel; bilinear operator form of the loss, attack/White/BiLevelAttackByBatchInject.py:_CwLoss).
The fake rows are the top-n of the scores of the last forward before the last step."""
import torch

from ..White.BiLevelAttackByBatchInject import _CwLoss
from ..White.DLAttack import masked_topk
from ._userlearn import UserLearningBiLevel


class FedRecAttack(UserLearningBiLevel):
    fresh_forward_for_rows = False

    def outer_loss(self, model, mask, topk):
        Pu, Pi = model()
        with torch.no_grad():
            top_idx, _ = masked_topk(Pu.detach(), Pi.detach(), mask, min(topk, self.itemNum))
        return _CwLoss.apply(Pu, Pi, top_idx, self.userNum, self.targetItem), Pu, Pi
