"""BiLevelAttackByBatchInject -- mirror of the reference's attack/White/BiLevelAttackByBatchInject.py (posionDataAttack
:56-136, project :138-152, fakeUserInject :154-186 = CLeaR's) on the MI355X kernels.

Surrogate loss (:75-94): the CW loss of CLeaR without the SFA term -- mean over (real user, target) of
<Pu[u], Pi[neg]> - <Pu[u], Pi[t]>, neg = successive pops from the tail of the user's masked top-k list.  The reference masks by
`scores - 10e8 * uiAdj2.todense()` on a device-resident U x I matrix; the streaming score+mask+top-k kernel gives the same lists.
The filler budget is spread over the outer epochs with plain top-n projection.
"""
import torch

from ... import ops
from ._bilevel import ScheduledBiLevel
from .CLeaR import _packed
from .DLAttack import masked_topk
from .PGA import cw_operator_from_topk


class _CwLoss(torch.autograd.Function):
    """CW loss as 1/2 X^T M X with the operator of PGA.cw_operator_from_topk (no U*T index lists, no atomics)."""

    @staticmethod
    def forward(ctx, Pu, Pi, top_idx, n_real, targets):
        X = _packed(Pu, Pi)
        Up, T = Pu.shape[0], len(targets)
        ranks = top_idx.shape[1] - 1 - torch.arange(T, device=X.device)
        neg = top_idx[:n_real][:, ranks].long()
        M, _ = cw_operator_from_topk(Up + Pi.shape[0], Up, n_real, targets, neg, X.device)
        G = ops.spmm(M, X)
        ctx.save_for_backward(G)
        ctx.Up = Up
        return 0.5 * (X * G).sum()

    @staticmethod
    def backward(ctx, g):
        G, = ctx.saved_tensors
        G = g * G
        return G[:ctx.Up], G[ctx.Up:], None, None, None


class BiLevelAttackByBatchInject(ScheduledBiLevel):
    def outer_loss(self, model, mask, topk):
        Pu, Pi = model()
        with torch.no_grad():
            top_idx, _ = masked_topk(Pu.detach(), Pi.detach(), mask, min(topk, self.itemNum), warm=self.last_top_idx)
            self.last_top_idx = top_idx
        return _CwLoss.apply(Pu, Pi, top_idx, self.userNum, self.targetItem), Pu, Pi

    def select(self, scores, n):
        out, idx = ops.topn_project_rows(scores.contiguous(), int(n))
        return out, idx.long()
