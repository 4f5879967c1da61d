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


class _CwLoss(torch.autograd.Function):
    """CW loss and its gradient on the packed table straight from the top-k lists (ops.cw_topk_term: no U*T index lists, no operator build,
    deterministic)."""

    @staticmethod
    def forward(ctx, Pu, Pi, top_idx, n_real, targets):
        X = _packed(Pu, Pi)
        Up = Pu.shape[0]
        tg = targets.to(X.device, torch.int64) if isinstance(targets, torch.Tensor) else torch.as_tensor(targets, device=X.device, dtype=torch.int64)
        loss, G, _ = ops.cw_topk_term(X.contiguous(), Up, n_real, top_idx.contiguous(), tg, want_w=False, check_range=False)
        ctx.save_for_backward(G)
        ctx.Up = Up
        return loss[0]

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
