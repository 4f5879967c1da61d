"""BiLevelAttackBatch -- mirror of the reference's attack/White/BiLevelAttackBatch.py (posionDataAttack :57-147,
relaxProject :165-183, fakeUserInject :185-218 = CLeaR's) on the MI355X kernels.

Surrogate loss (:91-100): CWloss = -mean over (real user, target) of <Pu[u], Pi[t]>.  The reference also builds a masked U x I
score matrix, its top-k and a list of negatives per pair (:76-90), but `neg_score` is commented out of the loss: those lines
consume no random numbers and influence nothing, so they are not executed here (the loss is a closed form in the column sums
of the two tables).

relaxProject (:165-183) as the reference EXECUTES it: the `try` branch draws, per fake user, n of the top-10n positions with
`random.sample`, stores them in a FLOAT tensor and fails in `scatter_` (index dtype); the bare `except` then rebuilds every row
from n random picks among its top-2n (a second `random.sample` per fake user) -- and the function returns that matrix
together with the indices of the FIRST, discarded draw.  Both draws and the mismatch are reproduced (the later epochs mask
and re-add the returned indices, :118-125).
"""
import random

import torch

from ._bilevel import ScheduledBiLevel


class BiLevelAttackBatch(ScheduledBiLevel):
    def outer_loss(self, model, mask, topk):
        Pu, Pi = model()
        t = torch.as_tensor(self.targetItem, device=Pi.device, dtype=torch.long)
        loss = -(Pu[:self.userNum].sum(0) * Pi[t].sum(0)).sum() / float(self.userNum * len(self.targetItem))
        return loss, Pu, Pi

    def relaxProject(self, mat, n):
        """`mat`: [F, I] scores (tensor, array or scipy matrix).  Returns ({0,1} matrix [F, I], indices [F, n])."""
        M = torch.as_tensor(mat.todense() if hasattr(mat, 'todense') else mat, dtype=torch.float32).to('cuda' if torch.cuda.is_available() else 'cpu')
        M = M.reshape(-1, M.shape[-1]).contiguous()
        F, I = M.shape
        n = int(n)
        if 10 * n > I:
            raise ValueError('relaxProject: 10*n exceeds the number of items (torch.topk fails in the reference too, and its fallback then returns an undefined name)')
        top10 = torch.topk(M, 10 * n, dim=1)[1].cpu()
        ind = torch.stack([top10[i, random.sample(list(range(10 * n)), n)] for i in range(F)])                   # first draw: returned
        top2 = torch.topk(M, 2 * n, dim=1)[1].cpu()
        picked = torch.stack([top2[i, random.sample(list(range(2 * n)), n)] for i in range(F)])                  # second draw: applied
        out = torch.zeros_like(M)
        out.scatter_(1, picked.to(M.device), 1.0)
        return out, ind.to(M.device)

    select = relaxProject
