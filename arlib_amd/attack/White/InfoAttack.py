"""InfoAttack -- mirror of the reference's attack/White/InfoAttack.py (posionDataAttack :55-140, relaxProject :158-176,
fakeUserInject :178-210 = CLeaR's, InfoNCEBatch :221-229) on the MI355X kernels.

Surrogate loss (:76-108): Loss = a*CW + b*Info with a = CW/(CW+Info), b = Info/(CW+Info) (detached), CW = CLeaR's CW term,
Info = mean over 256-item batches of the batch means of -log(exp(s_jj/T) / sum_i exp(s_ij/T)), s = normalised ORIGINAL item
table (taken before the fake users are injected, fixed) x normalised current item table, T = 0.2.

Reproduced as executed:
* the "interacted" mask is ONE element: `nozeroInd = uiAdj2.indices` is the 1-D CSR column array, so
  `scores[nozeroInd[0], nozeroInd[1]] = -10e8` writes scores[indices[0], indices[1]] only (:69-70);
* relaxProject draws twice per fake user (n of the top-2n positions, float index tensor, `scatter_` fails; the bare `except`
  draws again and applies that draw); the fake rows come from a fresh forward AFTER the last surrogate step (:110-114).

The item-item InfoNCE (I x I logits: 10^10 at 100 K items) is evaluated in 256-column panels with the gradient accumulated
in the same pass, so nothing of size I x I (or I x 256 per panel for autograd) is kept.
"""
import random
from copy import deepcopy

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from ... import ops
from ...util.metrics import AttackMetric
from .._common import init_graph, with_fake_rows
from ...util.optim import Adam        # torch.optim.Adam, stepped by arl_adam_dense_f32
from .BiLevelAttackByBatchInject import _CwLoss
from .CLeaR import CLeaR
from .DLAttack import masked_topk


class _ItemInfoNCE(torch.autograd.Function):
    """InfoLoss of InfoAttack.py:96-101 and its gradient w.r.t. the current item table, panel by panel."""

    @staticmethod
    def forward(ctx, Pi, view1, temperature, bs):
        with torch.no_grad():
            v1 = F.normalize(view1, dim=1)
            nrm = Pi.norm(dim=1, keepdim=True).clamp_min(1e-12)
            v2 = Pi / nrm
            I = Pi.shape[0]
            k = (I + bs - 1) // bs
            if Pi.is_cuda and Pi.shape[1] in ops.NCE_ALLROWS_WIDTHS and Pi.dtype == torch.float32 and temperature >= ops.NCE_ALLROWS_MIN_TAU:
                # fused form (arl_nce_allrows_*): ttl_j = sum_i exp(<v2_j, v1_i>/T) over ALL items i is a log-sum-exp with v2 as the batch side;
                # only v2 carries gradient.  Batch j // bs has n_j items: weight of item j = 1 / (n_j k).
                v1c, v2c = v1.contiguous(), v2.contiguous()
                lse, dA, _ = ops.nce_allrows(v2c, v1c, temperature, want_dV=False)
                n_of = torch.full((I,), float(bs), device=Pi.device)
                if I % bs:
                    n_of[(I // bs) * bs:] = float(I % bs)
                w = 1.0 / (n_of * k)
                loss = (w * (lse - (v1c * v2c).sum(-1) / temperature)).sum()
                g = (dA - v1c) * (w / temperature)[:, None]                # d loss / d v2
                G = (g - v2c * (v2c * g).sum(-1, keepdim=True)) / nrm      # through x / |x|
                ctx.save_for_backward(G)
                return loss
            loss = torch.zeros((), dtype=torch.float32, device=Pi.device)
            G = torch.empty_like(Pi)
            for b in range(0, I, bs):
                v2b = v2[b:b + bs]
                n = v2b.shape[0]
                pos = torch.exp((v1[b:b + n] * v2b).sum(-1) / temperature)
                E = torch.exp((v1 @ v2b.T) / temperature)                  # [I, n]
                ttl = E.sum(0)
                loss += (-torch.log(pos / ttl)).mean() / k
                g = ((E / ttl) .T @ v1 - v1[b:b + n]) / (temperature * n * k)       # d loss / d v2b
                G[b:b + n] = (g - v2b * (v2b * g).sum(-1, keepdim=True)) / nrm[b:b + n]   # through x / |x|
        ctx.save_for_backward(G)
        return loss

    @staticmethod
    def backward(ctx, g):
        G, = ctx.saved_tensors
        return g * G, None, None, None


class InfoAttack(CLeaR):
    def __init__(self, arg, data):
        super().__init__(arg, data)
        self.batchSize = 256

    def surrogate_loss(self, model, single_mask, topk, view1):
        """One evaluation of Loss = a*CW + b*Info (InfoAttack.py:62-105); returns (Loss, CW, Info)."""
        Pu, Pi = model()
        with torch.no_grad():
            top_idx, _ = masked_topk(Pu.detach(), Pi.detach(), single_mask, min(topk, self.itemNum))
        cw = _CwLoss.apply(Pu, Pi, top_idx, self.userNum, self.targetItem)
        info = _ItemInfoNCE.apply(Pi, view1, 0.2, self.batchSize)
        with torch.no_grad():
            tot = info + cw
            self.a, self.b = cw / tot, info / tot
        return self.a * cw + self.b * info, cw, info

    @staticmethod
    def single_element_mask(ui, device):
        """The mask the reference actually applies: element (indices[0], indices[1]) of the CSR column array (InfoAttack.py:69-70)."""
        m = sp.csr_matrix(ui)
        rp = np.zeros(m.shape[0] + 1, np.int32)
        cols = np.zeros(1, np.int32)
        if m.nnz >= 2 and m.indices[0] < m.shape[0]:
            rp[int(m.indices[0]) + 1:] = 1
            cols[0] = int(m.indices[1])
        return torch.from_numpy(rp).to(device), torch.from_numpy(cols).to(device)

    def posionDataAttack(self, recommender):
        with torch.no_grad():
            _, Pi0 = recommender.model()
            view1 = Pi0.detach().clone()
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
            mask = self.single_element_mask(uiAdj2, view1.device)
            for _ in range(self.outerEpoch):
                loss, _, _ = self.surrogate_loss(tmpRecommender.model, mask, topk, view1)
                print('loss:{}'.format(loss))
                optimizer_attack.zero_grad()
                loss.backward()
                optimizer_attack.step()
            with torch.no_grad():
                Pu, Pi = tmpRecommender.model()                              # fresh forward after the last step (InfoAttack.py:110)
                fake = torch.as_tensor(self.fakeUser, device=Pu.device)
                scores = (Pu[fake] @ Pi.T).contiguous()
            rows, _ = self.relaxProject(scores, self.maliciousFeedbackNum)
            rows[:, self.targetItem] = 1
            uiAdj2 = with_fake_rows(uiAdj2, self.userNum, rows.cpu().numpy())
            uiAdj = uiAdj2.copy()
            init_graph(recommender.model, uiAdj, Up, self.itemNum, n_real=self.userNum)
            recommender.train(Epoch=self.innerEpoch, optimizer=optimizer, evalNum=1)
            targetHitRate = AttackMetric(recommender, self.targetItem, [topk]).hitRate()[0]
            if targetHitRate > bestTargetHitRate:
                bestAdj = uiAdj.copy()
                bestTargetHitRate = targetHitRate
            uiAdj = bestAdj.copy()
            print('BiLevel epoch {} is over\n'.format(epoch + 1))
        self.interact = bestAdj
        return self.interact

    def relaxProject(self, mat, n):
        """({0,1} matrix [F, I] with n random picks among each row's top-2n (second draw), indices of the first draw)."""
        M = torch.as_tensor(mat.todense() if hasattr(mat, 'todense') else mat, dtype=torch.float32).to('cuda' if torch.cuda.is_available() else 'cpu')
        M = M.reshape(-1, M.shape[-1]).contiguous()
        n = int(n)
        if 2 * n > M.shape[1]:
            raise ValueError('relaxProject: 2*n exceeds the number of items')
        top2 = torch.topk(M, 2 * n, dim=1)[1].cpu()
        ind = torch.stack([top2[i, random.sample(list(range(2 * n)), n)] for i in range(M.shape[0])])
        picked = torch.stack([top2[i, random.sample(list(range(2 * n)), n)] for i in range(M.shape[0])])
        out = torch.zeros_like(M)
        out.scatter_(1, picked.to(M.device), 1.0)
        return out, ind.to(M.device)
