"""DLAttack -- mirror of the reference's attack/White/DLAttack.py (posionDataAttack :51-125, project :127-132,
fakeUserInject :134-163) on the MI355X kernels: one fake user at a time, surrogate fine-tuning with BPR, filler items =
top-n of (score * decaying popularity prior p).

What changed mechanically: the U x I score matrix the reference materialises on the HOST every outer epoch
(DLAttack.py:73-83) is replaced by the streaming score+mask+top-k kernel; the surrogate's BPR steps run on the fused
training engine; the `Pu @ Pi.T` recomputed inside every mini-batch only to feed a constant into the printed loss
(DLAttack.py:102-103; it carries no gradient) is not recomputed.
"""
from copy import deepcopy

import numpy as np
import scipy.sparse as sp
import torch

from ... import ops
from ...util.sampler import next_batch_pairwise, device_epoch
from .._common import AttackBase, DEVICE, symmetric_adjacency, init_graph, rebuild_interaction_matrix, reinit_with_tables, cw_pairs, with_fake_rows, append_rows


def device_mask(ui_mat, device=DEVICE):
    """CSR image (rowptr int32, sorted item ids int32) on the device of the nonzero pattern of a U x I interaction matrix:
    the `interacted -> -10e8` mask of DLAttack.py:76-80 / CLeaR.py:78-80.  Built once per outer loop -- the pattern does
    not change between the surrogate's steps."""
    m = sp.csr_matrix(ui_mat)
    m.eliminate_zeros(); m.sort_indices()
    rp = torch.from_numpy(m.indptr.astype(np.int32)).to(device)
    mc = torch.from_numpy(m.indices.astype(np.int32) if m.nnz else np.zeros(1, np.int32)).to(device)
    return rp, mc


def masked_topk(Pu, Pi, mask, k, warm=None):
    """top-k of Pu @ Pi.T with interacted entries set to -10e8 (DLAttack.py:73-83 / CLeaR.py:75-82), streamed.
    `mask` is a device_mask() pair or anything scipy can turn into a U x I sparse matrix; `warm` = the previous step's lists
    for the same users and mask (the surrogate moved a little since): result-neutral, about a third faster."""
    rp, mc = mask if isinstance(mask, tuple) else device_mask(mask, Pu.device)
    if warm is not None and tuple(warm.shape) != (Pu.shape[0], k):
        warm = None
    return ops.score_mask_topk(Pu.contiguous(), Pi.contiguous(), k, rp, mc, warm_idx=warm)


class DLAttack(AttackBase):
    def __init__(self, arg, data):
        super().__init__(arg, data)
        self.batchSize = 256

    def posionDataAttack(self, recommender):
        self.fakeUser = list(range(self.userNum, self.userNum + self.fakeUserNum))
        # bound to the model that fakeUserInject() is about to replace: the surrogate "retrain" below moves nothing (quirk Q4)
        optimizer = torch.optim.Adam(recommender.model.parameters(), lr=recommender.args.lRate / 10)
        topk = min(recommender.topN)
        p = torch.ones(self.itemNum, device=DEVICE)
        sigma = 0.8
        uiAdj = None
        for user in self.fakeUser:
            self.fakeUserInject(recommender, user)
            uiAdj = recommender.data.matrix()           # rebuilt from training_data: earlier fake users keep only their targets (quirk Q6)
            tmpRecommender = deepcopy(recommender)
            uiAdj2 = sp.csr_matrix(uiAdj, copy=True)
            U_now = tmpRecommender.data.user_num
            init_graph(tmpRecommender.model, uiAdj2, U_now, self.itemNum, n_real=self.userNum)
            tmpRecommender.train(Epoch=self.innerEpoch, optimizer=optimizer, evalNum=5)
            optimizer_attack = torch.optim.Adam(tmpRecommender.model.parameters(), lr=recommender.args.lRate)
            mask = device_mask(uiAdj2)
            top_idx = None
            for _ in range(self.outerEpoch):
                with torch.no_grad():
                    Pu, Pi = tmpRecommender.model()
                    top_idx, _ = masked_topk(Pu, Pi, mask, min(topk, self.itemNum), warm=top_idx)
                    users, pos, neg = cw_pairs(top_idx, self.userNum, self.targetItem, pop=False)
                    # CW term of DLAttack.py:92-101: computed on detached tensors there, i.e. a logged constant
                    self.last_cw_loss = float(((Pu[users] * Pi[neg]).sum(1) - (Pu[users] * Pi[pos]).sum(1)).mean())
                tmpRecommender.train_batches(device_epoch(self.data, tmpRecommender.args.batch_size, DEVICE, tmpRecommender.data.user_num, self.itemNum), optimizer_attack)
            with torch.no_grad():
                Pu, Pi = tmpRecommender.model()
                r = (Pu[user, :] @ Pi.T) * p
            m, ind = self.project(r, self.maliciousFeedbackNum)
            uiAdj2 = with_fake_rows(uiAdj2, user, m.cpu().numpy().reshape(1, -1))     # `user` is the row just appended
            p[ind] = p[ind] * sigma
            if p.max() < 1:
                p = torch.ones(self.itemNum, device=DEVICE)
            init_graph(recommender.model, uiAdj2, recommender.data.user_num, self.itemNum, n_real=self.userNum)
            uiAdj = uiAdj2
        self.interact = uiAdj
        return self.interact

    def project(self, mat, n):
        """top-n of a score vector -> ({0,1} vector, indices) (DLAttack.py:127-132)."""
        v = torch.as_tensor(mat, dtype=torch.float32, device=DEVICE).reshape(1, -1).contiguous()
        out, idx = ops.topn_project_rows(v, int(n))
        return out[0], idx[0].long()

    def fakeUserInject(self, recommender, user):
        """One more user whose only interactions are the targets; re-init the recommender and keep the old tables
        (DLAttack.py:134-163)."""
        Pu, Pi = recommender.model()
        data = recommender.data
        data.user_num += 1
        data.user['fakeuser{}'.format(data.user_num)] = len(data.user)
        data.id2user[len(data.user) - 1] = 'fakeuser{}'.format(data.user_num)
        append_rows(data, [(data.id2user[user], data.id2item[i]) for i in self.targetItem])
        _, _, data.interaction_mat = rebuild_interaction_matrix(data)
        reinit_with_tables(recommender, Pu, Pi)
