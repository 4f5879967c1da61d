"""Shared pieces of the white-box attack mirrors (the reference repeats them in every attack/White/*.py)."""
import numpy as np
import scipy.sparse as sp
import torch

from ..util.tool import targetItemSelect

DEVICE = 'cuda'


class AttackBase:
    """Constructor contract of every reference attack (e.g. attack/White/PGA.py:17-52): attributes `targetItem`
    (internal ids), `recommenderGradientRequired`, `recommenderModelRequired`, fake-user / filler budgets."""
    recommenderGradientRequired = False
    recommenderModelRequired = True

    def __init__(self, arg, data):
        self.data = data
        self.interact = data.matrix()
        self.userNum, self.itemNum = self.interact.shape
        self.targetItem = [data.item[i.strip()] for i in targetItemSelect(data, arg)]
        self.Epoch, self.innerEpoch, self.outerEpoch = arg.Epoch, arg.innerEpoch, arg.outerEpoch
        self.maliciousUserSize = arg.maliciousUserSize
        self.maliciousFeedbackSize = arg.maliciousFeedbackSize
        if self.maliciousFeedbackSize == 0:
            self.maliciousFeedbackNum = int(self.interact.sum() / data.user_num)
        elif self.maliciousFeedbackSize >= 1:
            self.maliciousFeedbackNum = self.maliciousFeedbackSize
        else:
            # the reference reads a non-existent self.item_num here and raises AttributeError (quirk Q3); we use itemNum
            self.maliciousFeedbackNum = int(self.maliciousFeedbackSize * self.itemNum)
        self.fakeUserNum = int(data.user_num * self.maliciousUserSize) if self.maliciousUserSize < 1 else int(self.maliciousUserSize)


def symmetric_adjacency(ui, n_users, n_items):
    """(U+I)^2 matrix [[0, R], [R^T, 0]] from a U x I (weighted) interaction matrix -- what the reference builds with
    `ui_adj[:U, U:] = uiAdj; ui_adj + ui_adj.T` (attack/White/PGA.py:79-83) without the O(nnz) lil assignment."""
    R = sp.csr_matrix(ui, dtype=np.float32)
    R.eliminate_zeros()
    return sp.bmat([[None, R], [R.T, None]], format='csr', dtype=np.float32) if R.shape == (n_users, n_items) else None


def rebuild_interaction_matrix(data):
    """interaction_mat from training_data with the current id maps (attack/White/PGA.py:185-192, CLeaR.py:192-199)."""
    if hasattr(data, '_ids'):
        u, i = data._ids()                                    # the sampler's int image when there is one (appends cost only the tail)
    else:
        u = np.fromiter((data.user[p[0]] for p in data.training_data), dtype=np.int64, count=len(data.training_data))
        i = np.fromiter((data.item[p[1]] for p in data.training_data), dtype=np.int64, count=len(data.training_data))
    return u, i, sp.csr_matrix((np.ones(len(u), np.float64), (u, i)), shape=(data.user_num, data.item_num), dtype=np.float32)


def reinit_with_tables(recommender, Pu, Pi):
    """recommender.__init__(args, data) then copy the old tables into the first rows (attack/White/PGA.py:61-65)."""
    recommender.__init__(recommender.args, recommender.data)
    with torch.no_grad():
        recommender.model.embedding_dict['user_emb'][:Pu.shape[0]] = Pu.detach().to(recommender.model.embedding_dict['user_emb'].device)
        recommender.model.embedding_dict['item_emb'][:] = Pi.detach().to(recommender.model.embedding_dict['item_emb'].device)
    recommender.model = recommender.model.cuda()


def cw_pairs(top_idx, n_real_users, targets, pop=True):
    """(users, pos_items, neg_items) of the CW loss: for every real user and every target, the negative is taken from the
    tail of the user's top-k list -- successive `.pop()`s in PGA/CLeaR (ranks k, k-1, ...; PGA.py:104-108, CLeaR.py:84-88),
    always the k-th entry in DLAttack (`top_items[u][-1]`, DLAttack.py:92-96)."""
    T = len(targets)
    k = top_idx.shape[1]
    users = torch.arange(n_real_users, device=top_idx.device).repeat_interleave(T)
    pos = torch.as_tensor(targets, device=top_idx.device, dtype=torch.long).repeat(n_real_users)
    if pop:
        ranks = (k - 1 - torch.arange(T, device=top_idx.device)).repeat(n_real_users)
    else:
        ranks = torch.full((n_real_users * T,), k - 1, device=top_idx.device, dtype=torch.long)
    neg = top_idx[users, ranks].long()
    return users, pos, neg


def with_fake_rows(ui, first_fake_row, block):
    """`uiAdj2[fake rows, :] = block` for fake users occupying the LAST rows (attack/White/CLeaR.py:130-135, DLAttack.py:118):
    a CSR vstack instead of a lil row assignment (lil construction/copies are O(nnz) Python objects)."""
    ui = sp.csr_matrix(ui)
    block = sp.csr_matrix(np.asarray(block, dtype=np.float32))
    if first_fake_row + block.shape[0] != ui.shape[0] or block.shape[1] != ui.shape[1]:
        raise ValueError('with_fake_rows: the block must cover the last rows of the matrix')
    return sp.vstack([ui[:first_fake_row], block], format='csr', dtype=np.float32)


def append_rows(data, rows):
    """data.training_data.append(row) for each row (attack/White/CLeaR.py:190-191, DLAttack.py:146-147)."""
    if hasattr(data, 'append_training_rows'):
        data.append_training_rows(rows)              # our DataLoader: keeps a pending sampler permutation pending
    else:
        data.training_data.extend(rows)


def init_graph(model, ui, n_users, n_items, n_real=None):
    """model._init_uiAdj(ui_adj + ui_adj.T) for the U' x I interaction matrix `ui` -- on the device when the model offers it
    (our encoders), through the (U'+I)^2 scipy matrix otherwise (any model with the reference's interface).  `n_real` = number of
    real users (the leading rows, unchanged over an attack's epochs): our encoders then only merge the fake users' rows into the
    device graph they already hold (ops.IncrementalBipartite)."""
    if hasattr(model, '_init_uiAdj_from_interactions') and ui.shape == (n_users, n_items):
        model._init_uiAdj_from_interactions(ui, n_real=n_real)
    else:
        model._init_uiAdj(symmetric_adjacency(ui, n_users, n_items))
