"""Ranking metrics with the reference's interface and output format (util/metrics.py:87-114 ranking_evaluation,
:125-208 AttackMetric).  Host-side bookkeeping around the top-k lists the GPU produces; AttackMetric replaces the
reference's per-user predict()+argsort loop by one streaming score+top-k kernel launch over all users.
"""
import math

import numpy as np


class RecommendMetric(object):
    @staticmethod
    def hits(origin, res):
        return {u: len(set(origin[u]).intersection(i[0] for i in res[u])) for u in origin}

    @staticmethod
    def hit_ratio(origin, hits):
        total = sum(len(origin[u]) for u in origin)
        return sum(hits.values()) / total

    @staticmethod
    def precision(hits, N):
        return sum(hits.values()) / (len(hits) * N)

    @staticmethod
    def recall(hits, origin):
        vals = [hits[u] / len(origin[u]) for u in hits]
        return sum(vals) / len(vals)

    @staticmethod
    def F1(prec, recall):
        return 2 * prec * recall / (prec + recall) if (prec + recall) != 0 else 0

    @staticmethod
    def NDCG(origin, res, N):
        total = 0
        for user in res:
            dcg = sum(1.0 / math.log(n + 2) for n, item in enumerate(res[user]) if item[0] in origin[user])
            idcg = sum(1.0 / math.log(n + 2) for n in range(min(len(origin[user]), N)))
            total += dcg / idcg
        return total / len(res)


def ranking_evaluation(origin, res, N):
    measure = []
    for n in N:
        predicted = {user: res[user][:n] for user in res}
        if len(origin) != len(predicted):
            print('The Lengths of test set and predicted set do not match!')
            exit(-1)
        hits = RecommendMetric.hits(origin, predicted)
        measure.append('Top ' + str(n) + '\n')
        measure.append('Hit Ratio:' + str(RecommendMetric.hit_ratio(origin, hits)) + '\n')
        measure.append('Precision:' + str(RecommendMetric.precision(hits, n)) + '\n')
        measure.append('Recall:' + str(RecommendMetric.recall(hits, origin)) + '\n')
        measure.append('NDCG:' + str(RecommendMetric.NDCG(origin, predicted, n)) + '\n')
    return measure


def test_set_image(data):
    """Per test user (in test_set order): how many test items it has, and the (row, internal item id) keys of those the model
    knows, sorted -- built once per data object (the test split never changes)."""
    img = getattr(data, '_arl_test_image', None)
    if img is None or img[0] != len(data.test_set):
        users = list(data.test_set)
        lens = np.array([len(data.test_set[u]) for u in users], np.int64)
        I = len(data.item)
        keys = np.array(sorted(r * I + data.item[it] for r, u in enumerate(users) for it in data.test_set[u] if it in data.item), np.int64)
        img = (len(users), users, lens, keys)
        data._arl_test_image = img
    return img


def ranking_evaluation_topk(data, idx, N):
    """The measure lines of ranking_evaluation(data.test_set, rec_list, N) computed from the top-k index array idx
    [test users in test_set order, >= max(N)] without building rec_list.  Same arithmetic in the same order (integer hit
    counts; rank-ordered DCG terms; the user-ordered Python float accumulations), hence the same strings."""
    _, users, lens, keys = test_set_image(data)
    I = len(data.item)
    idx = np.asarray(idx, np.int64)
    pk = np.arange(idx.shape[0], dtype=np.int64)[:, None] * I + idx
    pos = np.searchsorted(keys, pk)
    H = keys[np.minimum(pos, max(len(keys) - 1, 0))] == pk if len(keys) else np.zeros(pk.shape, bool)
    total = int(lens.sum())
    measure = []
    for n in N:
        Hn = H[:, :n]
        hits = Hn.sum(1).astype(np.int64)
        s_hits = int(hits.sum())
        w = [1.0 / math.log(r + 2) for r in range(n)]
        dcg = np.zeros(idx.shape[0], np.float64)
        for r in range(min(n, Hn.shape[1])):
            dcg = dcg + np.where(Hn[:, r], w[r], 0.0)
        pref = [0]
        for r in range(n):
            pref.append(pref[-1] + w[r])
        idcg = np.array([pref[m] for m in np.minimum(lens, n).tolist()], np.float64)
        ndcg_total = 0
        for v in (dcg / idcg).tolist():
            ndcg_total += v
        measure.append('Top ' + str(n) + '\n')
        measure.append('Hit Ratio:' + str(s_hits / total) + '\n')
        measure.append('Precision:' + str(s_hits / (len(hits) * n)) + '\n')
        measure.append('Recall:' + str(sum((hits / lens).tolist()) / len(hits)) + '\n')
        measure.append('NDCG:' + str(ndcg_total / len(hits)) + '\n')
    return measure


class AttackMetric(object):
    """targetItem: internal item ids.  Rankings are over ALL items without masking interacted ones, as in the
    reference (np.argsort(-score)[:k], util/metrics.py:141)."""

    def __init__(self, recommendModel, targetItem, top=[10]):
        self.recommendModel = recommendModel
        self.targetItem = targetItem
        self.top = top
        self._rank = None

    def _top(self):
        if self._rank is None:
            import torch
            from .. import ops
            rm = self.recommendModel
            uid = torch.tensor([rm.data.user[u] for u in rm.data.user], dtype=torch.long, device=rm.user_emb.device)
            Pu, Pi = rm.user_emb[uid].contiguous(), rm.item_emb.contiguous()
            k = min(max(self.top), Pi.shape[0])
            if Pu.shape[1] % 4 == 0 and k <= 128:
                idx, _ = ops.score_mask_topk(Pu, Pi, k)
            else:
                idx = torch.topk(Pu @ Pi.T, k)[1]
            self._rank = idx.cpu().numpy()
        return self._rank

    def _hits(self):
        """hit[u, r] = True when the r-th ranked item of user u is a target."""
        return np.isin(self._top(), np.asarray(self.targetItem))

    def precision(self):
        h = self._hits()
        return [float(h[:, :k].sum() / (h.shape[0] * k)) for k in self.top]

    def hitRate(self):
        h = self._hits()
        return [float((h[:, :k].any(1) / len(self.targetItem)).sum() / h.shape[0]) for k in self.top]

    def recall(self):
        h = self._hits()
        return [float(h[:, :k].sum() / (h.shape[0] * len(self.targetItem))) for k in self.top]

    def NDCG(self):
        h = self._hits()
        out = []
        for k in self.top:
            disc = 1.0 / np.log2(2 + np.arange(k))
            idcg = disc[:min(k, len(self.targetItem))].sum()
            out.append(float((h[:, :k] * disc).sum() / (idcg * h.shape[0])))
        return out
