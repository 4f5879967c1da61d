"""Host-side data object with the reference's DataLoader interface (util/DataLoader.py:7-177): first-seen id maps,
dict-of-dict train/val/test sets, the (U+I)^2 bipartite adjacency, its D^-1/2 A D^-1/2 normalisation and the U x I
interaction matrix.  Same attribute and method names, built with vectorised numpy/scipy instead of Python loops.
This is glue around the hot path (text in, scipy out); the device-side graph is built from `norm_adj` by the models.
"""
from collections import defaultdict

import numpy as np
import scipy.sparse as sp

from .FileIO import FileIO


class DataLoader():
    # `training_data` is the reference's list of [user, item, rating] rows, shuffled in place by the sampler every epoch
    # (util/sampler.py:9).  Re-ordering 10^6..10^8 Python references per epoch costs more than the epoch's GPU work, so the
    # sampler (which shuffles its int image natively) only records the permutation here; the list is brought up to date the
    # moment anybody reads the attribute.
    @property
    def training_data(self):
        pend = self.__dict__.get('_td_pending')
        if pend is not None:
            td = self.__dict__['_td']
            td[:] = [td[k] for k in pend.tolist()]
            self.__dict__['_td_pending'] = None
        return self.__dict__['_td']

    @training_data.setter
    def training_data(self, rows):
        self.__dict__['_td'] = rows
        self.__dict__['_td_pending'] = None

    def _defer_td_permutation(self, perm):
        """new_list[j] = old_list[perm[j]], applied lazily (composes with a permutation that is still pending)."""
        pend = self.__dict__.get('_td_pending')
        self.__dict__['_td_pending'] = np.array(perm, dtype=np.int64) if pend is None else pend[perm]

    def append_training_rows(self, rows):
        """training_data.extend(rows) without forcing a pending permutation: the new rows go to the end either way."""
        td, pend = self.__dict__['_td'], self.__dict__.get('_td_pending')
        n = len(td)
        td.extend(rows)
        if pend is not None:
            self.__dict__['_td_pending'] = np.concatenate([pend, np.arange(n, len(td), dtype=np.int64)])

    def _raw_training_data(self):
        """(list in its last materialised order, pending permutation or None) -- for the sampler's own bookkeeping."""
        return self.__dict__['_td'], self.__dict__.get('_td_pending')

    def __init__(self, args=None, training_data=None, val_data=None, test_data=None, dataName=None):
        if args is not None:
            base = args.data_path + args.dataset
            training_data = FileIO.load_data_set(base + args.training_data)
            val_data = FileIO.load_data_set(base + args.val_data)
            test_data = FileIO.load_data_set(base + args.test_data)
            dataName = args.dataset
        self.training_data = training_data
        self.val_data = val_data if val_data is not None else []
        self.test_data = test_data if test_data is not None else []
        self.dataName = dataName
        self.user, self.item, self.id2user, self.id2item = {}, {}, {}, {}
        self.training_set_u, self.training_set_i = defaultdict(dict), defaultdict(dict)
        self.val_set, self.test_set = defaultdict(dict), defaultdict(dict)
        self.val_set_item, self.test_set_item = set(), set()
        for user, item, rating in self.training_data:           # util/DataLoader.py:33-43 first-seen ids
            if user not in self.user:
                self.user[user] = len(self.user)
                self.id2user[self.user[user]] = user
            if item not in self.item:
                self.item[item] = len(self.item)
                self.id2item[self.item[item]] = item
            self.training_set_u[user][item] = rating
            self.training_set_i[item][user] = rating
        for src, dst, seen in ((self.val_data, self.val_set, self.val_set_item), (self.test_data, self.test_set, self.test_set_item)):
            for user, item, rating in src:                      # util/DataLoader.py:44-55 (users unseen in training are skipped)
                if user in self.user:
                    dst[user][item] = rating
                    seen.add(item)
        self.user_num = len(self.training_set_u)
        self.item_num = len(self.training_set_i)
        self.ui_adj = self._bipartite_adjacency()
        self.norm_adj = self.normalize_graph_mat(self.ui_adj)
        self.interaction_mat = self._interaction_matrix()

    # ---- matrices
    def _ids(self):
        """(user ids, item ids) of training_data in its current order.  Served from the sampler's int image (kept in step with
        the list by util/sampler.py, extended on appends) once it exists; the O(nnz) Python mapping otherwise."""
        if getattr(self, '_arl_sampler', None) is not None:
            from .sampler import _shadow
            p = _shadow(self).pairs
            return p[:, 0].astype(np.int64), p[:, 1].astype(np.int64)
        u = np.fromiter((self.user[r[0]] for r in self.training_data), dtype=np.int64, count=len(self.training_data))
        i = np.fromiter((self.item[r[1]] for r in self.training_data), dtype=np.int64, count=len(self.training_data))
        return u, i

    def _bipartite_adjacency(self, self_connection=False):
        n = self.user_num + self.item_num
        u, i = self._ids()
        half = sp.csr_matrix((np.ones(len(u), np.float32), (u, i + self.user_num)), shape=(n, n), dtype=np.float32)
        adj = half + half.T
        if self_connection:
            adj += sp.eye(n)
        return adj

    def normalize_graph_mat(self, adj_mat):
        shape = adj_mat.get_shape()
        rowsum = np.array(adj_mat.sum(1))
        with np.errstate(divide='ignore'):
            d_inv = np.power(rowsum, -0.5 if shape[0] == shape[1] else -1.0).flatten()
        d_inv[np.isinf(d_inv)] = 0.
        d_mat = sp.diags(d_inv)
        out = d_mat.dot(adj_mat)
        return out.dot(d_mat) if shape[0] == shape[1] else out

    def convert_to_laplacian_mat(self, adj_mat):
        r, c = adj_mat.get_shape()
        rows, cols = adj_mat.nonzero()
        half = sp.csr_matrix((adj_mat.data, (rows, cols + r)), shape=(r + c, r + c), dtype=np.float32)
        return self.normalize_graph_mat(half + half.T)

    def _interaction_matrix(self):
        u, i = self._ids()
        return sp.csr_matrix((np.ones(len(u), np.float64), (u, i)), shape=(self.user_num, self.item_num), dtype=np.float32)

    def matrix(self):
        return self._interaction_matrix()

    # ---- accessors
    def get_user_id(self, u):
        return self.user.get(u)

    def get_item_id(self, i):
        return self.item.get(i)

    def training_size(self):
        return len(self.user), len(self.item), len(self.__dict__['_td'])

    def val_size(self):
        return len(self.val_set), len(self.val_set_item), len(self.val_data)

    def test_size(self):
        return len(self.test_set), len(self.test_set_item), len(self.test_data)

    def contain(self, u, i):
        return u in self.user and i in self.training_set_u[u]

    def contain_user(self, u):
        return u in self.user

    def contain_item(self, i):
        return i in self.item

    def user_rated(self, u):
        return list(self.training_set_u[u].keys()), list(self.training_set_u[u].values())

    def item_rated(self, i):
        return list(self.training_set_i[i].keys()), list(self.training_set_i[i].values())

    def row(self, u):
        vec = np.zeros(len(self.item))
        for it, r in self.training_set_u[self.id2user[u]].items():
            vec[self.item[it]] = r
        return vec

    def col(self, i):
        vec = np.zeros(len(self.user))
        for us, r in self.training_set_i[self.id2item[i]].items():
            vec[self.user[us]] = r
        return vec

    # ---- array views used by the device path (no counterpart in the reference) ------------------------------------
    @classmethod
    def from_arrays(cls, train, val=None, test=None, dataName=None):
        """Build from integer/float triples (u, i, r) arrays instead of text files; raw ids are stringified exactly like
        FileIO.load_data_set would read them, so id assignment (first-seen order) is identical to a file round trip."""
        def rows(t):
            if t is None:
                return []
            u, i, r = t
            return [[str(a), str(b), float(c)] for a, b, c in zip(np.asarray(u).tolist(), np.asarray(i).tolist(), np.asarray(r).tolist())]
        return cls(training_data=rows(train), val_data=rows(val), test_data=rows(test), dataName=dataName)

    def pairs_array(self):
        """int32 [nnz, 2] (user id, item id) in the CURRENT order of training_data (the sampler shuffles it in place)."""
        u, i = self._ids()
        return np.stack([u, i], 1).astype(np.int32)

    def membership_csr(self):
        """training_set_u as (rowptr int64 [U0+1], sorted item ids int32): the negative sampler's rejection set.  Users added
        after construction (fake users) are not in training_set_u and therefore reject nothing, as in the reference."""
        users = [u for u in self.training_set_u if len(self.training_set_u[u])]
        n_rows = (max(self.user[u] for u in users) + 1) if users else 0
        counts = np.zeros(n_rows + 1, np.int64)
        chunks = [None] * n_rows
        for u in users:
            ids = np.fromiter((self.item[i] for i in self.training_set_u[u]), dtype=np.int32)
            ids.sort()
            chunks[self.user[u]] = ids
            counts[self.user[u] + 1] = len(ids)
        items = np.concatenate([c for c in chunks if c is not None]) if users else np.zeros(1, np.int32)
        return np.cumsum(counts), items

    def device_graph(self, device='cuda'):
        """The normalised adjacency as an arlib_amd.ops.CSRGraph on `device`."""
        from .. import ops
        m = sp.csr_matrix(self.norm_adj, dtype=np.float32)
        m.sort_indices()
        return ops.CSRGraph(m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float32), device)

    # pickling: drop the sampler's cached int image (rebuilt on demand)
    def __getstate__(self):
        self.training_data                                       # materialise a pending permutation first
        st = dict(self.__dict__)
        st.pop('_arl_sampler', None)
        return st

    def __deepcopy__(self, memo):
        """copy.deepcopy(recommender) is how the attacks fork a surrogate (attack/White/CLeaR.py:66, DLAttack.py:62).  A generic
        deep copy walks ~8 Python objects per interaction (12 s at 1.6 M interactions); nothing mutates the rows or the inner
        dicts in place (the sampler permutes the LIST, attacks append to it and add keys to the id maps), so the copy owns
        its containers and shares the leaves."""
        import copy
        new = object.__new__(type(self))
        memo[id(self)] = new
        shared_leaves = ('training_set_u', 'training_set_i', 'val_set', 'test_set')
        for k, v in self.__dict__.items():
            if k == '_td':
                new.__dict__['_td'] = list(v)                    # a pending permutation stays pending on both sides
            elif k == '_td_pending':
                new.__dict__['_td_pending'] = None if v is None else v.copy()
            elif k in ('val_data', 'test_data'):
                setattr(new, k, v)                               # never written after construction
            elif k in shared_leaves:
                c = type(v)(v.default_factory) if isinstance(v, defaultdict) else type(v)()
                c.update(v)
                setattr(new, k, c)
            elif k in ('user', 'item', 'id2user', 'id2item'):
                setattr(new, k, dict(v))
            elif k in ('val_set_item', 'test_set_item'):
                setattr(new, k, set(v))
            elif k == '_arl_sampler':
                from .sampler import PairSampler
                new._arl_sampler = PairSampler(v.pairs.copy(), v.n_items, (v.memb_rowptr, v.memb_items))
            elif k == '_arl_memb':
                new._arl_memb = v
            else:
                setattr(new, k, copy.deepcopy(v, memo))
        return new
