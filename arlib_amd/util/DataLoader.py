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
    ARRAY_NATIVE_MIN_NNZ = 2_000_000

    @classmethod
    def from_arrays(cls, train, val=None, test=None, dataName=None, array_native=None):
        """Build from integer/float triples (u, i, r) arrays instead of text files; raw ids are stringified exactly like
        FileIO.load_data_set would read them, so id assignment (first-seen order) is identical to a file round trip.
        array_native (default: from ARRAY_NATIVE_MIN_NNZ interactions on) returns an ArrayDataLoader: the same object surface
        over numpy arrays, its Python containers materialised only when somebody reads them."""
        if array_native is None:
            array_native = len(np.asarray(train[0])) >= cls.ARRAY_NATIVE_MIN_NNZ
        if array_native and cls is DataLoader:
            return ArrayDataLoader(train, val, test, dataName)

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


class _RowDicts:
    """Read-only mapping  raw id -> {raw id: rating}  over a CSR (the reference's training_set_u / training_set_i dict-of-dicts,
    util/DataLoader.py:41-42), rows built on demand.  Iteration order = first-seen order of the keys, as a dict filled in file order."""

    def __init__(self, key_names, key_ids, rowptr, cols, vals, col_names):
        self._names, self._ids, self._rp, self._cols, self._vals, self._colnames = key_names, key_ids, rowptr, cols, vals, col_names

    def __len__(self):
        return len(self._names)

    def __iter__(self):
        return iter(self._names)

    def __contains__(self, k):
        return k in self._ids and self._ids[k] < len(self._rp) - 1

    def keys(self):
        return list(self._names)

    def __getitem__(self, k):
        r = self._ids.get(k)
        if r is None or r >= len(self._rp) - 1:
            return {}                                              # defaultdict(dict) semantics for an unknown key (read side)
        b, e = int(self._rp[r]), int(self._rp[r + 1])
        cn = self._colnames
        return {cn[c]: v for c, v in zip(self._cols[b:e].tolist(), self._vals[b:e].tolist())}

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return ((k, self[k]) for k in self._names)


class LazyNormAdj:
    """data.norm_adj of an ArrayDataLoader: the normalised (U+I)^2 adjacency (util/DataLoader.py:57-87) that is only assembled
    where it is consumed -- on the device by the encoders (`device_graph`), as scipy for anybody who asks (`to_scipy`)."""

    def __init__(self, data):
        self._data = data
        n = data.user_num + data.item_num
        self.shape = (n, n)

    def get_shape(self):
        return self.shape

    def device_graph(self, device='cuda'):
        import torch
        from .. import ops
        u, i = self._data._sorted_pairs()
        return ops.bipartite_graph(torch.from_numpy(u).to(device), torch.from_numpy(i).to(device), self._data.user_num, self._data.item_num, device=device)

    def to_scipy(self):
        return self._data.normalize_graph_mat(self._data.ui_adj)

    def tocsr(self):
        return self.to_scipy().tocsr()


class ArrayDataLoader(DataLoader):
    """DataLoader with the same attributes and methods, built from (u, i, r) arrays in O(nnz) numpy: at 3.2e7 interactions the
    reference's list of rows and dict-of-dict sets (util/DataLoader.py:8-55) are ~10^8 Python objects.  The int32 pair array is the
    primary image (`pair_sampler`, what next_batch_pairwise shuffles in place of the list); `training_data`, `training_set_u/i`,
    `ui_adj`, `norm_adj`, `interaction_mat` are materialised when read.  Ids are assigned in first-seen order, raw ids stringified,
    exactly as a file round trip would (tested against the list-based loader)."""

    def __init__(self, train, val=None, test=None, dataName=None):
        from .sampler import PairSampler, build_membership
        tu, ti, tr = (np.asarray(a) for a in train)
        self.dataName = dataName

        def first_seen(raw):
            uniq, first, inv = np.unique(raw, return_index=True, return_inverse=True)
            order = np.argsort(first, kind='stable')              # unique values in first-seen order
            rank = np.empty(len(uniq), np.int64); rank[order] = np.arange(len(uniq))
            return uniq[order], rank[inv]
        raw_u, uid = first_seen(tu)
        raw_i, iid = first_seen(ti)
        name_u, name_i = [str(x) for x in raw_u.tolist()], [str(x) for x in raw_i.tolist()]
        self.user, self.item = dict(zip(name_u, range(len(name_u)))), dict(zip(name_i, range(len(name_i))))
        self.id2user, self.id2item = dict(enumerate(name_u)), dict(enumerate(name_i))
        self.user_num, self.item_num = len(name_u), len(name_i)
        pairs = np.stack([uid, iid], 1).astype(np.int32)
        self._rating = np.asarray(tr, np.float64).copy()           # in the CURRENT order of the pairs (permuted with them)
        self._uniform_rating = bool(len(self._rating) == 0 or np.all(self._rating == self._rating[0]))
        memb = build_membership(pairs, self.user_num)
        self._arl_memb = memb
        self.pair_sampler = PairSampler(pairs, self.item_num, memb)
        # training_set_u / training_set_i as CSR (last rating wins for a repeated pair, as dict assignment does)
        key = uid * self.item_num + iid
        _, last = np.unique(key[::-1], return_index=True)
        keep = np.sort(len(key) - 1 - last)
        ku, ki, kr = uid[keep], iid[keep], self._rating[keep]

        def csr(rows, cols, n_rows):
            o = np.argsort(rows, kind='stable')                   # columns stay in first-seen (file) order inside a row
            rp = np.zeros(n_rows + 1, np.int64); np.cumsum(np.bincount(rows, minlength=n_rows), out=rp[1:])
            return rp, cols[o].astype(np.int32), kr[o]
        self.training_set_u = _RowDicts(name_u, self.user, *csr(ku, ki, self.user_num), name_i)
        self.training_set_i = _RowDicts(name_i, self.item, *csr(ki, ku, self.item_num), name_u)
        self.val_data, self.test_data = [], []
        self.val_set, self.val_set_item = self._held_out(val)
        self.test_set, self.test_set_item = self._held_out(test)
        self._ui_adj = self._norm_adj = self._interaction_mat = None

    def _held_out(self, t):
        """val/test split as the reference's dict-of-dicts (users unseen in training are skipped, util/DataLoader.py:44-55)."""
        out, seen = defaultdict(dict), set()
        if t is None:
            return out, seen
        u, i, r = (np.asarray(a).tolist() for a in t)
        for a, b, c in zip(u, i, r):
            a, b = str(a), str(b)
            if a in self.user:
                out[a][b] = float(c)
                seen.add(b)
        return out, seen

    # ---- the list view of the reference (materialised on demand; the pair array stays the primary image)
    @property
    def training_data(self):
        p = self.pair_sampler.pairs
        iu, ii = self.id2user, self.id2item
        return [[iu[a], ii[b], c] for (a, b), c in zip(p.tolist(), self._rating.tolist())]

    @training_data.setter
    def training_data(self, rows):
        raise AttributeError('ArrayDataLoader: training_data is a view of the pair array; use append_training_rows()')

    def append_training_rows(self, rows):
        """training_data.extend(rows): O(len(rows)) -- the pair / rating arrays live in buffers that grow geometrically (an attack appends
        once per fake user: thousands of calls on a 3e7-row array)."""
        from .sampler import PairSampler
        if not len(rows):
            return
        tail = np.array([[self.user[r[0]], self.item[r[1]]] for r in rows], np.int32).reshape(-1, 2)
        if tail.min() < 0 or tail[:, 1].max() >= len(self.item):
            raise ValueError('append_training_rows: ids out of range')
        rt = np.array([float(r[2]) if len(r) > 2 else 1.0 for r in rows], np.float64)
        sh = self.pair_sampler
        n, m = sh.nnz, len(rows)
        buf = self.__dict__.get('_pairs_buf')
        if buf is None or buf.shape[0] < n + m or sh.pairs.base is not buf:
            cap = max(n + m, int(1.25 * n) + 1024)
            buf = np.empty((cap, 2), np.int32); buf[:n] = sh.pairs
            rbuf = np.empty(cap, np.float64); rbuf[:n] = self._rating
            self.__dict__['_pairs_buf'], self.__dict__['_rating_buf'] = buf, rbuf
        rbuf = self.__dict__['_rating_buf']
        buf[n:n + m] = tail; rbuf[n:n + m] = rt
        ns = PairSampler.__new__(PairSampler)                         # same membership sets (fixed at construction, util/DataLoader.py:41), longer pair list
        ns.pairs, ns.n_items, ns.memb_rowptr, ns.memb_items = buf[:n + m], len(self.item), sh.memb_rowptr, sh.memb_items
        self.pair_sampler = ns
        self._rating = rbuf[:n + m]
        self._uniform_rating = self._uniform_rating and bool(np.all(rt == self._rating[0]))
        self._ui_adj = self._norm_adj = self._interaction_mat = None

    def _permute_ratings(self, order):
        """Called by the sampler after an epoch shuffle with the permutation it applied to the pairs (skipped while all ratings are equal)."""
        self._rating = self._rating[order]

    def _ids(self):
        p = self.pair_sampler.pairs
        return p[:, 0].astype(np.int64), p[:, 1].astype(np.int64)

    def _sorted_pairs(self):
        """Unique (user, item) pairs sorted user-major (int64 arrays): the pattern of the interaction matrix."""
        u, i = self._ids()
        key = np.unique(u * self.item_num + i)
        return key // self.item_num, key % self.item_num

    def training_size(self):
        return len(self.user), len(self.item), self.pair_sampler.nnz

    def membership_csr(self):
        return self._arl_memb

    def contain(self, u, i):
        return u in self.user and i in self.training_set_u[u]

    # ---- matrices, built when read (attacks assign ui_adj / norm_adj / interaction_mat: plain setters)
    @property
    def ui_adj(self):
        if self._ui_adj is None:
            self._ui_adj = self._bipartite_adjacency()
        return self._ui_adj

    @ui_adj.setter
    def ui_adj(self, m):
        self._ui_adj = m

    @property
    def norm_adj(self):
        return LazyNormAdj(self) if self._norm_adj is None else self._norm_adj

    @norm_adj.setter
    def norm_adj(self, m):
        self._norm_adj = m

    @property
    def interaction_mat(self):
        if self._interaction_mat is None:
            self._interaction_mat = self._interaction_matrix()
        return self._interaction_mat

    @interaction_mat.setter
    def interaction_mat(self, m):
        self._interaction_mat = m

    def __getstate__(self):
        return dict(self.__dict__)

    def __deepcopy__(self, memo):
        import copy
        from .sampler import PairSampler
        new = object.__new__(type(self))
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == 'pair_sampler':
                new.pair_sampler = PairSampler(v.pairs.copy(), v.n_items, (v.memb_rowptr, v.memb_items))
            elif k in ('training_set_u', 'training_set_i', '_arl_memb', 'val_data', 'test_data'):
                new.__dict__[k] = v                               # read-only images
            elif k in ('user', 'item', 'id2user', 'id2item'):
                new.__dict__[k] = dict(v)
            elif k == '_rating':
                new.__dict__[k] = v.copy()
            elif k in ('_pairs_buf', '_rating_buf'):
                continue                                          # the copy starts without spare capacity
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new
