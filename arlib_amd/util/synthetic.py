"""SYN-v1: deterministic synthetic user x item interaction graphs and the array-native data
object used at scales where the reference's dict/list DataLoader (util/DataLoader.py) cannot exist.

All randomness comes from a counter-based hash, h(stream, index) = splitmix64(splitmix64(seed ^ stream*PHI) ^ index),
so any implementation (numpy here) produces the same graph bit for bit:
  * user degree  deg_u = clamp(round(exp(mu + sigma*z)), 4, 2048), sigma = 1, mu = ln(mean_deg) - sigma^2/2,
    z = Box-Muller of two hash uniforms (streams 1, 2, index u)
  * the k-th draw overall picks item pi[floor(I * r^2)], r = uniform(stream 3, index k): popularity density ~ x^-1/2;
    pi = argsort of h(stream 4, j) (a hash-derived permutation of item ids)
  * every item j additionally gets the edge (h(stream 5, j) mod U, j), so no item is isolated
  * per-user sort + de-duplication; interactions are emitted user-major (the pre-shuffle `training_data` order).
"""
import numpy as np

PHI = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x):
    x = (x + PHI).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_u64(seed, stream, index):
    with np.errstate(over='ignore'):
        base = splitmix64(np.array([np.uint64(seed) ^ (np.uint64(stream) * PHI)], dtype=np.uint64))[0]
        return splitmix64(np.asarray(index, dtype=np.uint64) ^ base)


def hash_uniform(seed, stream, index):
    """uniform in (0,1): top 53 bits, shifted by half an ulp so 0 is excluded."""
    return ((hash_u64(seed, stream, index) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def syn_v1_pairs(n_users, n_items, mean_deg=32.0, seed=2018, sigma=1.0, deg_min=4, deg_max=2048):
    """Returns int32 [nnz,2] (user, item) pairs sorted user-major, de-duplicated."""
    U, I = int(n_users), int(n_items)
    uidx = np.arange(U, dtype=np.uint64)
    u1, u2 = hash_uniform(seed, 1, uidx), hash_uniform(seed, 2, uidx)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    mu = np.log(mean_deg) - 0.5 * sigma * sigma
    deg = np.clip(np.rint(np.exp(mu + sigma * z)), deg_min, min(deg_max, I)).astype(np.int64)
    total = int(deg.sum())
    pi = np.argsort(hash_u64(seed, 4, np.arange(I, dtype=np.uint64)), kind='stable').astype(np.int64)
    keys = np.empty(total + I, np.int64)
    # draws, in blocks to bound temporaries
    owners = np.repeat(np.arange(U, dtype=np.int64), deg)
    blk = 1 << 24
    for s in range(0, total, blk):
        e = min(total, s + blk)
        r = hash_uniform(seed, 3, np.arange(s, e, dtype=np.uint64))
        it = pi[np.minimum((I * r * r).astype(np.int64), I - 1)]
        keys[s:e] = owners[s:e] * I + it
    del owners
    cover = (hash_u64(seed, 5, np.arange(I, dtype=np.uint64)) % np.uint64(U)).astype(np.int64)
    keys[total:] = cover * I + np.arange(I, dtype=np.int64)
    keys = np.unique(keys)
    pairs = np.empty((len(keys), 2), np.int32)
    pairs[:, 0] = keys // I
    pairs[:, 1] = keys % I
    return pairs


def syn_v1_pairs_native(n_users, n_items, mean_deg=32.0, seed=2018, sigma=1.0, deg_min=4, deg_max=2048):
    """The same pair list from the C++ generator in libarlib_amd.so (arl_syn_v1_pairs): int32 [nnz, 2]."""
    from .. import _lib
    L = _lib.lib()
    need = L.arl_syn_v1_pairs(int(n_users), int(n_items), float(mean_deg), int(seed), float(sigma), int(deg_min), int(deg_max), None, 0)
    if need < 0:
        raise ValueError('arl_syn_v1_pairs: bad arguments (%d)' % need)
    out = np.empty((need, 2), np.int32)
    got = L.arl_syn_v1_pairs(int(n_users), int(n_items), float(mean_deg), int(seed), float(sigma), int(deg_min), int(deg_max), out.ctypes.data, need)
    assert got == need
    return out


def graph_digest_native(pairs):
    from .. import _lib
    p = np.ascontiguousarray(pairs, dtype=np.int32)
    return int(_lib.lib().arl_graph_digest(p.ctypes.data, len(p)))


def graph_digest(pairs):
    """Order-sensitive 64-bit digest of the pair list (cross-implementation check)."""
    p = np.ascontiguousarray(pairs, dtype=np.int32).astype(np.uint64)
    with np.errstate(over='ignore'):
        h = splitmix64(p[:, 0] * np.uint64(0x100000001B3) ^ splitmix64(p[:, 1]) ^ np.arange(len(p), dtype=np.uint64))
        return int(np.bitwise_xor.reduce(h) ^ np.uint64(len(p)))


def bipartite_csr_from_sorted_pairs(pairs, n_users, n_items):
    """(U+I)^2 symmetric adjacency pattern in CSR (int64 rowptr, int32 col) from user-major sorted, unique pairs:
    same matrix as util/DataLoader.py:57-71 builds with scipy (all weights 1)."""
    U, I = int(n_users), int(n_items)
    u = pairs[:, 0].astype(np.int64)
    i = pairs[:, 1].astype(np.int64)
    nnz = len(u)
    du = np.bincount(u, minlength=U)
    di = np.bincount(i, minlength=I)
    rowptr = np.zeros(U + I + 1, np.int64)
    np.cumsum(np.concatenate([du, di]), out=rowptr[1:])
    col = np.empty(2 * nnz, np.int32)
    col[:nnz] = (i + U).astype(np.int32)                       # user rows: item columns ascending (pairs sorted)
    order = np.argsort(pairs[:, 1], kind='stable')              # item rows: users ascending within an item
    col[nnz:] = pairs[order, 0]
    return rowptr, col


class InteractionData:
    """Array-native stand-in for the reference DataLoader at scale: identity id maps, int32 pair array,
    membership CSR for the negative sampler.  `pair_sampler` is what next_batch_pairwise() drives."""

    def __init__(self, pairs, n_users, n_items):
        from .sampler import PairSampler, build_membership
        self.user_num, self.item_num = int(n_users), int(n_items)
        self.pairs0 = np.ascontiguousarray(pairs, dtype=np.int32)
        memb = build_membership(self.pairs0, self.user_num)
        self.pair_sampler = PairSampler(self.pairs0.copy(), self.item_num, memb)

    @property
    def nnz(self):
        return self.pairs0.shape[0]

    def training_size(self):
        return self.user_num, self.item_num, self.nnz

    def adjacency_pattern(self):
        return bipartite_csr_from_sorted_pairs(self.pairs0, self.user_num, self.item_num)


def syn_v1(n_users, n_items, mean_deg=32.0, seed=2018):
    return InteractionData(syn_v1_pairs(n_users, n_items, mean_deg, seed), n_users, n_items)
