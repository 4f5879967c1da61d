"""Driver for tools/sanitize_host.sh: exercises the sanitized host library and oracle through ctypes (incl. edge cases: empty
inputs, single elements, users without items, full-range samples) and checks results against CPython / numpy."""
import ctypes as C, os, random, sys
import numpy as np
d = os.environ['ARL_ASAN_DIR']
H = C.CDLL(os.path.join(d, 'libarl_host_asan.so'))
O = C.CDLL(os.path.join(d, 'liboracle_asan.so'))
vp = lambda a: a.ctypes.data_as(C.c_void_p)
i64 = C.c_int64


def mt_from_python():
    return np.array(random.getstate()[1], dtype=np.uint32)


# MT seeding == CPython's init_by_array
for seed in (0, 1, 2018, 2 ** 40 + 5):
    key = []
    s = seed
    while True:
        key.append(s & 0xFFFFFFFF); s >>= 32
        if not s:
            break
    st = np.zeros(625, np.uint32)
    assert H.arl_mt_seed(vp(st), vp(np.array(key, np.uint32)), i64(len(key))) == 0
    random.seed(seed)
    assert np.array_equal(st, mt_from_python()), seed
# shuffle == random.shuffle, for sizes incl. 0 and 1
for n in (0, 1, 2, 7, 1000, 44212):
    random.seed(5)
    st = mt_from_python()
    pairs = np.stack([np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32) * 3], 1).copy() if n else np.zeros((1, 2), np.int32)
    assert H.arl_sampler_shuffle(vp(st), vp(pairs), i64(n)) == 0
    ref = list(range(n)); random.shuffle(ref)
    assert pairs[:n, 0].tolist() == ref, n
    assert np.array_equal(st, mt_from_python())
# the same from a buffer that is only 4-byte aligned (the swap then moves the two int32 halves separately), across the block size of the look-ahead
for n in (3, 511, 512, 513, 5000):
    random.seed(6)
    st = mt_from_python()
    raw = np.zeros(2 * n + 1, np.int32)
    pairs = raw[1:].reshape(n, 2)
    pairs[:, 0] = np.arange(n); pairs[:, 1] = np.arange(n) * 3
    assert pairs.ctypes.data % 8 == 4
    assert H.arl_sampler_shuffle(vp(st), C.c_void_p(pairs.ctypes.data), i64(n)) == 0
    ref = list(range(n)); random.shuffle(ref)
    assert pairs[:, 0].tolist() == ref and pairs[:, 1].tolist() == [3 * x for x in ref], n
    assert np.array_equal(st, mt_from_python())
# next_batch: rejection against membership, users without any item, window at the very end
rng = np.random.default_rng(0)
U, I, nnz = 50, 40, 600
pu = rng.integers(0, U - 5, nnz).astype(np.int32); pi = rng.integers(0, I, nnz).astype(np.int32)     # users 45..49 never appear
key = np.unique(pu.astype(np.int64) * I + pi)
rowptr = np.zeros(U - 5 + 1, np.int64); np.add.at(rowptr, key // I + 1, 1); rowptr = np.cumsum(rowptr)
items = (key % I).astype(np.int32)
pairs = np.stack([pu, pi], 1).copy()
pairs[-3:, 0] = [47, 48, 49]                                                                            # rows beyond memb_rows
random.seed(9); st = mt_from_python()
out = np.zeros((3, nnz), np.int32)
H.arl_sampler_next_batch.argtypes = [C.c_void_p, C.c_void_p, i64, i64, C.c_int32, C.c_void_p, C.c_void_p, i64, C.c_void_p, C.c_void_p, C.c_void_p]
assert H.arl_sampler_next_batch(vp(st), vp(pairs), 0, nnz, I, vp(rowptr), vp(items), U - 5, vp(out[0]), vp(out[1]), vp(out[2])) == 0
member = set(key.tolist())
assert all((int(u) * I + int(n)) not in member for u, n in zip(out[0], out[2]))
assert H.arl_sampler_next_batch(vp(st), vp(pairs), nnz - 1, 1, I, vp(rowptr), vp(items), U - 5, vp(out[0]), vp(out[1]), vp(out[2])) == 0
assert H.arl_sampler_next_batch(vp(st), vp(pairs), 0, 0, I, vp(rowptr), vp(items), U - 5, vp(out[0]), vp(out[1]), vp(out[2])) == 0
# random.sample(range(n), k): both algorithms, k = 0 and k = n
H.arl_mt_sample_range.argtypes = [C.c_void_p, i64, i64, C.c_int32, C.c_void_p, C.c_void_p]
import math
for n, k in ((10, 10), (10, 0), (1, 1), (500, 450), (100000, 9), (64, 33)):
    setsize = 21 + (4 ** math.ceil(math.log(k * 3, 4)) if k > 5 else 0)
    pool = n <= setsize
    random.seed(3); st = mt_from_python()
    o = np.zeros(max(k, 1), np.int32); scratch = np.zeros(max(n if pool else (n + 31) // 32, 1), np.int32)
    assert H.arl_mt_sample_range(vp(st), n, k, int(pool), vp(o), vp(scratch)) == 0
    assert o[:k].tolist() == random.sample(range(n), k), (n, k)
print('host library: ok')
# the oracle through its own Python wrapper, pointed at the sanitized build: the compositions the golden tests pin, on small
# irregular inputs (isolated nodes, duplicate batch entries, a 1-row InfoNCE, k == I top-k)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as OR
O.orc_mt_random.restype = C.c_double
O.orc_mt_randbelow.restype = C.c_uint32
O.orc_num_threads.restype = C.c_int
OR._LIB = O
U, I, d = 30, 12, 8
u = rng.integers(0, U - 3, 200); it = rng.integers(0, I - 2, 200)                 # 3 users and 2 items stay isolated
# plan helper of the register-blocked SpMM: exact fill (n == n_bins * cap), ragged, empty, rejected inputs
H.arl_lpt_deal.argtypes = [i64, C.c_void_p, i64, i64, C.c_void_p, C.c_void_p]
for n, cap in ((64, 16), (65, 16), (1, 32), (0, 16), (5000, 32)):
    wts = np.sort(rng.integers(0, 1000, n)).astype(np.int32)[::-1].copy()
    nb = (n + cap - 1) // cap
    b = np.zeros(max(n, 1), np.int32); sl = np.zeros(max(n, 1), np.int32)
    assert H.arl_lpt_deal(n, vp(wts), nb, cap, vp(b), vp(sl)) == 0
    if n:
        assert b[:n].max() < nb and sl[:n].max() < cap and np.bincount(b[:n]).max() <= cap
assert H.arl_lpt_deal(10, vp(np.arange(10, dtype=np.int32)), 1, 16, vp(np.zeros(10, np.int32)), vp(np.zeros(10, np.int32))) == -4     # ascending weights
assert H.arl_lpt_deal(40, vp(np.zeros(40, np.int32)), 2, 16, vp(np.zeros(40, np.int32)), vp(np.zeros(40, np.int32))) == -4            # does not fit
print('arl_lpt_deal: ok')
rowptr, col, w = OR.bipartite_csr(u, it, U, I)
val = OR.norm_adj_values(rowptr, col, w)
assert np.isfinite(val).all()
E0 = rng.normal(size=(U + I, d)).astype(np.float32)
out = OR.lightgcn_forward((rowptr, col, val), E0, 3)
bu = np.array([0, 0, 1, 5], np.int32); bp = np.array([1, 1, 2, 3], np.int32); bn = np.array([4, 4, 1, 0], np.int32)
lb, lr_, G = OR.bpr_l2(out, U, bu, bp, bn, 1e-4)
g0 = OR.lightgcn_backward((rowptr, col, val), G, 3)
assert np.isfinite(g0).all() and np.isfinite(lb)
l, d1, d2 = OR.infonce(out[:1], out[1:2], 0.2)
idx, sc = OR.score_mask_topk(out[:U], out[U:], I, (np.zeros(U + 1, np.int64), np.zeros(0, np.int32)))
assert idx.shape == (U, I)
st = OR.TrainState(E0[:U], E0[U:], (rowptr, col, val), 2, 1e-4, 0.005)
st.step(bu, bp, bn)
print('oracle: ok')
