"""CPU tests: host logic of the product (DataLoader mirror, drop-in sampler, metrics) and the C-ABI surface
(the library loads and exports every symbol include/arlib_amd.h declares).  No GPU compute."""
import ctypes
import os
import random
import re
import numpy as np
import pytest
from conftest import golden, ROOT


def make_data(array_native=False):
    from arlib_amd.util.DataLoader import DataLoader
    g = golden('ml100k_data.npz')
    return DataLoader.from_arrays((g['train_u'], g['train_i'], g['train_r']), (g['val_u'], g['val_i'], g['val_r']),
                                  (g['test_u'], g['test_i'], g['test_r']), dataName='ml-100k', array_native=array_native)


def test_abi_exports_match_header():
    from arlib_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'arlib_amd.h')).read()
    declared = set(re.findall(r'\b(arl_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    l = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(l, name), name
    assert _lib.lib().arl_abi_version() == _lib.ABI_VERSION
    # argument validation happens before any device work: NULL pointers / bad sizes are rejected with ARL_E_* codes
    assert _lib.lib().arl_sgd_dense_f32(None, None, 4, 0.1, None) == -1
    assert _lib.lib().arl_bpr_l2_workspace_bytes(2048) == 4 * 4 * 2048
    assert _lib.lib().arl_sampler_shuffle(None, None, 5) == -1


def test_missing_library_fails_loudly(monkeypatch):
    from arlib_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libarlib_amd.so')
    with pytest.raises(_lib.ArlError):
        _lib.lib()


def test_dataloader_matches_reference_ids_and_adjacency(ml100k):
    data = make_data()
    assert data.training_size() == (ml100k['U'], ml100k['I'], ml100k['nnz'])
    ids = np.array([[data.user[r[0]], data.item[r[1]]] for r in data.training_data], np.int32)
    assert np.array_equal(ids, ml100k['pairs0'])                       # first-seen id assignment
    g = golden('g3_adj.npz')
    na = data.norm_adj.tocsr(); na.sort_indices()
    assert np.array_equal(na.indptr, g['norm_indptr']) and np.array_equal(na.indices, g['norm_indices'])
    assert np.allclose(na.data, g['norm_data'], rtol=1e-6, atol=0)
    assert data.matrix().shape == (ml100k['U'], ml100k['I']) and data.matrix().nnz == ml100k['nnz']
    assert data.get_user_id('253') == 0 and data.get_item_id('465') == 0 and data.get_user_id('nope') is None


def test_dropin_sampler_consumes_python_random_bit_exact(ml100k):
    from arlib_amd.util.sampler import next_batch_pairwise
    g = golden('g1_sampler.npz')
    data = make_data()
    random.seed(2018)
    for ep in range(2):
        bs = list(next_batch_pairwise(data, 2048))
        assert [len(bs), len(bs[-1][0])] == list(g['ep%d_nb' % ep])
        assert np.array_equal(np.concatenate([b[0] for b in bs]), g['ep%d_u' % ep])
        assert np.array_equal(np.concatenate([b[1] for b in bs]), g['ep%d_p' % ep])
        assert np.array_equal(np.concatenate([b[2] for b in bs]), g['ep%d_n' % ep])
        # the Python list itself was shuffled in place, like the reference's (Q7 carry-over)
        assert [data.user[r[0]] for r in data.training_data[:50]] == list(g['ep%d_u' % ep][:50])
    assert random.random() == float(g['next_random'][0])
    assert tuple(int(x) for x in g['mt_state_after'])[:5] != ()      # fixture sanity
    # appended fake-user interactions (attack/White/CLeaR.py:190-191): empty training_set_u -> never rejected
    data.user['fakeuser0'] = len(data.user); data.id2user[len(data.user) - 1] = 'fakeuser0'
    for it in ('465', '222'):
        data.training_data.append(('fakeuser0', it))
    random.seed(3)
    state = random.getstate()
    bs = list(next_batch_pairwise(data, 4096))
    assert sum(len(b[0]) for b in bs) == ml100k['nnz'] + 2
    fu = data.user['fakeuser0']
    assert sum(int((b[0] == fu).sum()) for b in bs) == 2


def test_metrics_and_topk_helpers():
    from arlib_amd.util.metrics import ranking_evaluation
    from arlib_amd.util.algorithm import find_k_largest
    origin = {'a': {'x': 1, 'y': 1}, 'b': {'z': 1}}
    res = {'a': [('x', 0.9), ('q', 0.5)], 'b': [('q', 0.3), ('z', 0.2)]}
    m = ranking_evaluation(origin, res, [2])
    vals = {s.split(':')[0]: float(s.split(':')[1]) for s in m[1:]}
    assert m[0] == 'Top 2\n' and abs(vals['Hit Ratio'] - 2 / 3) < 1e-12 and abs(vals['Precision'] - 0.5) < 1e-12
    assert abs(vals['Recall'] - 0.75) < 1e-12
    ids, sc = find_k_largest(3, np.array([0.1, 0.9, 0.5, 0.7, -1.0]))
    assert ids == [1, 3, 2] and sc == [0.9, 0.7, 0.5]


def test_synthetic_generator_is_deterministic_and_valid():
    from arlib_amd.util import synthetic as S
    p = S.syn_v1_pairs(2000, 300, mean_deg=16, seed=2018)
    assert S.graph_digest(p) == S.graph_digest(S.syn_v1_pairs(2000, 300, mean_deg=16, seed=2018))
    assert S.graph_digest(p) != S.graph_digest(S.syn_v1_pairs(2000, 300, mean_deg=16, seed=2019))
    key = p[:, 0].astype(np.int64) * 300 + p[:, 1]
    assert np.all(np.diff(key) > 0)                                     # user-major, strictly increasing, no duplicates
    assert set(np.unique(p[:, 1])) == set(range(300))                   # every item covered
    assert np.bincount(p[:, 0], minlength=2000).min() >= 1
    rowptr, col = S.bipartite_csr_from_sorted_pairs(p, 2000, 300)
    from oracle import oracle as O
    rp2, col2, _ = O.bipartite_csr(p[:, 0], p[:, 1], 2000, 300)
    assert np.array_equal(rowptr, rp2) and np.array_equal(col, col2)
    d = S.InteractionData(p, 2000, 300)
    from arlib_amd.util.sampler import MTState
    mt = MTState.from_seed(5)
    st = O.mt_seed(5)
    pairs_o = p.copy()
    memb = O.build_membership(p, 2000)
    for (u, pp, n), (ou, op, on) in zip(d.pair_sampler.epoch(mt, 777), O.next_batch_pairwise(st, pairs_o, 777, 300, memb)):
        assert np.array_equal(u, ou) and np.array_equal(pp, op) and np.array_equal(n, on)
    assert np.array_equal(mt.words, st)


def test_deferred_list_permutation_and_cheap_deepcopy(ml100k):
    """The sampler only records each epoch's permutation of data.training_data; reading the attribute applies what is pending
    (two epochs compose).  copy.deepcopy(data) -- how the attacks fork a surrogate -- owns its list and id maps, shares the rows."""
    import copy
    from arlib_amd.util.sampler import next_batch_pairwise
    g = golden('g1_sampler.npz')
    data = make_data()
    rows_before = {id(r) for r in data._raw_training_data()[0]}
    random.seed(2018)
    for ep in range(2):
        for _ in next_batch_pairwise(data, 2048):
            pass
        assert data._raw_training_data()[1] is not None                     # nothing read the list: still pending
    fork = copy.deepcopy(data)                                               # the pending permutation is carried, not applied
    assert data._raw_training_data()[1] is not None and fork._raw_training_data()[1] is not None
    ids = [data.user[r[0]] for r in data.training_data]
    assert ids[:2048] == list(g['ep1_u'][:2048]) and len(ids) == ml100k['nnz']
    assert {id(r) for r in data.training_data} == rows_before                # same row objects, re-ordered
    assert fork.training_data == data.training_data and fork.training_data is not data.training_data
    assert fork.training_data[0] is data.training_data[0] and fork.user == data.user and fork.user is not data.user
    fork.user['fakeuser0'] = len(fork.user)
    fork.training_data.append(('fakeuser0', '465'))
    assert 'fakeuser0' not in data.user and len(data.training_data) == ml100k['nnz']
    # the fork's sampler image came along and is extended by the appended tail only
    u, i = fork._ids()
    assert len(u) == ml100k['nnz'] + 1 and u[-1] == fork.user['fakeuser0'] and list(u[:100]) == ids[:100]
    # a third epoch on the original continues the reference's stream exactly as if the list had been shuffled eagerly
    eager = make_data()
    random.seed(2018)
    for ep in range(3):
        last = [b for b in next_batch_pairwise(eager, 2048)]
        eager.training_data                                                  # force materialisation every epoch
    random.seed(2018)
    lazy = make_data()
    for ep in range(3):
        last_lazy = [b for b in next_batch_pairwise(lazy, 2048)]
    assert all(np.array_equal(a[k], b[k]) for a, b in zip(last, last_lazy) for k in range(3))
    assert [r[0] for r in lazy.training_data] == [r[0] for r in eager.training_data]


def test_vectorised_ranking_evaluation_is_string_identical():
    """evaluate() computes its measure lines from the top-k index array; they must be the reference's strings exactly."""
    from arlib_amd.util.metrics import ranking_evaluation, ranking_evaluation_topk
    data = make_data()
    rng = np.random.default_rng(0)
    users = list(data.test_set)
    I = len(data.item)
    idx = np.stack([rng.permutation(I)[:50] for _ in users])
    for r, u in enumerate(users):                                           # plant some hits at assorted ranks
        its = [data.item[i] for i in data.test_set[u] if i in data.item]
        for j, it in enumerate(its[:3]):
            if it not in idx[r]:
                idx[r, (7 * j + r) % 50] = it
    rec = {u: [(data.id2item[int(i)], 0.0) for i in idx[r]] for r, u in enumerate(users)}
    assert ranking_evaluation_topk(data, idx, [10, 20, 50]) == ranking_evaluation(data.test_set, rec, [10, 20, 50])


def test_append_while_a_permutation_is_pending(ml100k):
    """An attack appends fake-user rows right after a training epoch: the sampler image grows by the tail only and the list,
    once read, is the shuffled order followed by the new rows."""
    from arlib_amd.util.sampler import next_batch_pairwise, _shadow
    data = make_data()
    random.seed(7)
    for _ in next_batch_pairwise(data, 4096):
        pass
    assert data._raw_training_data()[1] is not None
    image_before = _shadow(data).pairs.copy()
    data.user['fakeuser0'] = len(data.user); data.id2user[len(data.user) - 1] = 'fakeuser0'
    data.append_training_rows([('fakeuser0', '465'), ('fakeuser0', '222')])
    assert data._raw_training_data()[1] is not None                          # still pending
    u, i = data._ids()
    assert np.array_equal(u[:-2], image_before[:, 0]) and list(u[-2:]) == [data.user['fakeuser0']] * 2
    assert data._raw_training_data()[1] is not None                          # the image was extended without touching the list
    rows = data.training_data
    assert [data.user[r[0]] for r in rows] == u.tolist() and [data.item[r[1]] for r in rows] == i.tolist()


def test_whole_epoch_sampling_is_the_same_stream(ml100k):
    """whole_epoch=True (one native call per epoch, used by the training loops) yields the reference's batches and leaves
    Python's RNG where the per-batch form leaves it."""
    from arlib_amd.util.sampler import next_batch_pairwise
    g = golden('g1_sampler.npz')
    data = make_data()
    random.seed(2018)
    for ep in range(2):
        bs = list(next_batch_pairwise(data, 2048, whole_epoch=True))
        assert [len(bs), len(bs[-1][0])] == list(g['ep%d_nb' % ep])
        for k, key in enumerate(('u', 'p', 'n')):
            assert np.array_equal(np.concatenate([b[k] for b in bs]), g['ep%d_%s' % (ep, key)])
    assert random.random() == float(g['next_random'][0])


@pytest.mark.parametrize('n', [0, 1, 2, 511, 512, 513, 1025, 5000])
def test_native_shuffle_is_random_shuffle_across_the_lookahead_block(n):
    """arl_sampler_shuffle draws the swap partners a block of 512 ahead (to prefetch their rows) and then swaps in random.shuffle's order: same
    permutation and same RNG end point as CPython for sizes around the block size, from an 8-byte- and a 4-byte-aligned buffer."""
    import ctypes as C
    from arlib_amd import _lib
    for shift in (0, 1):
        raw = np.zeros(2 * max(n, 1) + 2, np.int32)
        pairs = raw[shift:shift + 2 * n].reshape(n, 2) if n else raw[:0].reshape(0, 2)
        if n:
            pairs[:, 0] = np.arange(n); pairs[:, 1] = 7 * np.arange(n)
        random.seed(99)
        st = np.array(random.getstate()[1], dtype=np.uint32)
        ptr = C.c_void_p(pairs.ctypes.data) if n else C.c_void_p(raw.ctypes.data)
        _lib.check(_lib.lib().arl_sampler_shuffle(st.ctypes.data_as(C.c_void_p), ptr, n), 'arl_sampler_shuffle')
        ref = list(range(n)); random.shuffle(ref)
        assert pairs[:, 0].tolist() == ref and pairs[:, 1].tolist() == [7 * x for x in ref]
        assert np.array_equal(st, np.array(random.getstate()[1], dtype=np.uint32))


@pytest.mark.parametrize('chunk_batches', [None, 3, 1])
def test_device_epoch_producer_thread_is_the_same_stream(ml100k, chunk_batches):
    """device_epoch (what train() iterates): the epoch's negatives are drawn in chunks by a producer thread behind the consumer.  Same batches as the
    reference's generator captured in g1 (two epochs, carry-over), Python's RNG left where the reference leaves it -- for one chunk (no thread), chunks of
    3 batches and of 1 -- and also when the consumer stops after two batches: the producer still finishes the epoch's RNG stream."""
    import torch
    from arlib_amd.util.sampler import device_epoch
    g = golden('g1_sampler.npz')
    data = make_data()
    random.seed(2018)
    for ep in range(2):
        st = {}
        bs = list(device_epoch(data, 2048, 'cpu', ml100k['U'], ml100k['I'], chunk_batches=chunk_batches, stats=st))
        assert [len(bs), len(bs[-1][0])] == list(g['ep%d_nb' % ep])
        for k, key in enumerate(('u', 'p', 'n')):
            assert np.array_equal(torch.cat([b[k] for b in bs]).numpy(), g['ep%d_%s' % (ep, key)])
        assert st['chunks'] == (1 if chunk_batches is None else -(-len(bs) // chunk_batches)) and st['first_batch_seconds'] > 0
    assert random.random() == float(g['next_random'][0])
    # early close: same RNG end point
    data2 = make_data()
    random.seed(2018)
    for ep in range(2):
        it = device_epoch(data2, 2048, 'cpu', chunk_batches=chunk_batches)
        first = [next(it), next(it)]
        it.close()
        assert np.array_equal(first[1][2].numpy(), g['ep%d_n' % ep][2048:4096])
    assert random.random() == float(g['next_random'][0])
    # a sampler result outside the tables surfaces in the consumer
    with pytest.raises(IndexError):
        list(device_epoch(make_data(), 2048, 'cpu', n_users=10, chunk_batches=chunk_batches))


@pytest.mark.parametrize('n,k', [(44212, 39790), (943, 94), (100000, 17), (30, 5), (10, 10), (5, 0)])
def test_native_sample_range_is_pythons(n, k):
    """random.sample(range(n), k) natively: both of CPython's algorithms (pool / rejection), same values, same RNG consumption."""
    from arlib_amd.util.sampler import sample_range
    random.seed(11); want = random.sample(range(n), k); after = random.random()
    random.seed(11); got = sample_range(n, k)
    assert got.tolist() == want and random.random() == after
    with pytest.raises(ValueError):
        sample_range(3, 4)


def test_lpt_deal_balances_and_respects_capacity():
    """Plan helper of the register-blocked SpMM (include/arlib_amd.h: arl_lpt_deal): against a direct Python restatement."""
    import ctypes as C, heapq
    from arlib_amd import _lib
    rng = np.random.default_rng(5)
    L = _lib.lib()
    for n, cap in ((1, 16), (100, 16), (1000, 32), (33, 32), (0, 32)):
        w = np.sort(rng.integers(0, 500, n)).astype(np.int32)[::-1].copy()
        n_bins = (n + cap - 1) // cap
        b = np.full(max(n, 1), -1, np.int32); s = np.full(max(n, 1), -1, np.int32)
        assert L.arl_lpt_deal(n, w.ctypes.data, n_bins, cap, b.ctypes.data, s.ctypes.data) == 0
        heap = [(0, k) for k in range(n_bins)]; fill = [0] * n_bins
        for r in range(n):
            load, k = heapq.heappop(heap)
            assert (b[r], s[r]) == (k, fill[k])
            fill[k] += 1
            if fill[k] < cap:
                heapq.heappush(heap, (load + int(w[r]), k))
        if n:
            assert np.bincount(b[:n], minlength=n_bins).max() <= cap
            assert len({(x, y) for x, y in zip(b[:n], s[:n])}) == n
    w = np.array([1, 5], np.int32); b = np.zeros(2, np.int32); s = np.zeros(2, np.int32)
    assert L.arl_lpt_deal(2, w.ctypes.data, 1, 16, b.ctypes.data, s.ctypes.data) == -4      # not sorted descending
    assert L.arl_lpt_deal(40, w.ctypes.data, 1, 16, b.ctypes.data, s.ctypes.data) == -4     # does not fit


def test_datasave_matches_per_entry_formatting(tmp_path):
    """util/tool.py:dataSave: the chunked writer against the reference's per-entry construction (restated here), with fake users
    missing from id2user, non-unit weights, an explicit zero and an empty row."""
    import scipy.sparse as sp
    from arlib_amd.util.tool import dataSave
    rng = np.random.default_rng(8)
    U, I = 37, 23
    dense = (rng.random((U, I)) < 0.2).astype(np.float32)
    dense[5] = 0; dense[U - 1, :4] = [0.5, 0.25, 1, 0.125]; dense[U - 2, 7] = 3
    m = sp.csr_matrix(dense)
    m.data[0] = 0.0                                      # stored zero: nonzero() skips it
    id2user = {u: 'u%d' % (u * 7) for u in range(U - 2)}
    id2item = {i: 'item_%d' % i for i in range(I)}
    want = []
    ind = m.nonzero()
    for i, j in zip(ind[0].tolist(), ind[1].tolist()):
        user = id2user[i] if i in id2user.keys() else 'fakeUser' + str(i)
        want.append('{} {} {}'.format(user, id2item[j], m[i, j]) + '\n')
    for chunk in (4_000_000, 7):
        path = tmp_path / ('out%d.txt' % chunk)
        dataSave(m, str(path), id2user, id2item, chunk=chunk)
        assert open(path).readlines() == want
    assert any(l.startswith('fakeUser%d ' % (U - 1)) and l.endswith(' 0.125\n') for l in want) and any(l.endswith(' 3.0\n') for l in want)


def test_header_is_plain_c(tmp_path):
    """include/arlib_amd.h is the FFI contract: it must compile as C99 (and as C++) on its own, and its two structs must have the layout
    the ctypes mirror in arlib_amd/_lib.py assumes."""
    import shutil, subprocess
    from arlib_amd import _lib
    hdr = os.path.join(ROOT, 'include', 'arlib_amd.h')
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    subprocess.run(['gcc', '-x', 'c', '-std=c99', '-fsyntax-only', '-Wall', '-Werror', hdr], check=True)
    subprocess.run(['g++', '-x', 'c++', '-std=c++17', '-fsyntax-only', '-Wall', '-Werror', hdr], check=True)
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include "arlib_amd.h"\nint main(void) { printf("%zu %zu %zu\\n", sizeof(arl_csr), sizeof(arl_blocked), sizeof(arl_tiled)); return 0; }\n')
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [ctypes.sizeof(_lib.arl_csr), ctypes.sizeof(_lib.arl_blocked), ctypes.sizeof(_lib.arl_tiled)]


@pytest.mark.parametrize('rpw,hub,split,split_hubs', [(32, 40, True, True), (32, 40, True, False), (16, 100000, False, True), (32, 7, True, True), (32, 7, True, False),
                                                    (16, 12, False, True)])
def test_blocked_plan_invariants_and_numpy_emulation(rpw, hub, split, split_hubs):
    """ops.BlockedPlan built on the CPU (torch ops + the host dealing helper; no kernel runs): every planned row sits in exactly one
    wave slot, every edge of a planned row is exactly one record of that wave with that slot, streams are padded to 64 with zero values
    and sorted by (column block, slot), waves of a set carry (nearly) equal edge counts -- and a numpy emulation of the kernel's
    arithmetic on the plan, plus the hub rows, reproduces A @ X.  With split_hubs the rows above the hub threshold are dealt as strided
    pieces (piece p of P = every P-th edge of the column-sorted row) whose raw sums meet in the split_* combine lists."""
    import torch
    import scipy.sparse as sp
    from arlib_amd import ops
    rng = np.random.default_rng(rpw + hub)
    U, I, d = 700, 150, 64
    deg = np.clip(rng.poisson(6, U), 0, I)
    us = np.repeat(np.arange(U), deg); its = np.floor(I * rng.random(len(us)) ** 2).astype(np.int64)
    key = np.unique(us * I + its); us, its = key // I, key % I
    R = sp.csr_matrix((rng.random(len(us)).astype(np.float32) + 0.5, (us, its)), shape=(U, I))
    Adj = sp.bmat([[None, R], [R.T, None]], format='csr').astype(np.float32)
    Adj.sort_indices()
    N = U + I
    A = ops.CSRGraph(Adj.indptr.astype(np.int64), Adj.indices.astype(np.int32), Adj.data, 'cpu', chunk=64)
    A.enable_blocked(split=U if split else None, rows_per_wave=rpw, hub=hub, col_block=32, split_hubs=split_hubs)
    bp = A.blocked
    X = rng.standard_normal((N, d)).astype(np.float32)
    Y = np.full((N, d), np.nan, np.float32)
    seen_rows, seen_edges = [], 0
    rowdeg = np.diff(Adj.indptr)
    for st in bp.sets:
        wp, wr = st['wave_ptr'].numpy(), st['wave_rows'].numpy()
        rc, rv = st['rec_col'].numpy(), st['rec_val'].numpy()
        assert np.all(wp % 64 == 0) and wp[0] == 0 and np.all(np.diff(wp) >= 0)
        loads = []
        partial = np.full((max(st['n_pieces'], 1), d), np.nan, np.float64)
        piece_edges = np.zeros(max(st['n_pieces'], 1), np.int64)
        piece_cols = {}
        for w in range(st['n_waves']):
            c, v = rc[wp[w]:wp[w + 1]], rv[wp[w]:wp[w + 1]]
            slot, colid = (c.astype(np.uint32) >> 24).astype(np.int64), (c & 0xffffff).astype(np.int64)
            real = v != 0
            if (~real).any() and real.any():                                       # padding repeats the wave's last real record
                assert np.all(c[~real] == c[real][-1])
            keyw = (colid[real] // 32) * rpw + slot[real]
            assert np.all(np.diff(keyw) >= 0) and slot.max(initial=0) < rpw       # sorted by (column block, slot)
            acc = np.zeros((rpw, d), np.float64)
            np.add.at(acc, slot, v[:, None].astype(np.float64) * X[colid])
            rows = wr[w]
            for sl, r in enumerate(rows):
                if r >= 0:
                    Y[r] = acc[sl]; seen_rows.append(int(r))
                    assert int((slot[real] == sl).sum()) == rowdeg[r]              # all of the row's edges are here, under its slot
                elif r < -1:
                    assert np.isnan(partial[-r - 2, 0])
                    partial[-r - 2] = acc[sl]; piece_edges[-r - 2] = int((slot[real] == sl).sum())
                    piece_cols[-r - 2] = np.sort(colid[real][slot[real] == sl])
                else:
                    assert not np.any(slot[real] == sl)
            loads.append(int(real.sum()))
            seen_edges += int(real.sum())
        if st['n_waves'] > 1 and hub >= 40:
            assert max(loads) - min(loads) <= max(rowdeg[rowdeg <= hub].max(), 1)                # longest-first dealing balances the waves
        if st['n_split']:
            assert split_hubs
            sr, sf, sc = st['split_row'].numpy(), st['split_first'].numpy(), st['split_count'].numpy()
            assert sf[0] == 0 and np.all(sf[1:] == np.cumsum(sc)[:-1]) and int(sc.sum()) == st['n_pieces'] and np.all(sc >= 2)
            for r, f, n in zip(sr, sf, sc):
                assert rowdeg[r] > hub and piece_edges[f:f + n].sum() == rowdeg[r] and piece_edges[f:f + n].max() <= hub
                Y[r] = partial[f:f + n].sum(0); seen_rows.append(int(r))
                rc_ = Adj.indices[Adj.indptr[r]:Adj.indptr[r + 1]]
                for k in range(n):                                                 # piece k = every n-th edge of the column-sorted row, from k
                    assert np.array_equal(piece_cols[f + k], rc_[k::n])
    hub_rows = bp._hub_rows.numpy()
    assert (len(hub_rows) == 0) if split_hubs else np.array_equal(np.sort(hub_rows), np.nonzero(rowdeg > hub)[0])
    assert sorted(seen_rows + hub_rows.tolist()) == list(range(N))                     # every row exactly once
    assert seen_edges + int(rowdeg[hub_rows].sum()) == Adj.nnz
    Y[hub_rows] = (Adj[hub_rows] @ X)
    ref = Adj @ X
    assert np.abs(Y - ref).max() <= 1e-4 * np.abs(ref).max()


def test_none_attack_protocol_matches_reference_run():
    """BASELINE config 1's attack leg (attack/Black/NoneAttack.py:7-40): constructor contract (target draw from Python's RNG, budgets, capability
    flags) and the identity posionDataAttack(), against the reference run recorded in g19 -- incl. the next random() afterwards (nothing else
    draws).  No GPU involved."""
    import random
    from types import SimpleNamespace
    import scipy.sparse as sp
    from arlib_amd.util.tool import seedSet
    from arlib_amd.attack.Black.NoneAttack import NoneAttack
    g = golden('g19_victims.npz')
    seedSet(2018)
    data = make_data()
    args = SimpleNamespace(maliciousUserSize=3, maliciousFeedbackSize=0, Epoch=1, innerEpoch=1, outerEpoch=1, attackTargetChooseWay='unpopular', targetSize=5)
    atk = NoneAttack(args, data)
    res = sp.csr_matrix(atk.posionDataAttack())
    assert list(atk.targetItem) == [int(t) for t in g['none_targets']]
    assert [atk.userNum, atk.itemNum, atk.fakeUserNum, atk.maliciousFeedbackNum, res.nnz] == [int(x) for x in g['none_sizes']]
    assert [int(atk.recommenderGradientRequired), int(atk.recommenderModelRequired)] == [int(x) for x in g['none_flags']]
    assert (res != sp.csr_matrix(data.matrix())).nnz == 0
    assert random.random() == float(g['none_next_random'][0])


def test_array_native_dataloader_equals_list_based_loader():
    """DataLoader.from_arrays(array_native=True) (numpy images, Python containers materialised on demand -- what 3.2e7-interaction graphs
    use) against the list/dict-based loader that mirrors util/DataLoader.py:8-55 on ml-100k: id maps in first-seen order, the train/val/test
    sets (key order included), matrices, the sampler's rejection sets, the batch stream drawn from Python's RNG, the in-place shuffle of
    training_data with ratings following their rows, appends and deepcopy."""
    import copy
    import random
    from arlib_amd.util.DataLoader import DataLoader, ArrayDataLoader
    from arlib_amd.util.sampler import next_batch_pairwise
    g = golden('ml100k_data.npz')
    tr, va, te = ((g[s + '_u'], g[s + '_i'], g[s + '_r']) for s in ('train', 'val', 'test'))
    a = DataLoader.from_arrays(tr, va, te, dataName='ml-100k', array_native=False)
    b = DataLoader.from_arrays(tr, va, te, dataName='ml-100k', array_native=True)
    assert isinstance(b, ArrayDataLoader) and type(a) is DataLoader
    assert a.user == b.user and a.item == b.item and a.id2user == b.id2user and a.id2item == b.id2item
    assert (a.user_num, a.item_num) == (b.user_num, b.item_num) and a.training_size() == b.training_size()
    assert dict(a.test_set) == dict(b.test_set) and list(a.test_set) == list(b.test_set) and a.test_set_item == b.test_set_item and dict(a.val_set) == dict(b.val_set)
    assert list(a.training_set_u) == list(b.training_set_u) and all(a.training_set_u[u] == b.training_set_u[u] for u in a.training_set_u)
    assert list(a.training_set_u['1']) == list(b.training_set_u['1'])                      # inner key order = file order
    assert list(a.training_set_i) == list(b.training_set_i) and all(a.training_set_i[i] == b.training_set_i[i] for i in list(a.training_set_i)[::7])
    assert a.contain('1', '61') == b.contain('1', '61') and a.contain('1', 'nope') == b.contain('1', 'nope') and a.user_rated('5') == b.user_rated('5')
    assert a.training_data == b.training_data
    assert (a.matrix() != b.matrix()).nnz == 0 and (a.ui_adj != b.ui_adj).nnz == 0 and abs(a.norm_adj - b.norm_adj.to_scipy()).max() < 1e-7
    ma, mb = a.membership_csr(), b.membership_csr()
    assert np.array_equal(ma[0], mb[0]) and np.array_equal(ma[1], mb[1])
    outs = []
    for d in (a, b):
        random.seed(2018)
        outs.append(([tuple(x.copy() for x in bt) for bt in next_batch_pairwise(d, 2048)], random.random()))
    assert outs[0][1] == outs[1][1] and len(outs[0][0]) == len(outs[1][0]) == 22
    assert all(np.array_equal(x, y) for ba, bb in zip(outs[0][0], outs[1][0]) for x, y in zip(ba, bb))
    assert a.training_data == b.training_data and a.training_data[0] != [str(g['train_u'][0]), str(g['train_i'][0]), float(g['train_r'][0])]   # shuffled alike
    # an attack's appends: new user id, rows at the end, matrices follow
    for d in (a, b):
        d.user['fakeuser0'] = len(d.user); d.id2user[len(d.user) - 1] = 'fakeuser0'; d.user_num += 1
        d.append_training_rows([['fakeuser0', d.id2item[3], 1.0], ['fakeuser0', d.id2item[9], 1.0]])
    assert a.training_data[-2:] == b.training_data[-2:] and (a.matrix() != b.matrix()).nnz == 0 and a.matrix().shape == (a.user_num, a.item_num)
    c = copy.deepcopy(b)
    c.append_training_rows([['fakeuser0', c.id2item[11], 1.0]])
    assert c.training_size()[2] == b.training_size()[2] + 1 and c.user == b.user and c.user is not b.user


def test_array_native_dataloader_at_scale_builds_in_seconds():
    """3.2 M synthetic interactions (a tenth of cfg2): the array-native loader is the default from 2 M interactions on and builds in
    O(nnz) numpy; spot checks of its lazy views against the arrays."""
    import time
    from arlib_amd.util import synthetic
    from arlib_amd.util.DataLoader import DataLoader, ArrayDataLoader
    pairs = synthetic.syn_v1_pairs(100_000, 20_000)
    t0 = time.perf_counter()
    d = DataLoader.from_arrays((pairs[:, 0], pairs[:, 1], np.ones(len(pairs), np.float32)), dataName='syn')
    dt = time.perf_counter() - t0
    assert isinstance(d, ArrayDataLoader) and dt < 60
    assert d.training_size() == (100_000, 20_000, len(pairs))
    u0 = d.id2user[0]
    row = pairs[pairs[:, 0] == int(u0), 1]
    assert list(d.training_set_u[u0]) == [str(x) for x in row.tolist()]
    assert d.matrix().nnz == len(pairs) and d.norm_adj.shape == (120_000, 120_000)


def test_syn_v1_native_generator_equals_numpy_generator():
    """the C++ SYN-v1 generator (arl_syn_v1_pairs) and the numpy one emit the same pair list -- compared element-wise and through
    the cross-language digest (arl_graph_digest / synthetic.graph_digest) -- for several sizes, seeds and degree laws."""
    from arlib_amd.util import synthetic as S
    for U, I, kw in ((3000, 400, {}), (20000, 3000, dict(mean_deg=12.0, seed=7)), (100_000, 20_000, {}), (1500, 50, dict(mean_deg=40.0, seed=3, sigma=0.5, deg_min=1, deg_max=64))):
        a, b = S.syn_v1_pairs(U, I, **kw), S.syn_v1_pairs_native(U, I, **kw)
        assert np.array_equal(a, b)
        assert S.graph_digest(a) == S.graph_digest_native(b) == S.graph_digest_native(a)
    assert S.graph_digest_native(a[::-1].copy()) != S.graph_digest_native(a)           # order-sensitive


def test_csrgraph_device_side_long_row_plan_equals_host_plan():
    """CSRGraph.from_device builds the long-row plan with static shapes and no host read (the CW operator CLeaR rebuilds every step):
    its real part must equal the host constructor's plan; the padding repeats the last real long row and holds only empty chunks."""
    import torch
    from arlib_amd import ops
    rng = np.random.default_rng(5)
    for n, chunk, longs in ((2000, 16, {5: 40, 77: 16, 1999: 100, 1000: 17}), (300, 8, {0: 9}), (50, 4, {49: 50, 48: 5, 3: 33})):
        deg = rng.integers(0, chunk, n)
        for r, v in longs.items():
            deg[r] = v
        rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        nnz = int(rp[-1])
        col, val = torch.zeros(nnz, dtype=torch.int32), torch.ones(nnz)
        g = ops.CSRGraph.from_device(torch.from_numpy(rp), col, val, nnz, chunk=chunk)
        h = ops.CSRGraph(rp, col.numpy(), val.numpy(), 'cpu', chunk=chunk, validate=False)
        nl, nc = h.n_long, h.n_chunks
        assert nl >= 1 and g.n_long >= nl and g.n_chunks >= nc
        for a, b in ((g.long_row, h.long_row), (g.long_first, h.long_first), (g.long_count, h.long_count)):
            assert torch.equal(a[:nl], b) and bool((a[nl:] == b[-1]).all())
        assert torch.equal(g.chunk_begin[:nc], h.chunk_begin) and torch.equal(g.chunk_end[:nc], h.chunk_end)
        assert bool((g.chunk_begin[nc:] == g.chunk_end[nc:]).all())
        assert torch.equal(g.rowptr, h.rowptr)
