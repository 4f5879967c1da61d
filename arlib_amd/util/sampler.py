"""Drop-in for the reference's util/sampler.py:4-30 (next_batch_pairwise), bit-exact.

The sequential MT19937 work (Fisher-Yates shuffle, per-sample rejection sampling) runs in the host part
of libarlib_amd.so (arlib_amd/csrc/arl_host.cpp).  It consumes CPython's *global* `random` state exactly
like the reference does: the state is read with random.getstate(), advanced in C, and written back with
random.setstate(), so code that interleaves other `random` calls stays in lock-step with the reference.
"""
import ctypes as C
import random
import numpy as np

from .. import _lib


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class MTState:
    """CPython `random` state as a 625-word uint32 array."""

    def __init__(self, words=None):
        self.words = np.zeros(625, np.uint32) if words is None else np.ascontiguousarray(words, dtype=np.uint32)

    @classmethod
    def from_seed(cls, seed):
        seed = abs(int(seed))
        key = []
        while True:
            key.append(seed & 0xFFFFFFFF)
            seed >>= 32
            if not seed:
                break
        key = np.array(key, np.uint32)
        st = cls()
        _lib.check(_lib.lib().arl_mt_seed(_vp(st.words), _vp(key), len(key)), 'arl_mt_seed')
        return st

    @classmethod
    def from_python(cls):
        return cls(np.array(random.getstate()[1], dtype=np.uint32))

    def to_python(self):
        st = random.getstate()
        random.setstate((st[0], tuple(int(x) for x in self.words), st[2]))


def sample_range(n, k):
    """random.sample(range(n), k) -- same values, same consumption of Python's global RNG -- as an int32 array, natively
    (the reference's graph augmentations draw 90 % of all edges this way every epoch, recommender/SGL.py:281-299)."""
    import math
    n, k = int(n), int(k)
    if not 0 <= k <= n:
        raise ValueError('Sample larger than population or is negative')
    setsize = 21
    if k > 5:
        setsize += 4 ** math.ceil(math.log(k * 3, 4))
    use_pool = n <= setsize
    out = np.empty(max(k, 1), np.int32)
    scratch = np.empty(max(n if use_pool else (n + 31) // 32, 1), np.int32)
    mt = MTState.from_python()
    _lib.check(_lib.lib().arl_mt_sample_range(_vp(mt.words), n, k, int(use_pool), _vp(out), _vp(scratch)), 'arl_mt_sample_range')
    mt.to_python()
    return out[:k]


def build_membership(pairs, n_users):
    """CSR image of training_set_u (util/DataLoader.py:41): sorted, de-duplicated item ids per user."""
    pairs = np.asarray(pairs)
    key = np.unique(pairs[:, 0].astype(np.int64) * (1 << 32) + pairs[:, 1].astype(np.int64))
    rowptr = np.zeros(n_users + 1, np.int64)
    np.add.at(rowptr, (key >> 32) + 1, 1)
    return np.cumsum(rowptr), (key & 0xFFFFFFFF).astype(np.int32)


class PairSampler:
    """Array-native sampler state: `pairs` int32 [nnz,2] (shuffled in place like data.training_data),
    membership CSR, item count."""

    def __init__(self, pairs, n_items, memb=None, n_users=None):
        self.pairs = np.ascontiguousarray(pairs, dtype=np.int32)
        if self.pairs.ndim != 2 or self.pairs.shape[1] != 2:
            raise ValueError('pairs must be [nnz,2]')
        self.n_items = int(n_items)
        if self.pairs.size and (self.pairs.min() < 0 or self.pairs[:, 1].max() >= self.n_items):
            raise ValueError('pair ids out of range')
        if memb is None:
            memb = build_membership(self.pairs, int(n_users if n_users is not None else self.pairs[:, 0].max() + 1))
        self.memb_rowptr = np.ascontiguousarray(memb[0], dtype=np.int64)
        self.memb_items = np.ascontiguousarray(memb[1], dtype=np.int32)
        if len(self.memb_items) == 0:
            self.memb_items = np.zeros(1, np.int32)

    @property
    def nnz(self):
        return self.pairs.shape[0]

    def shuffle(self, mt, also=None):
        """random.shuffle(training_data).  `also`: optional int32 [nnz,2] array permuted identically."""
        before = mt.words.copy() if also is not None else None
        _lib.check(_lib.lib().arl_sampler_shuffle(_vp(mt.words), _vp(self.pairs), self.nnz), 'arl_sampler_shuffle')
        if also is not None:
            _lib.check(_lib.lib().arl_sampler_shuffle(_vp(before), _vp(also), self.nnz), 'arl_sampler_shuffle')

    def batch(self, mt, begin, count, out=None):
        """One batch into `out` (int32 [3,count], e.g. a pinned-host tensor's numpy view) -> rows u, p, n."""
        if begin < 0 or count < 0 or begin + count > self.nnz:
            raise IndexError('batch window outside the training pairs')
        if out is None:
            out = np.empty((3, count), np.int32)
        if out.dtype != np.int32 or out.shape != (3, count) or not out.flags.c_contiguous:
            raise ValueError('out must be C-contiguous int32 [3,count]')
        _lib.check(_lib.lib().arl_sampler_next_batch(_vp(mt.words), _vp(self.pairs), begin, count, self.n_items, _vp(self.memb_rowptr),
                                                     _vp(self.memb_items), len(self.memb_rowptr) - 1, _vp(out[0]), _vp(out[1]), _vp(out[2])),
                   'arl_sampler_next_batch')
        return out

    def epoch(self, mt, batch_size):
        """Generator over one epoch: shuffle, then consecutive batches (last one ragged)."""
        self.shuffle(mt)
        b = 0
        while b < self.nnz:
            cnt = min(batch_size, self.nnz - b)
            yield self.batch(mt, b, cnt)
            b += cnt


def _rows(data):
    """(list, pending permutation): row j of the CURRENT order is list[perm[j]] (perm None = identity)."""
    if hasattr(data, '_raw_training_data'):
        return data._raw_training_data()
    return data.training_data, None


def _in_sync(sh, data, n=None):
    """Does the int image still describe the first `n` rows of data.training_data?  Sampled check (first, last, 64 random rows):
    the only writers of that list are this module (which permutes image and list together) and the attacks (which append)."""
    td, perm = _rows(data)
    n = sh.nnz if n is None else n
    if n == 0:
        return True
    if n > len(td):
        return False
    idx = np.unique(np.concatenate([[0, n - 1], np.random.default_rng(n).integers(0, n, 64)]))
    try:
        for j in idx.tolist():
            r = td[j if perm is None else int(perm[j])]
            if data.user[r[0]] != sh.pairs[j, 0] or data.item[r[1]] != sh.pairs[j, 1]:
                return False
        return True
    except KeyError:
        return False


def _shadow(data):
    """int32 image of data.training_data for the list-based DataLoader API.  Kept across appends (an attack adding fake-user
    interactions, attack/White/CLeaR.py:190-191, only costs the mapping of the new tail) and rebuilt from scratch otherwise."""
    sh = getattr(data, '_arl_sampler', None)
    if sh is not None and sh.n_items == len(data.item) and sh.nnz == len(_rows(data)[0]) and _in_sync(sh, data):
        return sh                                                  # steady state: nothing touched the list since the last epoch
    raw, perm = _rows(data)
    if (sh is not None and sh.n_items == len(data.item) and sh.nnz < len(raw) and _in_sync(sh, data)
            and (perm is None or np.array_equal(perm[sh.nnz:], np.arange(sh.nnz, len(raw))))):
        # rows appended since the last epoch (a deferred permutation never moves them: DataLoader.append_training_rows)
        tail = np.array([[data.user[r[0]], data.item[r[1]]] for r in raw[sh.nnz:]], np.int32).reshape(-1, 2)
        sh = PairSampler(np.concatenate([sh.pairs, tail]), len(data.item), (sh.memb_rowptr, sh.memb_items))
        data._arl_sampler = sh
        return sh
    td = data.training_data                                        # (brings a deferred permutation up to date)
    pairs = np.array([[data.user[r[0]], data.item[r[1]]] for r in td], np.int32).reshape(-1, 2)
    memb = getattr(data, '_arl_memb', None)
    if memb is None:
        # training_set_u is fixed at DataLoader construction (util/DataLoader.py:41); users added later have an empty set
        if hasattr(data, 'membership_csr'):
            memb = data.membership_csr()
        else:
            tsu = data.training_set_u
            us = [u for u in tsu if len(tsu[u])]
            mp = np.array([[data.user[u], data.item[i]] for u in us for i in tsu[u]], np.int32).reshape(-1, 2)
            n_rows = (max(data.user[u] for u in us) + 1) if us else 0
            memb = build_membership(mp, n_rows) if len(mp) else (np.zeros(1, np.int64), np.zeros(1, np.int32))
        data._arl_memb = memb
    sh = PairSampler(pairs, len(data.item), memb)
    data._arl_sampler = sh
    return sh


def _begin_epoch(data):
    """The epoch's in-place shuffle (util/sampler.py:9) on the int image, with the Python list / ratings kept in the same order.  Returns
    (sampler, mt): `mt` = the RNG state AFTER the shuffle, not yet written back to Python's `random`."""
    track = None
    if hasattr(data, 'pair_sampler'):               # array-native data (synthetic.InteractionData, DataLoader.ArrayDataLoader)
        sh = data.pair_sampler
        order = None
        if hasattr(data, '_permute_ratings') and not getattr(data, '_uniform_rating', True):
            track = np.stack([np.arange(sh.nnz, dtype=np.int32), np.zeros(sh.nnz, np.int32)], 1)      # ratings follow their pairs
    else:
        sh = _shadow(data)
        order = np.stack([np.arange(sh.nnz, dtype=np.int32), np.zeros(sh.nnz, np.int32)], 1)
    mt = MTState.from_python()
    sh.shuffle(mt, also=order if track is None else track)
    if track is not None:
        data._permute_ratings(track[:, 0])
    if order is not None:                            # keep the Python list in the same (shuffled) order: in-place carry-over
        if hasattr(data, '_defer_td_permutation'):
            data._defer_td_permutation(order[:, 0])  # applied when somebody reads data.training_data
        else:
            td = data.training_data
            td[:] = [td[k] for k in order[:, 0]]
    return sh, mt


def next_batch_pairwise(data, batch_size, whole_epoch=False):
    """Generator with the reference's signature and semantics (util/sampler.py:4-30): shuffles
    data.training_data in place, then yields (u_idx, i_idx, j_idx) per batch -- here int32 numpy arrays
    (index a tensor with them exactly as with the reference's Python lists).

    whole_epoch=True draws the negatives of ALL batches in one native call right after the shuffle.  The numbers are the
    same (the stream is consumed sample by sample either way); what changes is WHEN Python's `random` state advances -- at
    once instead of batch by batch -- so it is only for loops that do not touch `random` between batches (our own training
    loops).  It removes the per-batch getstate/setstate round trip (~0.2 ms per batch)."""
    sh, mt = _begin_epoch(data)
    if whole_epoch and sh.nnz:
        allb = sh.batch(mt, 0, sh.nnz)
    mt.to_python()
    b = 0
    while b < sh.nnz:
        cnt = min(batch_size, sh.nnz - b)
        if whole_epoch:
            out = allb[:, b:b + cnt]
            b += cnt
            yield np.ascontiguousarray(out[0]), np.ascontiguousarray(out[1]), np.ascontiguousarray(out[2])
            continue
        mt = MTState.from_python()
        out = sh.batch(mt, b, cnt)
        mt.to_python()
        b += cnt
        yield out[0], out[1], out[2]


EPOCH_CHUNK_BATCHES = 512        # batches per chunk of device_epoch's producer


def device_epoch(data, batch_size, device, n_users=None, n_items=None, chunk_batches=None, stats=None):
    """One epoch of next_batch_pairwise as DEVICE int32 tensors, the sampler running BEHIND the consumer: the epoch's shuffle is the only
    serial part; the negatives are then drawn in chunks of `chunk_batches` batches by ONE producer thread (the MT state is single-owner; the
    native call releases the GIL), range-checked, staged in one of two pinned host buffers and copied to their slice of the epoch's [3, nnz]
    device image on a copy stream of their own -- while the consumer's GPU steps run on the chunks already there.  The consumer's stream waits
    for a chunk's copy event, never the host for the device.  The numbers are those of the reference's generator (util/sampler.py:4-30: the
    stream is consumed sample by sample in the same order); Python's `random` state is written back when the generator ends or is closed --
    having consumed the WHOLE epoch, also after an early close -- so, as with whole_epoch=True, this is for loops that leave `random` alone
    between batches (our training loops).  stats (optional dict): 'first_batch_seconds' = shuffle + first chunk, 'chunks', 'producer_seconds'."""
    import queue
    import threading
    import time
    import torch
    t_begin = time.perf_counter()
    sh, mt = _begin_epoch(data)
    nnz = sh.nnz
    if nnz == 0:
        mt.to_python()
        return
    dev = torch.device(device)
    cuda = dev.type == 'cuda'
    if cuda and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    B = int(batch_size)
    n_batches = (nnz + B - 1) // B
    cb = int(chunk_batches or EPOCH_CHUNK_BATCHES)
    n_chunks = (n_batches + cb - 1) // cb
    image = torch.empty(3, nnz, dtype=torch.int32, device=dev)
    span = min(cb * B, nnz)
    stage = [torch.empty(3, span, dtype=torch.int32, pin_memory=cuda) for _ in range(min(2, n_chunks))]
    side = torch.cuda.Stream(device=dev) if cuda else None
    if cuda:
        # the image's block may be one the consumer's stream is still reading under its previous owner (the last steps of the previous epoch
        # read THAT epoch's image): uploads start behind everything the consumer has enqueued so far, and the block is not handed out again
        # before the copy stream is done with it
        side.wait_stream(torch.cuda.current_stream(dev))
        image.record_stream(side)
    copied = [None] * len(stage)                      # event of the last copy out of each staging buffer
    ready = queue.Queue()
    cancel = threading.Event()                        # (never skips sampling: the RNG stream must end where the reference's does)
    t_prod = [0.0]

    def produce():
        try:
            if cuda:
                torch.cuda.set_device(dev)
            for c in range(n_chunks):
                t0 = time.perf_counter()
                b0 = c * cb * B
                cnt = min(cb * B, nnz - b0)
                buf = stage[c % len(stage)]
                if copied[c % len(stage)] is not None:
                    copied[c % len(stage)].synchronize()          # the copy that last read this staging buffer (blocks the producer only)
                out = buf.numpy()[:, :cnt] if cnt == span else np.empty((3, cnt), np.int32)
                sh.batch(mt, b0, cnt, out=out)
                if n_users is not None and (int(out[0].max()) >= n_users or int(out[0].min()) < 0):
                    raise IndexError('sampler produced a user index outside the embedding table')
                if n_items is not None and (int(out[1:].max()) >= n_items or int(out[1:].min()) < 0):
                    raise IndexError('sampler produced an item index outside the embedding table')
                if cnt != span:
                    buf.numpy()[:, :cnt] = out
                ev = None
                if not cancel.is_set():                           # a closed consumer needs no more uploads
                    if cuda:
                        with torch.cuda.stream(side):
                            image[:, b0:b0 + cnt].copy_(buf[:, :cnt], non_blocking=True)
                            ev = side.record_event()
                        copied[c % len(stage)] = ev
                    else:
                        image[:, b0:b0 + cnt].copy_(buf[:, :cnt])
                t_prod[0] += time.perf_counter() - t0
                ready.put((c, ev))
            ready.put(('done', None))
        except BaseException as e:                                # surfaces in the consumer
            ready.put(('error', e))

    def drain():
        """Let the producer finish the epoch (the RNG stream's end point), then hand the state back to Python's `random`."""
        if th is not None:
            cancel.set()
            th.join()
        mt.to_python()
        if stats is not None:
            stats['producer_seconds'] = t_prod[0]
            stats['chunks'] = n_chunks

    th = None
    try:
        if n_chunks == 1:
            produce()                                             # small epochs: no thread
        else:
            th = threading.Thread(target=produce, name='arl-sampler', daemon=True)
            th.start()
        b = 0
        for c in range(n_chunks):
            tag, ev = ready.get()
            if tag == 'error':
                raise ev
            if ev is not None:
                torch.cuda.current_stream(dev).wait_event(ev)     # the consumer's stream, not the host, waits for the upload
            if c == 0 and stats is not None:
                stats['first_batch_seconds'] = time.perf_counter() - t_begin
            end = min(b + cb * B, nnz)
            while b < end:
                cnt = min(B, nnz - b)
                yield image[0, b:b + cnt], image[1, b:b + cnt], image[2, b:b + cnt]
                b += cnt
    finally:
        drain()
