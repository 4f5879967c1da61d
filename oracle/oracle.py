"""Python face of the CPU oracle (ctypes over oracle/liboracle.so + numpy glue).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/arl_oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from arlib_amd/.

Parity pin: tests/test_oracle_golden.py checks every entry point against vectors captured from the
reference itself (tests/golden/gen_golden.py).  Reference citations are relative to /root/reference.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'arl_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'liboracle.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_mt_random.restype = C.c_double
        _LIB.orc_mt_randbelow.restype = C.c_uint32
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ------------------------------------------------------------------ CPython random (MT19937)
def mt_seed(seed):
    """random.seed(int) (util/tool.py:102).  Returns the 625-word state (= random.getstate()[1])."""
    seed = abs(int(seed))
    key = []
    while True:
        key.append(seed & 0xFFFFFFFF)
        seed >>= 32
        if not seed:
            break
    key = np.array(key, np.uint32)
    st = np.zeros(625, np.uint32)
    lib().orc_mt_seed_by_array(_p(st), _p(key), C.c_int64(len(key)))
    return st


def mt_random(st):
    return lib().orc_mt_random(_p(st))


def mt_randbelow(st, n):
    return lib().orc_mt_randbelow(_p(st), C.c_uint32(n))


def build_membership(pairs, n_users):
    """training_set_u as a CSR with sorted, de-duplicated item ids (util/DataLoader.py:41)."""
    pairs = np.asarray(pairs, np.int64)
    key = np.unique(pairs[:, 0] * (1 << 32) + pairs[:, 1])
    u = key >> 32
    items = (key & 0xFFFFFFFF).astype(np.int32)
    rowptr = np.zeros(n_users + 1, np.int64)
    np.add.at(rowptr, u + 1, 1)
    return np.cumsum(rowptr), items


def shuffle_pairs(st, pairs):
    assert pairs.dtype == np.int32 and pairs.flags.c_contiguous
    lib().orc_shuffle_pairs(_p(st), _p(pairs), C.c_int64(pairs.shape[0]))


def sample_batch(st, pairs, begin, count, n_items, memb):
    u = np.empty(count, np.int32); p = np.empty(count, np.int32); n = np.empty(count, np.int32)
    rowptr, items = memb
    lib().orc_sample_batch(_p(st), _p(pairs), C.c_int64(begin), C.c_int64(count), C.c_int32(n_items),
                           _p(rowptr), _p(items), C.c_int64(len(rowptr) - 1), _p(u), _p(p), _p(n))
    return u, p, n


def next_batch_pairwise(st, pairs, batch_size, n_items, memb):
    """util/sampler.py:4-30 on int arrays; `pairs` is shuffled in place like data.training_data."""
    shuffle_pairs(st, pairs)
    nnz = pairs.shape[0]
    b = 0
    while b < nnz:
        cnt = min(batch_size, nnz - b)
        yield sample_batch(st, pairs, b, cnt, n_items, memb)
        b += cnt


# ------------------------------------------------------------------ adjacency
def bipartite_csr(u, i, U, I, w=None):
    """(U+I)^2 symmetric adjacency R (+) R^T as CSR with sorted columns; duplicate (u,i) pairs are summed
    exactly like scipy's csr_matrix((ratings,(row,col))) does (util/DataLoader.py:57-71)."""
    u = np.asarray(u, np.int64); i = np.asarray(i, np.int64)
    w = np.ones(len(u), np.float32) if w is None else np.asarray(w, np.float32)
    N = U + I
    r = np.concatenate([u, i + U]); c = np.concatenate([i + U, u]); ww = np.concatenate([w, w])
    key = r * N + c
    order = np.argsort(key, kind='stable')
    key = key[order]; ww = ww[order]
    uniq, start = np.unique(key, return_index=True)
    wsum = np.add.reduceat(ww.astype(np.float64), start).astype(np.float32) if len(key) else ww
    rows = uniq // N
    col = (uniq % N).astype(np.int32)
    rowptr = np.zeros(N + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr), col, wsum


def norm_adj_values(rowptr, col, w):
    """util/DataLoader.py:73-87 / recommender/LightGCN.py:212-215."""
    val = np.empty(len(col), np.float32)
    lib().orc_norm_adj_values(C.c_int64(len(rowptr) - 1), _p(_i64(rowptr)), _p(_i32(col)), _p(_f32(w)), _p(val))
    return val


# ------------------------------------------------------------------ propagation
def spmm(csr, X, alpha=1.0, beta=0.0, Z=None, f32acc=False):
    rowptr, col, val = csr
    X = _f32(X)
    n, d = len(rowptr) - 1, X.shape[1]
    Y = np.empty((n, d), np.float32)
    fn = lib().orc_spmm_csr_f32acc if f32acc else lib().orc_spmm_csr
    fn(C.c_int64(n), _p(rowptr), _p(col), _p(val), _p(X), C.c_int64(d), C.c_float(alpha), C.c_float(beta),
       _p(_f32(Z)) if Z is not None else None, _p(Y))
    return Y


def simgcl_perturb(E, noise, eps):
    E = _f32(E).copy()
    lib().orc_simgcl_perturb(_p(E), _p(_f32(noise)), C.c_int64(E.shape[0]), C.c_int64(E.shape[1]), C.c_float(eps))
    return E


def lightgcn_forward(csr, E0, L, skip0=False, noises=None, eps=0.1, return_layers=False, f32acc=False):
    """recommender/LightGCN.py:230-240 (mean of E_0..E_L); skip0=True is SimGCL (recommender/SimGCL.py:198-210:
    mean of E_1..E_L, optional additive perturbation after each hop)."""
    E = _f32(E0)
    layers = [E]
    acc = np.zeros_like(E, dtype=np.float64) if skip0 else E.astype(np.float64)
    for k in range(L):
        E = spmm(csr, E, f32acc=f32acc)
        if noises is not None:
            E = simgcl_perturb(E, noises[k], eps)
        layers.append(E)
        acc += E
    out = (acc / (L if skip0 else L + 1)).astype(np.float32)
    return (out, layers) if return_layers else out


def lightgcn_backward(csr, G, L, skip0=False, f32acc=False):
    """Backward of the above w.r.t. E0 (adjacency symmetric): LightGCN dE0 = (1/(L+1)) sum_{k=0..L} A^k G,
    SimGCL dE0 = (1/L) sum_{k=1..L} A^k G (noise/sign carry no gradient).  Horner form."""
    G = _f32(G)
    acc = G
    if skip0:
        for _ in range(L - 1):
            acc = spmm(csr, acc, 1.0, 1.0, G, f32acc=f32acc)
        return spmm(csr, acc, 1.0 / L, f32acc=f32acc)
    for k in range(L):
        last = (k == L - 1)
        s = 1.0 / (L + 1) if last else 1.0
        acc = spmm(csr, acc, s, s, G, f32acc=f32acc)
    if L == 0:
        return G
    return acc


# ------------------------------------------------------------------ losses / optimizers
def bpr_l2(emb, item_off, ui, pi, ni, reg, want_grad=True):
    emb = _f32(emb)
    lb = C.c_float(); lr_ = C.c_float()
    G = np.zeros_like(emb) if want_grad else None
    lib().orc_bpr_l2_fwd_bwd(_p(emb), C.c_int64(emb.shape[1]), C.c_int64(item_off), _p(_i32(ui)), _p(_i32(pi)),
                             _p(_i32(ni)), C.c_int64(len(ui)), C.c_float(reg), C.byref(lb), C.byref(lr_), _p(G))
    return lb.value, lr_.value, G


def adam_step(p, g, m, v, lr, t, b1=0.9, b2=0.999, eps=1e-8):
    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().orc_adam_step(_p(p), _p(_f32(g)), _p(m), _p(v), C.c_int64(p.size), C.c_float(lr), C.c_float(b1),
                        C.c_float(b2), C.c_float(eps), C.c_int64(t))


def sgd_step(p, g, lr):
    lib().orc_sgd_step(_p(p), _p(_f32(g)), C.c_int64(p.size), C.c_float(lr))


def infonce(v1, v2, tau, want_grad=True):
    v1 = _f32(v1); v2 = _f32(v2)
    loss = C.c_float()
    d1 = np.empty_like(v1) if want_grad else None
    d2 = np.empty_like(v2) if want_grad else None
    lib().orc_infonce_fwd_bwd(_p(v1), _p(v2), C.c_int64(v1.shape[0]), C.c_int64(v1.shape[1]), C.c_float(tau),
                              C.byref(loss), _p(d1), _p(d2))
    return loss.value, d1, d2


# ------------------------------------------------------------------ attack primitives
def sddmm_rows_dense(dY, X, rows, col_off, n_cols, out=None):
    dY = _f32(dY); X = _f32(X); rows = _i32(rows)
    if out is None:
        out = np.zeros((len(rows), n_cols), np.float32)
    lib().orc_sddmm_rows_dense(_p(dY), _p(X), C.c_int64(X.shape[1]), _p(rows), C.c_int64(len(rows)),
                               C.c_int64(col_off), C.c_int64(n_cols), _p(out))
    return out


def score_mask_topk(Pu, Pi, k, mask=None):
    Pu = _f32(Pu); Pi = _f32(Pi)
    U, I = Pu.shape[0], Pi.shape[0]
    idx = np.empty((U, k), np.int32); val = np.empty((U, k), np.float32)
    rp, mc = (None, None) if mask is None else (_i64(mask[0]), _i32(mask[1]))
    lib().orc_score_mask_topk(_p(Pu), _p(Pi), C.c_int64(U), C.c_int64(I), C.c_int64(Pu.shape[1]), _p(rp), _p(mc),
                              C.c_int64(k), _p(idx), _p(val))
    return idx, val


def topn_project_rows(M, n):
    M = _f32(M)
    out = np.empty_like(M); idx = np.empty((M.shape[0], n), np.int32)
    lib().orc_topn_project_rows(_p(M), C.c_int64(M.shape[0]), C.c_int64(M.shape[1]), C.c_int64(n), _p(out), _p(idx))
    return out, idx


def pga_update(S, grad, dinv_rows=None, dinv_cols=None):
    S = _f32(S).copy()
    lib().orc_pga_update(_p(S), _p(_f32(grad)), _p(_f32(dinv_rows)) if dinv_rows is not None else None,
                         _p(_f32(dinv_cols)) if dinv_cols is not None else None, C.c_int64(S.shape[0]), C.c_int64(S.shape[1]))
    return S


def num_threads():
    return lib().orc_num_threads()


# ------------------------------------------------------------------ whole training step (LightGCN.py:47-64)
class TrainState:
    """Tables + Adam state for the oracle's restatement of one `train()` inner-loop iteration."""

    def __init__(self, user0, item0, csr, L, reg, lr, skip0=False, optimizer='adam'):
        self.U, self.d = user0.shape
        self.I = item0.shape[0]
        self.E0 = np.concatenate([_f32(user0), _f32(item0)], 0)
        self.m = np.zeros_like(self.E0); self.v = np.zeros_like(self.E0)
        self.csr, self.L, self.reg, self.lr, self.skip0, self.optimizer = csr, L, reg, lr, skip0, optimizer
        self.t = 0
        self.f32acc = False          # True only for the timed CPU baseline (fp32 accumulators, like torch CPU)

    def forward(self):
        if self.L == 0 or self.csr is None:          # GMF: recommender/GMF.py:174-175
            return self.E0
        return lightgcn_forward(self.csr, self.E0, self.L, self.skip0, f32acc=self.f32acc)

    def grad(self, ui, pi, ni):
        out = self.forward()
        lb, lr_, G = bpr_l2(out, self.U, ui, pi, ni, self.reg)
        if self.L == 0 or self.csr is None:
            return lb + lr_, G
        return lb + lr_, lightgcn_backward(self.csr, G, self.L, self.skip0, f32acc=self.f32acc)

    def step(self, ui, pi, ni):
        loss, g = self.grad(ui, pi, ni)
        self.t += 1
        if self.optimizer == 'adam':
            adam_step(self.E0, g, self.m, self.v, self.lr, self.t)
        else:
            sgd_step(self.E0, g, self.lr)
        return loss


# ------------------------------------------------------------------ white-box attack compositions (numpy over the C pieces)
def cw_pairs(top_idx, n_real_users, targets, pop=True):
    """attack/White/PGA.py:104-108 / CLeaR.py:84-88 (`top_items[u].pop()` per target: ranks k, k-1, ...) and
    DLAttack.py:92-96 (`top_items[u][-1]`: always rank k)."""
    T, k = len(targets), top_idx.shape[1]
    users = np.repeat(np.arange(n_real_users), T)
    pos = np.tile(np.asarray(targets), n_real_users)
    ranks = np.tile(k - 1 - np.arange(T), n_real_users) if pop else np.full(n_real_users * T, k - 1)
    return users, pos, top_idx[users, ranks].astype(np.int64)


def cw_loss_grad(out, Up, users, pos, neg):
    """CWloss = mean(neg_score - pos_score) (PGA.py:109-116) and dL/d(out)."""
    import scipy.sparse as sp
    users = np.asarray(users, np.int64); pos = np.asarray(pos, np.int64); neg = np.asarray(neg, np.int64)
    ue, pe, ne = out[users], out[Up + pos], out[Up + neg]
    loss = float(np.mean((ue.astype(np.float64) * ne).sum(1) - (ue.astype(np.float64) * pe).sum(1)))
    c = 1.0 / len(users)
    # the three scatter-adds (du += c (n - p), dn += c u, dp -= c u) as one sparse product: duplicates are summed by the
    # COO -> CSR conversion, in float64 (np.add.at does the same arithmetic two orders of magnitude slower)
    n = out.shape[0]
    rows = np.concatenate([users, users, Up + neg, Up + pos])
    cols = np.concatenate([Up + neg, Up + pos, users, users])
    vals = np.concatenate([np.full(len(users), c), np.full(len(users), -c)] * 2)
    G = sp.csr_matrix((vals, (rows, cols)), shape=(n, n)) @ out.astype(np.float64)
    return loss, np.asarray(G, np.float32)


def sfa_l1_loss_grad(H, r0):
    """spectral_feature_augmentation(H, 1) + F.l1_loss(SFA, H) (CLeaR.py:98-125) on the explicit H, and dloss/dH by a literal
    reverse pass over the same graph (nothing is detached in the reference, r included).  float64 throughout."""
    H = np.asarray(H, np.float64); r0 = np.asarray(r0, np.float64)
    q = H @ r0
    r = H.T @ q                       # k = 1 power iteration: r = H^T H r0
    Rm = np.outer(r, r)
    P = H @ Rm
    n2 = float(r @ r)                 # torch.norm(r) ** 2
    Xm = P / n2
    D = (H - Xm) - H                  # SFA - target of the l1_loss
    loss = float(np.abs(D).mean())
    gD = np.sign(D) / D.size
    gXm = -gD                         # d/dH through `H -` and through the l1 target cancel exactly
    gP = gXm / n2
    gn2 = -float((gXm * P).sum()) / (n2 * n2)
    gH = gP @ Rm.T
    gRm = H.T @ gP
    gr = (gRm + gRm.T) @ r + 2.0 * r * gn2
    gH += np.outer(q, gr)             # r = H^T q
    gq = H @ gr
    gH += np.outer(gq, r0)            # q = H r0
    return loss, gH


def clear_loss_grad(out, Up, users, pos, neg, r0):
    """lossall = CWloss + sfaloss of one CLeaR surrogate step (CLeaR.py:89-126) and dlossall/d(out); H = cat(u, p, n) rows."""
    cw, G = cw_loss_grad(out, Up, users, pos, neg)
    rows = np.concatenate([np.asarray(users), Up + np.asarray(pos), Up + np.asarray(neg)])
    sfa, gH = sfa_l1_loss_grad(out[rows], r0)
    G = G.astype(np.float64)
    np.add.at(G, rows, gH)
    return cw, sfa, G.astype(np.float32)


def pga_weighted_graph(real_indptr, real_indices, U, F, I, S):
    """(U+F+I)^2 normalised adjacency of the real interactions plus the weighted fake block S (entries with S != 0 only:
    scipy's `ui_adj + ui_adj.T` drops explicit zeros) -- what PGA.py:93-97 hands to _init_uiAdj.  Returns (csr, dinv)."""
    ru = np.repeat(np.arange(U), np.diff(real_indptr[:U + 1]))
    fu, fi = np.nonzero(S)
    u = np.concatenate([ru, fu + U]); i = np.concatenate([real_indices[:len(ru)], fi])
    w = np.concatenate([np.ones(len(ru), np.float32), S[fu, fi].astype(np.float32)])
    rowptr, col, ww = bipartite_csr(u, i, U + F, I, w)
    val = norm_adj_values(rowptr, col, ww)
    rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))
    deg = np.zeros(len(rowptr) - 1, np.float32)
    np.add.at(deg, rows, ww)
    with np.errstate(divide='ignore'):
        dinv = np.where(deg > 0, 1.0 / np.sqrt(deg), 0.0).astype(np.float32)
    return (rowptr, col, val), dinv


def pga_step(real_indptr, real_indices, U, F, I, S, E0, L, users, pos, neg):
    """One projected-gradient step on the fake block (PGA.py:92-142).  Returns (scaled gradient block, new S, CW loss)."""
    csr, dinv = pga_weighted_graph(real_indptr, real_indices, U, F, I, S)
    Up = U + F
    out, E = lightgcn_forward(csr, E0, L, return_layers=True)
    loss, G = cw_loss_grad(out, Up, users, pos, neg)
    s = 1.0 / (L + 1)
    Gs = (G * s).astype(np.float32)
    dE = [None] * (L + 1)
    dE[L] = Gs
    for k in range(L - 1, 0, -1):
        dE[k] = spmm(csr, dE[k + 1], 1.0, 1.0, Gs)
    rows = np.arange(U, Up, dtype=np.int32)
    block = np.zeros((F, I), np.float32)
    for k in range(L):
        sddmm_rows_dense(dE[k + 1], E[k], rows, Up, I, out=block)
        sddmm_rows_dense(E[k], dE[k + 1], rows, Up, I, out=block)
    grad = block * dinv[U:Up, None] * dinv[None, Up:] * (S != 0)
    return grad, pga_update(S, block, dinv[U:Up], dinv[Up:]), loss


# ------------------------------------------------------------------ NGCF forward (recommender/NGCF.py:197-212)
def ngcf_forward(csr, E0, W1s, W2s, slope=0.01):
    """per layer: t = E W1; E' = leaky_relu(A t + t + ((A E) * E) W2); mean of L+1 layers -- written with BOTH sparse hops,
    exactly as the reference does (the product uses A(E W1) = (A E) W1 to save one)."""
    E = _f32(E0)
    acc = E.astype(np.float64)
    for W1, W2 in zip(W1s, W2s):
        t = (E.astype(np.float64) @ W1.astype(np.float64)).astype(np.float32)
        z = spmm(csr, t).astype(np.float64) + t + ((spmm(csr, E).astype(np.float64) * E) @ W2.astype(np.float64))
        E = np.where(z > 0, z, slope * z).astype(np.float32)
        acc += E
    return (acc / (len(W1s) + 1)).astype(np.float32)
