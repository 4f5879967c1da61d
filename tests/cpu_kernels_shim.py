"""Test double for `arlib_amd.ops` on CPU tensors, backed by the oracle: lets the world_size-2 gloo tests exercise the
user-sharded engine's shard arithmetic and collectives without a GPU.  Test infrastructure only."""
import numpy as np
import torch
from oracle import oracle as O


class CSRGraph:
    def __init__(self, rowptr, col, val, device, chunk=512, validate=True, n_cols=None):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float32)
        self.n_rows = len(self.rowptr) - 1
        self.n_cols = self.n_rows if n_cols is None else n_cols
        self.nnz = len(self.col)
        assert self.nnz == 0 or (self.col.min() >= 0 and self.col.max() < self.n_cols)


def spmm(A, X, alpha=1.0, beta=0.0, Z=None, out=None, row_scale=None, rows_from=0):
    assert X.shape[0] == A.n_cols
    if row_scale is not None:       # alpha * diag(row_scale) (A X) + beta * Z
        y = O.spmm((A.rowptr, A.col, A.val), X.numpy()).astype(np.float64) * (alpha * row_scale.numpy().astype(np.float64))[:, None]
        if Z is not None and beta != 0.0:
            y += beta * Z.numpy()
        y = y.astype(np.float32)
    else:
        y = O.spmm((A.rowptr, A.col, A.val), X.numpy(), alpha, beta, Z.numpy() if (Z is not None and beta != 0.0) else None)
    if out is None:
        return torch.from_numpy(y)
    out.copy_(torch.from_numpy(y))
    return out


def bpr_l2_partial(emb, item_off, u, p, n, B_global, workspace, sums_out):
    e = emb.numpy().astype(np.float64)
    ui, pi, ni = u.numpy(), p.numpy() + item_off, n.numpy() + item_off
    x = (e[ui] * e[pi]).sum(1) - (e[ui] * e[ni]).sum(1)
    s = 1.0 / (1.0 + np.exp(-x))
    workspace[:len(ui)] = torch.from_numpy((-(s * (1 - s)) / ((1e-7 + s) * B_global)).astype(np.float32))
    sums_out[0] = float(np.sum(-np.log(1e-7 + s))); sums_out[1] = float((e[ui] ** 2).sum()); sums_out[2] = float((e[pi] ** 2).sum())
    return sums_out


def bpr_l2_backward(emb, item_off, u, p, n, reg, norms4, G, workspace, upstream=1.0):
    e = emb.numpy().astype(np.float64)
    ui, pi, ni = u.numpy(), p.numpy() + item_off, n.numpy() + item_off
    g = workspace[:len(ui)].numpy().astype(np.float64)[:, None]
    nu, np_ = float(norms4[2]), float(norms4[3])
    Gn = np.zeros(e.shape, np.float64)
    np.add.at(Gn, ui, g * (e[pi] - e[ni]) + reg * e[ui] / nu)
    np.add.at(Gn, pi, g * e[ui] + reg * e[pi] / np_)
    np.add.at(Gn, ni, -g * e[ui])
    G.add_(torch.from_numpy(Gn.astype(np.float32)))
    return G


def adam_dense(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8):
    pn, mn, vn = p.numpy(), m.numpy(), v.numpy()           # share memory with the tensors
    O.adam_step(pn, np.ascontiguousarray(g.numpy()), mn, vn, lr, step, betas[0], betas[1], eps)


# ---- sparse-batch primitives (same contracts as arlib_amd.ops)
def spmm_rows(A, X, rows, layers=(), alpha=1.0, nsplit=16, out=None, workspace=None, check_range=True, row_weight=None):
    r = rows.numpy().astype(np.int64)
    y = O.spmm((A.rowptr, A.col, A.val), X.numpy())[r].astype(np.float64)
    for t in layers:
        y += t.numpy()[r]
    if row_weight is not None:
        y *= row_weight.numpy().astype(np.float64)[:, None]
    res = torch.from_numpy((alpha * y).astype(np.float32))
    if out is not None:
        out.copy_(res)
        return out
    return res


def spmm_flagged(A, X, xflags=None, alpha=1.0, beta=0.0, Z=None, zflags=None, out=None):
    Xn = X.numpy()
    if xflags is not None:          # bitmap
        bits = xflags.numpy().view(np.uint32)
        rows = np.arange(Xn.shape[0])
        on = (bits[rows >> 5] >> (rows & 31)) & 1
        assert np.all(Xn[on == 0] == 0), 'operand must be zero on rows whose bit is clear'
    Zn = None
    if beta != 0.0:
        Zn = Z.numpy().copy()
        if zflags is not None:
            assert np.all(Zn[zflags.numpy() == 0] == 0), 'Z must be zero on unflagged rows'
    y = torch.from_numpy(O.spmm((A.rowptr, A.col, A.val), Xn, alpha, beta, Zn))
    if out is None:
        return y
    out.copy_(y)
    return out


def spmm_adam(A, X, alpha, beta, Z, P, M, V, lr, step, betas=(0.9, 0.999), eps=1e-8, zflags=None):
    g = O.spmm((A.rowptr, A.col, A.val), X.numpy(), alpha, beta, Z.numpy() if beta != 0.0 else None)
    p, m, v = (np.ascontiguousarray(t.numpy()) for t in (P, M, V))
    O.adam_step(p, g, m, v, lr, step, betas[0], betas[1], eps)
    P.copy_(torch.from_numpy(p)); M.copy_(torch.from_numpy(m)); V.copy_(torch.from_numpy(v))


def bpr_l2_fwd_bwd(emb, item_off, u, p, n, reg, G=None, upstream=1.0, workspace=None, loss_out=None, check_range=True, distinct_rows=False):
    lb, lr_, Gn = O.bpr_l2(emb.numpy(), item_off, u.numpy(), p.numpy(), n.numpy(), reg)
    if G is not None:
        G.add_(torch.from_numpy(Gn))
    e = emb.numpy().astype(np.float64)
    loss_out[0], loss_out[1] = lb, lr_
    loss_out[2] = float(np.sqrt((e[u.numpy()] ** 2).sum())); loss_out[3] = float(np.sqrt((e[p.numpy() + item_off] ** 2).sum()))
    return loss_out


def gather_rows(src, idx, check_range=True):
    return src[idx.long()].clone()


def scatter_add_rows(dst, idx, src, scale=1.0, check_range=True):
    dst.index_add_(0, idx.long(), src * scale)
    return dst


def mark_rows_(flags, idx, value, check_range=True):
    flags[idx.long()] = value
    return flags


def mark_bits_(bits, idx, set_, n_nodes, check_range=True):
    b = bits.numpy().view(np.uint32)
    for i in np.unique(idx.numpy()).astype(np.int64):
        w, m = int(i) >> 5, 1 << (int(i) & 31)
        b[w] = np.uint32((int(b[w]) | m) if set_ else (int(b[w]) & (~m & 0xFFFFFFFF)))
    return bits


def zero_rows_(dst, idx, check_range=True):
    dst[idx.long()] = 0
    return dst


def simgcl_perturb_(E, noise, eps):
    E.copy_(torch.from_numpy(O.simgcl_perturb(E.numpy().copy(), noise.numpy(), eps)))
    return E


def infonce_fwd_bwd(v1, v2, tau, want_grad=True, upstream=1.0):
    loss, d1, d2 = O.infonce(v1.numpy(), v2.numpy(), tau, want_grad)
    t = lambda a: None if a is None else torch.from_numpy(a * np.float32(upstream))
    return torch.tensor([loss], dtype=torch.float32), t(d1), t(d2)


def score_mask_topk(Pu, Pi, k, mask_rowptr=None, mask_col=None, exact=False, warm_idx=None, item_order=None):
    mask = None if mask_rowptr is None else (mask_rowptr.numpy().astype(np.int64), mask_col.numpy())
    idx, val = O.score_mask_topk(Pu.numpy(), Pi.numpy(), k, mask)
    return torch.from_numpy(idx.astype(np.int32)), torch.from_numpy(val.astype(np.float32))


class SfaStages:
    """numpy restatement of arl_sfa_stage{1,2,3}_f32 (the weighted closed form that tests/test_oracle_attacks.py pins against the literal
    reverse pass): partial sums over the given rows; the caller all-reduces between the stages."""

    def __init__(self, X, w, r0):
        self.X, self.w, self.r0 = X.numpy().astype(np.float64), w.numpy().astype(np.float64), r0.numpy().astype(np.float64)

    def stage1(self):
        self.q = self.X @ self.r0
        return torch.from_numpy((self.X.T @ (self.w * self.q)).astype(np.float32))

    def stage2(self, r):
        rr = r.numpy().astype(np.float64)
        self.s = self.X @ rr
        a = self.X.T @ (self.w * np.sign(self.s))
        return torch.from_numpy(np.concatenate([a, [(self.w * np.abs(self.s)).sum()]]).astype(np.float32))

    def stage3(self, r, a_s, numel_h, out=None, scale=1.0, accumulate=False):
        rr, a, S = r.numpy().astype(np.float64), a_s.numpy()[:-1].astype(np.float64), float(a_s.numpy()[-1])
        A, Q = np.abs(rr).sum(), rr @ rr
        g_r = ((A / Q) * a + (S / Q) * np.sign(rr) - (2 * S * A / Q ** 2) * rr) / numel_h
        G = self.w[:, None] * ((A / (numel_h * Q)) * np.sign(self.s)[:, None] * rr[None, :] + self.q[:, None] * g_r[None, :] + (self.X @ g_r)[:, None] * self.r0[None, :])
        G = torch.from_numpy((scale * G).astype(np.float32))
        if out is not None:
            out.copy_(out + G if accumulate else G)
            G = out
        return torch.tensor([S * A / (numel_h * Q)], dtype=torch.float32), G


def ngcf_combine(P, E, out=None):
    return torch.cat([P + E, P * E], 1)


def ngcf_act_(Z, acc=None, slope=0.01):
    Z.copy_(torch.where(Z > 0, Z, slope * Z))
    if acc is not None:
        acc.add_(Z)
    return Z


def ngcf_act_bwd(gOut, Out, slope=0.01):
    return gOut * torch.where(Out > 0, torch.ones_like(Out), torch.full_like(Out, slope))


def ngcf_combine_bwd(gST, P, E):
    d = P.shape[1]
    gS, gT = gST[:, :d], gST[:, d:]
    return gS + gT * E, gS + gT * P


def rows_axpy_unique_(dst, src, idx, alpha=1.0, check_range=True, dup_bits=None):
    r = torch.unique(idx.long())
    dst[r] += alpha * src[r]
    return dst


def shard_batch_prep(u, p, n, u0, u1, out=None):
    Ul = u1 - u0
    lu = (u - u0).clamp(0, max(Ul - 1, 0)).to(torch.int32)
    own = torch.cat([((u >= u0) & (u < u1)).to(torch.float32), torch.ones(2 * u.numel())])
    item_rows = torch.cat([p, n]).to(torch.int32)
    return lu, own, item_rows, torch.cat([lu, item_rows + Ul]).to(torch.int32)


def batch_rows_set_(G, flags, bits, idx, src, scale=1.0, check_range=True, row_scale=None, dup_bits=None):
    if row_scale is not None:
        src = src * row_scale[:, None]
    scatter_add_rows(G, idx, src, scale)
    mark_rows_(flags, idx, 1)
    mark_bits_(bits, idx, True, G.shape[0])


def batch_rows_clear_(G, flags, bits, idx, check_range=True, dup_bits=None):
    zero_rows_(G, idx)
    mark_rows_(flags, idx, 0)
    mark_bits_(bits, idx, False, G.shape[0])


def sddmm_rows_dense(dY, X, rows, col_off, n_cols, out=None):
    res = O.sddmm_rows_dense(dY.numpy(), X.numpy(), rows.numpy(), col_off, n_cols, out=None if out is None else out.numpy())
    return out if out is not None else torch.from_numpy(res)


def pga_update_(S, grad, dinv_rows=None, dinv_cols=None):
    S.copy_(torch.from_numpy(O.pga_update(S.numpy(), grad.numpy(), None if dinv_rows is None else dinv_rows.numpy(), None if dinv_cols is None else dinv_cols.numpy())))
    return S


def cw_topk_term(X, n_user_rows, n_real, top_idx, targets, c=None, want_w=True, check_range=True):
    """float64 restatement of ops.cw_topk_term (attack/White/CLeaR.py:83-95): loss, dL/dX on every row, SFA row multiplicities."""
    Xn = X.numpy().astype(np.float64)
    Up, n_real = int(n_user_rows), int(n_real)
    k, tg = top_idx.shape[1], targets.numpy().astype(np.int64)
    T = len(tg)
    c = 1.0 / (max(n_real, 1) * T) if c is None else float(c)
    neg = top_idx.numpy()[:n_real][:, [k - 1 - t for t in range(T)]].astype(np.int64).reshape(n_real, T)
    G = np.zeros_like(Xn); loss = 0.0
    for t in range(T):
        G[:n_real] += c * (Xn[Up + neg[:, t]] - Xn[Up + tg[t]])
        np.add.at(G, Up + neg[:, t], c * Xn[:n_real])
        G[Up + tg[t]] -= c * Xn[:n_real].sum(0)
        loss += c * ((Xn[:n_real] * Xn[Up + neg[:, t]]).sum() - (Xn[:n_real] * Xn[Up + tg[t]]).sum())
    w = np.zeros(Xn.shape[0]); w[:n_real] = T
    np.add.at(w, Up + neg.reshape(-1), 1.0); np.add.at(w, Up + tg, float(n_real))
    return (torch.tensor([loss], dtype=torch.float32), torch.from_numpy(G.astype(np.float32)), torch.from_numpy(w.astype(np.float32)) if want_w else None)
