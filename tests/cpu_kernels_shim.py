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


def spmm(A, X, alpha=1.0, beta=0.0, Z=None, out=None):
    assert X.shape[0] == A.n_cols
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
