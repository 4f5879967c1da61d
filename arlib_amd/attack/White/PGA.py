"""PGA -- projected-gradient poisoning attack; mirror of the reference's attack/White/PGA.py on the MI355X kernels.

Same protocol and numbers (posionDataAttack(recommender) -> scipy (U+F) x I matrix), different mechanics:
  * the reference rebuilds the (U+F+I)^2 adjacency in scipy and re-uploads a COO tensor for every one of the
    ceil(I/128) gradient steps (PGA.py:93-97).  Here the pattern (real edges + DENSE fake-user rows/columns) is
    built once per outer epoch; a step only rewrites the 2*F*I fake-edge weights on the device and re-normalises
    there (arl_norm_adj_values_f32).  Zero-weight fake edges contribute nothing, so results are identical.
  * the reference takes autograd.grad w.r.t. ALL E adjacency values, densifies an N x N matrix and slices F rows
    (PGA.py:117-134).  Here only the F x I block is ever computed: sum over layers of two row-restricted SDDMMs,
      grad[f,j] = dinv[f] dinv[U'+j] ( sum_k <dE_{k+1}[f], E_k[U'+j]> + <dE_{k+1}[U'+j], E_k[f]> ),
    with dE_k = G/(L+1) + A dE_{k+1} (adjacency symmetric) and G the gradient of the CW loss w.r.t. the output.
  * posionDataAttack applies the poisoned operator in factors (FactoredFakeGraph): the real edges keep fixed values and a fixed hop plan,
    the fake block is dense algebra, so a step rewrites nothing of size nnz.  FakeBlockGraph (all edges in one re-normalised CSR) is the
    form the reference-trace test drives; both are tested against each other.
"""
from copy import deepcopy

import numpy as np
import scipy.sparse as sp
import torch

from ... import ops
from .._common import AttackBase, DEVICE, symmetric_adjacency, init_graph, rebuild_interaction_matrix, reinit_with_tables, cw_pairs


class FakeBlockGraph:
    """Device CSR of the (U'+I)^2 adjacency (U' = U + F) whose F fake-user rows (and the matching F columns of every item
    row) are dense, so the fake block S [F, I] maps to two fixed index sets of the edge-weight array."""

    def __init__(self, ui_real, n_real, n_fake, n_items, device=DEVICE, emb_size=None):
        U, F, I = int(n_real), int(n_fake), int(n_items)
        Up = U + F
        R = sp.csr_matrix(ui_real, dtype=np.float32)[:U]
        R.eliminate_zeros(); R.sort_indices()
        Rt = R.T.tocsr(); Rt.sort_indices()
        deg_u = np.diff(R.indptr).astype(np.int64)
        deg_i = np.diff(Rt.indptr).astype(np.int64)
        rowptr = np.zeros(Up + I + 1, np.int64)
        np.cumsum(np.concatenate([deg_u, np.full(F, I, np.int64), deg_i + F]), out=rowptr[1:])
        nnz = int(rowptr[-1])
        col = np.empty(nnz, np.int32); w = np.zeros(nnz, np.float32)
        col[:R.nnz] = R.indices + Up; w[:R.nnz] = R.data
        fb = int(rowptr[U])
        col[fb:fb + F * I] = np.tile(np.arange(I, dtype=np.int32) + Up, F)
        ib = int(rowptr[Up])
        # item rows: real users ascending, then the F fake users
        item_start = rowptr[Up:Up + I] - ib
        real_pos = (np.arange(Rt.nnz, dtype=np.int64) + np.repeat(np.arange(I, dtype=np.int64) * F, deg_i)) + ib
        col[real_pos] = Rt.indices; w[real_pos] = Rt.data
        fake_pos = (rowptr[Up + 1:Up + I + 1] - F)[None, :] + np.arange(F, dtype=np.int64)[:, None]     # [F, I]
        col[fake_pos] = (U + np.arange(F, dtype=np.int32))[:, None]
        self.U, self.F, self.I, self.Up, self.N = U, F, I, Up, Up + I
        self.device = torch.device(device)
        self.rowptr_d = torch.from_numpy(rowptr.astype(np.int32)).to(self.device)
        self.col_d = torch.from_numpy(col).to(self.device)
        self.erow_d = torch.repeat_interleave(torch.arange(Up + I, dtype=torch.int32, device=self.device), (self.rowptr_d[1:] - self.rowptr_d[:-1]).long(), output_size=nnz)
        self.w = torch.from_numpy(w).to(self.device)
        self.fwd_lo = fb
        self.bwd_idx = torch.from_numpy(fake_pos.reshape(-1)).to(self.device)
        base = np.zeros(Up + I, np.float32); base[:U] = np.asarray(R.sum(1)).ravel(); base[Up:] = np.asarray(R.sum(0)).ravel()
        self.base_rowsum = torch.from_numpy(base).to(self.device)          # weighted degrees of the real interactions
        self.graph = ops.CSRGraph(rowptr, self.col_d, torch.zeros(nnz, device=self.device), self.device)
        if emb_size is not None:            # the pattern never changes: large graphs get the register-blocked hop plan once (the dense fake rows are dealt as strided pieces)
            ops.auto_blocked(self.graph, emb_size, split=Up)
        self.fake_rows = torch.arange(U, Up, dtype=torch.int32, device=self.device)
        self.dinv = None

    def set_block(self, S):
        """Write S [F, I] into both directions of the fake edges and re-normalise on device (LightGCN.py:212-215)."""
        flat = S.reshape(-1)
        self.w[self.fwd_lo:self.fwd_lo + flat.numel()] = flat
        self.w[self.bwd_idx] = flat
        # row sums of the fixed pattern: real users keep their degree, a fake user has its S row, an item its real degree + its S column
        # (PGA.py:93-97 recomputes all 77 M-edge sums with scipy); then the edge-parallel value pass
        rs = self.base_rowsum.clone()
        rs[self.U:self.Up] = S.sum(1)
        rs[self.Up:] += S.sum(0)
        self.dinv = torch.where(rs > 0, 1.0 / torch.sqrt(rs), torch.zeros_like(rs))
        val = ops.norm_vals_coo(self.erow_d, self.col_d, self.w, self.dinv)
        self.graph = self.graph.with_values(val)
        return self.graph


class FactoredFakeGraph:
    """The same operator as FakeBlockGraph.set_block(S).graph, applied in factors:  A_hat X = D^-1/2 ( W_real (D^-1/2 X) + fake block ).
    W_real is the un-normalised adjacency of the REAL interactions only (fixed: its blocked hop plan and its values are built once), the
    F x I fake block enters as two small dense products (S (D^-1/2 X)_items for the fake users' rows, S^T (D^-1/2 X)_fake for the items'),
    and the degrees follow from S in O(F I).  Nothing of size nnz is touched when S changes: no 77 M-edge renormalisation, no
    re-binding of the plan's values, and the hops do not carry the 2 F I fake edges through the sparse kernels.  Row-scalings are
    element-wise passes over the [N, d] operand.  Results agree with FakeBlockGraph to fp32 rounding (different association)."""

    def __init__(self, ui_real, n_real, n_fake, n_items, device=DEVICE, emb_size=None):
        U, F, I = int(n_real), int(n_fake), int(n_items)
        Up = U + F
        R = sp.csr_matrix(ui_real, dtype=np.float32)[:U]
        R.eliminate_zeros(); R.sort_indices()
        Rt = R.T.tocsr(); Rt.sort_indices()
        rowptr = np.zeros(Up + I + 1, np.int64)
        np.cumsum(np.concatenate([np.diff(R.indptr), np.zeros(F, np.int64), np.diff(Rt.indptr)]), out=rowptr[1:])
        col = np.concatenate([R.indices.astype(np.int32) + Up, Rt.indices.astype(np.int32)])
        val = np.concatenate([R.data, Rt.data]).astype(np.float32)
        self.U, self.F, self.I, self.Up, self.N = U, F, I, Up, Up + I
        self.device = torch.device(device)
        self.W = ops.CSRGraph(rowptr, col, val, self.device, validate=False)
        if emb_size is not None:
            ops.auto_blocked(self.W, emb_size, split=Up)
        base = np.zeros(Up + I, np.float32); base[:U] = np.asarray(R.sum(1)).ravel(); base[Up:] = np.asarray(R.sum(0)).ravel()
        self.base_rowsum = torch.from_numpy(base).to(self.device)
        self.fake_rows = torch.arange(U, Up, dtype=torch.int32, device=self.device)
        self.S = self.dinv = None

    def set_block(self, S):
        self.S = S if S.is_contiguous() else S.contiguous()
        rs = self.base_rowsum.clone()
        rs[self.U:self.Up] = S.sum(1)
        rs[self.Up:] += S.sum(0)
        self.dinv = torch.where(rs > 0, 1.0 / torch.sqrt(rs), torch.zeros_like(rs))
        self._dcol = self.dinv[:, None].contiguous()
        return self

    def hop(self, X, alpha=1.0, beta=0.0, Z=None, items_only=False):
        """alpha * (A_hat @ X) + beta * Z.   items_only: the caller reads the fake users' and the items' rows only (the last backward hop of a
        PGA step: its output feeds nothing but the F x I block gradient) -- the real users' rows may be left unwritten, which saves the
        user-row launch of the blocked plan (half a hop)."""
        Xs = X * self._dcol
        U, Up = self.U, self.Up
        Y = ops.spmm(self.W, Xs, alpha, beta, Z, row_scale=self.dinv, rows_from=Up if items_only else 0)    # alpha D^-1/2 (W Xs) + beta Z in the epilogue
        if items_only:                                                     # W has no edges on the fake users' rows: what the skipped launch would have written
            if beta != 0.0:
                torch.mul(Z[U:Up], beta, out=Y[U:Up])
            else:
                Y[U:Up].zero_()
        # the fake block as two hand-written dense products (row slices of contiguous tables are contiguous)
        ops.fake_block_rows_(self.S, Xs[Up:], Y[U:Up], rscale=self.dinv[U:Up], alpha=alpha)
        ops.fake_block_cols_(self.S, Xs[U:Up], Y[Up:], rscale=self.dinv[Up:], alpha=alpha)
        return Y


def _hop(graph, X, alpha=1.0, beta=0.0, Z=None, items_only=False):
    """One application of the poisoned normalised adjacency: a CSRGraph (FakeBlockGraph) or a FactoredFakeGraph."""
    if hasattr(graph, 'hop'):
        return graph.hop(X, alpha, beta, Z, items_only=items_only)
    return ops.spmm(graph, X, alpha, beta, Z)


def cw_loss_and_grad(out, Up, users, pos, neg):
    """CWloss = mean(<Pu_u, Pi_neg> - <Pu_u, Pi_pos>) (PGA.py:109-116) and its gradient w.r.t. the propagated table."""
    u32, p32, n32 = users.to(torch.int32), (pos + Up).to(torch.int32), (neg + Up).to(torch.int32)
    ue, pe, ne = ops.gather_rows(out, u32), ops.gather_rows(out, p32), ops.gather_rows(out, n32)
    loss = ((ue * ne).sum(1) - (ue * pe).sum(1)).mean()
    c = 1.0 / users.numel()
    G = torch.zeros_like(out)
    ops.scatter_add_rows(G, u32, ne - pe, c, check_range=False)
    ops.scatter_add_rows(G, n32, ue, c, check_range=False)
    ops.scatter_add_rows(G, p32, ue, -c, check_range=False)
    return loss, G


def cw_operator(n_nodes, Up, users, pos, neg, device):
    """The CW loss is bilinear in the propagated table: L = 1/2 out^T M out with the fixed sparse symmetric M built from the
    (user, target, negative) triples of one inner epoch (PGA.py:104-116).  Then dL/d(out) = M out is ONE SpMM -- no atomics
    on the five target rows that every user pair hits -- and L = 1/2 <out, M out>."""
    c = 1.0 / users.numel()
    u, p, n = users.long(), pos.long() + Up, neg.long() + Up
    rows = torch.cat([u, u, n, p])
    cols = torch.cat([n, p, u, u]).to(torch.int32)
    vals = torch.cat([torch.full_like(u, c, dtype=torch.float32), torch.full_like(u, -c, dtype=torch.float32)] * 2)
    order = torch.sort(rows, stable=True)[1]
    rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=rows.device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_nodes), 0)
    return ops.CSRGraph(rowptr.cpu().numpy(), cols[order].contiguous(), vals[order].contiguous(), device)


def cw_operator_from_topk(n_nodes, Up, n_real, targets, neg, device):
    """The same operator as cw_operator() built from its structure instead of a 4*U*T-entry sort + histogram (which is what a
    per-step rebuild, as in CLeaR, cannot afford: the five target rows alone serialise a million atomic increments each):
      * a real user's row is [neg(u,0..T-1) (+c), targets (-c)]: 2T entries at a fixed stride, no sort;
      * item rows = the negatives grouped by item (one stable sort of U*T keys; row starts by binary search, no atomics)
        followed, for a target, by its dense row of all real users (-c).
    neg: [n_real, T] int64 item ids.  Returns (CSRGraph, neg_counts[I] int64 -- how often each item closes a top-k list)."""
    T = len(targets)
    I = n_nodes - Up
    c = 1.0 / (n_real * T)
    dev = neg.device
    # (a device tensor is taken as it is: building one from a Python list is a pageable host-to-device copy, i.e. a stream synchronisation per call)
    tg = targets.to(dev, torch.int64) if isinstance(targets, torch.Tensor) else torch.as_tensor(targets, device=dev, dtype=torch.int64)
    ar_u = torch.arange(n_real, device=dev, dtype=torch.int64)
    flat = neg.reshape(-1).to(torch.int32)                   # 32-bit keys: half the radix passes of an int64 sort
    sorted_items, order = torch.sort(flat, stable=True)
    neg_ptr = torch.searchsorted(sorted_items, torch.arange(I + 1, device=dev, dtype=torch.int32))
    sorted_items = sorted_items.long()
    neg_cnt = neg_ptr[1:] - neg_ptr[:-1]
    item_len = neg_cnt.clone()
    item_len[tg] += n_real
    n_user_entries = 2 * T * n_real
    rowptr = torch.empty(n_nodes + 1, dtype=torch.int64, device=dev)
    rowptr[:n_real + 1] = 2 * T * torch.arange(n_real + 1, device=dev, dtype=torch.int64)
    rowptr[n_real + 1:Up + 1] = n_user_entries
    rowptr[Up + 1:] = n_user_entries + torch.cumsum(item_len, 0)
    nnz = 2 * n_user_entries
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    val = torch.empty(nnz, dtype=torch.float32, device=dev)
    col[:n_user_entries] = torch.cat([neg + Up, (tg + Up).expand(n_real, T)], 1).reshape(-1).to(torch.int32)
    val[:n_user_entries] = torch.cat([torch.full((n_real, T), c, device=dev), torch.full((n_real, T), -c, device=dev)], 1).reshape(-1)
    item_start = rowptr[Up:Up + I]
    pos = item_start[sorted_items] + (torch.arange(n_real * T, device=dev, dtype=torch.int64) - neg_ptr[sorted_items])
    col[pos] = (order // T).to(torch.int32)
    val[pos] = c
    tpos = ((item_start[tg] + neg_cnt[tg])[:, None] + ar_u[None, :]).reshape(-1)
    col[tpos] = ar_u.to(torch.int32).repeat(T)
    val[tpos] = -c
    if n_real > ops.DEFAULT_CHUNK and rowptr.is_cuda:
        # every target's row holds its n_real dense entries: at least one long row, so the plan can be built on the device -- no host read,
        # the step that rebuilds this operator (CLeaR) never waits for the stream
        return ops.CSRGraph.from_device(rowptr, col, val, nnz, long_entries=n_user_entries if 2 * T <= ops.DEFAULT_CHUNK else None), neg_cnt      # user rows hold 2T entries: only item rows can be long
    return ops.CSRGraph(rowptr, col, val, device, validate=False), neg_cnt


def cw_loss_and_grad_op(M, out, scale=1.0):
    """L = 1/2 out^T M out and scale * dL/d(out) = scale * M out (the scale folded into the SpMM's epilogue)."""
    G = ops.spmm(M, out, alpha=scale)
    return (0.5 / scale) * (out * G).sum(), G


def pga_block_gradient(graph, fake_rows, Up, I, E0, L, G):
    """Returns the un-normalised F x I block sum_k <dE_{k+1}[f], E_k[U'+j]> + <E_k[f], dE_{k+1}[U'+j]> for the LightGCN mean."""
    E = [E0]
    for k in range(L):
        E.append(_hop(graph, E[k]))
    s = 1.0 / (L + 1)
    Gs = G * s
    dE = [None] * (L + 1)
    dE[L] = Gs
    for k in range(L - 1, 0, -1):
        dE[k] = _hop(graph, dE[k + 1], 1.0, 1.0, Gs, items_only=(k == 1))     # dE[1] is read on the fake users' and the items' rows only
    block = torch.zeros(fake_rows.numel(), I, dtype=torch.float32, device=E0.device)
    for k in range(L):
        ops.sddmm_rows_dense(dE[k + 1], E[k], fake_rows, Up, I, out=block)
        ops.sddmm_rows_dense(E[k], dE[k + 1], fake_rows, Up, I, out=block)
    return block, E


def pga_step_block(graph, fake_rows, Up, I, E0, L, M):
    """Forward (layers kept), CW gradient through the operator M, backward, and the F x I block -- one PGA gradient step
    minus the update (PGA.py:99-134)."""
    E = [E0]
    for k in range(L):
        E.append(_hop(graph, E[k]))
    s = 1.0 / (L + 1)
    out = ops.tables_sum(E, s)                                            # the mean over layers in one pass
    loss, Gs = cw_loss_and_grad_op(M, out, s)                             # Gs = dL/d(out) / (L + 1): what every layer receives
    dE = [None] * (L + 1)
    dE[L] = Gs
    for k in range(L - 1, 0, -1):
        dE[k] = _hop(graph, dE[k + 1], 1.0, 1.0, Gs, items_only=(k == 1))     # dE[1] is read on the fake users' and the items' rows only
    block = torch.zeros(fake_rows.numel(), I, dtype=torch.float32, device=E0.device)
    for k in range(L):
        ops.sddmm_rows_dense(dE[k + 1], E[k], fake_rows, Up, I, out=block)
        ops.sddmm_rows_dense(E[k], dE[k + 1], fake_rows, Up, I, out=block)
    return block, loss


class PGA(AttackBase):
    def __init__(self, arg, data):
        super().__init__(arg, data)
        self.batchSize = 128

    def posionDataAttack(self, recommender):
        Pu, Pi = recommender.model()
        n_pop = int(Pi.shape[0] * 0.05)
        maxRecNumItemInd = torch.topk(torch.as_tensor(np.asarray(self.interact.sum(0)).ravel()), n_pop)[1].numpy()
        self.maxRecNumItemInd = maxRecNumItemInd
        # optimizer bound to the OLD model's parameters; recommender.__init__ below replaces the model, so this
        # train() never moves the new tables (reference quirk Q4, PGA.py:59-67) -- reproduced as is
        optimizer = torch.optim.SGD(recommender.model.parameters(), lr=recommender.args.lRate / 10)
        self.dataUpdate(recommender)
        reinit_with_tables(recommender, Pu, Pi)
        newAdj = recommender.data.matrix()
        self.controlledUser = list(range(self.userNum, self.userNum + self.fakeUserNum))
        recommender.train(Epoch=self.Epoch, optimizer=optimizer, evalNum=5)
        originRecommender = deepcopy(recommender)
        U, F, I = self.userNum, self.fakeUserNum, self.itemNum
        # fake block S: targets 1, popular items one random weight per fake user (PGA.py:69-73)
        S = torch.zeros(F, I, dtype=torch.float32)
        for f in range(F):
            S[f, self.targetItem] = 1
            S[f, maxRecNumItemInd] = torch.rand([1]).item()
        S = S.to(DEVICE)
        recommender = deepcopy(originRecommender)
        optimizer = torch.optim.Adam(recommender.model.parameters(), lr=recommender.args.lRate / 10)
        real = sp.csr_matrix(newAdj[:U])
        fg = FactoredFakeGraph(real, U, F, I, emb_size=getattr(recommender.model, 'latent_size', None))
        L = getattr(recommender.model, 'n_prop_layers', 0)
        for epoch in range(self.outerEpoch):
            # outer optimisation: victim retrain on the current poisoned graph
            uiAdj = sp.vstack([real, sp.csr_matrix(S.cpu().numpy())]).tocsr()
            init_graph(recommender.model, uiAdj, U + F, I)
            recommender.train(Epoch=self.Epoch, optimizer=optimizer, evalNum=3)
            # inner optimisation on a frozen copy of the victim's tables
            E0 = recommender.model._pack().detach().clone()
            S2 = S.clone()
            for _ in range(self.innerEpoch):
                pairs = None
                losses = []                                  # printed after the loop: reading each loss as it is produced would stall the launch queue once per step
                for batch in range(0, I, self.batchSize):
                    graph = fg.set_block(S2)
                    if L == 0:
                        raise ValueError('PGA differentiates through the graph propagation; the victim has no propagation layers')
                    if pairs is None:
                        out = E0.clone()
                        E = E0
                        for k in range(L):
                            E = _hop(graph, E)
                            out += E
                        out /= (L + 1)
                        top_idx, _ = ops.score_mask_topk(out[:U + F].contiguous(), out[U + F:].contiguous(), min(50, I))     # no interacted mask (PGA.py:101-102)
                        pairs = cw_operator(U + F + I, U + F, *cw_pairs(top_idx, U, self.targetItem, pop=True), device=E0.device)
                    block, loss = pga_step_block(graph, fg.fake_rows, U + F, I, E0, L, pairs)
                    ops.pga_update_(S2, block, fg.dinv[U:U + F].contiguous(), fg.dinv[U + F:].contiguous())
                    losses.append(loss)
                for b, lo in enumerate(torch.stack(losses).tolist() if losses else []):      # the reference's lines (PGA.py:140), same order, one device read
                    print('>> batchNum:{} Loss:{}'.format(b, lo))
            proj, _ = ops.topn_project_rows(S2, int(self.maliciousFeedbackSize * I))
            proj[:, self.targetItem] = 1
            S = proj
            print('attack step {} is over\n'.format(epoch + 1))
        self.interact = sp.vstack([real, sp.csr_matrix(S.cpu().numpy())]).tolil()
        return self.interact

    def project(self, mat, n):
        """Per-row top-n -> {0,1} (PGA.py:153-167); accepts a scipy or dense matrix like the reference."""
        M = torch.as_tensor(np.asarray(mat.todense() if hasattr(mat, 'todense') else mat), dtype=torch.float32, device=DEVICE).contiguous()
        out, _ = ops.topn_project_rows(M, n)
        return out.cpu()

    def dataUpdate(self, recommender):
        """Append F fake users to the id maps and rebuild ui_adj / norm_adj / interaction_mat (PGA.py:169-192)."""
        data = recommender.data
        data.user_num += self.fakeUserNum
        for i in range(self.fakeUserNum):
            data.user['fakeuser{}'.format(i)] = len(data.user)
            data.id2user[len(data.user) - 1] = 'fakeuser{}'.format(i)
        u, it, inter = rebuild_interaction_matrix(data)
        n = data.user_num + data.item_num
        half = sp.csr_matrix((np.ones(len(u), np.float32), (u, it + data.user_num)), shape=(n, n), dtype=np.float32)
        data.ui_adj = half + half.T
        data.norm_adj = data.normalize_graph_mat(data.ui_adj)
        data.interaction_mat = inter
