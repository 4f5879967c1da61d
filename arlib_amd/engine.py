"""Device-resident training engine for the GMF / LightGCN / SimGCL hot loop.

One `step()` is one iteration of the reference's inner loop (recommender/LightGCN.py:47-64, GMF.py:39-54):
full-graph propagation, BPR + L2 on the gathered batch rows, backward through the propagation and a dense
Adam (or SGD) update of both embedding tables -- as 2L SpMM launches, three small loss kernels and (for
Adam) no separate optimizer launch at all: the update is fused into the epilogue of the last backward hop.

HBM layout (all fp32, row-major, allocated once):
    E0  [N,d]  the two embedding tables as ONE buffer: users rows [0,U), items rows [U,N)   (torch.cat eliminated)
    m,v [N,d]  Adam moments
    S   [N,d]  running layer sum, becomes the propagated output `out` in place
    Ea,Eb [N,d] ping-pong hop buffers (forward hops, then backward Horner accumulators)
    G   [N,d]  dL/d(out): zero except <= 3B rows (scatter-added by the BPR kernel)
Backward uses the symmetry of the normalised adjacency (Horner form):
    acc = G; repeat L-1 times: acc = G + A acc;  dE0 = (G + A acc) / (L+1).
"""
import torch

from . import ops


def mean_adjacency_gradient(A, E0, G, L, s, out, noises=None, eps=0.1):
    """out[e] += dL/d(value of stored entry e of the symmetric graph A) for out_tables = s * sum of layers E_k (k >= 1 always in the sum), E_{k+1} = A E_k
    (+ a gradient-free perturbation when `noises` is given), from G = dL/d(out_tables): dE_L = s G, dE_k = s G + A dE_{k+1}, entry (i, j) receives
    sum_{k=0}^{L-1} <dE_{k+1}[i], E_k[j]>.  L - 1 forward hops, L - 1 backward hops, L products over the pattern (ops.sddmm_csr)."""
    if L == 0:
        return out
    E = [E0]
    for k in range(L - 1):
        nxt = ops.spmm(A, E[-1])
        if noises is not None:
            ops.simgcl_perturb_(nxt, noises[k], eps)
        E.append(nxt)
    G = G.contiguous()
    acc = G
    for j in range(L):
        ops.sddmm_csr(A, acc, E[L - 1 - j], s, out=out)
        if j < L - 1:
            acc = ops.spmm(A, acc, 1.0, 1.0, G)
    return out


class PropagationEngine:
    def __init__(self, graph, n_users, n_items, emb_size, n_layers, reg, lr, device, skip_layer0=False,
                 optimizer='adam', betas=(0.9, 0.999), eps=1e-8, table=None, schedule='auto'):
        self.A = graph
        self.U, self.I, self.d, self.L = int(n_users), int(n_items), int(emb_size), int(n_layers)
        self.N = self.U + self.I
        self.reg, self.lr, self.betas, self.eps = float(reg), float(lr), betas, eps
        self.skip0 = bool(skip_layer0)
        self.optimizer = optimizer
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ops._lib.ArlError('PropagationEngine needs a GPU device (no CPU fallback)')
        if graph is None and self.L > 0:
            raise ValueError('n_layers > 0 needs a graph')
        if graph is not None and graph.n_rows != self.N:
            raise ValueError('graph has %d rows, expected U+I=%d' % (graph.n_rows, self.N))
        if self.skip0 and self.L < 1:
            raise ValueError('skip_layer0 needs n_layers >= 1')
        if schedule not in ('auto', 'csr', 'blocked'):
            raise ValueError("schedule must be 'auto', 'csr' or 'blocked'")
        # full-table hops: register-blocked schedule (ops.BlockedPlan) for large graphs at d = 64, row-per-group CSR kernel otherwise
        if schedule != 'csr':
            ops.auto_blocked(graph, emb_size, split=self.U, force=(schedule == 'blocked'))
        if graph is not None and graph.nnz >= ops.BLOCKED_MIN_NNZ:
            graph.enable_masked_order()              # flag-masked hops take the rows sorted by length
        z = lambda: torch.zeros(self.N, self.d, dtype=torch.float32, device=self.device)
        self.E0 = z() if table is None else table
        if self.E0.shape != (self.N, self.d) or self.E0.dtype != torch.float32 or not self.E0.is_contiguous():
            raise ValueError('table must be contiguous fp32 [U+I, d]')
        self.m, self.v, self.G = z(), z(), z()
        if self.L > 0:
            self.S, self.Ea, self.Eb = z(), z(), z()
        else:
            self.S = self.E0
        self.t = 0
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._ws = None

    # views with the reference's names
    @property
    def user_emb(self):
        return self.E0[:self.U]

    @property
    def item_emb(self):
        return self.E0[self.U:]

    def forward(self, noises=None, eps=0.1, out=None):
        """Propagated tables [N,d] (recommender/LightGCN.py:230-240; SimGCL.py:198-210 when skip_layer0).
        `noises`: optional list of L [N,d] tensors (SimGCL perturbed view, injected or from torch.rand)."""
        L, A = self.L, self.A
        if L == 0:
            return self.E0
        S = self.S if out is None else out
        if noises is None and not self.skip0:
            if L == 1:
                return ops.spmm(A, self.E0, 0.5, 0.5, self.E0, out=S)
            ops.spmm_layersum(A, self.E0, self.E0, S, self.Ea)
            cur, nxt = self.Ea, self.Eb
            for _ in range(L - 2):
                ops.spmm_layersum(A, cur, S, S, nxt)
                cur, nxt = nxt, cur
            s = 1.0 / (L + 1)
            return ops.spmm(A, cur, s, s, S, out=S)
        # SimGCL family: mean over hops 1..L, optional perturbation after each hop
        cur, nxt, x = self.Ea, self.Eb, self.E0
        for k in range(L):
            ops.spmm(A, x, out=cur)
            if noises is not None:
                ops.simgcl_perturb_(cur, noises[k], eps)
            if k == 0:
                if self.skip0:
                    S.copy_(cur)
                else:
                    torch.add(self.E0, cur, out=S)
            else:
                S.add_(cur)
            x, cur, nxt = cur, nxt, cur
        S.mul_(1.0 / (L if self.skip0 else L + 1))
        return S

    def backward_to_table(self, G, out=None):
        """dL/dE0 from dL/d(out) (= G, [N,d]) without touching the optimizer state."""
        L, A = self.L, self.A
        if L == 0:
            return G
        out = self.Ea if out is None else out
        if self.skip0:
            acc = G
            bufs = [self.Ea, self.Eb]
            for k in range(L - 1):
                dst = bufs[k % 2]
                ops.spmm(A, acc, 1.0, 1.0, G, out=dst)
                acc = dst
            dst = self.Eb if acc is self.Ea else self.Ea
            return ops.spmm(A, acc, 1.0 / L, out=dst)
        acc = G
        bufs = [self.Ea, self.Eb]
        for k in range(L):
            last = k == L - 1
            s = 1.0 / (L + 1) if last else 1.0
            dst = bufs[k % 2]
            ops.spmm(A, acc, s, s, G, out=dst)
            acc = dst
        return acc

    def adjacency_gradient(self, G, out=None, noises=None, eps=0.1):
        """dL/d(values of A) from dL/d(out) = G for out = mean of the layers E_k, E_{k+1} = A E_k (+ a perturbation that carries no gradient when
        `noises` is given: the SimGCL views, recommender/SimGCL.py:198-210): what autograd leaves in `sparse_norm_adj.grad` when the reference sets
        `sparse_norm_adj.requires_grad = True` (recommender/LightGCN.py:41-43,58-59; SimGCL.py:44-47,62-63).  With s = 1/(L+1) (layers 0..L) or
        1/L (skip_layer0: layers 1..L), dE_L = s G, dE_k = s G + A dE_{k+1} (A symmetric; k >= 1 is in the mean either way), the gradient on a
        stored entry (i, j) is sum_k <dE_{k+1}[i], E_k[j]>: L - 1 forward hops, L - 1 backward hops and L products over the pattern
        (ops.sddmm_csr), ACCUMULATED into `out` ([nnz] fp32, CSR order).  noises: the [hop] tables of the forward this gradient belongs to."""
        L, A = self.L, self.A
        if out is None:
            out = torch.zeros(A.col.numel(), dtype=torch.float32, device=G.device)
        return mean_adjacency_gradient(A, self.E0, G, L, 1.0 / L if self.skip0 else 1.0 / (L + 1), out, noises, eps)

    # ---- the propagated mean on a row subset, with its backward: the sparse-batch schedule of step() for callers that keep autograd and
    # their own optimiser (an extra loss term, an optimiser over one table only, gradient capture)
    def forward_rows(self, rows):
        """mean(E_0 .. E_L)[rows] (LightGCN): L-1 full hops and the last hop on the listed rows only.  rows: int32 node ids, duplicates allowed."""
        L, A = self.L, self.A
        if L == 0 or self.skip0 or L > 8:
            raise ValueError('forward_rows: LightGCN mean over layers 0..L with 1 <= L <= 8')
        n = rows.numel()
        self._sparse_buffers(max(n // 3, 1) if n % 3 == 0 else n)          # node-sized buffers (flags, bits, hops); batch-sized ones are not used here
        layers = [self.E0]
        for k in range(L - 1):
            ops.spmm(A, layers[-1], out=self.hops[k])
            layers.append(self.hops[k])
        return ops.spmm_rows(A, layers[-1], rows, layers, 1.0 / (L + 1), nsplit=self.nsplit, check_range=False)

    def backward_rows(self, rows, g_rows):
        """dL/dE0 [N, d] from dL/d(forward_rows(rows)) = g_rows: scatter into the (all-zero) gradient table, flag-masked first hop, Horner
        over the remaining hops; the sparse state is cleared again before returning."""
        L, A = self.L, self.A
        s = 1.0 / (L + 1)
        if getattr(self, '_G_dirty', True):
            self.G.zero_()
            self._G_dirty = False
        ops.batch_rows_set_(self.G, self.flags, self.bits, rows, g_rows.contiguous(), 1.0, check_range=False, dup_bits=self.dup_bits)
        if L == 1:
            out = ops.spmm_flagged(A, self.G, self.bits, s, s, self.G, self.flags)
        else:
            acc = ops.spmm_flagged(A, self.G, self.bits, 1.0, 1.0, self.G, self.flags, out=self.hops[0])
            for k in range(1, L - 1):
                acc = ops.spmm_flagged(A, acc, None, 1.0, 1.0, self.G, self.flags, out=self.hops[k % 2 if len(self.hops) == 2 else k])
            out = ops.spmm_flagged(A, acc, None, s, s, self.G, self.flags)
        ops.batch_rows_clear_(self.G, self.flags, self.bits, rows, check_range=False, dup_bits=self.dup_bits)
        return out

    def loss_and_grad_out(self, out, u, p, n):
        self.G.zero_()
        self._G_dirty = True          # the sparse step keeps G all-zero between calls; dense users leave it dirty
        if self._ws is None or self._ws.numel() < 4 * u.numel():
            self._ws = torch.empty(4 * u.numel(), dtype=torch.float32, device=self.device)
        ops.bpr_l2_fwd_bwd(out, self.U, u, p, n, self.reg, self.G, workspace=self._ws, loss_out=self.loss_out, check_range=False)
        return self.loss_out

    def grad(self, u, p, n):
        """(loss_out, dL/dE0) for one batch; no parameter update."""
        out = self.forward()
        lo = self.loss_and_grad_out(out, u, p, n)
        return lo, self.backward_to_table(self.G)

    def step(self, u, p, n, rows=None):
        """One training iteration on a device batch (int32 tensors, already range-checked by the caller).
        Returns the device tensor [bpr, reg_term, ||U_b||, ||P_b||]; loss = [0]+[1] (no host sync here).

        Sparse-batch form (LightGCN + Adam): the loss reads the propagated table on <= 3B rows only and its gradient G is
        non-zero on those rows only, so
          * the last forward hop is evaluated on the batch rows alone (arl_spmm_csr_rows_f32), and the running layer sum
            is never materialised for the other N - 3B rows;
          * the first backward hop A.G gathers only edges whose source row is flagged (arl_spmm_csr_flagged_f32);
          * G is kept all-zero between steps (batch rows are cleared afterwards) instead of memset per step.
        Same numbers as `step_dense` (which evaluates all 2L hops on the full graph), 2 of the 2L full hops cheaper.
        `rows` = cat(u, U+p, U+n) may be passed in when the caller has it precomputed."""
        L, A = self.L, self.A
        if self.optimizer != 'adam' or L == 0 or self.skip0 or L > 8:
            return self.step_dense(u, p, n)
        B = u.numel()
        if rows is None:
            rows = torch.cat([u, p + self.U, n + self.U])
        self._sparse_buffers(B)
        if getattr(self, '_G_dirty', True):
            self.G.zero_()
            self._G_dirty = False
        s = 1.0 / (L + 1)
        # forward: L-1 full hops, last hop on the batch rows
        layers = [self.E0]
        for k in range(L - 1):
            ops.spmm(A, layers[-1], out=self.hops[k])
            layers.append(self.hops[k])
        ops.spmm_rows(A, layers[-1], rows, layers, s, nsplit=self.nsplit, out=self.out_c, workspace=self.rows_ws, check_range=False)
        # loss + compact per-sample gradients (rows [0,B) users, [B,2B) positives, [2B,3B) negatives of out_c)
        self.Gc.zero_()
        ops.bpr_l2_fwd_bwd(self.out_c, B, self.ar, self.ar, self.arB, self.reg, self.Gc, workspace=self._ws, loss_out=self.loss_out, check_range=False, distinct_rows=True)
        ops.batch_rows_set_(self.G, self.flags, self.bits, rows, self.Gc, 1.0, check_range=False, dup_bits=self.dup_bits)      # duplicates accumulate in order; rows marked
        # backward (Horner): first hop gathers flagged rows only; G is read through the flags everywhere
        self.t += 1
        if L == 1:
            ops.spmm_flagged(A, self.G, self.bits, s, s, self.G, self.flags, out=self.hops[0])
            ops.adam_dense(self.E0, self.hops[0], self.m, self.v, self.lr, self.t, self.betas, self.eps)
        else:
            acc = ops.spmm_flagged(A, self.G, self.bits, 1.0, 1.0, self.G, self.flags, out=self.hops[0])
            for k in range(1, L - 1):
                acc = ops.spmm_flagged(A, acc, None, 1.0, 1.0, self.G, self.flags, out=self.hops[k % 2 if len(self.hops) == 2 else k])
            ops.spmm_adam(A, acc, s, s, self.G, self.E0, self.m, self.v, self.lr, self.t, self.betas, self.eps, zflags=self.flags)
        ops.batch_rows_clear_(self.G, self.flags, self.bits, rows, check_range=False, dup_bits=self.dup_bits)
        return self.loss_out

    # ---- NGCF (recommender/NGCF.py:47-64,197-212): the whole training iteration without autograd or a torch optimizer.  Per layer ONE hop
    # P = A E (A(E W1) = (A E) W1) and the fp32-MFMA dense part (ops.ngcf_dense_*); sparse-batch schedule as in step(): the LAST layer is
    # evaluated on the <= 3B batch rows only (row-subset hop + a [3B, 2d] dense part), its backward re-enters the table through the flag-masked
    # hop; the Adam update of the tables is the epilogue of the last backward hop, the d x d weights get the dense Adam kernel.
    def init_ngcf(self, weights):
        """weights: [(W1_l, W2_l)] device tensors / Parameters, one pair per layer (updated in place by step_ngcf)."""
        if len(weights) != self.L or any(tuple(w.shape) != (self.d, self.d) for pair in weights for w in pair):
            raise ValueError('init_ngcf: one pair of [d, d] weights per layer')
        self.ngcf_W = [(a.data if hasattr(a, 'data') else a, b.data if hasattr(b, 'data') else b) for a, b in weights]
        z = lambda: torch.zeros(self.d, self.d, dtype=torch.float32, device=self.device)
        self.ngcf_m = [(z(), z()) for _ in weights]
        self.ngcf_v = [(z(), z()) for _ in weights]

    def step_ngcf(self, u, p, n, rows=None, slope=0.01, capture=None):
        """capture: optional dict; receives the table gradient ('table') and the weight gradients ('W') of this step (diagnostics / tests:
        the Adam update then runs as a separate dense launch instead of the last hop's epilogue -- same numbers)."""
        L, A, d = self.L, self.A, self.d
        if L < 1 or not hasattr(self, 'ngcf_W') or d not in ops.NGCF_DENSE_WIDTHS:
            raise ValueError('step_ngcf: needs n_layers >= 1, init_ngcf() and d in %s' % (ops.NGCF_DENSE_WIDTHS,))
        B = u.numel()
        if rows is None:
            rows = torch.cat([u, p + self.U, n + self.U])
        self._sparse_buffers(B)
        if getattr(self, '_G_dirty', True):
            self.G.zero_()
            self._G_dirty = False
        s = 1.0 / (L + 1)
        Wcat = [torch.cat([a, b], 0) for a, b in self.ngcf_W]
        # forward: L-1 full layers, the last one on the batch rows
        egos, Ps = [self.E0], []
        for l in range(L - 1):
            P = ops.spmm(A, egos[-1])
            Ps.append(P)
            egos.append(ops.ngcf_dense_fwd(P, egos[-1], Wcat[l], slope))
        # same edge-range split as the autograd route (NGCF._LastLayerRows): leaky_relu's derivative jumps at 0, so a pre-activation within
        # rounding of 0 can land on either side of the kink under a different summation order and change that row's gradient by ~1 % --
        # with one association for both routes they stay bit-identical (and both sit on the reference's side on its captured steps)
        P_r = ops.spmm_rows(A, egos[-1], rows, (), 1.0, nsplit=ops.ROWS_NSPLIT, check_range=False)
        E_r = ops.gather_rows(egos[-1], rows, check_range=False)
        out_r = ops.ngcf_dense_fwd(P_r, E_r, Wcat[L - 1], slope)
        acc = out_r + E_r
        for e in egos[:-1]:
            acc += ops.gather_rows(e, rows, check_range=False)
        acc *= s
        self.Gc.zero_()
        ops.bpr_l2_fwd_bwd(acc, B, self.ar, self.ar, self.arB, self.reg, self.Gc, workspace=self._ws, loss_out=self.loss_out, check_range=False, distinct_rows=True)
        Gs = self.Gc * s                                         # every layer's batch rows receive this share of dL/d(out) directly
        # backward
        self.t += 1
        gP_r, gE_r, gW = ops.ngcf_dense_bwd(Gs, out_r, P_r, E_r, Wcat[L - 1], slope)
        gWs = [None] * L
        gWs[L - 1] = gW
        ops.batch_rows_set_(self.G, self.flags, self.bits, rows, gP_r, 1.0, check_range=False, dup_bits=self.dup_bits)
        g = ops.spmm_flagged(A, self.G, self.bits)              # A^T (rows' gP scattered) = A (...), A symmetric
        ops.batch_rows_clear_(self.G, self.flags, self.bits, rows, check_range=False, dup_bits=self.dup_bits)
        ops.scatter_add_rows(g, rows, gE_r + Gs, 1.0, check_range=False)
        if L == 1:
            if capture is not None:
                capture['table'] = g.clone()
            ops.adam_dense(self.E0, g, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        for l in range(L - 2, -1, -1):
            gP, gE, gWs[l] = ops.ngcf_dense_bwd(g, egos[l + 1], Ps[l], egos[l], Wcat[l], slope)
            ops.scatter_add_rows(gE, rows, Gs, 1.0, check_range=False)
            if l > 0:
                g = ops.spmm(A, gP, 1.0, 1.0, gE)
            elif capture is not None:
                capture['table'] = ops.spmm(A, gP, 1.0, 1.0, gE)
                ops.adam_dense(self.E0, capture['table'], self.m, self.v, self.lr, self.t, self.betas, self.eps)
            else:
                ops.spmm_adam(A, gP, 1.0, 1.0, gE, self.E0, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        if capture is not None:
            capture['W'] = [w.clone() for w in gWs]
        for l in range(L):
            for k in range(2):
                ops.adam_dense(self.ngcf_W[l][k], gWs[l][k * d:(k + 1) * d], self.ngcf_m[l][k], self.ngcf_v[l][k], self.lr, self.t, self.betas, self.eps)
        return self.loss_out

    def step_simgcl(self, u, p, n, cl_rate=0.2, tau=0.2, eps=0.1, noises=None):
        """One SimGCL training iteration (recommender/SimGCL.py:51-63,198-219) on the sparse-batch schedule.

        Reference: 3 forwards (clean + two perturbed views) of L hops each and autograd through all of them = 6L full-graph
        hops.  Here:
          * hop 1 (A E0) is shared by the three forwards (the perturbation is added after it);
          * every forward's last hop is only consumed on batch rows (BPR rows; InfoNCE rows are the unique batch users /
            positive items) -> row-subset hops;
          * the perturbation sign(E)*normalize(noise)*eps carries no gradient, so all three forwards have the SAME linear
            backward operator (1/L) sum_{k=1..L} A^k: the three sparse output gradients are summed first and ONE Horner pass
            (flag-masked first hop, Adam fused into the last) replaces three.
        L = 2: 2 full hops + 1 masked + 3 row-subset hops instead of 12 full hops.
        noises: optional [view][hop] full [N,d] tensors (parity tests); default: uniform noise drawn inside the perturbation kernel, one
        stream per hop and view (the reference's rand_like calls).  Returns (loss_out, cl_loss) device tensors."""
        if not self.skip0 or self.optimizer != 'adam':
            raise ValueError('step_simgcl needs a skip_layer0 engine with Adam')
        L, A, U, N, d = self.L, self.A, self.U, self.N, self.d
        B = u.numel()
        self._sparse_buffers(B)
        if getattr(self, '_G_dirty', True):
            self.G.zero_(); self._G_dirty = False
        inv = 1.0 / L
        rows = torch.cat([u, p + U, n + U])
        uidx = torch.unique(u.long())
        iidx = torch.unique(p.long()) + U
        rows_cl = torch.cat([uidx, iidx]).to(torch.int32)
        nu = uidx.numel()
        # noise: injected tables (parity tests) or, by default, drawn inside the perturbation kernel from a seed taken once from torch's global
        # generator and a running stream number -- one "draw" per hop and view like the reference's rand_like calls, but no [N, d] noise table,
        # no clone of the operand, and the compact last-hop rows get exactly the values a full-table draw would have given them
        rng_mode = noises is None
        if rng_mode:
            if getattr(self, '_noise_seed', None) is None:
                self._noise_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                self._noise_stream = 0
            stream0 = self._noise_stream
            self._noise_stream += 2 * L
        rnd = (lambda v, k: noises[v][k]) if noises is not None else None
        def perturb_full(src, v, k, out=None):                          # hop-k table of view v
            if rng_mode:
                return ops.simgcl_perturb_rng(src, eps, self._noise_seed, stream0 + v * L + k, out=out)
            dst = src.clone() if out is None else out.copy_(src)
            return ops.simgcl_perturb_(dst, rnd(v, k), eps)
        def perturb_rows(src, v, k, sel):                               # compact rows `sel` of the hop-k table of view v, in place
            if rng_mode:
                return ops.simgcl_perturb_rng(src, eps, self._noise_seed, stream0 + v * L + k, out=src, row_ids=sel)
            return ops.simgcl_perturb_(src, rnd(v, k)[sel.long()].contiguous(), eps)
        # ---- forwards
        E1 = ops.spmm(A, self.E0, out=self.hops[0])                       # shared first hop
        def finish(first, view):                                            # hops 2..L on table `first`; returns compact rows [rows_sel, d]
            sel = rows if view is None else rows_cl
            layers, cur = [first], first
            for k in range(1, L - 1):
                nxt = ops.spmm(A, cur)
                if view is not None:
                    perturb_full(nxt, view, k, out=nxt)
                layers.append(nxt); cur = nxt
            if L == 1:
                return ops.gather_rows(first, sel, check_range=False)
            if view is None:
                return ops.spmm_rows(A, cur, sel, layers, inv, nsplit=self.nsplit, check_range=False)
            last = ops.spmm_rows(A, cur, sel, (), 1.0, nsplit=self.nsplit, check_range=False)
            perturb_rows(last, view, L - 1, sel)
            for t in layers:
                last += ops.gather_rows(t, sel, check_range=False)
            return last * inv
        out_c = finish(E1, None)
        self.Gc.zero_()
        ops.bpr_l2_fwd_bwd(out_c, B, self.ar, self.ar, self.arB, self.reg, self.Gc, workspace=self._ws, loss_out=self.loss_out, check_range=False, distinct_rows=True)
        ops.scatter_add_rows(self.G, rows, self.Gc, 1.0, check_range=False)
        views = []
        for v in (0, 1):
            E1p = perturb_full(E1, v, 0)
            views.append(finish(E1p, v))
        lu, du1, du2 = ops.infonce_fwd_bwd(views[0][:nu].contiguous(), views[1][:nu].contiguous(), tau)
        li, di1, di2 = ops.infonce_fwd_bwd(views[0][nu:].contiguous(), views[1][nu:].contiguous(), tau)
        cl_loss = cl_rate * (lu[0] + li[0])
        gcl = torch.cat([du1 + du2, di1 + di2], 0)                         # both views differentiate through the same operator
        ops.scatter_add_rows(self.G, rows_cl, gcl, cl_rate, check_range=False)
        allrows = torch.cat([rows, rows_cl])
        ops.mark_rows_(self.flags, allrows, 1, check_range=False)
        ops.mark_bits_(self.bits, allrows, True, N, check_range=False)
        # ---- one backward pass: acc = G; (L-1) x: acc = G + A acc; g = A acc / L
        self.t += 1
        acc, first = self.G, True
        for k in range(L - 1):
            dst = self.hops[1] if acc is not self.hops[1] else self.hops[0]
            ops.spmm_flagged(A, acc, self.bits if first else None, 1.0, 1.0, self.G, self.flags, out=dst)
            acc, first = dst, False
        if first:       # L == 1: the only hop gathers the sparse G itself
            tmp = ops.spmm_flagged(A, self.G, self.bits, inv, 0.0, None, None, out=self.hops[1])
            ops.adam_dense(self.E0, tmp, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        else:
            ops.spmm_adam(A, acc, inv, 0.0, None, self.E0, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        ops.batch_rows_clear_(self.G, self.flags, self.bits, allrows, check_range=False, dup_bits=self.dup_bits)
        return self.loss_out, cl_loss

    def step_xsimgcl(self, u, p, n, cl_rate=0.2, tau=0.1, eps=0.1, layer_cl=1, noises=None):
        """One XSimGCL training iteration (recommender/XSimGCL.py:62-75,205-223): ONE perturbed forward; BPR + L2 on the mean of
        layers 1..L; InfoNCE (temperature tau) between that mean and the layer-`layer_cl` output at the batch's unique users and
        unique positive items.  Sparse-batch schedule: L-1 full hops + a row-subset hop forward; backward
            acc_L = c_L,  acc_k = c_k + A acc_{k+1},  dE0 = A acc_1,   c_k = G_mean / L + [k == layer_cl] G_cl
        with the first hop gathering flagged rows only and Adam fused into the last (default L=2: 2 full hops in all).
        noises: optional [hop] full [N,d] tensors (parity tests); default: uniform noise drawn inside the perturbation kernel, one stream
        per hop (the reference's rand_like).  Returns (loss_out, cl_loss)."""
        if not self.skip0 or self.optimizer != 'adam':
            raise ValueError('step_xsimgcl needs a skip_layer0 engine with Adam')
        L, A, U, N, d = self.L, self.A, self.U, self.N, self.d
        if not (1 <= layer_cl <= L):
            raise ValueError('layer_cl must be in [1, L]')
        B = u.numel()
        self._sparse_buffers(B)
        if getattr(self, '_G_dirty', True):
            self.G.zero_(); self._G_dirty = False
        inv = 1.0 / L
        rows = torch.cat([u, p + U, n + U])
        uidx = torch.unique(u.long())
        iidx = torch.unique(p.long()) + U
        rows_cl = torch.cat([uidx, iidx]).to(torch.int32)
        nu = uidx.numel()
        sel = torch.cat([rows, rows_cl])                                    # compact rows: [0,3B) BPR, then the CL rows
        rng_mode = noises is None                                           # default: noise drawn inside the perturbation kernel (step_simgcl)
        if rng_mode:
            if getattr(self, '_noise_seed', None) is None:
                self._noise_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                self._noise_stream = 0
            stream0 = self._noise_stream
            self._noise_stream += L
        # ---- forward
        layers, cur = [], self.E0
        for k in range(L - 1):
            nxt = ops.spmm(A, cur, out=self.hops[k % len(self.hops)] if L <= 3 else None)
            if rng_mode:
                ops.simgcl_perturb_rng(nxt, eps, self._noise_seed, stream0 + k, out=nxt)
            else:
                ops.simgcl_perturb_(nxt, noises[k], eps)
            layers.append(nxt); cur = nxt
        last = ops.spmm_rows(A, cur, sel, (), 1.0, nsplit=self.nsplit, check_range=False)
        if rng_mode:
            ops.simgcl_perturb_rng(last, eps, self._noise_seed, stream0 + L - 1, out=last, row_ids=sel)
        else:
            ops.simgcl_perturb_(last, noises[L - 1][sel.long()].contiguous(), eps)
        mean_c = last.clone()
        for t in layers:
            mean_c += ops.gather_rows(t, sel, check_range=False)
        mean_c *= inv
        cl_c = last[3 * B:] if layer_cl == L else ops.gather_rows(layers[layer_cl - 1], rows_cl, check_range=False)
        # ---- losses and compact gradients
        self.Gc.zero_()
        ops.bpr_l2_fwd_bwd(mean_c[:3 * B].contiguous(), B, self.ar, self.ar, self.arB, self.reg, self.Gc, workspace=self._ws, loss_out=self.loss_out,
                           check_range=False, distinct_rows=True)
        mcl = mean_c[3 * B:]
        lu, du1, du2 = ops.infonce_fwd_bwd(mcl[:nu].contiguous(), cl_c[:nu].contiguous(), tau)
        li, di1, di2 = ops.infonce_fwd_bwd(mcl[nu:].contiguous(), cl_c[nu:].contiguous(), tau)
        cl_loss = cl_rate * (lu[0] + li[0])
        g_mean_cl = torch.cat([du1, di1], 0)                                # w.r.t. the mean at the CL rows
        g_layer = torch.cat([du2, di2], 0) * cl_rate                        # w.r.t. the layer-`layer_cl` output at the CL rows
        # G <- G_mean / L  (sparse rows; G is all-zero between steps)
        ops.scatter_add_rows(self.G, rows, self.Gc, inv, check_range=False)
        ops.scatter_add_rows(self.G, rows_cl, g_mean_cl, cl_rate * inv, check_range=False)
        allrows = sel
        ops.mark_rows_(self.flags, allrows, 1, check_range=False)
        ops.mark_bits_(self.bits, allrows, True, N, check_range=False)
        # ---- backward
        self.t += 1
        bufs = [self.hops[1], self.hops[0]] if len(self.hops) >= 2 else [torch.empty_like(self.E0), torch.empty_like(self.E0)]
        if L == 1:
            src = self.G
            if layer_cl == 1:                                               # acc_1 = G + G_cl: both sparse, on the same flagged rows
                ops.scatter_add_rows(self.G, rows_cl, g_layer, 1.0, check_range=False)
            tmp = ops.spmm_flagged(A, src, self.bits, 1.0, 0.0, None, None, out=bufs[0])
            ops.adam_dense(self.E0, tmp, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        else:
            if layer_cl == L:                                               # acc_L = G + G_cl (sparse); later levels add G alone
                first_src = self.G.clone()
                ops.scatter_add_rows(first_src, rows_cl, g_layer, 1.0, check_range=False)
            else:
                first_src = self.G
            acc = ops.spmm_flagged(A, first_src, self.bits, 1.0, 1.0, self.G, self.flags, out=bufs[0])       # level L-1
            level = L - 1
            while True:
                if level == layer_cl:
                    ops.scatter_add_rows(acc, rows_cl, g_layer, 1.0, check_range=False)
                if level == 1:
                    break
                dst = bufs[1] if acc is bufs[0] else bufs[0]
                acc = ops.spmm_flagged(A, acc, None, 1.0, 1.0, self.G, self.flags, out=dst)
                level -= 1
            ops.spmm_adam(A, acc, 1.0, 0.0, None, self.E0, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        ops.batch_rows_clear_(self.G, self.flags, self.bits, allrows, check_range=False, dup_bits=self.dup_bits)
        return self.loss_out, cl_loss

    def step_sgl(self, u, p, n, view1, view2, cl_rate=0.2, tau=0.2):
        """One SGL training iteration (recommender/SGL.py:54-64,231-256) on the sparse-batch schedule.  Three LightGCN passes --
        the clean graph for BPR + L2, two perturbed graphs (`view1`, `view2`: CSRGraphs of this epoch's edge/node dropout) for ONE
        InfoNCE over the batch's unique users and unique positive items -- each with L-1 full hops + a row-subset hop forward and a
        flag-masked first hop backward; the three table gradients are summed and applied by one dense Adam.  6(L-1) full hops
        instead of the 6L of three autograd passes.  Returns (loss_out, cl_loss)."""
        if self.skip0 or self.optimizer != 'adam' or self.L < 1:
            raise ValueError('step_sgl needs a LightGCN-style engine (layers 0..L averaged) with Adam')
        L, U, N, d = self.L, self.U, self.N, self.d
        B = u.numel()
        self._sparse_buffers(B)
        if getattr(self, '_G_dirty', True):
            self.G.zero_(); self._G_dirty = False
        s = 1.0 / (L + 1)
        rows = torch.cat([u, p + U, n + U])
        rows_cl = torch.cat([torch.unique(u.long()), torch.unique(p.long()) + U]).to(torch.int32)
        if not hasattr(self, '_sgl_acc'):
            self._sgl_acc = torch.empty_like(self.E0)
            self._sgl_hops = [torch.empty_like(self.E0) for _ in range(max(L - 1, 1))]

        def forward_rows(graph, sel):                                        # mean of layers 0..L at the rows `sel`
            layers = [self.E0]
            for k in range(L - 1):
                layers.append(ops.spmm(graph, layers[-1], out=self._sgl_hops[k]))
            return ops.spmm_rows(graph, layers[-1], sel, layers, s, nsplit=self.nsplit, check_range=False)

        def backward_into(graph, sel, grad_c, dst, accumulate):             # dst (+)= dL/dE0 of a pass whose output gradient is grad_c at rows sel
            ops.batch_rows_set_(self.G, self.flags, self.bits, sel, grad_c, 1.0, check_range=False, dup_bits=self.dup_bits)
            beta, Z = (1.0, dst) if accumulate else (0.0, None)
            if L == 1:
                tmp = ops.spmm_flagged(graph, self.G, self.bits, s, s, self.G, self.flags, out=self._sgl_hops[0])
            else:
                acc = ops.spmm_flagged(graph, self.G, self.bits, 1.0, 1.0, self.G, self.flags, out=self._sgl_hops[0])
                for k in range(1, L - 1):
                    acc = ops.spmm_flagged(graph, acc, None, 1.0, 1.0, self.G, self.flags, out=self._sgl_hops[k])
                tmp = ops.spmm_flagged(graph, acc, None, s, s, self.G, self.flags, out=self.hops[0])
            if accumulate:
                dst.add_(tmp)
            else:
                dst.copy_(tmp)
            ops.batch_rows_clear_(self.G, self.flags, self.bits, sel, check_range=False, dup_bits=self.dup_bits)

        # clean pass: BPR + L2 on the batch rows
        out_c = forward_rows(self.A, rows)
        self.Gc.zero_()
        ops.bpr_l2_fwd_bwd(out_c, B, self.ar, self.ar, self.arB, self.reg, self.Gc, workspace=self._ws, loss_out=self.loss_out, check_range=False, distinct_rows=True)
        # the two views at the contrastive rows, one InfoNCE over users and items together
        v1 = forward_rows(view1, rows_cl)
        v2 = forward_rows(view2, rows_cl)
        lcl, d1, d2 = ops.infonce_fwd_bwd(v1, v2, tau)
        cl_loss = cl_rate * lcl[0]
        # backward: three passes into one gradient table, then Adam
        backward_into(self.A, rows, self.Gc, self._sgl_acc, False)
        backward_into(view1, rows_cl, d1 * cl_rate, self._sgl_acc, True)
        backward_into(view2, rows_cl, d2 * cl_rate, self._sgl_acc, True)
        self.t += 1
        ops.adam_dense(self.E0, self._sgl_acc, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        return self.loss_out, cl_loss

    def _sparse_buffers(self, B):
        """Buffers of the sparse-batch step.  The node-sized ones exist once; the batch-sized ones are kept per batch size (an epoch has two:
        the full batches and the last one), so that the last batch of every epoch does not re-allocate them."""
        if getattr(self, '_sparse_B', None) == B:
            return
        dev, d = self.device, self.d
        if not hasattr(self, '_sb_cache'):
            self._sb_cache = {}
            self.flags = torch.zeros(self.N, dtype=torch.uint8, device=dev)          # byte per row: read once per OUTPUT row (epilogue)
            self.bits = torch.zeros((self.N + 31) // 32, dtype=torch.int32, device=dev)   # bit per node: read once per EDGE (masked hop)
            self.dup_bits = torch.zeros_like(self.bits)                                   # rows a batch names more than once (ordered accumulation)
            self.nsplit = 32            # edge ranges per batch row in the row-subset hop (cfg2 sweep: 8: 0.56 ms, 16: 0.33, 32/64: 0.21, 128: 0.36)
            # forward needs E_1..E_{L-1} alive at the same time; Ea/Eb cover L <= 3
            self.hops = [self.Ea, self.Eb] + [torch.empty_like(self.Ea) for _ in range(max(0, self.L - 3))]
        sb = self._sb_cache.get(B)
        if sb is None:
            ar = torch.arange(B, dtype=torch.int32, device=dev)
            sb = {'Gc': torch.zeros(3 * B, d, dtype=torch.float32, device=dev), 'out_c': torch.empty(3 * B, d, dtype=torch.float32, device=dev),
                  'rows_ws': torch.empty(3 * B * self.nsplit * d, dtype=torch.float32, device=dev), 'ar': ar, 'arB': ar + B,
                  '_ws': torch.empty(4 * B, dtype=torch.float32, device=dev)}
            if len(self._sb_cache) >= 8:                   # unusual callers with many batch sizes: drop the oldest set
                del self._sb_cache[next(iter(self._sb_cache))]
            self._sb_cache[B] = sb
        self.Gc, self.out_c, self.rows_ws, self.ar, self.arB, self._ws = sb['Gc'], sb['out_c'], sb['rows_ws'], sb['ar'], sb['arB'], sb['_ws']
        self._sparse_B = B

    def step_dense(self, u, p, n):
        """Reference-shaped step: all 2L hops over the full graph (kept for A/B comparison and for SGD / SimGCL / GMF)."""
        L, A = self.L, self.A
        out = self.forward()
        lo = self.loss_and_grad_out(out, u, p, n)
        self.t += 1
        if self.optimizer == 'adam' and L > 0 and not self.skip0:
            acc = self.G
            bufs = [self.Ea, self.Eb]
            for k in range(L - 1):
                dst = bufs[k % 2]
                ops.spmm(A, acc, 1.0, 1.0, self.G, out=dst)
                acc = dst
            s = 1.0 / (L + 1)
            ops.spmm_adam(A, acc, s, s, self.G, self.E0, self.m, self.v, self.lr, self.t, self.betas, self.eps)
            return lo
        g = self.backward_to_table(self.G)
        self.apply_grad(g)
        return lo

    def apply_grad(self, g):
        if self.optimizer == 'adam':
            ops.adam_dense(self.E0, g, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        elif self.optimizer == 'sgd':
            ops.sgd_dense(self.E0, g, self.lr)
        else:
            raise ValueError('unknown optimizer %r' % (self.optimizer,))

    # --- accounting: algorithmic bytes of one train step
    def step_bytes(self, B):
        E, N, d, L = (self.A.nnz if self.A is not None else 0), self.N, self.d, self.L
        return L * (16 * E + 8 * N + 24 * N * d) + 28 * N * d + 24 * B * d
