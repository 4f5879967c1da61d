"""User-sharded LightGCN training step for the 8 GPUs of one node.  One process per GPU; `torch.distributed`
(backend "nccl" = RCCL over xGMI) provides the collective.  The reference has no multi-GPU path at all; this is new design.

Partition
  * users are split into `world` contiguous blocks; rank r owns user rows [u0, u1): their embedding rows, Adam moments and
    CSR rows.  The item table (I x d; 25.6 MB at cfg2) and its Adam moments are REPLICATED.
  * per-rank buffers are [Ul + I, d]: local users first, then all items -- the same packed layout as the single-GPU
    engine, so every kernel is reused unchanged.
  * the local graph is two rectangular CSR blocks over that packed index space:
      A_u : Ul rows, gathers item rows      (exact: a user's neighbours are all items, all replicated)
      A_i : I  rows, gathers LOCAL user rows (partial sums over this rank's users)
    with the GLOBAL degree normalisation 1/sqrt(deg_u deg_i).
Per propagation hop (forward and backward alike; the adjacency is symmetric):
      launch A_i  ->  async sum-all-reduce of the I x d partial (25.6 MB)  ||  launch A_u on the compute stream  ->  wait.
  That is the one real exchange step of the path: 2L + 1 all-reduces of I*d fp32 per training step (L forward hops, the
  batch gradient's item rows, L backward hops), plus one 3-float all-reduce for the batch-wide loss sums.  The item-side gradient therefore arrives already reduced and every rank applies
  the identical Adam update to its replica ("RCCL all-reduce of item-embedding grads", north_star).
The BPR batch is global (same bit-exact sampler stream on every rank); a rank takes the samples whose user it owns.

The collective and the kernel set are injected (`comm`, `kernels`) so the shard arithmetic can be exercised with
world_size-2 gloo processes on CPU in the test-suite; the defaults are RCCL and the HIP kernels, nothing else.
"""
import contextlib

import numpy as np
import torch


class _TimedWork:
    """all_reduce handle whose wait() is bracketed by two events on the compute stream: their distance is the time the compute stream
    stood still for the collective (exposed communication), whatever ran concurrently before."""

    def __init__(self, work, comm):
        self.work, self.comm = work, comm

    def wait(self):
        c = self.comm
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.work.wait()
        e1.record()
        c.waits.append((e0, e1))
        return True


class _StreamWork:
    """Handle of an exchange enqueued on the communication stream: wait() makes the CALLER's current stream wait for it (no host block)."""

    def __init__(self, done_event):
        self.done = done_event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.done)
        return True


class TorchDistComm:
    """torch.distributed sum-all-reduce (NCCL backend = RCCL on ROCm).  measure=True records, for every wait on a collective, how long
    the compute stream was blocked by it (exposed_ms() sums them after a synchronize; bench.py --gpus N prints it per rank).

    item_exchange = 'direct' (or ARL_ITEM_EXCHANGE=direct): GPU buffers of at least `direct_min_bytes` go through the C ABI's
    arl_allreduce_item_f32 -- a direct reduce-scatter + all-gather over the xGMI links on a communication stream of its own --
    instead of the library all-reduce; everything else (the 3-float loss sums, CPU tensors) stays with torch.distributed."""

    def __init__(self, group=None, measure=False, item_exchange=None, n_chunks=4, direct_min_bytes=1 << 20):
        import os
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.measure = bool(measure)
        self.waits = []
        self.item_exchange = item_exchange or os.environ.get('ARL_ITEM_EXCHANGE', 'torch')
        if self.item_exchange not in ('torch', 'direct'):
            raise ValueError("item_exchange must be 'torch' or 'direct'")
        self.n_chunks, self.direct_min_bytes = int(n_chunks), int(direct_min_bytes)
        self._native = None

    # ---- native exchange (built lazily: needs an initialised process group to carry the communicator id)
    def _native_comm(self, device):
        if self._native is None:
            import ctypes as C
            import os
            from . import _lib
            L = _lib.lib()
            cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')          # the RCCL instance torch itself runs on
            _lib.check(L.arl_comm_load(cand.encode() if os.path.exists(cand) else None), 'arl_comm_load')
            rank, world = self.dist.get_rank(self.group), self.dist.get_world_size(self.group)
            ident = C.create_string_buffer(128)
            if rank == 0:
                _lib.check(L.arl_comm_unique_id(ident), 'arl_comm_unique_id')
            box = [bytes(ident.raw)]
            self.dist.broadcast_object_list(box, src=self.dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            handle = C.c_void_p()
            dev_index = device.index if device.index is not None else torch.cuda.current_device()
            _lib.check(L.arl_comm_init(box[0], rank, world, dev_index, C.byref(handle)), 'arl_comm_init')
            self._native = dict(L=L, handle=handle, world=world, stream=torch.cuda.Stream(device=device), ws=None)
        return self._native

    def _direct(self, t):
        import ctypes as C
        from . import _lib
        nat = self._native_comm(t.device)
        L, n = nat['L'], t.numel()
        need = L.arl_allreduce_item_workspace_bytes(n, nat['world'], self.n_chunks)
        side = nat['stream']
        if nat['ws'] is None or nat['ws'].numel() * 4 < need:
            # the receive workspace is used on the communication stream only: tell the allocator, so that a replaced workspace is not handed out
            # again while an exchange enqueued on that stream still reads it
            nat['ws'] = torch.empty(max(need // 4, 4), dtype=torch.float32, device=t.device)
            nat['ws'].record_stream(side)
        side.wait_event(torch.cuda.current_stream().record_event())          # the producer of `t` runs on the caller's stream
        _lib.check(L.arl_allreduce_item_f32(nat['handle'], C.c_void_p(t.data_ptr()), n, self.n_chunks, C.c_void_p(nat['ws'].data_ptr()),
                                            C.c_void_p(side.cuda_stream)), 'arl_allreduce_item_f32')
        return _StreamWork(side.record_event())

    def _use_direct(self, t):
        return (self.item_exchange == 'direct' and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                and t.numel() * 4 >= self.direct_min_bytes and t.data_ptr() % 16 == 0)

    def close(self):
        if self._native is not None:
            self._native['L'].arl_comm_destroy(self._native['handle'])
            self._native = None

    def all_reduce_async(self, t):
        if self._use_direct(t):
            w = self._direct(t)
            return _TimedWork(w, self) if self.measure else w
        w = self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        return _TimedWork(w, self) if self.measure else w

    def all_reduce(self, t):
        if self._use_direct(t):
            self.all_reduce_async(t).wait()
            return t
        if self.measure:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            e1.record()
            self.waits.append((e0, e1))
            return t
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def exposed_ms(self, reset=True):
        """Sum over the recorded waits of the compute-stream stall (call after torch.cuda.synchronize())."""
        total = sum(a.elapsed_time(b) for a, b in self.waits)
        n = len(self.waits)
        if reset:
            self.waits = []
        return total, n


def shard_bounds(n_users, world):
    """Contiguous user blocks, sizes differing by at most one."""
    base, rem = divmod(int(n_users), int(world))
    starts = [r * base + min(r, rem) for r in range(world + 1)]
    return starts


def build_local_blocks(pairs, n_users, n_items, rank, world, normalise=True):
    """Host-side (numpy) construction of the rank's two CSR blocks from the user-major sorted pair list.
    Returns dict(u0, u1, Au=(rowptr,col,val), Ai=(rowptr,col,val)) with columns in the packed [Ul + I] index space."""
    pairs = np.asarray(pairs)
    U, I = int(n_users), int(n_items)
    u_all = pairs[:, 0].astype(np.int64); i_all = pairs[:, 1].astype(np.int64)
    deg_u = np.bincount(u_all, minlength=U).astype(np.float32)
    deg_i = np.bincount(i_all, minlength=I).astype(np.float32)
    with np.errstate(divide='ignore'):
        du = np.where(deg_u > 0, 1.0 / np.sqrt(deg_u), 0.0).astype(np.float32)       # util/DataLoader.py:77-78 (isinf -> 0)
        di = np.where(deg_i > 0, 1.0 / np.sqrt(deg_i), 0.0).astype(np.float32)
    b = shard_bounds(U, world)
    u0, u1 = b[rank], b[rank + 1]
    Ul = u1 - u0
    lo, hi = np.searchsorted(u_all, u0, 'left'), np.searchsorted(u_all, u1, 'left')      # pairs are user-major sorted
    lu = u_all[lo:hi] - u0
    li = i_all[lo:hi]
    val = ((du[u_all[lo:hi]] * np.float32(1.0)) * di[li]).astype(np.float32)             # (dinv[r]*w)*dinv[c], w = 1
    if not normalise:                 # raw 0/1 blocks + the degrees (ShardedPGA normalises in factors, the fake block changing every step)
        val = np.ones(hi - lo, np.float32)
    rp_u = np.zeros(Ul + 1, np.int64)
    np.cumsum(np.bincount(lu, minlength=Ul), out=rp_u[1:])
    col_u = (li + Ul).astype(np.int32)
    order = np.argsort(li, kind='stable')
    rp_i = np.zeros(I + 1, np.int64)
    np.cumsum(np.bincount(li, minlength=I), out=rp_i[1:])
    col_i = lu[order].astype(np.int32)
    val_i = ((di[li[order]] * np.float32(1.0)) * du[u_all[lo:hi][order]]).astype(np.float32)
    if not normalise:
        val_i = np.ones(hi - lo, np.float32)
    return dict(u0=u0, u1=u1, Au=(rp_u, col_u, val), Ai=(rp_i, col_i, val_i), deg_u=deg_u[u0:u1].copy(), deg_i=deg_i)


class ShardedPropagationEngine:
    """Rank-local state + step() of the user-sharded LightGCN (mean of L+1 layers) + BPR/L2 + dense Adam."""

    def __init__(self, blocks, n_users, n_items, emb_size, n_layers, reg, lr, device, rank, world, table, chunk=512,
                 comm=None, kernels=None, betas=(0.9, 0.999), eps=1e-8, skip_layer0=False, schedule='auto', two_streams=False):
        if kernels is None:
            from . import ops as kernels         # the HIP kernels; fails loudly if libarlib_amd.so is missing
        self.k = kernels
        self.comm = comm if comm is not None else TorchDistComm()
        self.rank, self.world = rank, world
        self.skip0 = bool(skip_layer0)             # SimGCL: layers 1..L averaged (step_simgcl); LightGCN: 0..L (step / step_sparse)
        self.two_streams = two_streams             # step_sparse: item-row and user-row kernels of a hop on two compute streams
        self._side = None
        self.U, self.I, self.d, self.L = int(n_users), int(n_items), int(emb_size), int(n_layers)
        if self.L < 1:
            raise ValueError('the sharded engine is for graph models (n_layers >= 1)')
        self.u0, self.u1 = blocks['u0'], blocks['u1']
        self.Ul = self.u1 - self.u0
        self.Nl = self.Ul + self.I
        self.reg, self.lr, self.betas, self.eps = float(reg), float(lr), betas, eps
        self.device = torch.device(device)
        rp, col, val = blocks['Au']
        self.Au = kernels.CSRGraph(rp, col, val, self.device, chunk=chunk, n_cols=self.Nl)
        rp, col, val = blocks['Ai']
        self.Ai = kernels.CSRGraph(rp, col, val, self.device, chunk=chunk, n_cols=self.Nl)
        # full hops of large shards at d = 64: register-blocked schedule (ops.BlockedPlan), as in engine.PropagationEngine
        if schedule not in ('auto', 'csr', 'blocked'):
            raise ValueError("schedule must be 'auto', 'csr' or 'blocked'")
        if schedule != 'csr' and hasattr(kernels, 'auto_blocked') and (schedule == 'blocked' or self.Au.nnz + self.Ai.nnz >= kernels.BLOCKED_MIN_NNZ):
            import os
            rpw = int(os.environ.get('ARL_SHARD_RPW', 32))             # developer knob (rows per wave of the shard plans)
            kernels.auto_blocked(self.Au, self.d, force=True, rows_per_wave=rpw, small_launch='user')
            kernels.auto_blocked(self.Ai, self.d, force=True, rows_per_wave=rpw, small_launch='item')
        for gph in (self.Au, self.Ai):                       # flag-masked hops take the rows sorted by length (ops.CSRGraph.enable_masked_order)
            if hasattr(gph, 'enable_masked_order') and gph.nnz >= 1_000_000:
                gph.enable_masked_order()
        table = torch.as_tensor(table, dtype=torch.float32)
        if table.shape != (self.U + self.I, self.d):
            raise ValueError('table must be the full [U+I, d] initial table (every rank slices its own rows)')
        z = lambda: torch.zeros(self.Nl, self.d, dtype=torch.float32, device=self.device)
        self.E0 = torch.cat([table[self.u0:self.u1], table[self.U:]], 0).to(self.device).contiguous()
        self.m, self.v, self.G, self.S, self.Ea, self.Eb = z(), z(), z(), z(), z(), z()
        self.t = 0
        self.ws = None
        self.sums = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=self.device)

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    @classmethod
    def from_pairs(cls, pairs, n_users, n_items, emb_size, n_layers, reg, lr, device, rank, world, table, **kw):
        return cls(build_local_blocks(pairs, n_users, n_items, rank, world), n_users, n_items, emb_size, n_layers, reg, lr, device, rank, world, table, **kw)

    # one hop: dst = alpha * (A src) + beta * Z for user rows (exact) and item rows (all-reduced over ranks).
    # Z is added on the item side BEFORE the reduction when `z_partial` (Z holds a per-rank partial, e.g. batch gradients),
    # AFTER it otherwise (Z is replicated, e.g. the running layer sum).
    def _hop(self, src, dst, alpha=1.0, beta=0.0, Z=None, z_partial=False):
        k, Ul = self.k, self.Ul
        di, du = dst[Ul:], dst[:Ul]
        if beta != 0.0 and z_partial:
            k.spmm(self.Ai, src, alpha, beta, Z[Ul:], out=di)
        else:
            k.spmm(self.Ai, src, alpha, out=di)
        work = self.comm.all_reduce_async(di)
        if beta != 0.0:
            k.spmm(self.Au, src, alpha, beta, Z[:Ul], out=du)
        else:
            k.spmm(self.Au, src, alpha, out=du)
        work.wait()
        if beta != 0.0 and not z_partial:
            di.add_(Z[Ul:], alpha=beta)
        return dst

    def forward(self):
        """Propagated tables for the local users and all items: mean(E_0..E_L) (recommender/LightGCN.py:230-240)."""
        if self.skip0:
            raise ValueError('forward()/step()/step_sparse() are the LightGCN mean over layers 0..L; a skip_layer0 engine uses step_simgcl()')
        L = self.L
        cur, nxt = self.Ea, self.Eb
        self._hop(self.E0, cur)
        torch.add(self.E0, cur, out=self.S)
        for _ in range(L - 1):
            self._hop(cur, nxt)
            self.S.add_(nxt)
            cur, nxt = nxt, cur
        self.S.mul_(1.0 / (L + 1))
        return self.S

    def forward_mean(self):
        """forward() for either encoder family: mean of layers 0..L (LightGCN) or 1..L (skip_layer0: SimGCL, unperturbed)."""
        if not self.skip0:
            return self.forward()
        L = self.L
        cur, nxt = self.Ea, self.Eb
        self._hop(self.E0, cur)
        self.S.copy_(cur)
        for _ in range(L - 1):
            self._hop(cur, nxt)
            self.S.add_(nxt)
            cur, nxt = nxt, cur
        self.S.mul_(1.0 / L)
        return self.S

    def backward_mean(self, G):
        """dL/dE0 (local users + item replica) from dL/d(out) = G whose ITEM rows are per-rank partials: one I x d all-reduce completes
        them, then the Horner form of the transposed mean (A symmetric): L hops, each with its item-row exchange."""
        L = self.L
        self.comm.all_reduce(G[self.Ul:])
        bufs = [self.Ea, self.Eb]
        acc = G
        if self.skip0:                                    # (1/L) sum_{k=1..L} A^k G
            for h in range(L - 1):
                acc = self._hop(acc, bufs[h % 2], 1.0, 1.0, G)
            return self._hop(acc, bufs[(L - 1) % 2], 1.0 / L)
        s = 1.0 / (L + 1)                                 # (1/(L+1)) sum_{k=0..L} A^k G
        for h in range(L):
            a = s if h == L - 1 else 1.0
            acc = self._hop(acc, bufs[h % 2], a, a, G, z_partial=False)
        return acc

    # ---- CLeaR's surrogate step (attack/White/CLeaR.py:73-129) on the user-sharded layout -- BASELINE config 4 (SimGCL + CLeaR, user-sharded):
    # forward mean; every rank ranks ITS users against the replicated items (masked top-k, no exchange); the CW pairs (real user x target,
    # negative = tail of that user's list) and the rows of H they generate belong to the user's rank, so
    #   CW   = sum over ranks of local pair sums / (n_real T)             (1 scalar in the final all-reduce)
    #   SFA  : the three weighted passes run on the local rows with the local multiplicities (users T; targets = local real users; negatives'
    #          histogram), cut at the two global reductions r (d floats) and [a | S] (d + 1 floats)  -> 2 tiny all-reduces
    #   dL/d(out): user rows complete on their owner, item rows per-rank partials -> the I x d all-reduce that opens backward_mean().
    # Then Adam on the local user block + the item replica (identical on every rank).
    def step_clear(self, targets, n_real, topk, mask_rowptr=None, mask_col=None, r0=None, warm_idx=None):
        """targets: item ids; n_real: number of real users (global ids [0, n_real): the fake users are the last rows); mask_*: CSR of the
        poisoned interactions over THIS rank's users; r0: the d-vector torch.randn(d) of CLeaR.py:100 (identical on every rank).
        Returns (cw, sfa) as floats-on-device [2] and this rank's top-k lists."""
        k, Ul, d, dev = self.k, self.Ul, self.d, self.device
        T = len(targets)
        out = self.forward_mean()
        top_idx, _ = self.score_topk(out, min(int(topk), self.I), mask_rowptr, mask_col, warm_idx=warm_idx)
        nl = int(min(max(int(n_real) - self.u0, 0), Ul))              # local real users: rows [0, nl)
        tg = torch.as_tensor(list(targets), dtype=torch.int64, device=dev)
        c = 1.0 / (float(n_real) * T)
        fused_cw = hasattr(k, 'cw_topk_term') and nl > 0               # the hand-written CW term (the oracle-backed CPU test double has none)
        if fused_cw:
            # local real users x targets against the replicated item rows: loss share, dL/d(out) (user rows complete, item rows this rank's partial)
            # and the SFA multiplicities of the local rows, in one kernel group
            lo, G, w = k.cw_topk_term(out.contiguous(), Ul, nl, top_idx.contiguous(), tg, c=c, check_range=False)
            cw_local = lo[0]
        else:
            G = torch.zeros_like(out)
            w = torch.zeros(self.Nl, dtype=torch.float32, device=dev)
            cw_local = torch.zeros((), dtype=torch.float32, device=dev)
        if nl and not fused_cw:
            ue = out[:nl]
            ranks = top_idx.shape[1] - 1 - torch.arange(T, device=dev)                  # successive .pop()s (CLeaR.py:84-88)
            neg = top_idx[:nl][:, ranks].long()                                         # [nl, T]
            tgt_rows = out[Ul + tg]                                                     # [T, d]
            sum_u = ue.sum(0)
            for t in range(T):
                nrows = (neg[:, t] + Ul).to(torch.int32).contiguous()
                ne = k.gather_rows(out, nrows, check_range=False)
                cw_local = cw_local + c * ((ue * ne).sum() - (sum_u * tgt_rows[t]).sum())
                G[:nl] += c * (ne - tgt_rows[t])
                k.scatter_add_rows(G, nrows, ue.contiguous(), c, check_range=False)
            G[Ul + tg] -= c * sum_u                                                     # every pair pulls its target: -c * sum of the local real users' rows
            w[:nl] = float(T)
            w[Ul:] = torch.bincount(neg.reshape(-1), minlength=self.I).to(torch.float32)
            w[Ul + tg] += float(nl)
        if r0 is None:
            raise ValueError('step_clear: r0 must be given (the same d-vector on every rank)')
        st = k.SfaStages(out.contiguous(), w, r0.to(dev, torch.float32).contiguous())
        r = st.stage1(); self.comm.all_reduce(r)
        a_s = st.stage2(r); self.comm.all_reduce(a_s)
        sfa, _ = st.stage3(r, a_s, 3 * int(n_real) * T * d, out=G, scale=1.0, accumulate=True)
        res = torch.stack([cw_local, torch.zeros((), device=dev)])
        self.comm.all_reduce(res)
        res[1] = sfa[0]
        grad = self.backward_mean(G)
        self.t += 1
        k.adam_dense(self.E0, grad, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        return res, top_idx

    # ---- NGCF (recommender/NGCF.py:197-212) on the user-sharded layout -- BASELINE config 5's training step.  Per layer ONE sparse hop
    # P = A E (its item rows all-reduced, as every hop here) and a row-local dense part E' = leaky_relu((P + E) W1 + (P * E) W2) with the
    # d x d weights REPLICATED; the item rows of E' are computed redundantly on every rank (they are replicas).  Backward: the weight
    # gradients are sums over rows -> local user rows + (rank 0 only) the item rows, one [2d, d] all-reduce per layer; the table gradient
    # goes back through the same hop (A symmetric).  Adam on the local user block, the item replica and the weight replicas.
    def init_ngcf(self, W1s, W2s, slope=0.01):
        """Attach the replicated layer weights (lists of [d, d] tensors, one pair per layer) and their Adam state."""
        dev = self.device
        self.W = [torch.cat([torch.as_tensor(a, dtype=torch.float32), torch.as_tensor(b, dtype=torch.float32)], 0).to(dev).contiguous() for a, b in zip(W1s, W2s)]
        if len(self.W) != self.L or any(w.shape != (2 * self.d, self.d) for w in self.W):
            raise ValueError('init_ngcf: one [d, d] pair of weights per layer')
        self.Wm = [torch.zeros_like(w) for w in self.W]; self.Wv = [torch.zeros_like(w) for w in self.W]
        self.slope = float(slope)

    def step_ngcf(self, u, p, n):
        k, L, Ul, d, dev = self.k, self.L, self.Ul, self.d, self.device
        B = u.numel()
        if not hasattr(self, 'W'):
            raise ValueError('step_ngcf: call init_ngcf(W1s, W2s) first')
        fused = hasattr(k, 'ngcf_dense_fwd') and d in getattr(k, 'NGCF_DENSE_WIDTHS', ())      # fp32-MFMA dense kernels (ops); the CPU test double has none
        saved = []
        ego = self.E0
        acc = self.E0.clone()
        for l in range(L):
            P = self._hop(ego, torch.empty_like(ego))
            if fused:
                ST, out_l = None, k.ngcf_dense_fwd(P, ego, self.W[l], self.slope)
            else:
                ST = k.ngcf_combine(P, ego)
                out_l = k.ngcf_act_(torch.mm(ST, self.W[l]), None, self.slope)
            saved.append((ego, P, ST, out_l))
            acc += out_l
            ego = out_l
        out = acc.mul_(1.0 / (L + 1))
        lu, lp, ln = self._local_batch(u, p, n)
        if self.ws is None or self.ws.numel() < 4 * max(B, 1):
            self.ws = torch.empty(4 * max(B, 1), dtype=torch.float32, device=dev)
        k.bpr_l2_partial(out, Ul, lu, lp, ln, B, self.ws, self.sums)
        self.comm.all_reduce(self.sums)
        nu_, np_ = torch.sqrt(self.sums[1]), torch.sqrt(self.sums[2])
        self.loss_out[0] = self.sums[0] / B
        self.loss_out[1] = self.reg * (nu_ + np_)
        self.loss_out[2] = nu_
        self.loss_out[3] = np_
        G = torch.zeros_like(out)
        if lu.numel():
            k.bpr_l2_backward(out, Ul, lu, lp, ln, self.reg, self.loss_out, G, self.ws)
        self.comm.all_reduce(G[Ul:])                                   # item rows: per-rank partials -> complete (replicated from here on)
        G.mul_(1.0 / (L + 1))                                          # d(out)/d(layer_k) = 1/(L+1) for every layer
        g_ego = G.clone()                                              # gradient reaching layer L's output
        gWs = [None] * L
        for l in range(L - 1, -1, -1):
            ego_l, P, ST, out_l = saved[l]
            # weight gradient: rows are partitioned (users) or replicated (items: counted on rank 0 only)
            if fused:
                gP, gE = torch.empty_like(P), torch.empty_like(P)
                gW = torch.zeros(2 * d, d, dtype=torch.float32, device=dev)
                go = g_ego.contiguous()
                for lo_, hi_, count in ((0, Ul, True), (Ul, self.Nl, self.rank == 0)):
                    if hi_ > lo_:
                        a, b, w_ = k.ngcf_dense_bwd(go[lo_:hi_], out_l[lo_:hi_], P[lo_:hi_], ego_l[lo_:hi_], self.W[l], self.slope)
                        gP[lo_:hi_] = a; gE[lo_:hi_] = b
                        if count:
                            gW += w_
            else:
                gZ = k.ngcf_act_bwd(g_ego.contiguous(), out_l, self.slope)
                gW = torch.mm(ST[:Ul].t(), gZ[:Ul]) if Ul else torch.zeros(2 * d, d, device=dev)
                if self.rank == 0:
                    gW = gW + torch.mm(ST[Ul:].t(), gZ[Ul:])
                gP, gE = k.ngcf_combine_bwd(torch.mm(gZ, self.W[l].t()), P, ego_l)
            self.comm.all_reduce(gW)
            gWs[l] = gW
            # g(ego_l) = A gP + gE (+ G/(L+1): layer l's own share of the mean); gP's item rows are replicas, so the hop's item-side partial
            # sums over local users are completed by its all-reduce and gE / G are added after it
            back = self._hop(gP.contiguous(), torch.empty_like(gP), 1.0, 1.0, gE, z_partial=False)
            g_ego = back.add_(G)
        self.t += 1
        k.adam_dense(self.E0, g_ego, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        for l in range(L):
            k.adam_dense(self.W[l], gWs[l].contiguous(), self.Wm[l], self.Wv[l], self.lr, self.t, self.betas, self.eps)
        return self.loss_out

    def _local_batch(self, u, p, n):
        sel = (u >= self.u0) & (u < self.u1)
        return (u[sel] - self.u0).to(torch.int32).contiguous(), p[sel].contiguous(), n[sel].contiguous()

    def step(self, u, p, n):
        """One training iteration on the GLOBAL batch (device int32 tensors, identical on every rank)."""
        k, L, Ul = self.k, self.L, self.Ul
        B = u.numel()
        out = self.forward()
        lu, lp, ln = self._local_batch(u, p, n)
        if self.ws is None or self.ws.numel() < 4 * max(B, 1):
            self.ws = torch.empty(4 * max(B, 1), dtype=torch.float32, device=self.device)
        k.bpr_l2_partial(out, Ul, lu, lp, ln, B, self.ws, self.sums)
        self.comm.all_reduce(self.sums)
        nu, np_ = torch.sqrt(self.sums[1]), torch.sqrt(self.sums[2])
        self.loss_out[0] = self.sums[0] / B
        self.loss_out[1] = self.reg * (nu + np_)
        self.loss_out[2] = nu
        self.loss_out[3] = np_
        self.G.zero_()
        if lu.numel():
            k.bpr_l2_backward(out, Ul, lu, lp, ln, self.reg, self.loss_out, self.G, self.ws)
        # backward, Horner form.  The user-side hop gathers ITEM rows of its operand, so G's item rows (per-rank partials:
        # each rank saw only its own samples) must be complete first: one more I x d all-reduce, after which G is
        # replicated on the item side and is added after each hop's reduction.
        self.comm.all_reduce(self.G[Ul:])
        self.t += 1
        acc = self.G
        bufs = [self.Ea, self.Eb]
        s = 1.0 / (L + 1)
        for h in range(L):
            last = h == L - 1
            dst = bufs[h % 2]
            a = s if last else 1.0
            self._hop(acc, dst, a, a, self.G, z_partial=False)
            acc = dst
        if hasattr(k, 'adam_dense'):
            k.adam_dense(self.E0, acc, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        return self.loss_out

    # ---- sparse-batch step (same idea as PropagationEngine.step): L-1 full hops + a row-subset hop forward, a flag-masked
    # first hop backward.  The compact batch rows [3B, d] are completed with ONE small all-reduce (users: the owner
    # contributes the row, everybody else zeros; items: partial sums), after which every rank evaluates the loss on the
    # whole batch redundantly (a few microseconds) -- so the batch-wide mean/norms and the item-side gradient need no
    # further collective.  Full-size all-reduces per step: 2(L-1) + 1 instead of 2L + 1.
    def step_sparse(self, u, p, n):
        # NOTE: batch indices must be range-checked by the caller (bench.py / the training loop do it once per batch chunk);
        # every op below runs with check_range=False so the step has no host synchronisation
        if self.skip0:
            raise ValueError('step_sparse() is the LightGCN step; a skip_layer0 engine uses step_simgcl()')
        k, L, Ul, d = self.k, self.L, self.Ul, self.d
        B = u.numel()
        dev = self.device
        s = 1.0 / (L + 1)
        if getattr(self, '_sp_B', None) != B:
            self.flags = torch.zeros(self.Nl, dtype=torch.uint8, device=dev)
            self.bits = torch.zeros((self.Nl + 31) // 32, dtype=torch.int32, device=dev)
            self.dup_bits = torch.zeros_like(self.bits)
            self.C = torch.zeros(3 * B, d, dtype=torch.float32, device=dev)
            self.Gc = torch.zeros(3 * B, d, dtype=torch.float32, device=dev)
            self.ar = torch.arange(B, dtype=torch.int32, device=dev)
            self.arB = self.ar + B
            self.ws = torch.empty(4 * B, dtype=torch.float32, device=dev)
            self.hops = [self.Ea, self.Eb] + [torch.empty_like(self.Ea) for _ in range(max(0, L - 3))]
            self._prep = None
            self.G.zero_()
            self._sp_B = B
        # static shapes, no host sync: samples of other ranks' users are kept with a clamped row id and a zero weight.  One launch builds
        # the local row ids, the ownership weights and the packed row lists (arl_shard_batch_prep_i32)
        self._prep = k.shard_batch_prep(u, p, n, self.u0, self.u1, out=self._prep)
        lu, own, item_rows, rows_l = self._prep
        item_rows_packed = rows_l[B:]
        # forward.  Software pipeline over hops: the item-row all-reduce of hop h runs behind A_u(h) AND A_i(h+1) -- the
        # item-side kernel of the next hop only gathers USER rows, which are local and already final.
        # Optional (two_streams=True; measured on one rank's share of cfg2: 1.26 -> 1.34 ms wall per step, so OFF by default): A_i(h) on the
        # main stream and A_u(h) on a side stream -- they read the same operand and write disjoint row blocks.  Events order exactly what depends:
        #   A_i(h+1) after A_u(h) (user rows of the new layer);  A_u(h) after the all-reduce of hop h-1 (item rows of its operand).
        two = self.two_streams and self.device.type == 'cuda'
        if two:
            main, side = torch.cuda.current_stream(), self._side_stream()
            side.wait_stream(main)                             # the previous step's updates of E0 / the sparse state
        on_side = (lambda: torch.cuda.stream(side)) if two else contextlib.nullcontext
        layers = [self.E0]
        pending = ev_u = ev_ar = None
        for h in range(L - 1):
            src, dst = layers[-1], self.hops[h]
            if two and ev_u is not None:
                main.wait_event(ev_u)                          # src's user rows come from the side stream
            k.spmm(self.Ai, src, out=dst[Ul:])                 # partial item rows (gathers user rows of src)
            if not two:
                # one compute stream: a collective's stream waits, at enqueue time, for everything the caller's stream holds -- so the exchange
                # of hop h is enqueued right after A_i(h), BEFORE A_u(h) is, and runs behind A_u(h) and A_i(h+1)
                if pending is not None:
                    pending.wait()                             # src's item rows are complete from here on
                pending = self.comm.all_reduce_async(dst[Ul:])
                k.spmm(self.Au, src, out=dst[:Ul])             # exact user rows (gathers item rows of src)
                layers.append(dst)
                continue
            with on_side():
                if pending is not None:
                    side.wait_event(ev_ar)
                    pending.wait()
                k.spmm(self.Au, src, out=dst[:Ul])
                ev_u = side.record_event()
            pending = self.comm.all_reduce_async(dst[Ul:])     # (main stream: after A_i(h) only -- A_u(h) is on the side stream)
            ev_ar = main.record_event()
            layers.append(dst)
        if two and ev_u is not None:
            main.wait_event(ev_u)                              # the row-subset hops below read the last layer's user rows
        X = layers[-1]
        # compact batch rows, already scaled by 1/(L+1): item rows = per-rank partial of the last hop + the layers' own rows, which are
        # replicas: layer j is contributed by rank j % world alone (its rows ride in the row-subset hop's epilogue); user rows = complete on the
        # owner, zero elsewhere (row weight `own`).  One [3B, d] all-reduce completes them.
        if pending is not None and L >= 2:
            # the last full hop's item rows (layers[-1][Ul:]) must be complete before they are read below as a replicated layer
            mine = [t[Ul:] for j, t in enumerate(layers[:-1]) if j % self.world == self.rank]
        else:
            mine = [t[Ul:] for j, t in enumerate(layers) if j % self.world == self.rank]
        k.spmm_rows(self.Ai, X, item_rows, mine, s, out=self.C[B:], check_range=False)      # gathers user rows only: overlaps the last full all-reduce
        if pending is not None:
            pending.wait()
            if L >= 2 and (L - 1) % self.world == self.rank:   # the newest layer's item rows, readable only now
                self.C[B:].add_(k.gather_rows(X, item_rows_packed, check_range=False), alpha=s)
        if Ul:
            k.spmm_rows(self.Au, X, lu, [t[:Ul] for t in layers], s, out=self.C[:B], check_range=False, row_weight=own[:B])
        else:
            self.C[:B].zero_()
        self.comm.all_reduce(self.C)
        # loss on the whole batch (identical on every rank), compact per-sample gradients; one launch puts them into G (foreign samples:
        # factor 0 on a clamped local row) and marks the rows
        self.Gc.zero_()
        k.bpr_l2_fwd_bwd(self.C, B, self.ar, self.ar, self.arB, self.reg, self.Gc, workspace=self.ws, loss_out=self.loss_out, check_range=False, distinct_rows=True)
        k.batch_rows_set_(self.G, self.flags, self.bits, rows_l, self.Gc, 1.0, check_range=False, row_scale=own, dup_bits=self.dup_bits)
        # backward (Horner).  hop 1: flag-masked gathers; later hops dense; G (complete on the item side, <= 2B distinct rows) is added to
        # the reduced item rows by a row-list kernel instead of a pass over the whole I x d block
        self.t += 1
        zu = self.flags[:Ul]
        acc = self.G
        pending, prev_items, prev_a = None, None, 1.0
        ev_u = ev_ar = None
        ev_g = main.record_event() if two else None                                # G, flags and bits are set (main stream)
        for h in range(L):
            last = h == L - 1
            a = s if last else 1.0
            dst = self.hops[h % len(self.hops)]
            if dst is acc:
                dst = self.hops[(h + 1) % len(self.hops)]
            xf = self.bits if h == 0 else None
            if two and ev_u is not None:
                main.wait_event(ev_u)                                              # acc's user rows come from the side stream
            k.spmm_flagged(self.Ai, acc, xf, a, 0.0, None, None, out=dst[Ul:])     # partial item rows (gathers acc's user rows)
            if not two:
                # same order as in the forward: complete acc's item rows, enqueue THIS hop's exchange, then launch A_u -- the exchange of the
                # last hop runs behind the user block's fused Adam hop
                if pending is not None:
                    pending.wait()
                    k.rows_axpy_unique_(prev_items, self.G, item_rows_packed, prev_a, check_range=False, dup_bits=self.dup_bits)
                pending, prev_items, prev_a = self.comm.all_reduce_async(dst[Ul:]), dst, a
                if last:
                    k.spmm_adam(self.Au, acc, a, a, self.G[:Ul], self.E0[:Ul], self.m[:Ul], self.v[:Ul], self.lr, self.t, self.betas, self.eps, zflags=zu)
                else:
                    k.spmm_flagged(self.Au, acc, xf, a, a, self.G[:Ul], zu, out=dst[:Ul])
                acc = dst
                continue
            with on_side():
                if pending is not None:                                            # complete acc's item rows before A_u reads them
                    side.wait_event(ev_ar)
                    pending.wait()
                    k.rows_axpy_unique_(prev_items, self.G, item_rows_packed, prev_a, check_range=False, dup_bits=self.dup_bits)
                else:
                    side.wait_event(ev_g)
                if last:
                    k.spmm_adam(self.Au, acc, a, a, self.G[:Ul], self.E0[:Ul], self.m[:Ul], self.v[:Ul], self.lr, self.t, self.betas, self.eps, zflags=zu)
                else:
                    k.spmm_flagged(self.Au, acc, xf, a, a, self.G[:Ul], zu, out=dst[:Ul])
                ev_u = side.record_event()
            pending, prev_items, prev_a = self.comm.all_reduce_async(dst[Ul:]), dst, a
            ev_ar = main.record_event()
            acc = dst
        pending.wait()
        k.rows_axpy_unique_(prev_items, self.G, item_rows_packed, prev_a, check_range=False, dup_bits=self.dup_bits)
        k.adam_dense(self.E0[Ul:], acc[Ul:], self.m[Ul:], self.v[Ul:], self.lr, self.t, self.betas, self.eps)
        if two:
            main.wait_event(ev_u)                                                  # the user block's fused Adam hop (side stream) reads G / flags
        k.batch_rows_clear_(self.G, self.flags, self.bits, rows_l, check_range=False, dup_bits=self.dup_bits)      # clear the sparse state
        return self.loss_out

    # ---- SimGCL (recommender/SimGCL.py:51-63,198-219) on the user-sharded layout -- BASELINE config 4's training step.
    # Three forwards (clean + two perturbed views; layer 0 is not averaged in), BPR + L2 on the clean one, InfoNCE between the
    # views at the batch's unique users and unique positive items.  The perturbation carries no gradient, so the three output
    # gradients share ONE backward operator (1/L) sum_{k=1..L} A^k and are summed first.
    # Exchanges: one I x d all-reduce per hop (3L forward + L backward), one [2*nu, d] all-reduce for the user rows of the
    # two views (each rank contributes the rows it owns), the 3 loss sums, and G's item rows before the backward pass.
    def step_simgcl(self, u, p, n, cl_rate=0.2, tau=0.2, eps=0.1, noises=None):
        """noises: optional [view][hop] tensors over the LOCAL packed rows [Ul + I, d].  By default user rows draw from the
        rank's generator and the replicated item rows from a generator every rank seeds identically (step counter)."""
        if not self.skip0:
            raise ValueError('step_simgcl needs skip_layer0=True (SimGCL averages layers 1..L)')
        k, L, Ul, d, dev = self.k, self.L, self.Ul, self.d, self.device
        B = u.numel()
        inv = 1.0 / L
        if not hasattr(self, 'S1'):
            self.S1, self.S2 = torch.zeros_like(self.S), torch.zeros_like(self.S)

        def noise(view, hop):
            if noises is not None:
                return noises[view][hop]
            g = torch.Generator(device=dev)
            g.manual_seed(0x51AC1 + 7919 * self.t + 2 * hop + view)                 # identical on every rank: replicated item rows
            item = torch.rand(self.I, d, generator=g, device=dev)
            return torch.cat([torch.rand(Ul, d, device=dev), item], 0)

        def forward(view, acc):
            cur, bufs = self.E0, [self.Ea, self.Eb]
            for h in range(L):
                dst = bufs[h % 2]
                self._hop(cur, dst)
                if view is not None:
                    if noises is None and hasattr(k, 'simgcl_perturb_rng'):
                        # noise drawn inside the kernel from (seed, step/view/hop stream, GLOBAL row id, column): the replicated item rows are
                        # bit-identical on every rank by construction, and the draw does not depend on the world size
                        if getattr(self, '_gid', None) is None:
                            self._gid = torch.cat([torch.arange(self.u0, self.u1, device=dev), torch.arange(self.U, self.U + self.I, device=dev)]).to(torch.int32)
                        k.simgcl_perturb_rng(dst, eps, 0x51AC1, (self.t * 2 + view) * L + h, out=dst, row_ids=self._gid)
                    else:
                        k.simgcl_perturb_(dst, noise(view, h), eps)
                if h == 0:
                    acc.copy_(dst)
                else:
                    acc.add_(dst)
                cur = dst
            return acc.mul_(inv)

        out = forward(None, self.S)
        v1, v2 = forward(0, self.S1), forward(1, self.S2)
        # ---- rec loss on the clean forward (as in step())
        lu, lp, ln = self._local_batch(u, p, n)
        if self.ws is None or self.ws.numel() < 4 * max(B, 1):
            self.ws = torch.empty(4 * max(B, 1), dtype=torch.float32, device=dev)
        k.bpr_l2_partial(out, Ul, lu, lp, ln, B, self.ws, self.sums)
        self.comm.all_reduce(self.sums)
        nu_, np_ = torch.sqrt(self.sums[1]), torch.sqrt(self.sums[2])
        self.loss_out[0] = self.sums[0] / B
        self.loss_out[1] = self.reg * (nu_ + np_)
        self.loss_out[2] = nu_
        self.loss_out[3] = np_
        self.G.zero_()
        if lu.numel():
            k.bpr_l2_backward(out, Ul, lu, lp, ln, self.reg, self.loss_out, self.G, self.ws)
        self.comm.all_reduce(self.G[Ul:])                                        # item rows: per-rank partials -> complete
        # ---- contrastive loss: unique users of the GLOBAL batch (rows live on their owners), unique positive items (replicated)
        uidx = torch.unique(u.long())
        iidx = torch.unique(p.long())
        nu = uidx.numel()
        own = (uidx >= self.u0) & (uidx < self.u1)
        loc = (uidx - self.u0).clamp_(0, max(Ul - 1, 0))
        Cv = torch.zeros(2 * nu, d, dtype=torch.float32, device=dev)
        if Ul:
            ownf = own.to(torch.float32).unsqueeze(1)
            Cv[:nu] = v1[loc] * ownf
            Cv[nu:] = v2[loc] * ownf
        self.comm.all_reduce(Cv)
        l_u, du1, du2 = k.infonce_fwd_bwd(Cv[:nu].contiguous(), Cv[nu:].contiguous(), tau)
        l_i, di1, di2 = k.infonce_fwd_bwd(v1[Ul + iidx].contiguous(), v2[Ul + iidx].contiguous(), tau)
        cl_loss = cl_rate * (l_u[0] + l_i[0])
        if Ul:
            self.G.index_add_(0, loc[own], (du1 + du2)[own] * cl_rate)           # only the owner holds the row
        self.G.index_add_(0, Ul + iidx, (di1 + di2) * cl_rate)                    # identical on every rank, after the reduction
        # ---- one backward pass for the three forwards, Adam on the local block + the item replica
        self.t += 1
        acc, bufs = self.G, [self.Ea, self.Eb]
        for h in range(L - 1):
            acc = self._hop(acc, bufs[h % 2], 1.0, 1.0, self.G)
        grad = self._hop(acc, bufs[(L - 1) % 2], inv)
        k.adam_dense(self.E0, grad, self.m, self.v, self.lr, self.t, self.betas, self.eps)
        return self.loss_out, cl_loss

    def score_topk(self, table, k, mask_rowptr=None, mask_col=None, warm_idx=None):
        """Masked top-k item lists of THIS rank's users from a propagated local table [Ul + I, d] (forward() / the SimGCL views):
        the scoring pass of the attack loops and of the evaluation is embarrassingly parallel over the user shards -- local user
        rows against the replicated item rows, no exchange.  mask_*: CSR over the local users."""
        Ul = self.Ul
        if Ul == 0:
            return (torch.zeros(0, k, dtype=torch.int32, device=self.device), torch.zeros(0, k, dtype=torch.float32, device=self.device))
        kw = {} if warm_idx is None else {'warm_idx': warm_idx}
        return self.k.score_mask_topk(table[:Ul].contiguous(), table[Ul:].contiguous(), k, mask_rowptr, mask_col, **kw)

    def gather_full_table(self):
        """[U+I, d] table assembled on every rank (tests / checkpoints): all-gather of the user blocks + the replica."""
        full = torch.zeros(self.U + self.I, self.d, dtype=torch.float32, device=self.device)
        full[self.u0:self.u1] = self.E0[:self.Ul]
        self.comm.all_reduce(full[:self.U])
        full[self.U:] = self.E0[self.Ul:]
        return full


class ShardedPGA:
    """PGA's gradient step w.r.t. the fake interactions (attack/White/PGA.py:92-142) on the user-sharded layout -- the attack-step half of
    BASELINE configs 3 / 4 at N > 1.  Users [0, U + F) are split over the ranks like everywhere here; the F fake users are the LAST rows, so
    they (and the fake block S [F, I], which only their rank ever holds) live on the last rank.  The poisoned operator is applied in factors,
    as attack.White.PGA.FactoredFakeGraph does on one GPU:   A_hat X = D^-1/2 ( W_real (D^-1/2 X) + fake block ),
      * W_real: the raw 0/1 blocks A_u / A_i of the REAL interactions (fixed; item-side partial sums all-reduced per hop, as in every hop here);
      * fake block: S (D^-1/2 X)_items for the fake users' rows and S^T (D^-1/2 X)_fake added to the item partial -- on the owner rank only;
      * degrees: real users fixed, fake users S's row sums (owner), items real degree + S's column sums (ONE [I]-float all-reduce per step).
    CW pairs (real user x target, negative = tail of the user's masked top-k list) stay with the user's rank; dL/d(out) item rows are
    per-rank partials (one I x d all-reduce).  The F x I block gradient needs rows of E_k / dE_k of the fake users and of all items: all on
    the owner.  Exchanges per step: 1 [I] + (2L - 1) [I, d] hop reductions + 1 [I, d] gradient + 1 scalar."""

    def __init__(self, pairs_real, n_real, n_fake, n_items, emb_size, n_layers, device, rank, world, table, comm=None, kernels=None, chunk=512):
        if kernels is None:
            from . import ops as kernels
        self.k, self.comm = kernels, comm if comm is not None else TorchDistComm()
        self.rank, self.world = rank, world
        self.U, self.F, self.I, self.d, self.L = int(n_real), int(n_fake), int(n_items), int(emb_size), int(n_layers)
        self.Up = self.U + self.F
        b = build_local_blocks(pairs_real, self.Up, self.I, rank, world, normalise=False)
        self.u0, self.u1 = b['u0'], b['u1']
        self.Ul, self.Nl = self.u1 - self.u0, self.u1 - self.u0 + self.I
        if shard_bounds(self.Up, world)[world - 1] > self.U:
            raise ValueError('ShardedPGA: the fake users must all live on the last rank (F <= users per rank)')
        self.owner = rank == world - 1
        self.f0 = self.U - self.u0 if self.owner else self.Ul            # local row of the first fake user
        self.device = torch.device(device)
        self.Au = kernels.CSRGraph(*b['Au'], self.device, chunk=chunk, n_cols=self.Nl)
        self.Ai = kernels.CSRGraph(*b['Ai'], self.device, chunk=chunk, n_cols=self.Nl)
        self.deg_u = torch.from_numpy(b['deg_u']).to(self.device)
        self.deg_i = torch.from_numpy(b['deg_i']).to(self.device)
        table = torch.as_tensor(table, dtype=torch.float32)
        if table.shape != (self.Up + self.I, self.d):
            raise ValueError('table must be the full [U + F + I, d] table')
        self.E0 = torch.cat([table[self.u0:self.u1], table[self.Up:]], 0).to(self.device).contiguous()
        self.fake_rows = torch.arange(self.f0, self.Ul, dtype=torch.int32, device=self.device)
        self.S = None

    def set_block(self, S):
        """S [F, I] on the owner (ignored elsewhere)."""
        if self.owner:
            self.S = torch.as_tensor(S, dtype=torch.float32).to(self.device).contiguous()

    def _degrees(self):
        cs = self.S.sum(0) if self.owner else torch.zeros(self.I, dtype=torch.float32, device=self.device)
        self.comm.all_reduce(cs)
        du = self.deg_u.clone()
        if self.owner:
            du[self.f0:] = self.S.sum(1)
        rs = torch.cat([du, self.deg_i + cs])
        self.dinv = torch.where(rs > 0, 1.0 / torch.sqrt(rs), torch.zeros_like(rs))
        self._dcol = self.dinv[:, None].contiguous()

    def _hop(self, X, alpha=1.0, beta=0.0, Z=None):
        """alpha * (A_hat X) + beta * Z on the local rows (Z's item rows replicated).  One element-wise pass (D^-1/2 X: the producer of X does not
        know the degrees of this step) and, when beta != 0, one I x d add after the reduction; the output-side D^-1/2, alpha and the user rows'
        beta * Z ride in the SpMM epilogues (row_scale), the fake block's products carry their own row scale."""
        k, Ul = self.k, self.Ul
        Xs = X * self._dcol
        Y = torch.empty_like(X)
        du, di = self.dinv[:Ul], self.dinv[Ul:]
        k.spmm(self.Ai, Xs, alpha, out=Y[Ul:], row_scale=di)            # alpha D_i^-1/2 (partial over the local real users)
        if self.owner and self.F:
            if hasattr(k, 'fake_block_cols_'):                          # + the fake users' contribution to every item row (hand-written product;
                k.fake_block_cols_(self.S, Xs[self.f0:Ul], Y[Ul:], rscale=di, alpha=alpha)      #   the oracle-backed CPU test double has no such kernel)
            else:
                Y[Ul:].add_((self.S.t() @ Xs[self.f0:Ul]) * di[:, None], alpha=alpha)
        work = self.comm.all_reduce_async(Y[Ul:])
        if beta != 0.0:
            k.spmm(self.Au, Xs, alpha, beta, Z[:Ul], out=Y[:Ul], row_scale=du)
        else:
            k.spmm(self.Au, Xs, alpha, out=Y[:Ul], row_scale=du)
        if self.owner and self.F:
            if hasattr(k, 'fake_block_rows_'):                          # the fake users' own rows
                k.fake_block_rows_(self.S, Xs[Ul:], Y[self.f0:Ul], rscale=du[self.f0:], alpha=alpha)
            else:
                Y[self.f0:Ul].add_((self.S @ Xs[Ul:]) * du[self.f0:, None], alpha=alpha)
        work.wait()
        if beta != 0.0:
            Y[Ul:].add_(Z[Ul:], alpha=beta)                             # replicated rows: once, after the reduction
        return Y

    def forward(self):
        self._degrees()
        E = [self.E0]
        out = self.E0.clone()
        for _ in range(self.L):
            E.append(self._hop(E[-1]))
            out += E[-1]
        out /= (self.L + 1)
        return out, E

    def step(self, targets, top_idx):
        """One gradient step on S.  top_idx: masked top-k lists of THIS rank's users from the inner epoch's forward (fixed over the epoch's
        steps, PGA.py:99-108).  Returns the CW loss (device scalar, identical on every rank); S is updated in place on the owner."""
        k, Ul, L, dev = self.k, self.Ul, self.L, self.device
        T = len(targets)
        out, E = self.forward()
        nl = int(min(max(self.U - self.u0, 0), Ul))
        tg = torch.as_tensor(list(targets), dtype=torch.int64, device=dev)
        c = 1.0 / (float(self.U) * T)
        G = torch.zeros_like(out)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        if nl:
            ue = out[:nl]
            ranks = top_idx.shape[1] - 1 - torch.arange(T, device=dev)
            neg = top_idx[:nl][:, ranks].long()
            tgt_rows = out[Ul + tg]
            sum_u = ue.sum(0)
            for t in range(T):
                nrows = (neg[:, t] + Ul).to(torch.int32).contiguous()
                ne = k.gather_rows(out, nrows, check_range=False)
                loss += c * ((ue * ne).sum() - (sum_u * tgt_rows[t]).sum())
                G[:nl] += c * (ne - tgt_rows[t])
                k.scatter_add_rows(G, nrows, ue.contiguous(), c, check_range=False)
            G[Ul + tg] -= c * sum_u
        self.comm.all_reduce(G[Ul:])
        self.comm.all_reduce(loss)
        s = 1.0 / (L + 1)
        Gs = G * s
        dE = [None] * (L + 1)
        dE[L] = Gs
        for kk in range(L - 1, 0, -1):
            dE[kk] = self._hop(dE[kk + 1], 1.0, 1.0, Gs)
        if self.owner and self.F:
            block = torch.zeros(self.F, self.I, dtype=torch.float32, device=dev)
            for kk in range(L):
                k.sddmm_rows_dense(dE[kk + 1].contiguous(), E[kk].contiguous(), self.fake_rows, Ul, self.I, out=block)
                k.sddmm_rows_dense(E[kk].contiguous(), dE[kk + 1].contiguous(), self.fake_rows, Ul, self.I, out=block)
            self.last_block = block
            k.pga_update_(self.S, block, self.dinv[self.f0:Ul].contiguous(), self.dinv[Ul:].contiguous())
        return loss[0]
