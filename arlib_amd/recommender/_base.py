"""Shared machinery of the GMF / LightGCN / SimGCL mirrors (the reference repeats it per file:
recommender/LightGCN.py:17-161,202-252, GMF.py:16-175, SimGCL.py:18-231).

* `GraphEncoder` -- nn.Module with the reference encoder's surface: callable -> (user_emb, item_emb), differentiable;
  `.embedding_dict['user_emb'|'item_emb']` real nn.Parameters (views of ONE packed [U+I,d] device buffer, so
  torch.cat is never needed); `.sparse_norm_adj` (device CSR); `._init_uiAdj(scipy)`; `.attack_emb(du, di)`; `.cuda()`.
* `Recommender` -- train()/save()/predict()/evaluate()/test() with the reference signatures.  train() runs the
  fused device engine when it owns the optimizer (or is handed a plain torch Adam/SGD over exactly these parameters)
  and otherwise drives the caller's optimizer through autograd with the same kernels.
"""
import itertools
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from .. import ops
from ..engine import PropagationEngine
from ..util.sampler import next_batch_pairwise, device_epoch
from ..util.loss import bpr_l2_loss, l2_reg_loss, InfoNCE
from ..util.metrics import ranking_evaluation, ranking_evaluation_topk
from ..util.optim import Adam as FusedAdam

DEVICE = 'cuda'


class SparseNormAdj:
    """Device image of the normalised adjacency (the reference's `sparse_norm_adj` COO tensor, LightGCN.py:247-252),
    kept as CSR with int32 indices.  `.requires_grad` / `.grad` mirror the tensor attributes attacks poke
    (attack/White/PGA.py:98,117): the gradient lives on `.values` (one fp32 per stored edge, CSR order)."""

    def __init__(self, mat):
        if hasattr(mat, 'device_graph') and not sp.issparse(mat):
            # array-native data (util/DataLoader.LazyNormAdj): assembled and normalised on the device, never as a scipy matrix
            g = mat.device_graph(DEVICE)
            self.shape = tuple(mat.shape)
            self._indptr = self._indices = None
            self.values, self.dinv, self._graph = g.val, g.dinv, g
            return
        m = sp.csr_matrix(mat, dtype=np.float32)
        m.sort_indices()
        self.shape = m.shape
        self.indptr, self.indices = m.indptr.astype(np.int64), m.indices.astype(np.int32)
        self.values = torch.from_numpy(m.data.astype(np.float32))
        self._graph = None

    # host images of the pattern: numpy arrays, fetched from the device graph the first time somebody reads them
    @property
    def indptr(self):
        if self._indptr is None:
            self._indptr = self._graph.rowptr.cpu().numpy().astype(np.int64)
        return self._indptr

    @indptr.setter
    def indptr(self, a):
        self._indptr = a

    @property
    def indices(self):
        if self._indices is None:
            self._indices = self._graph.col.cpu().numpy()
        return self._indices

    @indices.setter
    def indices(self, a):
        self._indices = a

    @property
    def requires_grad(self):
        return self.values.requires_grad

    @requires_grad.setter
    def requires_grad(self, flag):
        self.values.requires_grad_(bool(flag))

    @property
    def grad(self):
        return self.values.grad

    def cuda(self):
        if not self.values.is_cuda:
            rg = self.values.requires_grad
            self.values = self.values.detach().to(DEVICE).requires_grad_(rg)
            self._graph = None
        return self

    def graph(self):
        self.cuda()
        if self._graph is None or self._graph.val.data_ptr() != self.values.data_ptr():
            if self._graph is None:
                self._graph = ops.CSRGraph(self.indptr, self.indices, self.values.detach(), self.values.device)
            else:
                self._graph = self._graph.with_values(self.values.detach())
        return self._graph

    def to_scipy(self):
        return sp.csr_matrix((self.values.detach().cpu().numpy(), self.indices, self.indptr), shape=self.shape)

    def __getstate__(self):
        self.indptr, self.indices                               # host images of the pattern travel, the device graph does not
        st = dict(self.__dict__)
        st['_graph'] = None
        st['values'] = self.values.detach().cpu()
        if st.get('dinv') is not None:
            st['dinv'] = st['dinv'].detach().cpu()
        return st


class TorchGraphInterface(object):
    @staticmethod
    def convert_sparse_mat_to_tensor(X):
        return SparseNormAdj(X)


class _Propagate(torch.autograd.Function):
    """(user_emb, item_emb) -> propagated (user_emb, item_emb); forward and backward are SpMM kernel launches."""

    @staticmethod
    def forward(ctx, user_emb, item_emb, enc, noises):
        eng = enc._engine()
        out = torch.empty_like(eng.E0)
        eng.forward(noises=noises, eps=enc.eps, out=out)
        ctx.enc, ctx.noises = enc, noises
        U = user_emb.shape[0]
        return out[:U], out[U:]

    @staticmethod
    def backward(ctx, g_user, g_item):
        eng = ctx.enc._engine()
        G = torch.cat([g_user, g_item], 0).contiguous()
        sink = getattr(ctx.enc, '_adj_sink', None)
        if sink is not None:            # train(requires_adjgrad=True): every forward through the adjacency leaves its share of sparse_norm_adj.grad here
            eng.adjacency_gradient(G, out=sink, noises=ctx.noises, eps=ctx.enc.eps)
        dE0 = eng.backward_to_table(G).clone()
        U = g_user.shape[0]
        return dE0[:U], dE0[U:], None, None


class _PropagateRows(torch.autograd.Function):
    """(user_emb, item_emb, rows) -> the propagated LightGCN mean on the listed rows only, [len(rows), d]; forward and backward on the
    sparse-batch schedule of the fused step (engine.forward_rows / backward_rows): 2L-2 full hops instead of 2L."""

    @staticmethod
    def forward(ctx, user_emb, item_emb, enc, rows):
        eng = enc._engine()
        ctx.enc, ctx.rows, ctx.U = enc, rows, user_emb.shape[0]
        return eng.forward_rows(rows)

    @staticmethod
    def backward(ctx, g_rows):
        dE0 = ctx.enc._engine().backward_rows(ctx.rows, g_rows)
        return dE0[:ctx.U], dE0[ctx.U:], None, None


class GraphEncoder(nn.Module):
    n_prop_layers = 0          # 0 = plain matrix factorisation
    skip_layer0 = False
    eps = 0.1

    def __init__(self, data, emb_size):
        super().__init__()
        self.data = data
        self.latent_size = self.emb_size = emb_size
        self.embedding_dict = self._init_model()
        self._eng = None

    # reference: recommender/LightGCN.py:222-228 (xavier_uniform_, user table first, on the CPU generator)
    def _init_model(self):
        init = nn.init.xavier_uniform_
        packed = torch.cat([init(torch.empty(self.data.user_num, self.latent_size)), init(torch.empty(self.data.item_num, self.latent_size))], 0)
        U = self.data.user_num
        return nn.ParameterDict({'user_emb': nn.Parameter(packed[:U]), 'item_emb': nn.Parameter(packed[U:])})

    def _init_uiAdj(self, ui_adj):
        """recommender/LightGCN.py:212-215: D^-1/2 A D^-1/2 of an arbitrary weighted symmetric adjacency (no inf guard in the
        reference; an isolated node simply has an empty row).  Row sums and scaling run on the GPU."""
        m = sp.csr_matrix(ui_adj, dtype=np.float32)
        m.sort_indices()
        adj = SparseNormAdj.__new__(SparseNormAdj)
        adj.shape = m.shape
        adj.indptr, adj.indices = m.indptr.astype(np.int64), m.indices.astype(np.int32)
        adj._graph = None
        if torch.cuda.is_available():
            w = torch.from_numpy(m.data.astype(np.float32)).to(DEVICE)
            val, dinv = ops.norm_adj_values(torch.from_numpy(m.indptr.astype(np.int32)).to(DEVICE), torch.from_numpy(adj.indices).to(DEVICE), w, m.shape[0])
            adj.values = val
            adj.dinv = dinv
        else:
            raise ops._lib.ArlError('_init_uiAdj needs the GPU (no CPU fallback)')
        self.sparse_norm_adj = adj
        self._eng = None

    def _init_uiAdj_from_interactions(self, ui, n_real=None):
        """Same result as `_init_uiAdj(ui_adj + ui_adj.T)` with ui_adj's upper-right block = `ui` (the U' x I, possibly weighted,
        interaction matrix every attack assembles: attack/White/PGA.py:79-83, CLeaR.py:66-71, DLAttack.py:57-61), but the
        (U'+I)^2 adjacency is never built on the host: the symmetric CSR is assembled and normalised on the device.
        `n_real`: rows [0, n_real) are the real users, whose interactions do not change between the calls of one attack -- the device
        image of that block (and its hop plan) is then kept and only the fake users' rows are merged in (ops.IncrementalBipartite);
        the block is recognised by size and checksum, anything else falls back to the full build."""
        m = ui if sp.isspmatrix_csr(ui) and ui.dtype == np.float32 else sp.csr_matrix(ui, dtype=np.float32)
        U, I = m.shape
        g = None
        if n_real is not None and 0 < n_real < U:
            g = self._incremental_graph(m, int(n_real))
        if g is None:
            m = sp.csr_matrix(m, dtype=np.float32, copy=(m is ui)); m.eliminate_zeros(); m.sort_indices()
            u = torch.from_numpy(np.repeat(np.arange(U, dtype=np.int64), np.diff(m.indptr))).to(DEVICE)
            g = ops.bipartite_graph(u, torch.from_numpy(m.indices.astype(np.int64)).to(DEVICE), U, I, weights=torch.from_numpy(m.data).to(DEVICE))
        adj = SparseNormAdj.__new__(SparseNormAdj)
        adj.shape = (U + I, U + I)
        adj._indptr = adj._indices = None                   # host images of the pattern are fetched only if somebody reads them
        adj.values, adj.dinv, adj._graph = g.val, g.dinv, g
        self.sparse_norm_adj = adj
        self._eng = None

    def _incremental_graph(self, m, n_real):
        """ops.IncrementalBipartite keyed by the real block's fingerprint; shared (not copied) by deep copies of this encoder."""
        U, I = m.shape
        F = U - n_real
        e_real = int(m.indptr[n_real])
        # fingerprint of the real block: edge count + strided samples of its three arrays, O(65 K) per call (the attacks never touch real
        # rows; a caller that does changes the count or, with overwhelming probability, a sample).  A block that arrives in non-canonical
        # form (unsorted / explicit zeros) never matches the canonical block's fingerprint and simply takes the full path every time.
        def fingerprint(indptr, indices, data, nnz):
            st = max(1, nnz // 65536)
            return (n_real, F, I, nnz, int(np.add.reduce(indptr[:n_real + 1:max(1, n_real // 4096)], dtype=np.int64)),
                    float(np.add.reduce(data[:nnz:st], dtype=np.float64)), int(np.add.reduce(indices[:nnz:st], dtype=np.int64)))
        inc = getattr(self, '_arl_inc', None)
        if inc is None or inc[0] != fingerprint(m.indptr, m.indices, m.data, e_real):
            R = sp.csr_matrix(m[:n_real], dtype=np.float32); R.eliminate_zeros(); R.sort_indices()
            u = torch.from_numpy(np.repeat(np.arange(n_real, dtype=np.int64), np.diff(R.indptr))).to(DEVICE)
            w = None if bool(np.all(R.data == 1.0)) else torch.from_numpy(R.data).to(DEVICE)
            inc = (fingerprint(R.indptr, R.indices, R.data, R.nnz),
                   ops.IncrementalBipartite(u, torch.from_numpy(R.indices.astype(np.int64)).to(DEVICE), n_real, F, I, DEVICE, weights=w,
                                            emb_size=self.latent_size if self.n_prop_layers else None))
            self._arl_inc = inc
        B = sp.csr_matrix(m[n_real:], dtype=np.float32); B.eliminate_zeros(); B.sort_indices()          # the fake users' rows: small
        fu = torch.from_numpy(np.repeat(np.arange(F, dtype=np.int64), np.diff(B.indptr)))
        return inc[1].update(fu, torch.from_numpy(B.indices.astype(np.int64)), None if bool(np.all(B.data == 1.0)) else torch.from_numpy(B.data))

    def attack_emb(self, users_emb_grad, items_emb_grad):
        with torch.no_grad():
            self.embedding_dict['user_emb'] += users_emb_grad
            self.embedding_dict['item_emb'] += items_emb_grad

    def cuda(self, device=None):
        self._pack()
        return self

    # ---- packed table: both Parameters must be adjacent views of one [U+I,d] device buffer
    def _pack(self):
        u, i = self.embedding_dict['user_emb'], self.embedding_dict['item_emb']
        U, d = u.shape
        es = u.element_size()
        ok = (u.is_cuda and i.is_cuda and u.is_contiguous() and i.is_contiguous() and i.data_ptr() == u.data_ptr() + U * d * es
              and u.untyped_storage().data_ptr() == i.untyped_storage().data_ptr())
        if not ok:
            packed = torch.cat([u.data.to(DEVICE), i.data.to(DEVICE)], 0).contiguous()
            u.data, i.data = packed[:U], packed[U:]
            self._eng = None
        base = torch.as_strided(u.data, (U + i.shape[0], d), (d, 1))
        return base

    def _graph(self):
        if self.n_prop_layers == 0:
            return None
        return ops.auto_blocked(self.sparse_norm_adj.graph(), self.latent_size, split=self.data.user_num)

    def _engine(self, reg=0.0, lr=0.0, optimizer='adam'):
        base = self._pack()
        g = self._graph()
        e = self._eng
        if e is None or e.E0.data_ptr() != base.data_ptr() or e.A is not g or e.L != self.n_prop_layers:
            e = PropagationEngine(g, self.data.user_num, self.data.item_num, self.latent_size, self.n_prop_layers, reg, lr, base.device,
                                  skip_layer0=self.skip_layer0, optimizer=optimizer, table=base)
            self._eng = e
        return e

    def forward(self, perturbed=False, noises=None):
        u, i = self.embedding_dict['user_emb'], self.embedding_dict['item_emb']
        if self.n_prop_layers == 0:
            return u, i
        self._pack()
        if perturbed and noises is None:
            N = u.shape[0] + i.shape[0]
            noises = [torch.rand(N, self.latent_size, device=u.device) for _ in range(self.n_prop_layers)]      # SimGCL.py:204
        return _Propagate.apply(u, i, self, noises)

    def forward_rows(self, rows):
        """Rows `rows` (int32 node ids: users, then U + items) of forward()'s output -- what a training step reads -- on the sparse-batch
        schedule.  Plain LightGCN mean (layers 0..L) only; other encoders do not define this."""
        u, i = self.embedding_dict['user_emb'], self.embedding_dict['item_emb']
        if self.n_prop_layers == 0:
            # index the two Parameters directly (as the reference does): no [U+I, d] copy, no dense gradient through a cat
            r = rows.long()
            U = u.shape[0]
            isu = r < U
            out = torch.empty(r.numel(), u.shape[1], dtype=u.dtype, device=u.device)
            ku, ki = torch.nonzero(isu).flatten(), torch.nonzero(~isu).flatten()
            return out.index_copy(0, ku, u[r[ku]]).index_copy(0, ki, i[r[ki] - U])
        self._pack()
        return _PropagateRows.apply(u, i, self, rows.contiguous())

    def __getstate__(self):
        st = dict(self.__dict__)
        st['_eng'] = None
        st.pop('_arl_inc', None)
        st.pop('_adj_sink', None)
        return st


class Recommender:
    """train()/save()/predict()/evaluate()/test() shared by the three models (reference: LightGCN.py:29-161)."""
    print_every = 1000
    has_extra_loss = False
    fused_extra_loss = False      # the model's extra loss has a fused engine step (SimGCL)
    adjgrad_through_views = False # train(requires_adjgrad=True) is covered although the loss has an extra term: that term's forwards also run through _Propagate (SimGCL)
    train_forward_perturbed = False   # the training forward is model(True) and hands extra outputs to _extra_loss (XSimGCL)

    @staticmethod
    def _rows_capable(model):
        """Encoders whose training forward can be evaluated on the batch rows alone: NGCF (own forward_rows) and the plain LightGCN mean
        (GraphEncoder.forward_rows; not the SimGCL family, whose mean skips layer 0 and whose views are perturbed)."""
        if not hasattr(model, 'forward_rows'):
            return False
        return type(model).forward_rows is not GraphEncoder.forward_rows or (not getattr(model, 'skip_layer0', False) and getattr(model, 'n_prop_layers', 0) <= 8
                                                                             and type(model).forward is GraphEncoder.forward)

    def _fused_step(self, eng, u, p, n):
        return eng.step(u, p, n)

    def _common_init(self, args, data, name):
        print('Recommender: ' + name)
        self.data, self.args = data, args
        self.bestPerformance, self.recOutput = [], []
        self.topN = [int(n) for n in self.args.topK.split(',')]
        self.max_N = max(self.topN)

    # ---- optimizer handling -------------------------------------------------------------------------------------
    def _params(self):
        return [self.model.embedding_dict['user_emb'], self.model.embedding_dict['item_emb']]

    def _fusable(self, optimizer):
        """The fused engine may stand in for `optimizer` iff it is a stock Adam/SGD over exactly this model's two tables."""
        if type(optimizer) not in (torch.optim.Adam, FusedAdam, torch.optim.SGD) or len(optimizer.param_groups) != 1:
            return None
        g = optimizer.param_groups[0]
        ps = g['params']
        mine = self._params()
        # nn.ParameterDict sorts plain-dict keys, so parameters() yields item_emb before user_emb: compare as a set
        if len(ps) != 2 or {id(ps[0]), id(ps[1])} != {id(mine[0]), id(mine[1])} or g.get('weight_decay', 0) != 0 or g.get('maximize', False):
            return None
        if type(optimizer) in (torch.optim.Adam, FusedAdam):
            if g.get('amsgrad', False) or g.get('capturable', False):
                return None
            return 'adam'
        if g.get('momentum', 0) != 0 or g.get('nesterov', False) or g.get('dampening', 0) != 0:
            return None
        return 'sgd'

    def _bind_optimizer_state(self, eng, optimizer, kind):
        """Share Adam moments between the engine and a torch optimizer so either can continue the other's run."""
        g = optimizer.param_groups[0]
        eng.lr = float(g['lr'])
        eng.optimizer = kind
        if kind != 'adam':
            return
        eng.betas, eng.eps = tuple(g['betas']), float(g['eps'])
        U = self.data.user_num
        params = self._params()
        if not any('exp_avg' in optimizer.state[p] for p in params):
            # a fresh optimizer (what the reference builds per train(optimizer=None) call, LightGCN.py:31) starts from zero moments and
            # step 0, whatever an earlier run left in the cached engine
            eng.m.zero_(); eng.v.zero_(); eng.t = 0
        for p, sl in zip(params, (slice(0, U), slice(U, None))):
            st = optimizer.state[p]
            if 'exp_avg' in st and st['exp_avg'].data_ptr() != eng.m[sl].data_ptr():
                eng.m[sl].copy_(st['exp_avg']); eng.v[sl].copy_(st['exp_avg_sq'])
                eng.t = int(st['step'])
            st['exp_avg'], st['exp_avg_sq'] = eng.m[sl], eng.v[sl]
            st.setdefault('step', torch.tensor(float(eng.t)))

    def _sync_optimizer_step(self, eng, optimizer, kind):
        if kind == 'adam':
            for p in self._params():
                optimizer.state[p]['step'] = torch.tensor(float(eng.t))

    # ---- training ------------------------------------------------------------------------------------------------
    l2_scale = 1.0          # NCL divides its L2 term by the batch size (NCL.py:147) ...
    l2_on_negatives = False # ... and also regularises the negative items' rows
    extra_loss_takes_outputs = False   # proxyLG's extra term is computed from the step's own forward outputs
    rows_forward = True                # use model.forward_rows(batch rows) in the training loop when the encoder offers it
    max_steps_per_epoch = None         # measurement knob (bench.py's class-API leg): stop an epoch after this many batches

    def _extra_loss(self, model, user_idx, pos_idx):
        return None

    def _on_epoch_start(self, model):
        pass

    def _after_backward(self, model, epoch, maxEpoch, gradIterationNum):
        pass

    def _on_epoch_end(self, model, epoch, maxEpoch, gradIterationNum):
        pass

    def _train_loop(self, Epoch, optimizer, evalNum, requires_embgrad=False, requires_adjgrad=False, gradIterationNum=10, force_autograd=False):
        self.bestPerformance = []
        model = self.model.cuda()
        fused_kind = None
        if optimizer is None:
            optimizer = torch.optim.Adam(model.parameters(), lr=self.args.lRate)
        if not requires_embgrad and not requires_adjgrad and not force_autograd:      # (force_autograd: a model that captures gradients its own way, SGL)
            fused_kind = self._fusable(optimizer)
            if self.has_extra_loss and not (fused_kind == 'adam' and self.fused_extra_loss):
                fused_kind = None
        self.optimizer = optimizer
        adj = None
        if requires_adjgrad and requires_embgrad and not hasattr(self, 'Matgrad'):
            # recommender/LightGCN.py:36-44: `if requires_embgrad: ... elif requires_adjgrad:` allocates Matgrad only when requires_embgrad is off, and
            # the loop's `self.Matgrad += ...` (:58-59) then fails: same error, raised before any work is spent
            raise AttributeError("'%s' object has no attribute 'Matgrad'" % type(self).__name__)
        if requires_adjgrad:
            # recommender/LightGCN.py:41-43: sparse_norm_adj.requires_grad = True, Matgrad = zeros(N, N).  The gradient of a sparse operand lives on its
            # stored entries, so Matgrad is kept as one value per entry of the pattern (CSR order) instead of N x N.
            if self.has_extra_loss and not self.adjgrad_through_views:
                raise NotImplementedError('requires_adjgrad is implemented for LightGCN, NGCF, SimGCL and XSimGCL')
            self._adjgrad_begin(model)
            adj = model.sparse_norm_adj
            adj.requires_grad = True
            self.Matgrad = torch.zeros(adj.values.numel(), dtype=torch.float32, device=DEVICE)
        if requires_embgrad:
            model.requires_grad = True
            self.usergrad = torch.zeros((self.data.user_num, self.args.emb_size), device=DEVICE)
            self.itemgrad = torch.zeros((self.data.item_num, self.args.emb_size), device=DEVICE)
        maxEpoch = Epoch if Epoch else self.args.maxEpoch
        # reference quirk Q4 (attack/White/PGA.py:59-67, DLAttack.py:53-68): an optimizer built on a model that
        # recommender.__init__ has since replaced owns none of the live parameters, so the reference's loop moves
        # nothing -- it only consumes the sampler's random stream and runs the per-epoch evaluation.  Same here,
        # without spending the forward/backward.
        mine = self._params()
        inert = not requires_embgrad and not requires_adjgrad and not force_autograd and not any(p is q for g in optimizer.param_groups for p in g['params'] for q in mine)
        eng = None
        if fused_kind:
            eng = model._engine(self.args.reg, self.args.lRate, fused_kind)
            eng.reg = float(self.args.reg)
            self._bind_optimizer_state(eng, optimizer, fused_kind)
        U, I = self.data.user_num, self.data.item_num
        for epoch in range(maxEpoch):
            self._on_epoch_start(model)                              # e.g. SGL draws its two graph views here, BEFORE the sampler shuffles (also when inert: same RNG stream)
            # nothing in this loop draws from Python's `random`: the epoch's negatives are sampled in one native call, checked and
            # uploaded once; the batches are views of that device image
            if inert:
                for _ in next_batch_pairwise(self.data, self.args.batch_size, whole_epoch=True):
                    pass
                batches = ()
            else:
                samp = {}
                batches = device_epoch(self.data, self.args.batch_size, DEVICE, U, I, stats=samp)
            it = iter(batches)
            t_first = time.perf_counter()
            first = next(it, None)                                   # the epoch's shuffle + the first chunk of negatives (host); later chunks are drawn behind the GPU steps
            t_first = time.perf_counter() - t_first
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t_loop, n_done = time.perf_counter(), 0
            for n, (u, p, ng) in enumerate(itertools.chain(() if first is None else (first,), it)):
                if self.max_steps_per_epoch is not None and n >= self.max_steps_per_epoch:
                    break
                n_done += 1
                if eng is not None:
                    lo = self._fused_step(eng, u, p, ng)
                    if n % self.print_every == 0:
                        print('training:', epoch + 1, 'batch', n, 'batch_loss:', float(lo[0] + lo[1]))
                    continue
                model.train()
                ul, pl, nl = u.long(), p.long(), ng.long()
                if (self.rows_forward and self._rows_capable(model) and not self.train_forward_perturbed and adj is None
                        and not (self.has_extra_loss and self.extra_loss_takes_outputs)):
                    # the loss reads the output on the batch rows only: encoders that can evaluate just those rows do (NGCF's last layer)
                    B = u.numel()
                    out_r = model.forward_rows(torch.cat([u, p + U, ng + U]).to(torch.int32))
                    user_emb, pos_item_emb, neg_item_emb = out_r[:B], out_r[B:2 * B], out_r[2 * B:]
                else:
                    outs = model(True) if self.train_forward_perturbed else model()
                    rec_user_emb, rec_item_emb = outs[0], outs[1]
                    if adj is not None:
                        rec_user_emb.retain_grad(); rec_item_emb.retain_grad()
                    user_emb, pos_item_emb, neg_item_emb = rec_user_emb[ul], rec_item_emb[pl], rec_item_emb[nl]
                batch_loss = bpr_l2_loss(user_emb, pos_item_emb, neg_item_emb, self.args.reg * self.l2_scale)
                if self.l2_on_negatives:
                    batch_loss = batch_loss + l2_reg_loss(self.args.reg * self.l2_scale, neg_item_emb)
                if self.has_extra_loss:
                    batch_loss = batch_loss + (self._extra_loss(model, ul, pl, *outs) if (self.train_forward_perturbed or self.extra_loss_takes_outputs)
                                               else self._extra_loss(model, ul, pl))
                optimizer.zero_grad()
                batch_loss.backward()
                if adj is not None:
                    # what autograd would add to sparse_norm_adj.grad; the reference never zeroes it (it is no optimizer parameter, so
                    # optimizer.zero_grad() passes it by): .grad is the running sum over the steps so far, and Matgrad adds THAT (LightGCN.py:58-59)
                    g_vals = self._adjgrad_step(model, rec_user_emb.grad, rec_item_emb.grad)
                    adj.values.grad = g_vals if adj.values.grad is None else adj.values.grad + g_vals
                    if maxEpoch - epoch < gradIterationNum:
                        self.Matgrad += adj.values.grad
                if requires_embgrad and maxEpoch - epoch < gradIterationNum:
                    self.usergrad += model.embedding_dict['user_emb'].grad
                    self.itemgrad += model.embedding_dict['item_emb'].grad
                self._after_backward(model, epoch, maxEpoch, gradIterationNum)
                optimizer.step()
                if n % self.print_every == 0:
                    print('training:', epoch + 1, 'batch', n, 'batch_loss:', batch_loss.item())
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            # wall clock of the epoch's batch loop alone (sampler draw, evaluation and the epoch-end forward excluded)
            # wall clock of the epoch's batch loop (evaluation and the epoch-end forward excluded).  The loop INCLUDES whatever time it waited for the
            # sampler's producer thread; the serial prologue (epoch shuffle + first chunk) is reported beside it
            self.last_train_stats = {'steps': n_done, 'loop_seconds': time.perf_counter() - t_loop, 'fused': eng is not None, 'first_batch_seconds': t_first}
            if hasattr(batches, 'close'):
                batches.close()                                      # lets the producer finish the epoch's RNG stream and writes Python's `random` state back
            if not inert:
                self.last_train_stats.update({'sampler_' + k: v for k, v in samp.items()})
            if eng is not None:
                self._sync_optimizer_step(eng, optimizer, fused_kind)
            self._on_epoch_end(model, epoch, maxEpoch, gradIterationNum)
            model.eval()
            with torch.no_grad():
                self.user_emb, self.item_emb = self._detached_forward()
            if epoch % evalNum == 0:
                self.evaluate(epoch)
        self.user_emb, self.item_emb = self.best_user_emb, self.best_item_emb
        if requires_adjgrad:
            self._adjgrad_end(model)
            block = self._adjgrad_block(adj)
            if requires_embgrad:
                return block, self.user_emb, self.item_emb, self.usergrad, self.itemgrad
            return block
        if requires_embgrad:
            return self.user_emb, self.item_emb, self.usergrad, self.itemgrad

    def _adjgrad_begin(self, model):
        """Prepare the encoder for train(requires_adjgrad=True).  The base form covers every encoder whose forwards all run through _Propagate
        (GraphEncoder.forward: LightGCN's mean over layers 0..L, SimGCL's clean pass and perturbed views over layers 1..L): each backward of such a
        forward adds its share of the adjacency's gradient to the encoder's sink (engine.adjacency_gradient)."""
        if not getattr(model, 'adjgrad_sink_capable', type(model).forward is GraphEncoder.forward):
            raise NotImplementedError('requires_adjgrad is implemented for the LightGCN, NGCF, SimGCL and XSimGCL propagations')
        nnz = model._engine().A.col.numel()
        model._adj_sink = torch.zeros(nnz, dtype=torch.float32, device=DEVICE)

    def _adjgrad_end(self, model):
        model._adj_sink = None

    def _adjgrad_step(self, model, g_user, g_item):
        """One step's gradient on the adjacency's stored entries (CSR order): what the step's backward passes left in the sink."""
        out = model._adj_sink.clone()
        model._adj_sink.zero_()
        return out

    def _adjgrad_block(self, adj):
        """(Matgrad + Matgrad.T)[:U, U:] of recommender/LightGCN.py:74-80 from the per-entry Matgrad: the value on (u, U + i) plus the value on its
        mirror entry (U + i, u).  Dense [U, I] like the reference's while that is at most 2^27 elements, a sparse COO tensor beyond."""
        U, I = self.data.user_num, self.data.item_num
        indptr, indices = adj.indptr, adj.indices
        n = len(indices)
        pos = sp.csr_matrix((np.arange(1, n + 1, dtype=np.int64), indices, indptr), shape=adj.shape)
        mirror = pos.T.tocsr()
        mirror.sort_indices()
        if mirror.nnz != n or not np.array_equal(mirror.indices, indices):
            raise ValueError('requires_adjgrad: the adjacency pattern is not symmetric')
        m = self.Matgrad
        nu = int(indptr[U])                                              # the user rows' entries come first in CSR order
        vals = m[:nu] + m[torch.from_numpy(mirror.data[:nu] - 1).to(m.device)]
        rows = torch.from_numpy(np.repeat(np.arange(U, dtype=np.int64), np.diff(indptr[:U + 1]))).to(m.device)
        cols = torch.from_numpy(indices[:nu].astype(np.int64) - U).to(m.device)
        if U * I <= 2 ** 27:
            block = torch.zeros(U, I, dtype=torch.float32, device=m.device)
            block[rows, cols] = vals
            return block
        return torch.sparse_coo_tensor(torch.stack([rows, cols]), vals, (U, I)).coalesce()

    def train_batches(self, batches, optimizer):
        """Run the BPR + L2 training step on an iterable of (u, p, n) batches with `optimizer` and no evaluation -- the
        surrogate fine-tuning loop of attack/White/DLAttack.py:84-106 (fused engine when the optimizer allows it)."""
        model = self.model.cuda()
        kind = None if self.has_extra_loss else self._fusable(optimizer)
        U, I = self.data.user_num, self.data.item_num
        eng = None
        if kind:
            eng = model._engine(self.args.reg, self.args.lRate, kind)
            eng.reg = float(self.args.reg)
            self._bind_optimizer_state(eng, optimizer, kind)
        last = None
        for user_idx, pos_idx, neg_idx in batches:
            if isinstance(user_idx, torch.Tensor):            # device_epoch(): already range-checked and resident
                u, p, ng = user_idx, pos_idx, neg_idx
            else:
                if int(user_idx.max()) >= U or int(max(pos_idx.max(), neg_idx.max())) >= I:
                    raise IndexError('batch index outside the embedding tables')
                u, p, ng = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(DEVICE) for x in (user_idx, pos_idx, neg_idx))
            if eng is not None:
                last = self._fused_step(eng, u, p, ng)
                continue
            outs = model(True) if self.train_forward_perturbed else model()
            rec_user_emb, rec_item_emb = outs[0], outs[1]
            loss = bpr_l2_loss(rec_user_emb[u.long()], rec_item_emb[p.long()], rec_item_emb[ng.long()], self.args.reg * self.l2_scale)
            if self.l2_on_negatives:
                loss = loss + l2_reg_loss(self.args.reg * self.l2_scale, rec_item_emb[ng.long()])
            if self.has_extra_loss:
                loss = loss + (self._extra_loss(model, u.long(), p.long(), *outs) if (self.train_forward_perturbed or self.extra_loss_takes_outputs)
                               else self._extra_loss(model, u.long(), p.long()))
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            last = loss
        if eng is not None:
            self._sync_optimizer_step(eng, optimizer, kind)
        return last

    def _detached_forward(self):
        with torch.no_grad():
            u, i = self.model()
            if self.model.n_prop_layers == 0:
                # reference quirk: Matrix_Factorization.forward returns the Parameters themselves (GMF.py:174-175), so
                # user_emb / best_user_emb ALIAS the live tables and "best epoch" always tracks the latest values
                return u.detach(), i.detach()
            return u.detach().clone(), i.detach().clone()

    def save(self):
        self.best_user_emb, self.best_item_emb = self._detached_forward()

    def predict(self, u):
        with torch.no_grad():
            u = self.data.get_user_id(u)
            score = torch.matmul(self.user_emb[u], self.item_emb.transpose(0, 1))
            return score.cpu().numpy()

    # ---- evaluation (reference: LightGCN.py:92-161).  The per-user python loop + numba heap are replaced by one
    # streaming score+mask+top-k kernel launch over all test users.
    def evaluate(self, epoch):
        print('Evaluating the model...')
        users, idx, _ = self._test_topk()                  # the per-epoch evaluation needs the measures only, not rec_list
        sys.stdout.write('\rProgress: [' + '+' * 50 + ']100%\n')
        measure = ranking_evaluation_topk(self.data, idx, [self.max_N]) if users else ranking_evaluation(self.data.test_set, {}, [self.max_N])
        performance = {}
        for m in measure[1:]:
            k, v = m.strip().split(':')
            performance[k] = float(v)
        if len(self.bestPerformance) > 0:
            count = 0
            for k in self.bestPerformance[1]:
                count += 1 if self.bestPerformance[1][k] > performance[k] else -1
            if count < 0:
                self.bestPerformance[1] = performance
                self.bestPerformance[0] = epoch + 1
                self.save()
        else:
            self.bestPerformance.append(epoch + 1)
            self.bestPerformance.append(performance)
            self.save()
        print('-' * 120)
        print('Real-Time Ranking Performance ' + ' (Top-' + str(self.max_N) + ' Item Recommendation)')
        measure = [m.strip() for m in measure[1:]]
        print('*Current Performance*')
        print('Epoch:', str(epoch + 1) + ',', '  |  '.join(measure))
        bp = '  |  '.join(k + ':' + str(self.bestPerformance[1][k]) for k in ('Hit Ratio', 'Precision', 'Recall', 'NDCG'))
        print('*Best Performance* ')
        print('Epoch:', str(self.bestPerformance[0]) + ',', bp)
        print('-' * 120)
        return measure

    def _test_topk(self):
        """(test users in test_set order, top-max_N item ids [n, k], scores) with interacted items masked (LightGCN.py:137-161)."""
        users = list(self.data.test_set)
        if not users:
            return users, np.zeros((0, 0), np.int64), np.zeros((0, 0), np.float32)
        uid = torch.tensor([self.data.user[u] for u in users], dtype=torch.long, device=self.user_emb.device)
        Pu = self.user_emb[uid].contiguous()
        Pi = self.item_emb.contiguous()
        mask = getattr(self.data, '_arl_test_mask', None)
        if mask is None or mask[0] != len(users):
            # interacted-item mask (candidates[rated] = -10e8, LightGCN.py:153-154) as CSR over the selected users; the training
            # sets of the test users never change (fake users are not test users), so it is built once per data object
            cols = [np.fromiter((self.data.item[it] for it in self.data.training_set_u[u]), dtype=np.int32) for u in users]
            rp = np.zeros(len(users) + 1, np.int32)
            np.cumsum([len(c) for c in cols], out=rp[1:])
            mc = np.concatenate([np.sort(c) for c in cols]) if rp[-1] else np.zeros(1, np.int32)
            mask = (len(users), rp, mc.astype(np.int32))
            self.data._arl_test_mask = mask
        _, rp, mc = mask
        k = min(self.max_N, Pi.shape[0])
        if Pu.shape[1] % 4 == 0 and k <= 128:
            warm = getattr(self, '_arl_eval_warm', None)                 # last evaluation's lists: same users, same mask
            if warm is not None and (tuple(warm.shape) != (len(users), k) or warm.device != Pu.device):
                warm = None
            idx, val = ops.score_mask_topk(Pu, Pi, k, torch.from_numpy(rp).to(Pu.device), torch.from_numpy(mc).to(Pu.device), warm_idx=warm)
            self._arl_eval_warm = idx
        else:       # embedding sizes the kernel does not cover: torch plumbing
            sc = Pu @ Pi.T
            rows = torch.from_numpy(np.repeat(np.arange(len(users)), np.diff(rp))).to(Pu.device)
            sc[rows, torch.from_numpy(mc[:rp[-1]].astype(np.int64)).to(Pu.device)] = -10e8
            val, idx = torch.topk(sc, k)
        return users, idx.cpu().numpy(), val.cpu().numpy()

    def test(self):
        users, idx, val = self._test_topk()
        rec_list = {}
        for r, user in enumerate(users):
            rec_list[user] = [(self.data.id2item[int(i)], float(s)) for i, s in zip(idx[r], val[r])]
        sys.stdout.write('\rProgress: [' + '+' * 50 + ']100%\n')
        return rec_list, ranking_evaluation_topk(self.data, idx, self.topN) if users else ranking_evaluation(self.data.test_set, rec_list, self.topN)
