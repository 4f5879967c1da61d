"""NGCF (simplified, as in the reference's recommender/NGCF.py:173-212): per layer
    E' = leaky_relu( A(E W1) + E W1 + ((A E) * E) W2 ),   mean of L+1 layers.
Since A(E W1) = (A E) W1 the layer needs ONE sparse hop, not the reference's two:  P = A E;  E' = leaky_relu((P + E) W1 + (P * E) W2).
The hop (forward and its backward A^T dY = A dY) is the HIP SpMM kernel; the layer is one autograd node (`_Layer`) whose dense part
runs on the matrix cores in hand-written kernels (arl_ngcf_dense_{fwd,dgrad,wgrad}_f32: exact-fp32 MFMA, the [N, 2d] operand [P + E | P * E]
formed in registers) for d in {16, 32, 64, 128}; other widths keep the element-wise kernels + library GEMM form.
"""
import torch
import torch.nn as nn

from .. import ops
from ._base import GraphEncoder, Recommender, TorchGraphInterface


class _Layer(torch.autograd.Function):
    """One NGCF layer  E' = leaky_relu((P + E) W1 + (P * E) W2),  P = A E,  forward and backward by hand:
    forward  : SpMM kernel, one element-wise pass building [S | T], ONE GEMM with [W1; W2], one activation pass (in place);
    backward : activation pass, two GEMMs (gST = gZ [W1; W2]^T and [gW1; gW2] = [S | T]^T gZ), one element-wise pass for
               (gP, gE), and the SpMM kernel with its AXPBY epilogue: gE_total = A gP + gE  (A is symmetric).
    Saved per layer: E, P, [S | T], E'  (the ATen expression graph keeps about twice that)."""

    @staticmethod
    def forward(ctx, ego, W1, W2, graph, slope):
        ego = ego.contiguous()
        P = ops.spmm(graph, ego)
        Wcat = torch.cat([W1, W2], 0)
        ctx.fused = ego.shape[1] in ops.NGCF_DENSE_WIDTHS
        if ctx.fused:
            out = ops.ngcf_dense_fwd(P, ego, Wcat, slope)
            ctx.save_for_backward(ego, P, out, Wcat)
        else:
            ST = ops.ngcf_combine(P, ego)
            out = ops.ngcf_act_(torch.mm(ST, Wcat), None, slope)
            ctx.save_for_backward(ego, P, ST, out, Wcat)
        ctx.graph, ctx.slope = graph, slope
        return out

    @staticmethod
    def backward(ctx, g_out):
        d = g_out.shape[1]
        if ctx.fused:
            ego, P, out, Wcat = ctx.saved_tensors
            gP, gE, gW = ops.ngcf_dense_bwd(g_out.contiguous(), out, P, ego, Wcat, ctx.slope)
        else:
            ego, P, ST, out, Wcat = ctx.saved_tensors
            gZ = ops.ngcf_act_bwd(g_out.contiguous(), out, ctx.slope)
            gW = torch.mm(ST.t(), gZ)
            gP, gE = ops.ngcf_combine_bwd(torch.mm(gZ, Wcat.t()), P, ego)
        sink = getattr(ctx.graph, 'grad_sink', None)
        if sink is not None:                                       # P = A ego: the adjacency's stored entries receive <gP[row], ego[col]> (train(requires_adjgrad=True))
            ops.sddmm_csr(ctx.graph, gP.contiguous(), ego, 1.0, out=sink)
        g_ego = ops.spmm(ctx.graph, gP, 1.0, 1.0, gE)
        return g_ego, gW[:d], gW[d:], None, None


class _LastLayerRows(torch.autograd.Function):
    """The LAST NGCF layer evaluated on the listed rows only (the training loss reads the mean over layers on <= 3B batch rows): the hop is
    the row-subset SpMM, the dense part runs on [3B, 2d]; backward scatters the rows' gP into an otherwise zero table and propagates it with
    the flag-masked hop (A symmetric), then adds the rows' gE.  Duplicate row ids are fine: their gradients accumulate."""

    @staticmethod
    def forward(ctx, ego, W1, W2, graph, slope, rows):
        ego = ego.contiguous()
        P = ops.spmm_rows(graph, ego, rows, (), 1.0, check_range=False)
        E = ops.gather_rows(ego, rows, check_range=False)
        Wcat = torch.cat([W1, W2], 0)
        ctx.fused = ego.shape[1] in ops.NGCF_DENSE_WIDTHS
        if ctx.fused:
            out = ops.ngcf_dense_fwd(P, E, Wcat, slope)
            ST = P                                             # placeholder (not read)
        else:
            ST = ops.ngcf_combine(P, E)
            out = ops.ngcf_act_(torch.mm(ST, Wcat), None, slope)
        ctx.save_for_backward(P, E, ST, out, Wcat, rows)
        ctx.graph, ctx.slope, ctx.shape = graph, slope, tuple(ego.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        P, E, ST, out, Wcat, rows = ctx.saved_tensors
        N, d = ctx.shape
        if ctx.fused:
            gP, gE, gW = ops.ngcf_dense_bwd(g_out.contiguous(), out, P, E, Wcat, ctx.slope)
        else:
            gZ = ops.ngcf_act_bwd(g_out.contiguous(), out, ctx.slope)
            gW = torch.mm(ST.t(), gZ)
            gP, gE = ops.ngcf_combine_bwd(torch.mm(gZ, Wcat.t()), P, E)
        Gp = torch.zeros(N, d, dtype=torch.float32, device=gP.device)
        ops.scatter_add_rows(Gp, rows, gP, 1.0, check_range=False)
        bits = torch.zeros((N + 31) // 32, dtype=torch.int32, device=gP.device)
        ops.mark_bits_(bits, rows, True, N, check_range=False)
        g_ego = ops.spmm_flagged(ctx.graph, Gp, bits)
        ops.scatter_add_rows(g_ego, rows, gE, 1.0, check_range=False)
        return g_ego, gW[:d], gW[d:], None, None, None


class NGCF_Encoder(GraphEncoder):
    def __init__(self, data, emb_size, n_layers):
        super().__init__(data, emb_size)
        self.layers = n_layers
        self.n_prop_layers = n_layers
        self.norm_adj = data.norm_adj
        init = nn.init.xavier_uniform_
        w = {}
        for i in range(self.layers):                     # same creation order as the reference (NGCF.py:180-182)
            w['w1_' + str(i)] = nn.Parameter(init(torch.empty(self.latent_size, self.latent_size)))
            w['w2_' + str(i)] = nn.Parameter(init(torch.empty(self.latent_size, self.latent_size)))
        self.W = nn.ParameterDict(w)
        self.sparse_norm_adj = TorchGraphInterface.convert_sparse_mat_to_tensor(self.norm_adj)

    def cuda(self, device=None):
        self._pack()
        for p in self.W.values():
            if not p.is_cuda:
                p.data = p.data.to('cuda')
        return self

    def forward(self):
        self.cuda()
        graph = self._graph()
        u, i = self.embedding_dict['user_emb'], self.embedding_dict['item_emb']
        ego = torch.cat([u, i], 0)
        acc = ego
        for k in range(self.layers):
            ego = _Layer.apply(ego, self.W['w1_' + str(k)], self.W['w2_' + str(k)], graph, 0.01)      # F.leaky_relu default slope
            acc = acc + ego
        out = acc / (self.layers + 1)
        U = self.data.user_num
        return out[:U], out[U:]

    def forward_rows(self, rows):
        """Rows `rows` (int32 node ids, users then U + items) of the mean over layers -- what a training step reads.  The first L-1 layers
        are full-table passes; the last one is evaluated on those rows alone (_LastLayerRows): one full hop, one dense [N, 2d] product and
        their backward counterparts less per step."""
        self.cuda()
        graph = self._graph()
        ego = torch.cat([self.embedding_dict['user_emb'], self.embedding_dict['item_emb']], 0)
        idx = rows.long()
        acc = ego[idx]
        if self.layers == 0:                              # no propagation layer: the plain rows of the ego table (as forward() and the reference)
            return acc
        for k in range(self.layers - 1):
            ego = _Layer.apply(ego, self.W['w1_' + str(k)], self.W['w2_' + str(k)], graph, 0.01)
            acc = acc + ego[idx]
        k = self.layers - 1
        acc = acc + _LastLayerRows.apply(ego, self.W['w1_' + str(k)], self.W['w2_' + str(k)], graph, 0.01, rows.contiguous())
        return acc / (self.layers + 1)


class NGCF(Recommender):
    def __init__(self, args, data):
        self._common_init(args, data, 'NGCF')
        self.model = NGCF_Encoder(self.data, self.args.emb_size, self.args.n_layers)

    # ---- fused route: a stock Adam over exactly this model's parameters (tables + the 2L weights) is replaced by engine.step_ngcf
    def _weight_params(self):
        return [self.model.W[n + str(k)] for k in range(self.model.layers) for n in ('w1_', 'w2_')]

    def _fusable(self, optimizer):
        g = optimizer.param_groups[0] if len(optimizer.param_groups) == 1 else None
        if (not isinstance(optimizer, torch.optim.Adam) or g is None or self.model.layers < 1 or self.model.latent_size not in ops.NGCF_DENSE_WIDTHS
                or g.get('weight_decay', 0) != 0 or g.get('amsgrad', False) or g.get('maximize', False) or g.get('capturable', False)):
            return None
        mine = self._params() + self._weight_params()
        if len(g['params']) != len(mine) or {id(q) for q in g['params']} != {id(q) for q in mine}:
            return None
        return 'adam'

    def _fused_step(self, eng, u, p, n):
        return eng.step_ngcf(u, p, n)

    def _bind_optimizer_state(self, eng, optimizer, kind):
        fresh = not any('exp_avg' in optimizer.state[q] for q in self._params() + self._weight_params())
        super()._bind_optimizer_state(eng, optimizer, kind)
        L = self.model.layers
        if not hasattr(eng, 'ngcf_W') or any(a.data_ptr() != self.model.W['w1_%d' % k].data_ptr() for k, (a, b) in enumerate(eng.ngcf_W)):
            eng.init_ngcf([(self.model.W['w1_%d' % k], self.model.W['w2_%d' % k]) for k in range(L)])
        elif fresh:
            for pair in eng.ngcf_m + eng.ngcf_v:
                for t in pair:
                    t.zero_()
        for k in range(L):
            for j, name in enumerate(('w1_', 'w2_')):
                st = optimizer.state[self.model.W[name + str(k)]]
                if 'exp_avg' in st and st['exp_avg'].data_ptr() != eng.ngcf_m[k][j].data_ptr():
                    eng.ngcf_m[k][j].copy_(st['exp_avg']); eng.ngcf_v[k][j].copy_(st['exp_avg_sq'])
                    eng.t = int(st['step'])
                st['exp_avg'], st['exp_avg_sq'] = eng.ngcf_m[k][j], eng.ngcf_v[k][j]
                st.setdefault('step', torch.tensor(float(eng.t)))

    def _sync_optimizer_step(self, eng, optimizer, kind):
        super()._sync_optimizer_step(eng, optimizer, kind)
        for q in self._weight_params():
            optimizer.state[q]['step'] = torch.tensor(float(eng.t))

    def _adjgrad_begin(self, model):
        # recommender/NGCF.py:39-42: every layer's P = A E is a product with the adjacency; their backward passes add into this buffer (_Layer.backward)
        g = model._graph()
        g.grad_sink = torch.zeros(g.col.numel(), dtype=torch.float32, device=g.col.device)

    def _adjgrad_end(self, model):
        model._graph().grad_sink = None

    def _adjgrad_step(self, model, g_user, g_item):
        sink = model._graph().grad_sink
        out = sink.clone()
        sink.zero_()
        return out

    def train(self, requires_adjgrad=False, requires_embgrad=False, gradIterationNum=10, Epoch=0, optimizer=None, evalNum=5):
        return self._train_loop(Epoch, optimizer, evalNum, requires_embgrad=requires_embgrad, requires_adjgrad=requires_adjgrad,
                                gradIterationNum=gradIterationNum)
