"""SGL: LightGCN + InfoNCE between two structurally perturbed views of the graph, re-drawn every epoch -- mirror of the
reference's recommender/SGL.py (class SGL :17-150, SGL_Encoder :180-260, GraphAugmentor :263-302; hyper-parameters hard-coded
there: L=2, lambda=0.2, drop rate 0.1, temperature 0.2, aug_type 2 -- which, through `if self.aug_type == 0 or 1`, means ONE
edge-dropped graph per view shared by all layers).

The augmentation draws `random.sample(range(E), int(0.9 E))` from Python's global RNG; the same draw (values and RNG
consumption) is made natively (`util.sampler.sample_range`), and the view's normalised adjacency is built on the device
(`ops.bipartite_graph`) instead of scipy COO -> torch COO."""
import numpy as np
import scipy.sparse as sp
import torch

from .. import ops
from ._base import GraphEncoder, Recommender, TorchGraphInterface
from ..util.loss import InfoNCE
from ..util.sampler import sample_range

DEVICE = 'cuda'


class GraphAugmentor(object):
    """(kept user ids, kept item ids) of the perturbed interaction matrix, in the order the reference's arrays have."""

    @staticmethod
    def _edges(sp_adj):
        m = sp.csr_matrix(sp_adj)
        m.eliminate_zeros(); m.sort_indices()
        return np.repeat(np.arange(m.shape[0], dtype=np.int64), np.diff(m.indptr)), m.indices.astype(np.int64)      # == sp_adj.nonzero()

    @staticmethod
    def edge_dropout(sp_adj, drop_rate):
        row_idx, col_idx = GraphAugmentor._edges(sp_adj)
        keep_idx = sample_range(len(row_idx), int(len(row_idx) * (1 - drop_rate)))                                 # SGL.py:293
        return row_idx[keep_idx], col_idx[keep_idx]

    @staticmethod
    def node_dropout(sp_adj, drop_rate):
        row_idx, col_idx = GraphAugmentor._edges(sp_adj)
        n_u, n_i = sp_adj.shape
        drop_user = sample_range(n_u, int(n_u * drop_rate))                                                        # SGL.py:270-271
        drop_item = sample_range(n_i, int(n_i * drop_rate))
        ku = np.ones(n_u, bool); ku[drop_user] = False
        ki = np.ones(n_i, bool); ki[drop_item] = False
        keep = ku[row_idx] & ki[col_idx]
        return row_idx[keep], col_idx[keep]


class _PropagateOn(torch.autograd.Function):
    """LightGCN pass (mean of layers 0..L) over an ARBITRARY normalised graph: the SpMM kernel forward, and -- the graph
    being symmetric -- the same kernel on the incoming gradient backward (Horner form)."""

    @staticmethod
    def forward(ctx, user_emb, item_emb, graph, L):
        E = torch.cat([user_emb, item_emb], 0).contiguous()
        acc, cur = E.clone(), E
        for _ in range(L):
            cur = ops.spmm(graph, cur)
            acc.add_(cur)
        acc.mul_(1.0 / (L + 1))
        ctx.graph, ctx.L, ctx.E0 = graph, L, E
        U = user_emb.shape[0]
        return acc[:U], acc[U:]

    @staticmethod
    def backward(ctx, g_u, g_i):
        L = ctx.L
        G = torch.cat([g_u, g_i], 0).contiguous()
        s = 1.0 / (L + 1)
        sink = getattr(ctx.graph, 'grad_sink', None)
        if sink is not None:            # train(requires_adjgrad=True): this VIEW's stored entries receive their gradient (the reference sets dropped_adj*.requires_grad, SGL.py:56-63)
            from ..engine import mean_adjacency_gradient
            mean_adjacency_gradient(ctx.graph, ctx.E0, G, L, s, sink)
        acc = G
        for k in range(L):
            a = s if k == L - 1 else 1.0
            acc = ops.spmm(ctx.graph, acc, a, a, G)
        if L == 0:
            acc = G
        U = g_u.shape[0]
        return acc[:U], acc[U:], None, None


class SGL_Encoder(GraphEncoder):
    def __init__(self, data, emb_size, drop_rate, n_layers, temp, aug_type):
        super().__init__(data, emb_size)
        self.drop_rate, self.temp, self.aug_type = drop_rate, temp, aug_type
        self.n_layers = self.n_prop_layers = n_layers
        self.norm_adj = data.norm_adj
        self.sparse_norm_adj = TorchGraphInterface.convert_sparse_mat_to_tensor(self.norm_adj)

    def graph_reconstruction(self):
        # `if self.aug_type == 0 or 1` in the reference (SGL.py:212) is always true: one perturbed graph, whatever aug_type
        return self.random_graph_augment()

    def random_graph_augment(self):
        """One perturbed view as a device graph (SGL.py:220-229): node dropout for aug_type 0, edge dropout for 1 and 2."""
        mat = self.data.interaction_mat
        if self.aug_type == 0:
            u, i = GraphAugmentor.node_dropout(mat, self.drop_rate)
        else:
            u, i = GraphAugmentor.edge_dropout(mat, self.drop_rate)
        key = np.sort(u * mat.shape[1] + i)                                   # csr_matrix((1, (u, i))) order
        ut = torch.from_numpy(key // mat.shape[1]).to(DEVICE)
        it = torch.from_numpy(key % mat.shape[1]).to(DEVICE)
        return ops.auto_blocked(ops.bipartite_graph(ut, it, self.data.user_num, self.data.item_num), self.latent_size, split=self.data.user_num)

    def forward(self, perturbed_adj=None):
        if perturbed_adj is None:
            return super().forward()
        u, i = self.embedding_dict['user_emb'], self.embedding_dict['item_emb']
        self._pack()
        if isinstance(perturbed_adj, list):
            raise NotImplementedError('per-layer views are unreachable in the reference (graph_reconstruction never returns a list)')
        return _PropagateOn.apply(u, i, perturbed_adj, self.n_prop_layers)

    def cal_cl_loss(self, idx, perturbed_mat1, perturbed_mat2):
        """recommender/SGL.py:248-256: ONE InfoNCE over the concatenated user and item rows of the two views."""
        dev = self.embedding_dict['user_emb'].device
        u_idx = torch.unique(torch.as_tensor(idx[0], device=dev).long())
        i_idx = torch.unique(torch.as_tensor(idx[1], device=dev).long())
        user_view_1, item_view_1 = self.forward(perturbed_mat1)
        user_view_2, item_view_2 = self.forward(perturbed_mat2)
        view1 = torch.cat((user_view_1[u_idx], item_view_1[i_idx]), 0)
        view2 = torch.cat((user_view_2[u_idx], item_view_2[i_idx]), 0)
        return InfoNCE(view1, view2, self.temp)


class SGL(Recommender):
    print_every = 100
    has_extra_loss = True
    fused_extra_loss = True

    def _fused_step(self, eng, u, p, n):
        lo, self.last_cl_loss = eng.step_sgl(u, p, n, self.dropped_adj1, self.dropped_adj2, cl_rate=self.cl_rate, tau=self.temp)
        return lo

    def __init__(self, args, data):
        self._common_init(args, data, 'SGL')
        self.n_layers = 2                   # hard-coded in the reference (SGL.py:30-35), args.n_layers is ignored
        self.cl_rate = 0.2
        self.aug_type = 2
        self.drop_rate = 0.1
        self.temp = 0.2
        self.model = SGL_Encoder(self.data, self.args.emb_size, self.drop_rate, self.n_layers, self.temp, self.aug_type)

    def _extra_loss(self, model, user_idx, pos_idx):
        return self.cl_rate * model.cal_cl_loss([user_idx, pos_idx], self.dropped_adj1, self.dropped_adj2)

    # ---- train(requires_adjgrad / requires_embgrad) the way the reference's SGL does it (recommender/SGL.py:39-95), which is NOT the other models' way:
    #   * the adjacency gradient is taken w.r.t. the two DROPPED graphs of the epoch; per step grad_mat* += dropped_adj*.grad, and `.grad` is never zeroed
    #     inside the epoch, so grad_mat adds the running sum; at the END of the epoch gradAll[U, I] += upper-right block of (grad_mat1 + grad_mat2)
    #     (no transpose added), while maxEpoch - epoch < gradIterationNum;
    #   * `elif requires_embgrad`: the tables' `.grad` of the epoch's LAST step is added once per epoch (both flags: the adjacency branch wins);
    #   * the two flags are independent `if`s at set-up, so both together work (the other models fail there).
    _grad_req = None

    def _on_epoch_start(self, model):
        self.dropped_adj1 = model.graph_reconstruction()      # two fresh views per epoch (SGL.py:50-51)
        self.dropped_adj2 = model.graph_reconstruction()
        if self._grad_req is not None and self._grad_req[0]:
            for g in (self.dropped_adj1, self.dropped_adj2):
                z = lambda: torch.zeros(g.col.numel(), dtype=torch.float32, device=g.col.device)
                g.grad_sink, g.grad_run, g.grad_mat = z(), z(), z()

    def _after_backward(self, model, epoch, maxEpoch, gradIterationNum):
        if self._grad_req is not None and self._grad_req[0]:
            for g in (self.dropped_adj1, self.dropped_adj2):
                g.grad_run += g.grad_sink                      # dropped_adj.grad: the running sum of the epoch's steps
                g.grad_sink.zero_()
                g.grad_mat += g.grad_run                       # grad_mat += dropped_adj.grad (SGL.py:75-77)

    def _on_epoch_end(self, model, epoch, maxEpoch, gradIterationNum):
        if self._grad_req is None or not maxEpoch - epoch < gradIterationNum:
            return
        U, I = self.data.user_num, self.data.item_num
        if self._grad_req[0]:
            for g in (self.dropped_adj1, self.dropped_adj2):
                nu = int(g.rowptr[U])                          # the user rows' entries come first: (u, U + i)
                rows = torch.repeat_interleave(torch.arange(U, device=g.col.device), (g.rowptr[1:U + 1] - g.rowptr[:U]).long(), output_size=nu)
                self.gradAll.index_put_((rows, g.col[:nu].long() - U), g.grad_mat[:nu], accumulate=True)
                g.grad_sink = None
        elif self._grad_req[1]:
            self.usergrad += model.embedding_dict['user_emb'].grad
            self.itemgrad += model.embedding_dict['item_emb'].grad

    def train(self, requires_adjgrad=False, requires_embgrad=False, gradIterationNum=10, Epoch=0, optimizer=None, evalNum=5):
        if not requires_adjgrad and not requires_embgrad:
            return self._train_loop(Epoch, optimizer, evalNum)
        U, I = self.data.user_num, self.data.item_num
        if requires_adjgrad:
            self.gradAll = torch.zeros(U, I, dtype=torch.float32, device=DEVICE)
        if requires_embgrad:
            self.usergrad = torch.zeros((U, self.args.emb_size), device=DEVICE)
            self.itemgrad = torch.zeros((self.data.item_num, self.args.emb_size), device=DEVICE)
        self._grad_req = (bool(requires_adjgrad), bool(requires_embgrad))
        try:
            self._train_loop(Epoch, optimizer, evalNum, gradIterationNum=gradIterationNum, force_autograd=True)
        finally:
            self._grad_req = None
        if requires_adjgrad and requires_embgrad:
            return self.gradAll, self.user_emb, self.item_emb, self.usergrad, self.itemgrad
        if requires_adjgrad:
            return self.gradAll
        return self.user_emb, self.item_emb, self.usergrad, self.itemgrad
