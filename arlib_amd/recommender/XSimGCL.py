"""XSimGCL: one noise-perturbed LightGCN pass (layers 1..L averaged) whose mean is contrasted with its own layer-`layer_cl`
output (InfoNCE, temperature 0.1) -- mirror of the reference's recommender/XSimGCL.py (class XSimGCL :18-160,
XSimGCL_Encoder :180-223; hyper-parameters hard-coded there: L=2, lambda=0.2, eps=0.1, layer_cl=1)."""
import torch

from .. import ops
from ._base import GraphEncoder, Recommender, TorchGraphInterface
from ..util.loss import InfoNCE


class _PropagateX(torch.autograd.Function):
    """(user_emb, item_emb) -> (mean_u, mean_i, cl_u, cl_i) of the perturbed pass (XSimGCL.py:205-223).  The perturbation carries
    no gradient, so with G the gradient of the mean and G_cl that of the layer-`layer_cl` output:
        acc_L = c_L,  acc_k = c_k + A acc_{k+1},  dE0 = A acc_1,   c_k = G / L + [k == layer_cl] G_cl   (A symmetric)."""

    @staticmethod
    def forward(ctx, user_emb, item_emb, enc, noises):
        eng = enc._engine()
        L, U = enc.n_prop_layers, user_emb.shape[0]
        cur, acc, cl = eng.E0, None, None
        for k in range(L):
            cur = ops.spmm(eng.A, cur)
            ops.simgcl_perturb_(cur, noises[k], enc.eps)
            acc = cur.clone() if acc is None else acc.add_(cur)
            if k == enc.layer_cl - 1:
                cl = cur if k < L - 1 else cur.clone()
        acc.mul_(1.0 / L)
        ctx.enc, ctx.noises = enc, noises
        return acc[:U], acc[U:], cl[:U], cl[U:]

    @staticmethod
    def backward(ctx, g_mu, g_mi, g_cu, g_ci):
        enc = ctx.enc
        eng = enc._engine()
        L = enc.n_prop_layers
        G = (torch.cat([g_mu, g_mi], 0) * (1.0 / L)).contiguous()
        Gcl = torch.cat([g_cu, g_ci], 0).contiguous()
        sink = getattr(enc, '_adj_sink', None)
        E = None
        if sink is not None:
            # train(requires_adjgrad=True): the stored entries of A receive sum_k <dE_k[row], E_{k-1}[col]> with dE_k = acc_k below and E_k the
            # perturbed layer tables of THIS forward (recomputed from its noise; XSimGCL.py:205-216)
            E, cur = [eng.E0], eng.E0
            for k in range(L - 1):
                cur = ops.spmm(eng.A, cur)
                ops.simgcl_perturb_(cur, ctx.noises[k], enc.eps)
                E.append(cur)
        acc = None
        for k in range(L, 0, -1):
            c = G + Gcl if k == enc.layer_cl else G
            acc = c if acc is None else ops.spmm(eng.A, acc, 1.0, 1.0, c)
            if sink is not None:
                ops.sddmm_csr(eng.A, acc.contiguous(), E[k - 1], 1.0, out=sink)
        dE0 = ops.spmm(eng.A, acc.contiguous())
        U = g_mu.shape[0]
        return dE0[:U], dE0[U:], None, None


class XSimGCL_Encoder(GraphEncoder):
    skip_layer0 = True
    adjgrad_sink_capable = True         # every training forward runs through _PropagateX, whose backward feeds the encoder's adjacency-gradient sink

    def __init__(self, data, emb_size, eps, n_layers, layer_cl):
        super().__init__(data, emb_size)
        self.eps = eps
        self.n_layers = self.n_prop_layers = n_layers
        self.layer_cl = layer_cl
        self.norm_adj = data.norm_adj
        self.sparse_norm_adj = TorchGraphInterface.convert_sparse_mat_to_tensor(self.norm_adj)

    def forward(self, perturbed=False, noises=None):
        """model() -> (user, item); model(True) -> (user, item, user_cl, item_cl) as in the reference.
        `noises`: optional [hop] tensors replacing torch.rand_like (parity tests)."""
        if not perturbed:
            return super().forward()
        u, i = self.embedding_dict['user_emb'], self.embedding_dict['item_emb']
        self._pack()
        if noises is None:
            N = u.shape[0] + i.shape[0]
            noises = [torch.rand(N, self.latent_size, device=u.device) for _ in range(self.n_prop_layers)]
        return _PropagateX.apply(u, i, self, noises)


class XSimGCL(Recommender):
    print_every = 100
    has_extra_loss = True
    fused_extra_loss = True
    adjgrad_through_views = True
    train_forward_perturbed = True          # the BPR term reads the PERTURBED pass (XSimGCL.py:66-68)

    def __init__(self, args, data):
        self._common_init(args, data, 'XSimGCL')
        self.n_layers = 2                   # hard-coded in the reference (XSimGCL.py:32-36), args.n_layers is ignored
        self.cl_rate = 0.2
        self.eps = 0.1
        self.layer_cl = 1
        self.temp = 0.1
        self.model = XSimGCL_Encoder(self.data, self.args.emb_size, self.eps, self.n_layers, self.layer_cl)

    def cal_cl_loss(self, idx, user_view1, user_view2, item_view1, item_view2):
        """recommender/XSimGCL.py:39-44."""
        dev = user_view1.device
        u_idx = torch.unique(torch.as_tensor(idx[0], device=dev).long())
        i_idx = torch.unique(torch.as_tensor(idx[1], device=dev).long())
        return InfoNCE(user_view1[u_idx], user_view2[u_idx], self.temp) + InfoNCE(item_view1[i_idx], item_view2[i_idx], self.temp)

    def _fused_step(self, eng, u, p, n):
        lo, self.last_cl_loss = eng.step_xsimgcl(u, p, n, cl_rate=self.cl_rate, tau=self.temp, eps=self.eps, layer_cl=self.layer_cl)
        return lo

    def _extra_loss(self, model, user_idx, pos_idx, rec_user, rec_item, cl_user, cl_item):
        return self.cl_rate * self.cal_cl_loss([user_idx, pos_idx], rec_user, cl_user, rec_item, cl_item)

    def train(self, requires_adjgrad=False, requires_embgrad=False, gradIterationNum=10, Epoch=0, optimizer=None, evalNum=5):
        return self._train_loop(Epoch, optimizer, evalNum, requires_embgrad=requires_embgrad, requires_adjgrad=requires_adjgrad,
                                gradIterationNum=gradIterationNum)
