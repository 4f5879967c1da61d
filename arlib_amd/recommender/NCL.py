"""NCL: LightGCN + structure-contrastive loss (2-hop context of a node against ALL nodes' initial embeddings) + prototype-contrastive
loss (k-means centroids, from the 6th epoch on) -- mirror of the reference's recommender/NCL.py (class NCL :20-166: e_step :52-56,
run_kmeans :58-73, ProtoNCE_loss :75-88, ssl_layer_loss :90-117, train :119-176; LGCN_Encoder :273-310; local InfoNCE :320-334).
Hyper-parameters are hard-coded there: n_layers = 2 for the context pass (the recommendation pass uses args.n_layers),
hyper_layers = 1, ssl_temp = 0.05, ssl_reg = 1e-6, alpha = 1.5, proto_reg = 1e-7, k = 2000.

As executed by the reference:
* the L2 term is divided by the batch size (NCL.py:147,157) -- `l2_scale`;
* the context pass runs on `data.norm_adj` (the graph the DataLoader built), not on the encoder's current `sparse_norm_adj`;
* `e_step` is sklearn's KMeans on the host from the global numpy RNG (same call here: same clusters), on the raw tables;
* both phases step the optimiser (the commented `optimizer.step()` lines sit above the live one).

The structure loss has B x U and B x I logits (2048 x 10^6 at cfg2): `_AllRowsNCE` walks the table in panels, twice (log-sum-exp,
then gradients), so nothing of that size is kept; at d = 64 on the GPU it runs as the fused all-rows kernels (ops.nce_allrows: exact
fp32 MFMA, no logits stored, no library GEMM), other widths keep two library GEMMs per panel and pass.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from ._base import Recommender, SparseNormAdj
from .LightGCN import LGCN_Encoder


class _Hop(torch.autograd.Function):
    """y = A x for a symmetric normalised adjacency (backward: the same kernel on the gradient)."""

    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        return ops.spmm(graph, x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.spmm(ctx.graph, g.contiguous()), None


class _AllRowsNCE(torch.autograd.Function):
    """sum_b -log( exp(<A_b, V_idx_b>/T) / sum_j exp(<A_b, V_j>/T) ) for normalised rows A [B, d] and the normalised table V [N, d]
    (ssl_layer_loss, NCL.py:96-103 / :109-115), with the gradients w.r.t. A and V, panel by panel."""
    PANEL = 65536
    FUSED = True          # False: the panel form (library GEMMs) also where the fused kernels apply -- A/B and tests

    @staticmethod
    def forward(ctx, A, V, idx, T):
        if _AllRowsNCE.FUSED and A.is_cuda and A.shape[1] in ops.NCE_ALLROWS_WIDTHS and A.dtype == torch.float32 and T >= ops.NCE_ALLROWS_MIN_TAU:
            # the fused form (arl_nce_allrows_*): no B x N logits, no library GEMM; exact fp32 products on the matrix cores
            with torch.no_grad():
                A_, V_ = A.contiguous(), V.contiguous()
                lse, dA, dV = ops.nce_allrows(A_, V_, T)
                Vi = V_[idx]
                loss = (lse - (A_ * Vi).sum(1) / T).sum()
                dA -= Vi
                dA /= T; dV /= T
                ops.scatter_add_rows(dV, idx.to(torch.int32).contiguous(), A_, -1.0 / T, check_range=False)
            ctx.save_for_backward(dA, dV)
            return loss
        with torch.no_grad():
            B, N = A.shape[0], V.shape[0]
            m = torch.full((B,), -float('inf'), device=A.device); s = torch.zeros(B, device=A.device)
            for j in range(0, N, _AllRowsNCE.PANEL):
                S = (A @ V[j:j + _AllRowsNCE.PANEL].T) / T
                mj = torch.maximum(m, S.max(1)[0])
                s = s * torch.exp(m - mj) + torch.exp(S - mj[:, None]).sum(1)
                m = mj
            lse = m + torch.log(s)
            pos = (A * V[idx]).sum(1) / T
            loss = (lse - pos).sum()
            dA = torch.zeros_like(A); dV = torch.zeros_like(V)
            for j in range(0, N, _AllRowsNCE.PANEL):
                Vp = V[j:j + _AllRowsNCE.PANEL]
                P = torch.exp((A @ Vp.T) / T - lse[:, None])
                dA += P @ Vp
                dV[j:j + _AllRowsNCE.PANEL] = P.T @ A
            dA -= V[idx]
            dV.index_add_(0, idx, -A)
            dA /= T; dV /= T
        ctx.save_for_backward(dA, dV)
        return loss

    @staticmethod
    def backward(ctx, g):
        dA, dV = ctx.saved_tensors
        return g * dA, g * dV, None, None


class _AllRowsNCEOfRaw(torch.autograd.Function):
    """_AllRowsNCE.apply(F.normalize(Xa), F.normalize(Xv), idx, T) with the four normalisation passes (forward and autograd) inside:
    arl_normalize_rows_f32 / _bwd_f32 around arl_nce_allrows_grad_f32; 1/T and the upstream gradient ride in the backward kernel."""

    @staticmethod
    def forward(ctx, Xa, Xv, idx, T):
        with torch.no_grad():
            Ya, na = ops.normalize_rows(Xa.contiguous())
            Yv, nv = ops.normalize_rows(Xv.contiguous())
            lse, dA, dV = ops.nce_allrows(Ya, Yv, T)
            Vi = Yv[idx]
            loss = (lse - (Ya * Vi).sum(1) / T).sum()
            dA -= Vi
            ops.scatter_add_rows(dV, idx.to(torch.int32).contiguous(), Ya, -1.0, check_range=False)
        ctx.save_for_backward(Ya, na, Yv, nv, dA, dV)
        ctx.T = T
        return loss

    @staticmethod
    def backward(ctx, g):
        Ya, na, Yv, nv, dA, dV = ctx.saved_tensors
        g1 = g.reshape(1).to(torch.float32).contiguous()
        # (not in place: the graph may be walked again, retain_graph=True)
        return ops.normalize_rows_bwd(Ya, na, dA, 1.0 / ctx.T, scale_dev=g1), ops.normalize_rows_bwd(Yv, nv, dV, 1.0 / ctx.T, scale_dev=g1), None, None


def all_rows_nce(Xa, Xv, idx, T):
    """sum_b -log softmax_j(<F.normalize(Xa)_b, F.normalize(Xv)_j>/T)[idx_b]  (ssl_layer_loss, NCL.py:96-103 / :109-115)."""
    if _AllRowsNCE.FUSED and Xa.is_cuda and Xa.shape[1] in ops.NCE_ALLROWS_WIDTHS and Xa.dtype == torch.float32 and T >= ops.NCE_ALLROWS_MIN_TAU:
        return _AllRowsNCEOfRaw.apply(Xa, Xv, idx, T)
    return _AllRowsNCE.apply(F.normalize(Xa), F.normalize(Xv), idx, T)


def InfoNCE(view1, view2, temperature, b_cos=True):
    """The module-local InfoNCE of the reference's NCL.py (:320-334): -mean(diag(log_softmax(v1 v2^T / T)))."""
    if b_cos:
        view1, view2 = F.normalize(view1, dim=1), F.normalize(view2, dim=1)
    return -torch.diag(F.log_softmax((view1 @ view2.T) / temperature, dim=1)).mean()


class NCL(Recommender):
    print_every = 10 ** 9
    has_extra_loss = True
    fused_extra_loss = False
    l2_on_negatives = True
    adjgrad_through_views = True      # train(requires_adjgrad=True): only the main forward runs on sparse_norm_adj (NCL.py:134); the structure term's hops use a fresh tensor (:135)

    def __init__(self, args, data):
        self._common_init(args, data, 'NCL')
        self.n_layers = 2
        self.cl_rate = 0.2
        self.eps = 0.1
        self.ssl_temp = 0.05
        self.ssl_reg = 1e-6
        self.hyper_layers = 1
        self.alpha = 1.5
        self.proto_reg = 1e-7
        self.k = 2000
        self.reg = self.args.reg
        self.batch_size = self.args.batch_size
        self.model = LGCN_Encoder(self.data, self.args.emb_size, self.args.n_layers)
        self.user_centroids = self.user_2cluster = self.item_centroids = self.item_2cluster = None
        self._epoch = -1
        self._clean = None

    @property
    def l2_scale(self):
        return 1.0 / self.batch_size

    # ---- prototypes ------------------------------------------------------------------------------------------------
    def e_step(self):
        self.user_centroids, self.user_2cluster = self.run_kmeans(self.model.embedding_dict['user_emb'].detach().cpu().numpy())
        self.item_centroids, self.item_2cluster = self.run_kmeans(self.model.embedding_dict['item_emb'].detach().cpu().numpy())

    def run_kmeans(self, x):
        from sklearn.cluster import KMeans
        kmeans = KMeans(n_clusters=self.k).fit(x)
        dev = self.model.embedding_dict['user_emb'].device
        return torch.Tensor(kmeans.cluster_centers_).to(dev), torch.LongTensor(kmeans.predict(x)).squeeze().to(dev)

    def ProtoNCE_loss(self, initial_emb, user_idx, item_idx):
        user_emb, item_emb = torch.split(initial_emb, [self.data.user_num, self.data.item_num])
        user2centroids = self.user_centroids[self.user_2cluster[user_idx]]
        loss_user = InfoNCE(user_emb[user_idx], user2centroids, self.ssl_temp) * self.batch_size
        item2centroids = self.item_centroids[self.item_2cluster[item_idx]]
        loss_item = InfoNCE(item_emb[item_idx], item2centroids, self.ssl_temp) * self.batch_size
        return self.proto_reg * (loss_user + loss_item)

    # ---- structure contrast ------------------------------------------------------------------------------------------
    def ssl_layer_loss(self, context_emb, initial_emb, user, item):
        U, I = self.data.user_num, self.data.item_num
        ctx_u, ctx_i = torch.split(context_emb, [U, I])
        ini_u, ini_i = torch.split(initial_emb, [U, I])
        loss_u = all_rows_nce(ctx_u[user], ini_u, user, self.ssl_temp)
        loss_i = all_rows_nce(ctx_i[item], ini_i, item, self.ssl_temp)
        return self.ssl_reg * (loss_u + self.alpha * loss_i)

    def _clean_graph(self):
        if self._clean is None or self._clean[0] is not self.data.norm_adj:
            self._clean = (self.data.norm_adj, SparseNormAdj(self.data.norm_adj).graph())
        return self._clean[1]

    def context_embeddings(self, model):
        """emb_list of NCL.py:137-142: E_0 and its n_layers propagations over data.norm_adj."""
        ego = torch.cat([model.embedding_dict['user_emb'], model.embedding_dict['item_emb']], 0)
        out = [ego]
        g = self._clean_graph()
        for _ in range(self.n_layers):
            ego = _Hop.apply(ego, g)
            out.append(ego)
        return out

    def _on_epoch_start(self, model):
        self._epoch += 1
        if self._epoch >= 5:
            self.e_step()

    def _extra_loss(self, model, user_idx, pos_idx):
        emb_list = self.context_embeddings(model)
        loss = self.ssl_layer_loss(emb_list[self.hyper_layers * 2], emb_list[0], user_idx, pos_idx)
        if self._epoch >= 5:
            loss = loss + self.ProtoNCE_loss(emb_list[0], user_idx, pos_idx)
        return loss

    def train(self, requires_adjgrad=False, requires_embgrad=False, gradIterationNum=10, Epoch=0, optimizer=None, evalNum=5):
        self._epoch = -1
        return self._train_loop(Epoch, optimizer, evalNum, requires_embgrad=requires_embgrad, requires_adjgrad=requires_adjgrad,
                                gradIterationNum=gradIterationNum)
