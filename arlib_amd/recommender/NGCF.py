"""NGCF (simplified, as in the reference's recommender/NGCF.py:173-212): per layer
    E' = leaky_relu( A(E W1) + E W1 + ((A E) * E) W2 ),   mean of L+1 layers.
Since A(E W1) = (A E) W1 the layer needs ONE sparse hop, not the reference's two:  P = A E;  E' = leaky_relu((P + E) W1 + (P * E) W2).
The hop (forward and its backward A^T dY = A dY) is the HIP SpMM kernel; the small dense d x d products and the element-wise
glue go through ATen (rocBLAS) this round -- a fused MFMA epilogue is the follow-up.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ._base import GraphEncoder, Recommender, TorchGraphInterface


class _Hop(torch.autograd.Function):
    """Y = A X with the normalised (symmetric) adjacency; backward is the same kernel on the incoming gradient."""

    @staticmethod
    def forward(ctx, X, graph):
        ctx.graph = graph
        return ops.spmm(graph, X.contiguous())

    @staticmethod
    def backward(ctx, gY):
        return ops.spmm(ctx.graph, gY.contiguous()), None


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
            P = _Hop.apply(ego, graph)
            ego = F.leaky_relu(torch.mm(P + ego, self.W['w1_' + str(k)]) + torch.mm(P * ego, self.W['w2_' + str(k)]))
            acc = acc + ego
        out = acc / (self.layers + 1)
        U = self.data.user_num
        return out[:U], out[U:]


class NGCF(Recommender):
    def __init__(self, args, data):
        self._common_init(args, data, 'NGCF')
        self.model = NGCF_Encoder(self.data, self.args.emb_size, self.args.n_layers)

    def _fusable(self, optimizer):
        return None                                       # extra dense weights: always the autograd route

    def train(self, requires_adjgrad=False, requires_embgrad=False, gradIterationNum=10, Epoch=0, optimizer=None, evalNum=5):
        return self._train_loop(Epoch, optimizer, evalNum, requires_embgrad=requires_embgrad, requires_adjgrad=requires_adjgrad,
                                gradIterationNum=gradIterationNum)
