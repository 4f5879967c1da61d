"""LightGCN: E_{k+1} = A_hat E_k, mean of L+1 layers, BPR + L2, Adam -- mirror of the reference's
recommender/LightGCN.py (class LightGCN :17-161, LGCN_Encoder :202-240) on the MI355X kernels."""
from ._base import GraphEncoder, Recommender, TorchGraphInterface


class LGCN_Encoder(GraphEncoder):
    def __init__(self, data, emb_size, n_layers):
        super().__init__(data, emb_size)
        self.layers = self.n_prop_layers = n_layers
        self.norm_adj = data.norm_adj
        self.sparse_norm_adj = TorchGraphInterface.convert_sparse_mat_to_tensor(self.norm_adj)


class LightGCN(Recommender):
    def __init__(self, args, data):
        self._common_init(args, data, 'LightGCN')
        self.model = LGCN_Encoder(self.data, self.args.emb_size, self.args.n_layers)

    def train(self, requires_adjgrad=False, requires_embgrad=False, gradIterationNum=10, Epoch=0, optimizer=None, evalNum=5):
        return self._train_loop(Epoch, optimizer, evalNum, requires_embgrad=requires_embgrad, requires_adjgrad=requires_adjgrad,
                                gradIterationNum=gradIterationNum)
