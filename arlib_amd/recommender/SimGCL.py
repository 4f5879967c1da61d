"""SimGCL: LightGCN without layer 0, two noise-perturbed views, InfoNCE (tau 0.2, eps 0.1, lambda 0.2, L hard-coded 2)
-- mirror of the reference's recommender/SimGCL.py (class SimGCL :18-160, SimGCL_Encoder :170-219)."""
import torch

from ._base import GraphEncoder, Recommender, TorchGraphInterface
from ..util.loss import InfoNCE


class SimGCL_Encoder(GraphEncoder):
    skip_layer0 = True

    def __init__(self, data, emb_size, eps, n_layers):
        super().__init__(data, emb_size)
        self.eps = eps
        self.n_layers = self.n_prop_layers = n_layers
        self.norm_adj = data.norm_adj
        self.sparse_norm_adj = TorchGraphInterface.convert_sparse_mat_to_tensor(self.norm_adj)

    def cal_cl_loss(self, idx, noises=None):
        """recommender/SimGCL.py:212-219.  `noises`: optional [view][hop] tensors replacing torch.rand (parity tests)."""
        dev = self.embedding_dict['user_emb'].device
        u_idx = torch.unique(torch.as_tensor(idx[0], device=dev).long())
        i_idx = torch.unique(torch.as_tensor(idx[1], device=dev).long())
        user_view_1, item_view_1 = self.forward(perturbed=True, noises=None if noises is None else noises[0])
        user_view_2, item_view_2 = self.forward(perturbed=True, noises=None if noises is None else noises[1])
        user_cl_loss = InfoNCE(user_view_1[u_idx], user_view_2[u_idx], 0.2)
        item_cl_loss = InfoNCE(item_view_1[i_idx], item_view_2[i_idx], 0.2)
        return user_cl_loss + item_cl_loss


class SimGCL(Recommender):
    print_every = 100
    has_extra_loss = True
    fused_extra_loss = True
    adjgrad_through_views = True

    def __init__(self, args, data):
        self._common_init(args, data, 'SimGCL')
        self.n_layers = 2          # hard-coded in the reference (SimGCL.py:31), args.n_layers is ignored
        self.cl_rate = 0.2
        self.eps = 0.1
        self.model = SimGCL_Encoder(self.data, self.args.emb_size, self.eps, self.n_layers)

    def _fused_step(self, eng, u, p, n):
        lo, self.last_cl_loss = eng.step_simgcl(u, p, n, cl_rate=self.cl_rate, tau=0.2, eps=self.eps)
        return lo

    def _extra_loss(self, model, user_idx, pos_idx):
        return self.cl_rate * model.cal_cl_loss([user_idx, pos_idx])

    def train(self, requires_adjgrad=False, requires_embgrad=False, gradIterationNum=10, Epoch=0, optimizer=None, evalNum=5):
        return self._train_loop(Epoch, optimizer, evalNum, requires_embgrad=requires_embgrad, requires_adjgrad=requires_adjgrad,
                                gradIterationNum=gradIterationNum)
