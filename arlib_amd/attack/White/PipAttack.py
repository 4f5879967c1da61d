"""PipAttack -- mirror of the reference's attack/White/PipAttack.py (MLP :18-30, init_popularity_mlp :75-106,
posionDataAttack :108-205, project :207-221, fakeUserInject :223-255 = CLeaR's) on the MI355X kernels.

Surrogate loss (:140-158): lossall = ExplicitPromotionLoss + 0.1 * PopularityPromotionLoss.
* ExplicitPromotionLoss = -mean over (real user, target) of <Pu[u], Pi[t]> (BiLevelAttackBatch's loss: the masked top-k and the
  negatives the reference still builds feed nothing).
* PopularityPromotionLoss = cross-entropy of a small MLP "popularity classifier" on the targets' interaction columns.  The MLP is
  trained once in __init__ and is not among the optimised parameters, and its input is the clean interaction matrix: the term is a
  CONSTANT of the surrogate step (no gradient).  It is evaluated once and added to the reported loss; what matters for a drop-in is
  that constructing the attack consumes torch's global RNG exactly as the reference does (three nn.Linear initialisations, ten
  shuffled DataLoader passes), because the fake users' embedding rows are drawn from that stream afterwards -- so the classifier is
  built and trained with the same torch calls on the CPU (it is a 942-input, 1412-sample problem at ml-100k; the dense I x U input the
  reference materialises bounds the attack to such sizes).
* Quirk reproduced: `np.argsort(self.interact.sum(0))` is a 1 x I np.matrix, so `sorteditem[-n:]` selects its only ROW: every item
  is labelled "popular" (:77-83).
"""
import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import TensorDataset, DataLoader

from ... import ops
from .CLeaR import CLeaR


class MLP(nn.Module):
    """Popularity classifier: input_size -> 128 -> 64 -> 2 with ReLU (PipAttack.py:18-30)."""

    def __init__(self, input_size):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(input_size, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 2))

    def forward(self, x):
        return self.layers(x)


class PipAttack(CLeaR):
    MAX_DENSE = 1 << 28          # elements of the dense I x U classifier input

    def __init__(self, arg, data):
        super().__init__(arg, data)
        self.realuserNum = self.userNum
        self.batchSize = 2048
        if self.userNum * self.itemNum > self.MAX_DENSE:
            raise MemoryError('PipAttack: the popularity classifier needs the dense %d x %d interaction matrix (as in the reference)' % (self.itemNum, self.userNum))
        self.popularity_model = MLP(self.realuserNum)
        self.init_popularity_mlp()
        self.alpha = 0.1

    def _item_columns(self):
        return torch.tensor(np.asarray(self.interact.T.todense()), dtype=torch.float32)

    def init_popularity_mlp(self):
        labels = torch.zeros(self.itemNum, 2, dtype=torch.float32)
        labels[:, 1] = 1.0                                              # every item ends up "popular" (module docstring)
        loader = DataLoader(TensorDataset(self._item_columns(), labels), batch_size=64, shuffle=True)
        criterion = nn.CrossEntropyLoss()
        opt = torch.optim.Adam(self.popularity_model.parameters(), lr=0.001)
        for epoch in range(10):
            for inputs, lab in loader:
                opt.zero_grad()
                loss = criterion(self.popularity_model(inputs), lab)
                loss.backward()
                opt.step()
            print(f'Epoch {epoch+1} popularityloss: {loss.item():.3f}')

    def popularity_promotion_loss(self):
        with torch.no_grad():
            out = self.popularity_model(self._item_columns()[self.targetItem])
            lab = torch.zeros(len(self.targetItem), 2); lab[:, 1] = 1.0
            return nn.CrossEntropyLoss()(out, lab)

    def surrogate_loss(self, model, uiAdj2, topk, r0=None, warm=None):
        """lossall of PipAttack.py:140-158; the CLeaR scaffold (posionDataAttack) calls this once per outer step."""
        Pu, Pi = model()
        t = torch.as_tensor(self.targetItem, device=Pi.device, dtype=torch.long)
        explicit = -(Pu[:self.userNum].sum(0) * Pi[t].sum(0)).sum() / float(self.userNum * len(self.targetItem))
        if not hasattr(self, '_pop_const'):
            self._pop_const = float(self.popularity_promotion_loss())
        self.last_top_idx = None
        return explicit + self.alpha * self._pop_const, Pu, Pi, explicit, self._pop_const
