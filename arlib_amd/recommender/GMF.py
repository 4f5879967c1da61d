"""GMF: plain matrix factorisation with BPR + L2 and Adam -- mirror of the reference's recommender/GMF.py
(class GMF :16-147, Matrix_Factorization :149-175) on the MI355X kernels."""
from ._base import GraphEncoder, Recommender, TorchGraphInterface  # noqa: F401


class Matrix_Factorization(GraphEncoder):
    n_prop_layers = 0

    def __init__(self, data, emb_size):
        super().__init__(data, emb_size)


class GMF(Recommender):
    def __init__(self, args, data):
        self._common_init(args, data, 'GMF')
        self.model = Matrix_Factorization(self.data, args.emb_size)

    def train(self, requires_embgrad=False, gradIterationNum=10, Epoch=0, optimizer=None, evalNum=5):
        return self._train_loop(Epoch, optimizer, evalNum, requires_embgrad=requires_embgrad, gradIterationNum=gradIterationNum)
