"""A_ra -- mirror of the reference's attack/Gray/A_ra.py (posionDataAttack :57-120) on the MI355X kernels: every outer step re-learns the
user table (5 epochs, Adam over `user_emb` only), then takes one step of
    loss = sum over targets t and n = 100 random user vectors a_j ~ N(0, sigma^2 I) of -log(sigmoid(<a_j, Pi[t]>) + 10e-8)
(:78-83; the vectors are drawn with torch.randn on the host from the global generator, then moved -- same here).  The fake rows are the
top-n of a fresh forward after the loop (:88-90)."""
import torch

from ._userlearn import UserLearningBiLevel


class A_ra(UserLearningBiLevel):
    fresh_forward_for_rows = True

    def __init__(self, arg, data):
        super().__init__(arg, data)
        self.sigma = 1
        self.n = 100

    def outer_loss(self, model, mask, topk):
        Pu, Pi = model()
        a = (torch.randn((self.n, Pi.shape[1])) * self.sigma).to(Pi.device)
        t = torch.as_tensor(self.targetItem, device=Pi.device, dtype=torch.long)
        loss = (-torch.log(torch.sigmoid(a @ Pi[t].T) + 10e-8)).sum()
        return loss, Pu, Pi
