"""torch.optim.Adam with the update of dense fp32 GPU parameters done by arl_adam_dense_f32 (one pass over p, g, m, v instead of the ~8
element-wise launches of the stock multi-tensor step: 1.4 -> 0.45 ms per step on the 1.1 M x 64 tables of a cfg2 surrogate).

Same constructor, same state layout (`step`, `exp_avg`, `exp_avg_sq`) and the same arithmetic as torch.optim.Adam with its default flags
(the reference's only use: `torch.optim.Adam(model.parameters(), lr=...)`, e.g. attack/White/CLeaR.py:72, recommender/LightGCN.py:33), so the
training loop's engine can share or take over its state exactly as it does for the stock class.  Anything else (amsgrad, weight decay,
maximize, capturable, sparse or CPU or non-fp32 parameters) falls through to the stock step.
"""
import torch

from .. import ops


class Adam(torch.optim.Adam):
    def _plain(self):
        for g in self.param_groups:
            if g.get('amsgrad') or g.get('weight_decay', 0) != 0 or g.get('maximize') or g.get('capturable') or g.get('differentiable'):
                return False
            if isinstance(g['lr'], torch.Tensor):
                return False
            for p in g['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse and p.grad.is_contiguous()
                        and p.grad.dtype == torch.float32):
                    return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if not self._plain():
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                ops.adam_dense(p, p.grad, st['exp_avg'], st['exp_avg_sq'], float(g['lr']), int(st['step']), tuple(g['betas']), float(g['eps']))
        return loss
