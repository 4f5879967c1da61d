"""Loss functions with the reference's names and semantics (util/loss.py:5-9, 25-29, 42-49), computed by the HIP
kernels and differentiable through torch.autograd (custom Functions; backward is another kernel, not autograd of
ATen ops).  Inputs are GPU fp32 tensors; there is no CPU fallback.
"""
import torch

from .. import ops


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


class _BprL2Rows(torch.autograd.Function):
    """bpr_loss(u,p,n) and/or reg*(||u||_F+||p||_F) on already-gathered [B,d] rows (fused kernel over a packed copy)."""

    @staticmethod
    def forward(ctx, user_emb, pos_emb, neg_emb, reg, which):
        B, d = user_emb.shape
        packed = torch.cat([user_emb, pos_emb, neg_emb], 0).contiguous()
        ar = torch.arange(B, dtype=torch.int32, device=packed.device)
        out = ops.bpr_l2_fwd_bwd(packed, B, ar, ar, ar + B, reg, None, check_range=False)
        ctx.save_for_backward(packed)
        ctx.reg, ctx.which, ctx.B = reg, which, B
        return out[0].clone() if which == 'bpr' else (out[1].clone() if which == 'reg' else out[0] + out[1])

    @staticmethod
    def backward(ctx, gout):
        (packed,) = ctx.saved_tensors
        B = ctx.B
        ar = torch.arange(B, dtype=torch.int32, device=packed.device)
        G = torch.zeros_like(packed)
        # the fused kernel produces d(bpr + reg-term); isolate one of them by a second call with reg = 0 when needed
        if ctx.which == 'both':
            ops.bpr_l2_fwd_bwd(packed, B, ar, ar, ar + B, ctx.reg, G, upstream=1.0, check_range=False, distinct_rows=True)
        elif ctx.which == 'bpr':
            ops.bpr_l2_fwd_bwd(packed, B, ar, ar, ar + B, 0.0, G, upstream=1.0, check_range=False, distinct_rows=True)
        else:
            ops.bpr_l2_fwd_bwd(packed, B, ar, ar, ar + B, ctx.reg, G, upstream=1.0, check_range=False, distinct_rows=True)
            G2 = torch.zeros_like(packed)
            ops.bpr_l2_fwd_bwd(packed, B, ar, ar, ar + B, 0.0, G2, upstream=1.0, check_range=False, distinct_rows=True)
            G -= G2
        G *= gout
        return G[:B], G[B:2 * B], G[2 * B:], None, None


def bpr_loss(user_emb, pos_item_emb, neg_item_emb):
    """util/loss.py:5-9: mean(-log(10e-8 + sigmoid(<u,p> - <u,n>)))"""
    return _BprL2Rows.apply(user_emb, pos_item_emb, neg_item_emb, 0.0, 'bpr')


class _FrobNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb):
        x = emb.contiguous()
        B = x.shape[0]
        ar = torch.arange(B, dtype=torch.int32, device=x.device)
        out = ops.bpr_l2_fwd_bwd(x, 0, ar, ar, ar, 1.0, None, check_range=False)     # out[2] = ||x||_F
        ctx.save_for_backward(x, out)
        return out[2].clone()

    @staticmethod
    def backward(ctx, gout):
        x, out = ctx.saved_tensors
        # torch.norm's backward gives 0 (not NaN) at the origin; bpr_bwd_kernel guards the same way
        return x * torch.where(out[2] > 0, gout / out[2], torch.zeros_like(gout))


def l2_reg_loss(reg, *args):
    """util/loss.py:25-29: reg * sum_k ||emb_k||_F  (un-squared Frobenius norms)."""
    total = 0
    for emb in args:
        total = total + (_FrobNorm.apply(emb) if emb.dim() == 2 and emb.is_cuda and emb.dtype == torch.float32 and emb.shape[0] > 0 else torch.norm(emb, p=2))
    return total * reg


def bpr_l2_loss(user_emb, pos_item_emb, neg_item_emb, reg):
    """bpr_loss(u,p,n) + l2_reg_loss(reg,u,p) in one fused forward and one fused backward kernel."""
    return _BprL2Rows.apply(user_emb, pos_item_emb, neg_item_emb, float(reg), 'both')


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, view1, view2, temperature):
        v1, v2 = view1.contiguous(), view2.contiguous()
        loss, d1, d2 = ops.infonce_fwd_bwd(v1, v2, float(temperature), want_grad=True)
        ctx.save_for_backward(d1, d2)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, gout):
        d1, d2 = ctx.saved_tensors
        return d1 * gout, d2 * gout, None


def InfoNCE(view1, view2, temperature):
    """util/loss.py:42-49."""
    return _InfoNCE.apply(view1, view2, temperature)
