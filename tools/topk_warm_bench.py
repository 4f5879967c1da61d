"""Cold vs warm-started score_mask_topk (candidates = the lists of a slightly different table, as between two surrogate steps)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
U, I, d, k = int(os.environ.get('U', 400000)), 100000, int(os.environ.get('D', 64)), int(os.environ.get('K', 50))
torch.manual_seed(0)
Pu = torch.randn(U, d, device='cuda') * 0.1
Pi = torch.randn(I, d, device='cuda') * 0.1
prev, _ = ops.score_mask_topk(Pu, Pi, k)
Pu2 = Pu + torch.randn_like(Pu) * float(os.environ.get('MOVE', 5e-4))
Pi2 = Pi + torch.randn_like(Pi) * float(os.environ.get('MOVE', 5e-4))
for name, w in (('cold', None), ('warm', prev)):
    ops.score_mask_topk(Pu2[:256].contiguous(), Pi2, k, warm_idx=None if w is None else w[:256].contiguous())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, val = ops.score_mask_topk(Pu2, Pi2, k, warm_idx=w)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%s: %.1f ms (%.1f TFLOP/s fp32-equivalent)' % (name, dt * 1e3, 2.0 * U * I * d / dt / 1e12))
    if w is None:
        ref = idx
    else:
        print('  identical to cold:', bool(torch.equal(idx, ref)), ' overlap with previous lists: %.3f' % (idx == prev).float().mean().item())
