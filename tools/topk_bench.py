"""Experiment harness: time score_mask_topk at a given U x I (random tables), for rocprofv3 counter passes."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
U, I, d, k = int(os.environ.get('U', 200000)), int(os.environ.get('I', 100000)), int(os.environ.get('D', 64)), int(os.environ.get('K', 50))
torch.manual_seed(0)
Pu = torch.randn(U, d, device='cuda') * 0.1
Pi = torch.randn(I, d, device='cuda') * 0.1
exact = os.environ.get('EXACT', '0') == '1'
ops.score_mask_topk(Pu[:256].contiguous(), Pi, k, exact=exact)
torch.cuda.synchronize()
t0 = time.perf_counter()
idx, val = ops.score_mask_topk(Pu, Pi, k, exact=exact)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('U=%d I=%d d=%d k=%d %s: %.1f ms, %.1f TFLOP/s (fp32-equivalent)' % (U, I, d, k, 'exact-f32' if exact else 'split-fp16', dt * 1e3, 2.0 * U * I * d / dt / 1e12))
if not exact and os.environ.get('CMP', '1') == '1':
    n = min(U, 20000)
    i2, v2 = ops.score_mask_topk(Pu[:n].contiguous(), Pi, k, exact=True)
    print('  vs exact on %d users: index agreement %.6f, max rel score diff %.2e' % (n, (i2 == idx[:n]).float().mean().item(), ((v2 - val[:n]).abs() / v2.abs().clamp_min(1e-12)).max().item()))
