import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
U, I, d, k = int(os.environ.get('U', 200000)), 100000, int(os.environ.get('D', 64)), int(os.environ.get('K', 50))
torch.manual_seed(0)
Pu = torch.randn(U, d, device='cuda') * 0.1
Pi = torch.randn(I, d, device='cuda') * 0.1
ops.score_mask_topk(Pu[:256].contiguous(), Pi, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
idx, val = ops.score_mask_topk(Pu, Pi, k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
v = val.view(U // 32, 32 * k)[:, :6].double().mean(0).cpu().numpy()
print('k=%d d=%d: %.1f ms; per-wave ticks: barrier %.0f  mfma %.0f  book %.0f  compact %.0f | loop %.0f total %.0f' % (k, d, dt * 1e3, *v))
