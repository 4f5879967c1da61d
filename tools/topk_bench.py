"""Experiment harness: time score_mask_topk at a given U x I (random tables), for rocprofv3 counter passes."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
U, I, d, k = int(os.environ.get('U', 200000)), int(os.environ.get('I', 100000)), int(os.environ.get('D', 64)), int(os.environ.get('K', 50))
torch.manual_seed(0)
Pu = torch.randn(U, d, device='cuda') * 0.1
Pi = torch.randn(I, d, device='cuda') * 0.1
ops.score_mask_topk(Pu[:256].contiguous(), Pi, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
idx, val = ops.score_mask_topk(Pu, Pi, k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('U=%d I=%d d=%d k=%d: %.1f ms, %.1f TFLOP/s' % (U, I, d, k, dt * 1e3, 2.0 * U * I * d / dt / 1e12))
