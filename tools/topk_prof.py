"""Per-wave section timers of score_mask_topk (barrier / MFMA chain / pre-filter+insert+compaction), in core clock ticks.
Needs the instrumented build:  make -C arlib_amd/csrc prof  &&  ARLIB_AMD_LIB=arlib_amd/lib/libarlib_amd_prof.so python3 tools/topk_prof.py
(the timers overwrite part of top_val, so this library is never the product)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
U, I, d, k = int(os.environ.get('U', 204800)), 100000, int(os.environ.get('D', 64)), int(os.environ.get('K', 50))
torch.manual_seed(0)
Pu = torch.randn(U, d, device='cuda') * 0.1
Pi = torch.randn(I, d, device='cuda') * 0.1
ops.score_mask_topk(Pu[:256].contiguous(), Pi, k, exact=os.environ.get('EXACT', '0') == '1')
torch.cuda.synchronize()
t0 = time.perf_counter()
idx, val = ops.score_mask_topk(Pu, Pi, k, exact=os.environ.get('EXACT', '0') == '1')
torch.cuda.synchronize()
dt = time.perf_counter() - t0
NW = int(os.environ.get('NW', 16 if d == 64 and os.environ.get('EXACT', '0') != '1' else 8))
v = val.view(U // (16 * NW), NW, 16 * k)[:, :, :9].double().mean(0).cpu().numpy()      # [wave, section]
print('k=%d d=%d: %.1f ms' % (k, d, dt * 1e3))
for w in range(NW):
    print('  wave %d: wait done %.0f  load wait %.0f  stash+signal %.0f  fetch issue %.0f  wait fill %.0f  mfma %.0f  book %.0f | loop %.0f total %.0f' % (w, v[w, 6], v[w, 7], v[w, 8], v[w, 0], v[w, 3], v[w, 1], v[w, 2], v[w, 4], v[w, 5]))
