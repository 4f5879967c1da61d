"""Per-wave section timers of score_mask_topk's second form (topk2_main_kernel), in clock ticks of s_memtime, averaged over workgroups.
Needs the instrumented build:  make -C arlib_amd/csrc variant NAME=t2prof DEFS=-DARL_TOPK2_PROF
    ARLIB_AMD_LIB=arlib_amd/lib/libarlib_amd_t2prof.so python3 tools/topk2_prof.py      (the timers overwrite top_val: never the product)"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
U, I, d, k = int(os.environ.get('U', 1 << 20)), 100000, 64, 50
torch.manual_seed(0)
Pu = torch.randn(U, d, device='cuda') * 0.1
Pi = torch.randn(I, d, device='cuda') * 0.1
ops.score_mask_topk(Pu[:512].contiguous(), Pi, k); torch.cuda.synchronize()
names = ['tiles(+mid slot)', 'appends', '-', 'end slot', 'barrier', 'sync merges', '#sync', '#pipelined', '#candidates', 'prologue+boot']
warm_idx = None
for what in ('cold', 'warm (same tables)', 'warm (tables moved by 5e-3 sigma)'):
    if what.endswith('sigma)'):
        Pu = Pu + 5e-4 * torch.randn_like(Pu); Pi = Pi + 5e-4 * torch.randn_like(Pi)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, val = ops.score_mask_topk(Pu, Pi, k, warm_idx=warm_idx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    warm_idx = idx
    v = val.view(U // 512, 16, 32 * k)[:, :, :10].double()
    m = v.mean(0).cpu().numpy()
    print('%s pass %.1f ms; per wave, mean over %d workgroups:' % (what, dt * 1e3, U // 512))
    for wv in (0, 15):
        print('  wave %2d: ' % wv + '  '.join('%s %.0f' % (names[i], m[wv, i]) for i in range(10)))
    print('  all waves: ' + '  '.join('%s %.0f' % (names[i], m[:, i].mean()) for i in range(10)))
