"""Round-2 hop experiments on the cfg2 graph (1M x 100K SYN-v1, d = 64): the register-blocked hop with the hub rows (a) on the chunked CSR
kernel (round 1), (b) dealt as strided pieces inside the blocked launches (split_hubs), for several hub thresholds / piece sizes.
    python3 tools/hop_experiments.py            env: U, I, CONFIGS="split,hub,piece;..."  """
import os, sys, time
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops, _lib
from arlib_amd.util import synthetic

U, I, d = int(os.environ.get('U', 1_000_000)), int(os.environ.get('I', 100_000)), int(os.environ.get('D', 64))
dev = 'cuda:0'
data = synthetic.syn_v1(U, I)
rowptr, col = data.adjacency_pattern()
val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(dev), torch.from_numpy(col).to(dev), torch.ones(len(col), device=dev), U + I)
N = U + I
X = torch.randn(N, d, device=dev)
Yr, Yb = torch.empty(N, d, device=dev), torch.zeros(N, d, device=dev)


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


A0 = ops.CSRGraph(rowptr, col, val, dev)
print('CSR hop: %.3f ms' % t(lambda: ops.spmm(A0, X, out=Yr)), flush=True)
default = 'auto;0,1024,1024,4,1'
default_old = '1,4096,4096,1,256;1,2048,2048,1,256;1,1024,1024,1,256;1,4096,4096,2,512;1,4096,4096,1,512;1,4096,4096,4,1024;1,4096,4096,1,1024;1,8192,8192,1,256'
for cfg in os.environ.get('CONFIGS', default).split(';'):
    if cfg == 'auto':
        sh, hub, piece, wpg, wm, co, cb = 1, None, None, None, None, 0, 1024
    else:
        vals = [int(x) for x in cfg.split(',')]
        sh, hub, piece, wpg, wm, co, cb = vals + [1, 1024, 1024, 4, 1, 0, 1024][len(vals):]
    A = ops.CSRGraph(rowptr, col, val, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    A.enable_blocked(split=U, hub=hub, split_hubs=bool(sh), piece=piece, wpg=wpg, wave_multiple=wm, col_order='degree' if co else None, col_block=cb)
    torch.cuda.synchronize()
    bp = A.blocked
    desc = ', '.join('%d rows / %d waves / %d edges / %d split rows in %d pieces' % (s['n_rows'], s['n_waves'], s['n_edges'], s['n_split'], s['n_pieces']) for s in bp.sets)
    Yb.zero_(); ops.spmm(A, X, out=Yb); torch.cuda.synchronize()
    err = ((Yb - Yr).abs().max() / Yr.abs().max()).item()
    st = ops._stream()
    parts = []
    for k in range(len(bp.sets)):
        s = bp.struct(k, d)
        parts.append(t(lambda: _lib.check(_lib.lib().arl_spmm_blocked_f32(C.byref(s), X.data_ptr(), d, 1.0, 0.0, None, None, Yb.data_ptr(), st), 'blocked')))
    hubt = t(lambda: _lib.check(_lib.lib().arl_spmm_csr_f32(C.byref(bp.hub._struct(d)), X.data_ptr(), d, 1.0, 0.0, None, Yb.data_ptr(), st), 'csr')) if bp.hub is not None else 0.0
    print('split_hubs=%s hub=%s piece=%s wpg=%s wave_multiple=%s col_order=%s col_block=%s: hop %.3f ms (sets %s, chunked hub rows %.3f; %d hub rows) err %.1e plan %.2fs [%s]'
          % (sh, hub, piece, wpg, wm, co, cb, t(lambda: ops.spmm(A, X, out=Yb)), ' + '.join('%.3f' % p for p in parts), hubt, bp.n_hub, err, time.perf_counter() - t0, desc), flush=True)
    del A, bp
