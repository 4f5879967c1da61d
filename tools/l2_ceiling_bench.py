"""What would perfect L2 reuse buy the register-blocked hop?  The cfg2 plan with every column folded into a 2 MB window of the operand
(col & 8191): same records, same waves, same instruction stream, but every gather is served by the XCD's L2.
    python3 tools/l2_ceiling_bench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from arlib_amd import ops, _lib
from arlib_amd.util import synthetic
U, I, d = 1_000_000, 100_000, 64
dev = 'cuda:0'
data = synthetic.syn_v1(U, I)
rowptr, col = data.adjacency_pattern()
N = U + I
X = torch.randn(N, d, device=dev); Y = torch.empty(N, d, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, fold in (('real columns', None), ('columns folded into 8192 rows (2 MB)', 8191), ('folded into 65536 rows (16 MB: L2 misses, Infinity Cache hits)', 65535)):
    c = col.copy()
    if fold is not None:
        eu = int(rowptr[U])
        c[:eu] = U + ((c[:eu] - U) & fold)
        c[eu:] = c[eu:] & fold
        # keep rows sorted and duplicate-free is not required by the kernels; the plan sorts by column block itself
    A = ops.CSRGraph(rowptr, c, np.ones(len(c), np.float32), dev, validate=False).enable_blocked(split=U)
    st = ops._stream()
    parts = []
    for k in range(len(A.blocked.sets)):
        s = A.blocked.struct(k, d)
        parts.append(t(lambda: _lib.check(_lib.lib().arl_spmm_blocked_f32(C.byref(s), X.data_ptr(), d, 1.0, 0.0, None, None, Y.data_ptr(), st), 'blocked')))
    print('%s: user rows %.3f ms, item rows %.3f ms' % (name, parts[0], parts[1]), flush=True)
