"""Register-blocked hop on ONE RANK'S SHARE of cfg2 at N = 8 (125 K local users x 100 K items, 4 M interactions): the two rectangular blocks of
dist_engine (A_u: local user rows gather replicated item rows; A_i: item rows gather local user rows) timed per launch over the plan's knobs --
rows per wave, loads in flight, waves per workgroup, hub threshold, column block.  The cfg2 defaults were tuned on 32 M-edge launches.
    python3 tools/shard_hop_sweep.py          env: U (default 125000), FULL=1 (also the whole cfg2 graph's two row sets)"""
import itertools, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from arlib_amd import ops, dist_engine, _lib
from arlib_amd.util import synthetic

U, I, d = int(os.environ.get('U', 125_000)), 100_000, 64
dev = torch.device('cuda', 0)
data = synthetic.syn_v1(U, I, 32.0, 2018)
blk = dist_engine.build_local_blocks(data.pairs0, U, I, 0, 1)
Nl = U + I
X = torch.randn(Nl, d, device=dev)
Y = torch.empty(Nl, d, device=dev)
st = ops._stream()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n          # microseconds


def hop_us(g, out):
    structs = [g.blocked.struct(k, d) for k in range(len(g.blocked.sets))]

    def run():
        for s in structs:
            _lib.check(_lib.lib().arl_spmm_blocked_f32(C.byref(s), X.data_ptr(), d, 1.0, 0.0, None, None, out.data_ptr(), st), 'blocked')
        if g.blocked.hub is not None:
            _lib.check(_lib.lib().arl_spmm_csr_f32(C.byref(g.blocked.hub._struct(d)), X.data_ptr(), d, 1.0, 0.0, None, out.data_ptr(), st), 'csr')
    return timeit(run)


print('share: %d local users x %d items, %d interactions; lib %s' % (U, I, len(data.pairs0), os.environ.get('ARLIB_AMD_LIB', 'default')))
base = {}
for name, (rp, col, val), out in (('A_u', blk['Au'], Y[:U]), ('A_i', blk['Ai'], Y[U:])):
    g0 = ops.CSRGraph(rp, col, val, dev, n_cols=Nl)
    t_csr = timeit(lambda: ops.spmm(g0, X, out=out))
    rows = []
    for rpw, unr, wpg, hub, cb in itertools.product((32, 16), (None, 16, 32), (None, 1, 2, 4), (None, 512, 4096), (1024, 256, 4096)):
        if (hub is not None or cb != 1024) and (rpw, unr, wpg) != (32, None, None):
            continue                                  # hub / column-block variations on the default wave shape only
        g = ops.CSRGraph(rp, col, val, dev, n_cols=Nl, validate=False)
        try:
            g.enable_blocked(rows_per_wave=rpw, unroll=unr, wpg=wpg, hub=hub, col_block=cb)
        except Exception as e:
            print('  %s rpw=%s unr=%s wpg=%s hub=%s cb=%s: %s' % (name, rpw, unr, wpg, hub, cb, e)); continue
        us = hop_us(g, out)
        s0 = g.blocked.sets[0]
        rows.append((us, rpw, unr, wpg, hub, cb, s0['n_waves'], s0['unroll'], s0['wpg'], s0['hub'], s0['n_split']))
        del g
    rows.sort()
    print('%s: CSR kernel %.1f us; %d edges = %.2f GB of 256-B gathers' % (name, t_csr, g0.nnz, g0.nnz * 256e-9))
    for r in rows[:6] + [x for x in rows if x[1:6] == (32, None, None, None, 1024)]:
        print('   %7.1f us  (%.1f TB/s gathered)  rpw=%s unroll=%s wpg=%s hub=%s col_block=%s -> waves %d, unroll %d, wpg %d, hub %d, split rows %d'
              % (r[0], g0.nnz * 256e-6 / r[0], *r[1:]))
    base[name] = rows
