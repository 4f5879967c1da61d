"""CPU-side enqueue cost of one sharded step (1-rank RCCL group): how long the host needs to issue a step, i.e. the floor
of the step time once the per-rank GPU work shrinks with N.   python3 tools/enqueue_cost.py"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29519')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
from arlib_amd import ops, dist_engine
from arlib_amd.util import synthetic
from arlib_amd.util.sampler import MTState

U, I, d, L, B = int(os.environ.get('U', 125_000)), 100_000, 64, 3, 2048       # default: one rank's share at N=8
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=dev)
data = synthetic.syn_v1(U, I, 32.0, 2018)
torch.manual_seed(2018)
E0 = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, d)), torch.nn.init.xavier_uniform_(torch.empty(I, d))], 0)
eng = dist_engine.ShardedPropagationEngine.from_pairs(data.pairs0, U, I, d, L, 1e-4, 0.005, dev, 0, 1, table=E0)
mt = MTState.from_seed(2018); s = data.pair_sampler; s.shuffle(mt)
hb = np.empty((40, 3, B), np.int32)
for k in range(40):
    s.batch(mt, k * B, B, out=hb[k])
b = torch.from_numpy(hb).to(dev)
n_launch = [0]
class Hook:
    on = True
    def begin(self, tag): n_launch[0] += 1; return None
    def end(self, tok): pass
for k in range(5):
    eng.step_sparse(b[k, 0], b[k, 1], b[k, 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(5, 35):
    eng.step_sparse(b[k, 0], b[k, 1], b[k, 2])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('U=%d: host enqueue %.3f ms/step, wall %.3f ms/step (30 steps)' % (U, 1e3 * (t1 - t0) / 30, 1e3 * (t2 - t0) / 30))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for k in range(35, 38):
        eng.step_sparse(b[k, 0], b[k, 1], b[k, 2])
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print('GPU kernels+memops per step: %.0f, GPU busy %.3f ms/step' % (len(ev) / 3, sum(e.device_time for e in ev) / 3e3))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    agg[e.name[:90]][0] += 1; agg[e.name[:90]][1] += e.device_time
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print('%6.1f us/step  x%-4.1f %s' % (t / 3, n / 3, name))
dist.destroy_process_group()
