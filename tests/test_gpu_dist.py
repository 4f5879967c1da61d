"""GPU test of the user-sharded engine with the real HIP kernels: two processes share the one GPU of the test box (RCCL
refuses two ranks on one device, so the collective is gloo staged through host memory here; the RCCL path itself is the
driver's multi-GPU bench).  Result must equal the single-GPU engine and the oracle."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from conftest import rel_err, RTOL, ROOT
from test_dist_cpu import small_problem, oracle_run

pytestmark = pytest.mark.gpu


class HostStagedComm:
    def __init__(self):
        import torch.distributed as dist
        self.dist = dist

    class _Done:
        def wait(self):
            return True

    def all_reduce(self, t):
        c = t.detach().cpu()
        self.dist.all_reduce(c)
        t.copy_(c)
        return t

    def all_reduce_async(self, t):
        self.all_reduce(t)
        return self._Done()


def _worker(rank, world, port, ret, sparse):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from arlib_amd.dist_engine import ShardedPropagationEngine
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    U, I, d, L, pairs, E0, batches = small_problem()
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cuda:0', rank, world, torch.from_numpy(E0), comm=HostStagedComm())
    losses = []
    for u, p, n in batches:
        lo = (eng.step_sparse if sparse else eng.step)(torch.from_numpy(u).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda())
        losses.append(float(lo[0] + lo[1]))
    full = eng.gather_full_table().cpu().numpy()
    if rank == 0:
        ret['table'], ret['losses'] = full, losses
    dist.destroy_process_group()


@pytest.mark.parametrize('sparse', [False, True])
def test_sharded_engine_two_ranks_hip_kernels(sparse):
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    U, I, d, L, pairs, E0, batches = small_problem()
    ref_table, ref_losses = oracle_run(U, I, d, L, pairs, E0, batches)
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, ret, sparse), nprocs=2, join=True)
    assert np.allclose(ret['losses'], ref_losses, rtol=RTOL, atol=0)
    assert rel_err(ret['table'], ref_table) < RTOL
