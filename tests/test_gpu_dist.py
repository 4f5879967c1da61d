"""GPU test of the user-sharded engine with the real HIP kernels: two processes share the one GPU of the test box (RCCL
refuses two ranks on one device, so the collective is gloo staged through host memory here; the RCCL path itself is the
driver's multi-GPU bench).  Result must equal the single-GPU engine and the oracle."""
import os
import socket
import sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from conftest import rel_err, RTOL, ROOT
from test_dist_cpu import small_problem, oracle_run

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _spawn(worker, args_after_port, nprocs=2):
    """mp.spawn with a fresh rendezvous port; one retry if the processes could not be brought up (port race, slow first CUDA init)."""
    ctx = mp.get_context('spawn')
    for attempt in range(2):
        ret = ctx.Manager().dict()
        try:
            mp.spawn(worker, args=(nprocs, _free_port(), ret) + tuple(args_after_port), nprocs=nprocs, join=True)
            return ret
        except Exception:
            if attempt == 1:
                raise


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


def _worker(rank, world, port, ret, sparse, d=16, schedule='auto'):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from arlib_amd.dist_engine import ShardedPropagationEngine
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    U, I, d, L, pairs, E0, batches = small_problem(d)
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cuda:0', rank, world, torch.from_numpy(E0), comm=HostStagedComm(), schedule=schedule)
    assert (eng.Au.blocked is not None) == (schedule == 'blocked')
    losses = []
    for u, p, n in batches:
        lo = (eng.step_sparse if sparse else eng.step)(torch.from_numpy(u).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda())
        losses.append(float(lo[0] + lo[1]))
    full = eng.gather_full_table().cpu().numpy()
    if rank == 0:
        ret['table'], ret['losses'] = full, losses
    dist.destroy_process_group()


@pytest.mark.parametrize('sparse,d,schedule', [(False, 16, 'auto'), (True, 16, 'auto'), (True, 64, 'blocked'), (False, 64, 'blocked')])
def test_sharded_engine_two_ranks_hip_kernels(sparse, d, schedule):
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    U, I, d, L, pairs, E0, batches = small_problem(d)
    ref_table, ref_losses = oracle_run(U, I, d, L, pairs, E0, batches)
    ret = _spawn(_worker, (sparse, d, schedule))
    assert np.allclose(ret['losses'], ref_losses, rtol=RTOL, atol=0)
    assert rel_err(ret['table'], ref_table) < RTOL


def _simgcl_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from arlib_amd.dist_engine import ShardedPropagationEngine
    from test_dist_cpu import simgcl_problem
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    U, I, d, L, pairs, E0, batch, noise = simgcl_problem()
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cuda:0', rank, world, torch.from_numpy(E0), comm=HostStagedComm(),
                                              skip_layer0=True)
    local = lambda a: torch.from_numpy(np.concatenate([a[eng.u0:eng.u1], a[U:]])).cuda()
    lo, cl = eng.step_simgcl(*(torch.from_numpy(x).cuda() for x in batch), noises=[[local(noise[v][h]) for h in range(L)] for v in range(2)])
    full = eng.gather_full_table().cpu().numpy()
    # the default noise path: replicated item rows must receive identical noise on every rank (tables stay replicas)
    eng.step_simgcl(*(torch.from_numpy(x).cuda() for x in batch))
    items = eng.E0[eng.Ul:].detach().cpu()
    gathered = [torch.zeros_like(items) for _ in range(world)]
    dist.all_gather(gathered, items)
    if rank == 0:
        ret['table'], ret['rec'], ret['cl'] = full, float(lo[0] + lo[1]), float(cl)
        ret['replica_diff'] = float((gathered[0] - gathered[1]).abs().max())
    dist.destroy_process_group()


def test_sharded_simgcl_two_ranks_hip_kernels():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    from test_dist_cpu import simgcl_problem, oracle_simgcl_step
    ref_table, ref_rec, ref_cl = oracle_simgcl_step(*simgcl_problem())
    ret = _spawn(_simgcl_worker, ())
    assert abs(ret['rec'] - ref_rec) <= RTOL * abs(ref_rec)
    assert abs(ret['cl'] - ref_cl) <= RTOL * abs(ref_cl)
    assert rel_err(ret['table'], ref_table) < RTOL
    assert ret['replica_diff'] == 0.0
