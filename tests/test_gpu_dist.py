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


class DeferredPoisonComm(HostStagedComm):
    """Overlap-hazard double: all_reduce_async() sets the partial aside, fills the buffer with NaN and returns; the reduction happens INSIDE
    wait() and only then lands in the buffer.  A kernel that reads a buffer whose all-reduce is still in flight reads NaN, and one that writes
    it loses its data to wait()'s copy -- either way the final tables differ from the oracle.  The pipelined steps (all-reduce of hop h in
    flight behind A_u(h) and A_i(h+1)) must come out unchanged."""

    class _Work:
        def __init__(self, comm, t, saved):
            self.comm, self.t, self.saved = comm, t, saved

        def wait(self):
            c = self.saved.cpu()
            self.comm.dist.all_reduce(c)
            self.t.copy_(c)
            return True

    def all_reduce_async(self, t):
        saved = t.detach().clone()
        t.fill_(float('nan'))
        return self._Work(self, t, saved)


def _comm(kind):
    return DeferredPoisonComm() if kind == 'deferred' else HostStagedComm()


def _worker(rank, world, port, ret, sparse, d=16, schedule='auto', comm='staged'):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from arlib_amd.dist_engine import ShardedPropagationEngine
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    U, I, d, L, pairs, E0, batches = small_problem(d)
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cuda:0', rank, world, torch.from_numpy(E0), comm=_comm(comm), schedule=schedule)
    assert (eng.Au.blocked is not None) == (schedule == 'blocked')
    losses = []
    for u, p, n in batches:
        lo = (eng.step_sparse if sparse else eng.step)(torch.from_numpy(u).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda())
        losses.append(float(lo[0] + lo[1]))
    full = eng.gather_full_table().cpu().numpy()
    if rank == 0:
        ret['table'], ret['losses'] = full, losses
    dist.destroy_process_group()


@pytest.mark.parametrize('sparse,d,schedule,comm', [(False, 16, 'auto', 'staged'), (True, 16, 'auto', 'staged'), (True, 64, 'blocked', 'staged'),
                                                   (False, 64, 'blocked', 'staged'), (True, 16, 'auto', 'deferred'), (True, 64, 'blocked', 'deferred'),
                                                   (False, 16, 'auto', 'deferred')])
def test_sharded_engine_two_ranks_hip_kernels(sparse, d, schedule, comm):
    """comm = 'deferred': the reduction completes only inside wait() and the buffer is NaN in between (DeferredPoisonComm) -- proves the
    pipelined schedule of step_sparse never touches an in-flight buffer."""
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    U, I, d, L, pairs, E0, batches = small_problem(d)
    ref_table, ref_losses = oracle_run(U, I, d, L, pairs, E0, batches)
    ret = _spawn(_worker, (sparse, d, schedule, comm))
    assert np.allclose(ret['losses'], ref_losses, rtol=RTOL, atol=0)
    assert rel_err(ret['table'], ref_table) < RTOL


def _simgcl_worker(rank, world, port, ret, comm='staged'):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from arlib_amd.dist_engine import ShardedPropagationEngine
    from test_dist_cpu import simgcl_problem
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    U, I, d, L, pairs, E0, batch, noise = simgcl_problem()
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cuda:0', rank, world, torch.from_numpy(E0), comm=_comm(comm),
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


@pytest.mark.parametrize('comm', ['staged', 'deferred'])
def test_sharded_simgcl_two_ranks_hip_kernels(comm):
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    from test_dist_cpu import simgcl_problem, oracle_simgcl_step
    ref_table, ref_rec, ref_cl = oracle_simgcl_step(*simgcl_problem())
    ret = _spawn(_simgcl_worker, (comm,))
    assert abs(ret['rec'] - ref_rec) <= RTOL * abs(ref_rec)
    assert abs(ret['cl'] - ref_cl) <= RTOL * abs(ref_cl)
    assert rel_err(ret['table'], ref_table) < RTOL
    assert ret['replica_diff'] == 0.0


def _clear_worker(rank, world, port, ret, skip0, comm='staged'):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from arlib_amd.dist_engine import ShardedPropagationEngine
    from test_dist_cpu import clear_problem, sp_mask
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    U, I, d, L, pairs, E0, n_real, targets, topk, r0, _ = clear_problem(skip0)
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cuda:0', rank, world, torch.from_numpy(E0), comm=_comm(comm), skip_layer0=skip0)
    m = sp_mask(pairs, U, I, eng.u0, eng.u1)
    res, _ = eng.step_clear(targets, n_real, topk, torch.from_numpy(m[0].astype(np.int32)).cuda(), torch.from_numpy(m[1]).cuda(), r0=torch.from_numpy(r0))
    full = eng.gather_full_table().cpu().numpy()
    if rank == 0:
        ret['table'], ret['cw'], ret['sfa'] = full, float(res[0]), float(res[1])
    dist.destroy_process_group()


@pytest.mark.parametrize('skip0,comm', [(True, 'staged'), (False, 'staged'), (True, 'deferred')])
def test_sharded_clear_step_two_ranks_hip_kernels(skip0, comm):
    """BASELINE config 4 (SimGCL + CLeaR, user-sharded) with the real kernels: masked top-k per shard, staged SFA (arl_sfa_stage1/2/3),
    item-row gradient exchange -- against the single-process oracle composition."""
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    from test_dist_cpu import clear_problem, oracle_clear_step
    ref_table, ref_cw, ref_sfa = oracle_clear_step(*clear_problem(skip0))
    ret = _spawn(_clear_worker, (skip0, comm))
    assert abs(ret['cw'] - ref_cw) <= RTOL * abs(ref_cw) and abs(ret['sfa'] - ref_sfa) <= RTOL * abs(ref_sfa)
    assert rel_err(ret['table'], ref_table) < RTOL


def _ngcf_worker(rank, world, port, ret, comm='staged'):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from arlib_amd.dist_engine import ShardedPropagationEngine
    from test_dist_cpu import ngcf_problem
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    U, I, d, L, pairs, E0, batches, W1, W2 = ngcf_problem()
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cuda:0', rank, world, torch.from_numpy(E0), comm=_comm(comm))
    eng.init_ngcf(W1, W2)
    losses = []
    for u, p, n in batches:
        lo = eng.step_ngcf(torch.from_numpy(u).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda())
        losses.append(float(lo[0] + lo[1]))
    full = eng.gather_full_table().cpu().numpy()
    if rank == 0:
        ret['table'], ret['losses'], ret['W'] = full, losses, [w.cpu().numpy() for w in eng.W]
    dist.destroy_process_group()


@pytest.mark.parametrize('comm', ['staged', 'deferred'])
def test_sharded_ngcf_two_ranks_hip_kernels(comm):
    """BASELINE config 5's training step (NGCF, user-sharded) with the real kernels against torch autograd on a dense fp64 graph."""
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    from test_dist_cpu import ngcf_problem, torch_ngcf_steps
    prob = ngcf_problem()
    ref_table, ref_W, ref_losses = torch_ngcf_steps(*prob)
    ret = _spawn(_ngcf_worker, (comm,))
    L, d = prob[3], prob[2]
    assert np.allclose(ret['losses'], ref_losses, rtol=RTOL, atol=0)
    assert rel_err(ret['table'], ref_table) < RTOL
    for l in range(L):
        assert rel_err(ret['W'][l][:d], ref_W[l]) < RTOL and rel_err(ret['W'][l][d:], ref_W[L + l]) < RTOL


def test_sfa_stages_equal_monolithic_call():
    """arl_sfa_stage1/2/3 chained without reductions = arl_sfa_l1_fwd_bwd_f32 bit for bit; split over two row blocks with the partial vectors
    added in between = the same loss and gradient to rounding."""
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    from arlib_amd import ops
    g = torch.Generator().manual_seed(4)
    n, d = 5000, 64
    X = (torch.randn(n, d, generator=g) * 0.2).cuda(); w = torch.randint(0, 4, (n,), generator=g).float().cuda(); r0 = torch.randn(d, generator=g).cuda()
    numel = int(w.sum().item()) * d
    loss, G = ops.sfa_l1(X, w, r0, numel)
    st = ops.SfaStages(X, w, r0)
    r = st.stage1(); a_s = st.stage2(r)
    loss2, G2 = st.stage3(r, a_s, numel)
    assert torch.equal(loss, loss2) and torch.equal(G, G2)
    h = 1777
    sa, sb = ops.SfaStages(X[:h].contiguous(), w[:h].contiguous(), r0), ops.SfaStages(X[h:].contiguous(), w[h:].contiguous(), r0)
    r = sa.stage1() + sb.stage1()
    a_s = sa.stage2(r) + sb.stage2(r)
    la, Ga = sa.stage3(r, a_s, numel); lb, Gb = sb.stage3(r, a_s, numel)
    assert abs(la.item() - loss.item()) <= 1e-5 * abs(loss.item()) and la.item() == lb.item()
    assert rel_err(torch.cat([Ga, Gb]).cpu().numpy(), G.cpu().numpy()) < 1e-5


def _pga_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from arlib_amd import ops
    from arlib_amd.dist_engine import ShardedPGA
    from test_dist_cpu import pga_problem
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    U, I, F, L, d, pairs, E0, targets, topk, S0 = pga_problem()
    eng = ShardedPGA(pairs, U, F, I, d, L, 'cuda:0', rank, world, torch.from_numpy(E0), comm=HostStagedComm())
    eng.set_block(S0)
    out, _ = eng.forward()
    top_idx, _ = ops.score_mask_topk(out[:eng.Ul].contiguous(), out[eng.Ul:].contiguous(), topk)
    losses = [float(eng.step(targets, top_idx)) for _ in range(2)]
    if rank == world - 1:
        ret['S'] = eng.S.cpu().numpy().copy()
    if rank == 0:
        ret['losses'] = losses
    dist.destroy_process_group()


def test_sharded_pga_two_ranks_hip_kernels():
    """PGA's gradient step on two user shards with the real kernels (raw-adjacency hops, SDDMM block, tanh/clamp update) against the oracle."""
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    from test_dist_cpu import pga_problem, oracle_pga_steps
    prob = pga_problem()
    ref_S, ref_losses, _ = oracle_pga_steps(*prob)
    ret = _spawn(_pga_worker, ())
    assert np.allclose(ret['losses'], ref_losses, rtol=RTOL, atol=0)
    assert rel_err(ret['S'], ref_S) < RTOL


def _rccl1_worker(rank, world, port, ret):
    """One rank on real RCCL: the direct item exchange behind the C ABI (communicator from a unique id, stream hand-over, workspace) and the
    sharded step routed through it.  With one rank the exchange moves nothing -- what runs here is everything AROUND it."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    from arlib_amd.dist_engine import ShardedPropagationEngine, TorchDistComm
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    comm = TorchDistComm(item_exchange='direct', direct_min_bytes=1024)
    t = torch.randn(1412 * 64, device='cuda'); ref = t.clone()
    w = comm.all_reduce_async(t)
    assert comm._native is not None and comm._native['world'] == 1           # went through arl_comm_init / arl_allreduce_item_f32
    w.wait(); torch.cuda.synchronize()
    ok = torch.equal(t, ref)
    U, I, d, L, pairs, E0, batches = small_problem(64)
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cuda:0', 0, 1, torch.from_numpy(E0), comm=comm)
    losses = []
    for u, p, n in batches:
        lo = eng.step_sparse(torch.from_numpy(u).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda())
        losses.append(float(lo[0] + lo[1]))
    ret['ok'], ret['table'], ret['losses'] = ok, eng.gather_full_table().cpu().numpy(), losses
    comm.close()
    dist.destroy_process_group()


def test_direct_item_exchange_one_rank_rccl():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    U, I, d, L, pairs, E0, batches = small_problem(64)
    ref_table, ref_losses = oracle_run(U, I, d, L, pairs, E0, batches)
    ret = _spawn(_rccl1_worker, (), nprocs=1)
    assert ret['ok']
    assert np.allclose(ret['losses'], ref_losses, rtol=RTOL, atol=0) and rel_err(ret['table'], ref_table) < RTOL
