"""CPU tests of the N>1 path: block construction ("virtual ranks" in one process) and the user-sharded engine under
world_size-2 gloo with the oracle-backed kernel shim.  The result must equal the single-process oracle run."""
import os
import socket
import sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from conftest import rel_err, RTOL, ROOT
from oracle import oracle as O


def free_port():
    """A rendezvous port nobody listens on right now (consecutive tests of one pytest process used to share a pid-derived port)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def small_problem(d=16):
    from arlib_amd.util import synthetic as S
    U, I, L, B = 600, 90, 3, 256
    pairs = S.syn_v1_pairs(U, I, mean_deg=10, seed=7)
    rng = np.random.default_rng(1)
    E0 = ((rng.random((U + I, d)) * 2 - 1) * 0.1).astype(np.float32)
    batches = []
    for k in range(3):
        sel = rng.integers(0, len(pairs), B)
        batches.append((pairs[sel, 0].copy(), pairs[sel, 1].copy(), rng.integers(0, I, B).astype(np.int32)))
    return U, I, d, L, pairs, E0, batches


def oracle_run(U, I, d, L, pairs, E0, batches):
    rowptr, col, w = O.bipartite_csr(pairs[:, 0], pairs[:, 1], U, I)
    st = O.TrainState(E0[:U], E0[U:], (rowptr, col, O.norm_adj_values(rowptr, col, w)), L, 1e-4, 0.005)
    losses = [st.step(*b) for b in batches]
    return st.E0, losses


@pytest.mark.parametrize('world', [2, 3, 8])
def test_virtual_ranks_partial_item_sums_add_up(world):
    from arlib_amd.dist_engine import build_local_blocks, shard_bounds
    U, I, d, L, pairs, E0, _ = small_problem()
    rowptr, col, w = O.bipartite_csr(pairs[:, 0], pairs[:, 1], U, I)
    full = O.spmm((rowptr, col, O.norm_adj_values(rowptr, col, w)), E0)
    assert shard_bounds(U, world)[-1] == U
    item_sum = np.zeros((I, d), np.float64)
    for r in range(world):
        b = build_local_blocks(pairs, U, I, r, world)
        Ul = b['u1'] - b['u0']
        X = np.concatenate([E0[b['u0']:b['u1']], E0[U:]])
        yu = O.spmm(b['Au'], X)
        assert rel_err(yu, full[b['u0']:b['u1']]) < 1e-6                 # user rows are exact on their owner
        item_sum += O.spmm(b['Ai'], X)
    assert rel_err(item_sum, full[U:]) < 1e-6                             # item rows = sum of per-rank partials


def sp_mask(pairs, U, I, u0, u1):
    """interacted-item CSR over users [u0, u1) (pairs are user-major sorted)"""
    sel = (pairs[:, 0] >= u0) & (pairs[:, 0] < u1)
    rp = np.zeros(u1 - u0 + 1, np.int64)
    np.cumsum(np.bincount(pairs[sel, 0] - u0, minlength=u1 - u0), out=rp[1:])
    return rp, np.ascontiguousarray(pairs[sel, 1].astype(np.int32)) if sel.any() else np.zeros(1, np.int32)


class DeferredPoisonGloo:
    """Overlap-hazard double for the CPU suite (the GPU suite has the same one over host-staged gloo): all_reduce_async() sets the partial aside
    and fills the buffer with NaN; the reduction runs inside wait().  A step that reads or writes a buffer whose all-reduce is in flight cannot
    produce the oracle's tables."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist

    class _Work:
        def __init__(self, dist, t, saved):
            self.dist, self.t, self.saved = dist, t, saved

        def wait(self):
            self.dist.all_reduce(self.saved)
            self.t.copy_(self.saved)
            return True

    def all_reduce(self, t):
        self.dist.all_reduce(t)
        return t

    def all_reduce_async(self, t):
        saved = t.detach().clone()
        t.fill_(float('nan'))
        return self._Work(self.dist, t, saved)


def _worker(rank, world, port, ret, sparse, deferred=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    import cpu_kernels_shim as shim
    from arlib_amd.dist_engine import ShardedPropagationEngine
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    U, I, d, L, pairs, E0, batches = small_problem()
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cpu', rank, world, torch.from_numpy(E0), kernels=shim,
                                              comm=DeferredPoisonGloo() if deferred else None)
    losses = []
    for u, p, n in batches:
        lo = (eng.step_sparse if sparse else eng.step)(torch.from_numpy(u), torch.from_numpy(p), torch.from_numpy(n))
        losses.append(float(lo[0] + lo[1]))
    full = eng.gather_full_table().numpy()
    # sharded scoring: every rank ranks its own users against the replicated items; rank-ordered concatenation = global lists
    out = eng.forward()
    m = sp_mask(pairs, U, I, eng.u0, eng.u1)
    idx, _ = eng.score_topk(out, 10, torch.from_numpy(m[0].astype(np.int32)), torch.from_numpy(m[1]))
    gathered = [None] * world
    dist.all_gather_object(gathered, idx.numpy())
    if rank == 0:
        ret['table'], ret['losses'], ret['top10'] = full, losses, np.concatenate(gathered)
    dist.destroy_process_group()


@pytest.mark.parametrize('world,sparse,deferred', [(2, False, False), (2, True, False), (3, True, False), (2, True, True), (3, True, True)])
def test_sharded_engine_gloo_matches_single_process_oracle(world, sparse, deferred):
    U, I, d, L, pairs, E0, batches = small_problem()
    ref_table, ref_losses = oracle_run(U, I, d, L, pairs, E0, batches)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = free_port()
    mp.spawn(_worker, args=(world, port, ret, sparse, deferred), nprocs=world, join=True)
    assert np.allclose(ret['losses'], ref_losses, rtol=RTOL, atol=0)
    assert rel_err(ret['table'], ref_table) < RTOL
    # the sharded masked top-10 equals the single-process one on the trained table
    rowptr, col, w = O.bipartite_csr(pairs[:, 0], pairs[:, 1], U, I)
    out = O.lightgcn_forward((rowptr, col, O.norm_adj_values(rowptr, col, w)), ref_table, L)
    ref_idx, _ = O.score_mask_topk(out[:U], out[U:], 10, sp_mask(pairs, U, I, 0, U))
    assert (ret['top10'] == ref_idx).mean() > 0.995


def simgcl_problem():
    U, I, d, _, pairs, E0, batches = small_problem()
    rng = np.random.default_rng(5)
    L = 2
    noise = rng.random((2, L, U + I, d)).astype(np.float32)          # [view][hop] global rows; every rank slices its own
    return U, I, d, L, pairs, E0, batches[0], noise


def oracle_simgcl_step(U, I, d, L, pairs, E0, batch, noise, cl_rate=0.2, tau=0.2, eps=0.1):
    """Single-process restatement of one SimGCL iteration (same composition test_oracle_golden pins on g5_simgcl)."""
    rowptr, col, w = O.bipartite_csr(pairs[:, 0], pairs[:, 1], U, I)
    csr = (rowptr, col, O.norm_adj_values(rowptr, col, w))
    bu, bp, bn = batch
    out = O.lightgcn_forward(csr, E0, L, skip0=True)
    lb, lr_, G = O.bpr_l2(out, U, bu, bp, bn, 1e-4)
    v1 = O.lightgcn_forward(csr, E0, L, skip0=True, noises=noise[0], eps=eps)
    v2 = O.lightgcn_forward(csr, E0, L, skip0=True, noises=noise[1], eps=eps)
    cl = 0.0
    G = G.astype(np.float64)
    for idx in (np.unique(bu), np.unique(bp) + U):
        l, d1, d2 = O.infonce(v1[idx], v2[idx], tau)
        cl += l
        G[idx] += cl_rate * (d1.astype(np.float64) + d2)
    grad = O.lightgcn_backward(csr, G.astype(np.float32), L, skip0=True)
    m = np.zeros_like(E0); v = np.zeros_like(E0); E = E0.copy()
    O.adam_step(E, grad, m, v, 0.005, 1)
    return E, lb + lr_, cl_rate * cl


def _simgcl_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    import cpu_kernels_shim as shim
    from arlib_amd.dist_engine import ShardedPropagationEngine
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    U, I, d, L, pairs, E0, batch, noise = simgcl_problem()
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cpu', rank, world, torch.from_numpy(E0), kernels=shim, skip_layer0=True)
    local = lambda a: torch.from_numpy(np.concatenate([a[eng.u0:eng.u1], a[U:]]))
    lo, cl = eng.step_simgcl(*(torch.from_numpy(x) for x in batch), noises=[[local(noise[v][h]) for h in range(L)] for v in range(2)])
    full = eng.gather_full_table().numpy()
    with pytest.raises(ValueError):
        eng.step_sparse(*(torch.from_numpy(x) for x in batch))       # a skip_layer0 engine refuses the LightGCN step
    if rank == 0:
        ret['table'], ret['rec'], ret['cl'] = full, float(lo[0] + lo[1]), float(cl)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_simgcl_step_gloo_matches_single_process_oracle(world):
    ref_table, ref_rec, ref_cl = oracle_simgcl_step(*simgcl_problem())
    mgr = mp.Manager()
    ret = mgr.dict()
    port = free_port()
    mp.spawn(_simgcl_worker, args=(world, port, ret), nprocs=world, join=True)
    assert abs(ret['rec'] - ref_rec) <= RTOL * abs(ref_rec)
    assert abs(ret['cl'] - ref_cl) <= RTOL * abs(ref_cl)
    assert rel_err(ret['table'], ref_table) < RTOL


# ---------------------------------------------------------------------------------------------- sharded attack / NGCF steps (BASELINE configs 4 and 5)
def clear_problem(skip0):
    U, I, d, _, pairs, E0, _ = small_problem()
    rng = np.random.default_rng(9)
    F, L, T, topk = 6, 2, 3, 10
    n_real = U - F                                   # the last F users play the fake users (their rows are ordinary interaction rows here)
    targets = [int(x) for x in rng.choice(I, T, replace=False)]
    r0 = rng.standard_normal(d).astype(np.float32)
    return U, I, d, L, pairs, E0, n_real, targets, topk, r0, skip0


def oracle_clear_step(U, I, d, L, pairs, E0, n_real, targets, topk, r0, skip0):
    """Single-process restatement of one CLeaR surrogate step with the oracle's pieces (the composition test_oracle_attacks pins on g7 / g19)."""
    rowptr, col, w = O.bipartite_csr(pairs[:, 0], pairs[:, 1], U, I)
    csr = (rowptr, col, O.norm_adj_values(rowptr, col, w))
    out = O.lightgcn_forward(csr, E0, L, skip0=skip0)
    idx, _ = O.score_mask_topk(out[:U], out[U:], topk, sp_mask(pairs, U, I, 0, U))
    users, pos, neg = O.cw_pairs(idx, n_real, np.array(targets), pop=True)
    cw, sfa, G = O.clear_loss_grad(out, U, users, pos, neg, r0)
    grad = O.lightgcn_backward(csr, G, L, skip0=skip0)
    m = np.zeros_like(E0); v = np.zeros_like(E0); E = E0.copy()
    O.adam_step(E, grad, m, v, 0.005, 1)
    return E, cw, sfa


def _clear_worker(rank, world, port, ret, skip0):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    import cpu_kernels_shim as shim
    from arlib_amd.dist_engine import ShardedPropagationEngine
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    U, I, d, L, pairs, E0, n_real, targets, topk, r0, _ = clear_problem(skip0)
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cpu', rank, world, torch.from_numpy(E0), kernels=shim, skip_layer0=skip0)
    m = sp_mask(pairs, U, I, eng.u0, eng.u1)
    res, _ = eng.step_clear(targets, n_real, topk, torch.from_numpy(m[0].astype(np.int32)), torch.from_numpy(m[1]), r0=torch.from_numpy(r0))
    full = eng.gather_full_table().numpy()
    if rank == 0:
        ret['table'], ret['cw'], ret['sfa'] = full, float(res[0]), float(res[1])
    dist.destroy_process_group()


@pytest.mark.parametrize('world,skip0', [(2, True), (3, True), (2, False)])
def test_sharded_clear_step_gloo_matches_single_process_oracle(world, skip0):
    """BASELINE config 4: the CLeaR surrogate step (CW + SFA through a SimGCL-shaped, skip-layer-0, or LightGCN-shaped mean) on user shards --
    local masked top-k, local CW pairs, the SFA reductions cut at their two global sums, item-row gradient all-reduced -- equals the
    single-process oracle composition."""
    prob = clear_problem(skip0)
    ref_table, ref_cw, ref_sfa = oracle_clear_step(*prob)
    ret = mp.Manager().dict()
    mp.spawn(_clear_worker, args=(world, free_port(), ret, skip0), nprocs=world, join=True)
    assert abs(ret['cw'] - ref_cw) <= RTOL * abs(ref_cw) and abs(ret['sfa'] - ref_sfa) <= RTOL * abs(ref_sfa)
    assert rel_err(ret['table'], ref_table) < RTOL


def ngcf_problem():
    U, I, d, _, pairs, E0, batches = small_problem()
    rng = np.random.default_rng(13)
    L = 2
    W1 = [(rng.standard_normal((d, d)) * 0.3).astype(np.float32) for _ in range(L)]
    W2 = [(rng.standard_normal((d, d)) * 0.3).astype(np.float32) for _ in range(L)]
    return U, I, d, L, pairs, E0, batches[:2], W1, W2


def torch_ngcf_steps(U, I, d, L, pairs, E0, batches, W1, W2, reg=1e-4, lr=0.005):
    """Independent single-process reference: the reference's own layer expression (recommender/NGCF.py:197-212, both sparse hops) on a dense
    fp64 adjacency with torch autograd and torch.optim.Adam."""
    rowptr, col, w = O.bipartite_csr(pairs[:, 0], pairs[:, 1], U, I)
    val = O.norm_adj_values(rowptr, col, w)
    N = U + I
    A = torch.zeros(N, N, dtype=torch.float64)
    A[torch.from_numpy(np.repeat(np.arange(N), np.diff(rowptr))), torch.from_numpy(col.astype(np.int64))] = torch.from_numpy(val.astype(np.float64))
    E = torch.nn.Parameter(torch.from_numpy(E0.astype(np.float64)))
    Ws = [torch.nn.Parameter(torch.from_numpy(x.astype(np.float64))) for x in W1 + W2]
    opt = torch.optim.Adam([E] + Ws, lr=lr)
    losses = []
    for bu, bp, bn in batches:
        ego = E; layers = [ego]
        for l in range(L):
            t = ego @ Ws[l]
            ego = torch.nn.functional.leaky_relu(A @ t + t + ((A @ ego) * ego) @ Ws[L + l])
            layers.append(ego)
        out = torch.stack(layers, 1).mean(1)
        u, p, n = (torch.from_numpy(x.astype(np.int64)) for x in (bu, bp, bn))
        ue, pe, ne = out[u], out[U + p], out[U + n]
        loss = -torch.log(1e-7 + torch.sigmoid((ue * pe).sum(1) - (ue * ne).sum(1))).mean() + reg * (torch.norm(ue) + torch.norm(pe))
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.detach()))
    return E.detach().numpy(), [x.detach().numpy() for x in Ws], losses


def _ngcf_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    import cpu_kernels_shim as shim
    from arlib_amd.dist_engine import ShardedPropagationEngine
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    U, I, d, L, pairs, E0, batches, W1, W2 = ngcf_problem()
    eng = ShardedPropagationEngine.from_pairs(pairs, U, I, d, L, 1e-4, 0.005, 'cpu', rank, world, torch.from_numpy(E0), kernels=shim)
    eng.init_ngcf(W1, W2)
    losses = []
    for u, p, n in batches:
        lo = eng.step_ngcf(torch.from_numpy(u), torch.from_numpy(p), torch.from_numpy(n))
        losses.append(float(lo[0] + lo[1]))
    full = eng.gather_full_table().numpy()
    Wl = [w.numpy().copy() for w in eng.W]
    gathered = [None] * world
    dist.all_gather_object(gathered, Wl[0])
    if rank == 0:
        ret['table'], ret['losses'], ret['W'] = full, losses, Wl
        ret['w_replica_diff'] = float(max(np.abs(g - gathered[0]).max() for g in gathered))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_ngcf_steps_gloo_match_torch_autograd_reference(world):
    """BASELINE config 5's training step: NGCF on user shards (one hop per layer with its item-row all-reduce, row-local dense part with
    replicated weights, weight gradients all-reduced) against the reference's layer expression under torch autograd on a dense fp64 graph."""
    prob = ngcf_problem()
    ref_table, ref_W, ref_losses = torch_ngcf_steps(*prob)
    ret = mp.Manager().dict()
    mp.spawn(_ngcf_worker, args=(world, free_port(), ret), nprocs=world, join=True)
    L, d = prob[3], prob[2]
    assert np.allclose(ret['losses'], ref_losses, rtol=RTOL, atol=0)
    assert rel_err(ret['table'], ref_table) < RTOL
    for l in range(L):
        assert rel_err(ret['W'][l][:d], ref_W[l]) < RTOL and rel_err(ret['W'][l][d:], ref_W[L + l]) < RTOL
    assert ret['w_replica_diff'] == 0.0


def pga_problem():
    from arlib_amd.util import synthetic as S_
    U, I, F, L, d, T, topk = 594, 90, 6, 2, 16, 3, 10
    pairs = S_.syn_v1_pairs(U, I, mean_deg=10, seed=11)
    rng = np.random.default_rng(17)
    E0 = ((rng.random((U + F + I, d)) * 2 - 1) * 0.1).astype(np.float32)
    targets = [int(x) for x in rng.choice(I, T, replace=False)]
    S0 = np.zeros((F, I), np.float32)
    S0[:, targets] = 1.0
    S0[:, rng.choice(I, 9, replace=False)] = 0.5
    S0[2, 5] = 0.0
    return U, I, F, L, d, pairs, E0, targets, topk, S0


def oracle_pga_steps(U, I, F, L, d, pairs, E0, targets, topk, S0, n_steps=2):
    """Single-process oracle: top-k lists from the forward of the FIRST step's graph (PGA.py:99-108), then n gradient steps (O.pga_step, the
    composition test_oracle_attacks pins on the reference's traced grad / S)."""
    rp = np.zeros(U + 1, np.int64); np.cumsum(np.bincount(pairs[:, 0], minlength=U), out=rp[1:])
    ri = pairs[:, 1].astype(np.int64)
    csr, _ = O.pga_weighted_graph(rp, ri, U, F, I, S0)
    out = O.lightgcn_forward(csr, E0, L)
    Up = U + F
    idx, _ = O.score_mask_topk(out[:Up], out[Up:], topk)
    users, pos, neg = O.cw_pairs(idx, U, np.array(targets), pop=True)
    S, losses = S0.copy(), []
    for _ in range(n_steps):
        grad, S, loss = O.pga_step(rp, ri, U, F, I, S, E0, L, users, pos, neg)
        losses.append(float(loss))
    return S, losses, idx


def _pga_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    import cpu_kernels_shim as shim
    from arlib_amd.dist_engine import ShardedPGA
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    U, I, F, L, d, pairs, E0, targets, topk, S0 = pga_problem()
    eng = ShardedPGA(pairs, U, F, I, d, L, 'cpu', rank, world, torch.from_numpy(E0), kernels=shim)
    eng.set_block(S0)
    out, _ = eng.forward()
    top_idx, _ = shim.score_mask_topk(out[:eng.Ul].contiguous(), out[eng.Ul:].contiguous(), topk)      # unmasked, as PGA.py:100-102
    losses = [float(eng.step(targets, top_idx)) for _ in range(2)]
    gathered = [None] * world
    dist.all_gather_object(gathered, top_idx.numpy())
    if rank == world - 1:
        ret['S'] = eng.S.numpy().copy()
    if rank == 0:
        ret['losses'], ret['top'] = losses, np.concatenate(gathered)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_pga_steps_gloo_match_single_process_oracle(world):
    """PGA's gradient step on user shards (factored poisoned operator, fake block on the last rank, item-side partials all-reduced) against the
    oracle's single-process pga_step: CW loss and the updated fake block after two steps."""
    prob = pga_problem()
    ref_S, ref_losses, ref_idx = oracle_pga_steps(*prob)
    ret = mp.Manager().dict()
    mp.spawn(_pga_worker, args=(world, free_port(), ret), nprocs=world, join=True)
    assert (ret['top'] == ref_idx).mean() > 0.995
    assert np.allclose(ret['losses'], ref_losses, rtol=RTOL, atol=0)
    assert rel_err(ret['S'], ref_S) < RTOL and (ret['S'] != prob[-1]).any()


# ---------------------------------------------------------------------------------------------- direct item exchange (arl_allreduce_item_f32)
def test_item_exchange_partition_tiles_the_buffer():
    """arl_item_exchange_range (the shard x chunk partition arl_allreduce_item_f32 sends by): every element in exactly one (shard, chunk) range,
    ranges in order, starts 16-byte aligned -- for ragged sizes, more ranks than 4-element groups, one element, nothing."""
    import ctypes as C
    from arlib_amd import _lib
    L = _lib.lib()
    for n in (0, 1, 3, 4, 5, 1023, 25_600_000, 1412 * 64, 7):
        for world in (1, 2, 3, 8):
            for chunks in (1, 4, 7):
                pos = 0
                for q in range(world):
                    for c in range(chunks):
                        lo, hi = C.c_int64(), C.c_int64()
                        assert L.arl_item_exchange_range(n, world, chunks, q, c, C.byref(lo), C.byref(hi)) == 0
                        assert lo.value == pos and hi.value >= lo.value and (lo.value % 4 == 0 or lo.value == n)
                        pos = hi.value
                assert pos == n
                need = L.arl_allreduce_item_workspace_bytes(n, world, chunks)
                biggest = max(1, -(-n // world))
                assert need >= 4 * (world - 1) * -(-biggest // chunks)
    lo, hi = C.c_int64(), C.c_int64()
    assert L.arl_item_exchange_range(10, 2, 1, 2, 0, C.byref(lo), C.byref(hi)) == -4          # shard out of range


def _direct_worker(rank, world, port, ret, n, chunks):
    """The schedule of arl_allreduce_item_f32 (direct reduce-scatter, owner sums in rank order, direct all-gather, chunk by chunk) replayed
    with gloo point-to-point calls over the SAME partition function: the result must equal a plain all-reduce."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch.distributed as dist
    from arlib_amd import _lib
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    L = _lib.lib()

    def rng(q, c):
        lo, hi = C.c_int64(), C.c_int64()
        assert L.arl_item_exchange_range(n, world, chunks, q, c, C.byref(lo), C.byref(hi)) == 0
        return lo.value, hi.value
    g = torch.Generator().manual_seed(100 + rank)
    buf = torch.randn(n, generator=g)
    ref = buf.clone(); dist.all_reduce(ref)
    for c in range(chunks):
        mlo, mhi = rng(rank, c)
        reqs, slots = [], {}
        for q in range(world):
            if q == rank:
                continue
            a, b = rng(q, c)
            if b > a:
                reqs.append(dist.isend(buf[a:b].clone(), q))
            if mhi > mlo:
                slots[q] = torch.empty(mhi - mlo)
                reqs.append(dist.irecv(slots[q], q))
        for r in reqs:
            r.wait()
        if mhi > mlo:
            acc = torch.zeros(mhi - mlo)
            for r in range(world):                                  # rank order, as shard_sum_kernel
                acc += buf[mlo:mhi] if r == rank else slots[r]
            buf[mlo:mhi] = acc
        reqs = []
        for q in range(world):
            if q == rank:
                continue
            a, b = rng(q, c)
            if mhi > mlo:
                reqs.append(dist.isend(buf[mlo:mhi].clone(), q))
            if b > a:
                reqs.append(dist.irecv(buf[a:b], q))
        for r in reqs:
            r.wait()
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, buf)
    if rank == 0:
        ret['err'] = float((buf - ref).abs().max() / ref.abs().max())
        ret['replicas_equal'] = all(torch.equal(gathered[0], x) for x in gathered)
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n,chunks', [(2, 1412 * 16, 4), (3, 1001, 3), (3, 5, 2)])
def test_direct_item_exchange_schedule_equals_all_reduce(world, n, chunks):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_direct_worker, args=(world, free_port(), ret, n, chunks), nprocs=world, join=True)
    assert ret['err'] < 1e-6 and ret['replicas_equal']             # one owner per element: every replica holds the same bits
