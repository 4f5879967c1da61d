"""Parity at BASELINE.json's full size (cfg2: SYN-v1 1M users x 100K items, nnz 32.1M, d=64, L=3) through size-independent
properties -- the oracle cannot run this size in seconds, so the checks are identities that hold for any graph:
  * A_hat (D^1/2 1) = D^1/2 1            (exact fixed point of the symmetric normalisation, every row incl. 100k-edge rows)
  * <y, A x> = <A y, x>  and linearity   (the backward pass relies on the symmetry)
  * blocked hop == CSR hop (all epilogues); sparse-batch step (blocked) == dense 2L-hop step (CSR) on real sampler batches;
    three steps against the oracle itself
  * sampler epoch: positives are a permutation of the training pairs (checksum), negatives never interacted, indices in range
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def cfg2():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    from arlib_amd import ops
    from arlib_amd.util import synthetic
    data = synthetic.syn_v1(1_000_000, 100_000, 32.0, 2018)
    U, I, nnz = data.training_size()
    rowptr, col = data.adjacency_pattern()
    col_d = torch.from_numpy(col).to(DEV)
    val, dinv = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).to(DEV), col_d, torch.ones(2 * nnz, device=DEV), U + I)
    A = ops.CSRGraph(rowptr, col_d, val, DEV)                                   # CSR schedule only (no blocked plan attached)
    Ab = ops.CSRGraph(rowptr, col_d, val, DEV, validate=False).enable_blocked(split=U)      # the headline schedule: register-blocked plan, explicit
    assert A.blocked is None and Ab.blocked is not None and len(Ab.blocked.sets) == 2
    return dict(data=data, U=U, I=I, nnz=nnz, rowptr=rowptr, col=col, A=A, Ab=Ab, dinv=dinv, ops=ops)


def test_cfg2_graph_is_the_same_from_both_generators(cfg2):
    """The benchmark graph itself: numpy and C++ SYN-v1 generators agree at the full size (32 095 419 pairs, cross-language digest)."""
    from arlib_amd.util import synthetic as S
    p = cfg2['data'].pairs0
    q = S.syn_v1_pairs_native(cfg2['U'], cfg2['I'], 32.0, 2018)
    assert len(p) == 32_095_419 and np.array_equal(p, q) and S.graph_digest(p) == S.graph_digest_native(q)


def test_normalised_adjacency_fixed_point(cfg2):
    ops, A = cfg2['ops'], cfg2['A']
    deg = torch.from_numpy(np.diff(cfg2['rowptr']).astype(np.float32)).to(DEV)
    x = torch.sqrt(deg)[:, None].repeat(1, 64).contiguous()                   # D^1/2 1 in every column
    y = ops.spmm(A, x)
    rel = ((y - x).abs().max() / x.abs().max()).item()
    assert rel < 1e-5, rel
    assert int(deg.max()) > 50_000                                            # a popular-item row with > 50k edges is covered


def test_blocked_hop_equals_csr_hop_full_size_all_epilogues(cfg2):
    """The headline kernel at the headline size against an INDEPENDENT schedule: spmm_blocked64_kernel (+ split rows) vs the CSR row kernel on
    the same operand, for the three epilogues the step uses (AXPBY with a flagged Z, layer sum, fused Adam), plus the fixed point on the
    blocked path.  An offset overflow or a lost record at 64 M records shows up here (the two schedules share no index arrays)."""
    ops, A, Ab, U, I = cfg2['ops'], cfg2['A'], cfg2['Ab'], cfg2['U'], cfg2['I']
    N = U + I
    plan = Ab.blocked
    assert plan.n_hub == 0 and sum(s['n_edges'] for s in plan.sets) == 2 * cfg2['nnz'] and sum(s['n_rows'] for s in plan.sets) == N      # every edge is a record
    assert sum(s['n_split'] for s in plan.sets) > 100                          # incl. the > 4096-edge item rows, dealt as pieces
    g = torch.Generator(device=DEV).manual_seed(5)
    X = torch.randn(N, 64, device=DEV, generator=g); Z = torch.randn(N, 64, device=DEV, generator=g)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    yc, yb = ops.spmm(A, X), ops.spmm(Ab, X)
    assert rel(yb, yc) < 1e-5
    assert torch.equal(yb, ops.spmm(Ab, X))                                   # deterministic
    # row-wise check too: the tensor-level bar could hide a wrong small-magnitude row
    rown = (yb - yc).norm(dim=1) / yc.norm(dim=1).clamp_min(1e-20)
    assert float(rown.max()) < 1e-4
    zf = (torch.rand(N, device=DEV, generator=g) < 0.01).to(torch.uint8)
    Zs = Z * zf[:, None]
    assert rel(ops.spmm_flagged(Ab, X, None, 0.25, 0.25, Zs, zf), ops.spmm_flagged(A, X, None, 0.25, 0.25, Zs, zf)) < 1e-5
    Sc, Sb = Z.clone(), Z.clone()
    ops.spmm_layersum(A, X, Sc, Sc); ops.spmm_layersum(Ab, X, Sb, Sb)
    assert rel(Sb, Sc) < 1e-5
    P = torch.randn(N, 64, device=DEV, generator=g) * 0.1; M = torch.randn(N, 64, device=DEV, generator=g) * 0.01; V = torch.rand(N, 64, device=DEV, generator=g) * 1e-4
    Pc, Mc, Vc, Pb, Mb, Vb = P.clone(), M.clone(), V.clone(), P.clone(), M.clone(), V.clone()
    ops.spmm_adam(A, X, 0.25, 0.25, Zs, Pc, Mc, Vc, 0.005, 7, zflags=zf)
    ops.spmm_adam(Ab, X, 0.25, 0.25, Zs, Pb, Mb, Vb, 0.005, 7, zflags=zf)
    assert rel(Pb, Pc) < 1e-5 and rel(Mb, Mc) < 1e-5 and rel(Vb, Vc) < 1e-5
    deg = torch.from_numpy(np.diff(cfg2['rowptr']).astype(np.float32)).to(DEV)
    x = torch.sqrt(deg)[:, None].repeat(1, 64).contiguous()
    # A_hat (D^1/2 1) = D^1/2 1 through the blocked plan: an all-positive sum, the worst case for a sequential fp32 chain (a piece of a
    # 50k-edge row adds up to 4096 terms into one register: 2.4e-5 measured; the CSR kernel's shorter chains give < 1e-5)
    assert rel(ops.spmm(Ab, x), x) < 1e-4


def test_three_steps_at_full_size_match_the_oracle(cfg2):
    """Product and oracle meeting at cfg2: three Adam steps of the LightGCN d=64 L=3 step (the engine on its default = blocked schedule) from
    the bench's start state against oracle/arl_oracle.c on the host cores (OpenMP; ~4 s per step on the GPU box) -- updated tables within
    1e-4 rel, losses within 1e-4."""
    from arlib_amd import engine
    from arlib_amd.util.sampler import MTState
    from oracle import oracle as O
    O.build()
    ops, Ab, U, I, nnz = cfg2['ops'], cfg2['Ab'], cfg2['U'], cfg2['I'], cfg2['nnz']
    torch.manual_seed(2018)
    E0 = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, 64)), torch.nn.init.xavier_uniform_(torch.empty(I, 64))], 0)
    eng = engine.PropagationEngine(Ab, U, I, 64, 3, 1e-4, 0.005, DEV, table=E0.to(DEV))
    assert eng.A.blocked is not None
    st = O.TrainState(E0[:U].numpy(), E0[U:].numpy(), (cfg2['rowptr'], cfg2['col'], Ab.val.cpu().numpy()), 3, 1e-4, 0.005)
    st.f32acc = True
    mt = MTState.from_seed(2018)
    sampler = cfg2['data'].pair_sampler
    sampler.shuffle(mt)
    for k in range(3):
        b = sampler.batch(mt, k * 2048, 2048)
        ref = st.step(b[0].copy(), b[1].copy(), b[2].copy())
        lo = eng.step(*(torch.from_numpy(np.ascontiguousarray(x)).to(DEV) for x in b)).cpu().numpy()
        assert abs(float(lo[0] + lo[1]) - ref) <= 1e-4 * abs(ref)
    got = eng.E0.cpu().numpy()
    assert float(np.abs(got - st.E0).max() / np.abs(st.E0).max()) < 1e-4
    moved = np.abs(st.E0 - E0.numpy()).max()
    assert float(np.abs(got - st.E0).max()) < 1e-2 * moved                     # error far below what three steps moved the table by


def test_spmm_adjoint_symmetry_and_linearity(cfg2):
    ops, A = cfg2['ops'], cfg2['A']
    g = torch.Generator(device=DEV).manual_seed(1)
    N = cfg2['U'] + cfg2['I']
    x = torch.randn(N, 64, device=DEV, generator=g); y = torch.randn(N, 64, device=DEV, generator=g)
    Ax, Ay = ops.spmm(A, x), ops.spmm(A, y)
    a, b = (y.double() * Ax.double()).sum().item(), (Ay.double() * x.double()).sum().item()
    assert abs(a - b) <= 1e-5 * max(abs(a), abs(b), 1.0)
    lin = ops.spmm(A, x + 0.5 * y)
    assert ((lin - (Ax + 0.5 * Ay)).abs().max() / lin.abs().max()).item() < 1e-5


def test_sparse_step_equals_dense_step_full_size(cfg2):
    """Sparse-batch step on the BLOCKED schedule against the dense 2L-hop step on the CSR schedule (independent kernels and index arrays)."""
    from arlib_amd import engine
    from arlib_amd.util.sampler import MTState
    ops, A, Ab, U, I = cfg2['ops'], cfg2['A'], cfg2['Ab'], cfg2['U'], cfg2['I']
    torch.manual_seed(2018)
    E0 = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, 64)), torch.nn.init.xavier_uniform_(torch.empty(I, 64))], 0).to(DEV)
    ea = engine.PropagationEngine(Ab, U, I, 64, 3, 1e-4, 0.005, DEV, table=E0.clone())
    eb = engine.PropagationEngine(A, U, I, 64, 3, 1e-4, 0.005, DEV, table=E0.clone(), schedule='csr')
    assert ea.A.blocked is not None and eb.A.blocked is None
    mt = MTState.from_seed(2018)
    sampler = cfg2['data'].pair_sampler
    sampler.shuffle(mt)
    for k in range(2):
        b = torch.from_numpy(sampler.batch(mt, k * 2048, 2048)).to(DEV)
        la = ea.step(b[0], b[1], b[2]).cpu().numpy(); lb = eb.step_dense(b[0], b[1], b[2]).cpu().numpy()
        assert np.allclose(la, lb, rtol=1e-4, atol=0)
    for x, y in ((ea.E0, eb.E0), (ea.m, eb.m), (ea.v, eb.v)):
        assert ((x - y).abs().max() / y.abs().max()).item() < 1e-4
    assert abs(float(la[0]) - 0.6931) < 2e-3                                  # ln 2 at initialisation scale


def test_sampler_epoch_properties_full_size(cfg2):
    from arlib_amd.util.sampler import MTState, PairSampler
    data, U, I, nnz = cfg2['data'], cfg2['U'], cfg2['I'], cfg2['nnz']
    s = PairSampler(data.pairs0.copy(), I, (data.pair_sampler.memb_rowptr, data.pair_sampler.memb_items))
    mt = MTState.from_seed(7)
    key0 = np.sort(data.pairs0[:, 0].astype(np.int64) * I + data.pairs0[:, 1])
    s.shuffle(mt)
    assert np.array_equal(np.sort(s.pairs[:, 0].astype(np.int64) * I + s.pairs[:, 1]), key0)       # a permutation of the training pairs
    assert not np.array_equal(s.pairs[:1000], data.pairs0[:1000])
    rp, items = s.memb_rowptr, s.memb_items
    for k in (0, 1, nnz // 2048 - 1):
        u, p, n = s.batch(mt, k * 2048, 2048)
        assert u.min() >= 0 and u.max() < U and n.min() >= 0 and n.max() < I
        assert np.array_equal(np.stack([u, p], 1), s.pairs[k * 2048:(k + 1) * 2048])
        for uu, nn in zip(u[:256], n[:256]):                                  # negatives are never interacted items
            row = items[rp[uu]:rp[uu + 1]]
            j = np.searchsorted(row, nn)
            assert not (j < len(row) and row[j] == nn)


def test_score_mask_topk_full_size_against_dense_rows(cfg2):
    """Streaming masked top-50 over 1M x 100K against dense fp32 scoring + torch.topk on sampled users; every returned item of
    EVERY user is un-interacted (membership via the CSR) and the lists are sorted."""
    ops, U, I, nnz = cfg2['ops'], cfg2['U'], cfg2['I'], cfg2['nnz']
    torch.manual_seed(3)
    X = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, 64)), torch.nn.init.xavier_uniform_(torch.empty(I, 64))], 0).to(DEV)
    X = ops.spmm(cfg2['A'], X)                                                 # propagated tables: popularity-skewed scores
    Pu, Pi = X[:U].contiguous(), X[U:].contiguous()
    rp = torch.from_numpy(cfg2['rowptr'][:U + 1].astype(np.int32)).to(DEV)
    mc = (cfg2['A'].col[:nnz] - U).to(torch.int32).contiguous()
    idx, val = ops.score_mask_topk(Pu, Pi, 50, rp, mc)
    assert bool((val[:, :-1] >= val[:, 1:]).all())
    # no interacted item anywhere: (user, item) keys of the result vs the sorted interaction keys
    keys = (torch.arange(U, device=DEV, dtype=torch.int64)[:, None] * I + idx.long()).flatten()
    inter = torch.repeat_interleave(torch.arange(U, device=DEV, dtype=torch.int64), (rp[1:] - rp[:-1]).long()) * I + mc.long()
    pos = torch.searchsorted(inter, keys).clamp_(max=inter.numel() - 1)
    assert not bool((inter[pos] == keys).any())
    sample = torch.from_numpy(np.random.default_rng(0).choice(U, 512, replace=False)).to(DEV)
    sc = Pu[sample] @ Pi.T
    for r, u in enumerate(sample.tolist()):
        sc[r, mc[rp[u]:rp[u + 1]].long()] = -10e8
    rv, ri = torch.topk(sc, 50)
    assert (ri == idx[sample].long()).float().mean().item() > 0.999           # ties / last-ulp orderings aside
    assert torch.allclose(rv, val[sample], rtol=1e-5, atol=1e-7)
    # size-independent properties: a warm start from the result changes nothing; the two forms of the fp16 stream (32 users per wave with the lists in the
    # outputs / 16 users per wave with the lists in registers) agree bit for bit on all 50 M entries; so does the table-order stream
    i_w, v_w = ops.score_mask_topk(Pu, Pi, 50, rp, mc, warm_idx=idx)
    assert torch.equal(i_w, idx) and torch.equal(v_w, val)
    ops.TOPK_FORM2 = False
    try:
        i_1, v_1 = ops.score_mask_topk(Pu, Pi, 50, rp, mc)
    finally:
        ops.TOPK_FORM2 = True
    assert torch.equal(i_1, idx) and torch.equal(v_1, val)
    i_t, v_t = ops.score_mask_topk(Pu, Pi, 50, rp, mc, item_order=None)
    assert torch.equal(i_t, idx) and torch.equal(v_t, val)


def test_sfa_full_size_against_float64_closed_form(cfg2):
    """CLeaR's SFA term at cfg2 multiplicities (5 per real user, U per target, histogram of negatives) vs the same closed
    form evaluated in float64 (the closed form itself is pinned against the literal restatement in the CPU suite)."""
    ops, U, I = cfg2['ops'], cfg2['U'], cfg2['I']
    g = torch.Generator().manual_seed(11)
    X = (torch.randn(U + I, 64, generator=g) * 0.1).to(DEV)
    w = torch.zeros(U + I, device=DEV)
    w[:U] = 5.0
    neg = torch.randint(0, I, (5 * U,), generator=g).to(DEV)
    w[U:] = torch.bincount(neg, minlength=I).float()
    w[U + torch.arange(5, device=DEV) * 977] += float(U)
    r0 = torch.randn(64, generator=g).to(DEV)
    numel = 3 * U * 5 * 64
    loss, G = ops.sfa_l1(X, w, r0, numel)
    Xd, wd, r0d = X.double(), w.double(), r0.double()
    q = Xd @ r0d; r = Xd.T @ (wd * q); s = Xd @ r
    S, A, Q = (wd * s.abs()).sum(), r.abs().sum(), r @ r
    a = Xd.T @ (wd * torch.sign(s))
    g_r = ((A / Q) * a + (S / Q) * torch.sign(r) - (2 * S * A / Q ** 2) * r) / numel
    Gd = wd[:, None] * ((A / (numel * Q)) * torch.sign(s)[:, None] * r[None, :] + q[:, None] * g_r[None, :] + (Xd @ g_r)[:, None] * r0d[None, :])
    assert abs(loss.item() - (S * A / (numel * Q)).item()) <= 1e-4 * abs((S * A / (numel * Q)).item())
    assert ((G.double() - Gd).norm() / Gd.norm()).item() < 1e-4


def test_pga_gradient_step_full_size_against_oracle(cfg2):
    """BASELINE config 3 at its size: ONE PGA gradient step w.r.t. the F = 64 fake users' 64 x 100 K interaction block on the cfg2 graph
    (attack/White/PGA.py:92-142) -- device re-normalisation of the poisoned graph in factors, L = 3 hop forward, CW gradient through the bilinear
    operator, backward, row-restricted SDDMMs, tanh / clamp update -- against the oracle's step on the host (numpy graph rebuild, OpenMP SpMM; ~15 s):
    the CW loss, the WHOLE scaled gradient block and the whole updated block S."""
    import scipy.sparse as sp
    from oracle import oracle as O
    from arlib_amd.attack.White.PGA import FactoredFakeGraph, _hop, cw_operator, pga_step_block
    from arlib_amd.attack._common import cw_pairs
    ops, data, U, I, nnz = cfg2['ops'], cfg2['data'], cfg2['U'], cfg2['I'], cfg2['nnz']
    F, L, d = 64, 3, 64
    O.build()
    real = sp.csr_matrix((np.ones(nnz, np.float32), (data.pairs0[:, 0], data.pairs0[:, 1])), shape=(U, I))
    deg_i = np.bincount(data.pairs0[:, 1], minlength=I)
    targets = [int(t) for t in np.argsort(deg_i, kind='stable')[:5]]
    popular = np.argsort(-deg_i, kind='stable')[:int(0.05 * I)]
    fg = FactoredFakeGraph(real, U, F, I, device=DEV, emb_size=d)
    assert fg.W.blocked is not None                                           # the hops run on the register-blocked schedule
    del real
    g = torch.Generator().manual_seed(2018)
    S = torch.zeros(F, I, device=DEV)
    S[:, targets] = 1.0
    S[:, torch.from_numpy(popular).to(DEV)] = torch.rand(F, len(popular), generator=g).to(DEV) * 0.9 + 0.05      # fractional weights (PGA.py:137-140)
    E0 = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, d), generator=g), torch.nn.init.xavier_uniform_(torch.empty(F, d), generator=g),
                    torch.nn.init.xavier_uniform_(torch.empty(I, d), generator=g)], 0).to(DEV)
    graph = fg.set_block(S)
    out = E0.clone(); E = E0
    for _ in range(L):
        E = _hop(graph, E); out += E
    out /= (L + 1)
    top_idx, _ = ops.score_mask_topk(out[:U + F].contiguous(), out[U + F:].contiguous(), 50)      # PGA ranks without an interacted mask (PGA.py:100-102)
    users, pos, neg = cw_pairs(top_idx, U, targets, pop=True)
    M = cw_operator(U + F + I, U + F, users, pos, neg, device=DEV)
    block, loss = pga_step_block(fg.set_block(S), fg.fake_rows, U + F, I, E0, L, M)
    dr, dc = fg.dinv[U:U + F].contiguous(), fg.dinv[U + F:].contiguous()
    grad = (block * dr[:, None] * dc[None, :] * (S != 0)).cpu().numpy()
    S_new = ops.pga_update_(S.clone(), block, dr, dc).cpu().numpy()
    rowptr, col = data.adjacency_pattern()
    g_ref, S_ref, cw_ref = O.pga_step(rowptr[:U + 1].astype(np.int64), (col[:nnz] - U).astype(np.int64), U, F, I, S.cpu().numpy(), E0.cpu().numpy(), L,
                                      users.cpu().numpy(), pos.cpu().numpy(), neg.cpu().numpy())
    assert abs(float(loss) - float(cw_ref)) <= 1e-4 * abs(float(cw_ref))
    scale = np.abs(g_ref).max()
    assert scale > 0 and np.abs(grad - g_ref).max() <= 1e-4 * scale                       # every entry of the 64 x 100 K block
    rn = np.sqrt((g_ref.astype(np.float64) ** 2).sum(1))
    assert (np.sqrt(((grad - g_ref).astype(np.float64) ** 2).sum(1)) / np.maximum(rn, 1e-3 * rn.max())).max() < 1e-4      # and every fake user's row at its own magnitude
    assert np.count_nonzero(grad[:, np.setdiff1d(np.arange(I), np.concatenate([popular, targets]))]) == 0             # nothing outside the block's pattern
    assert np.abs(S_new - S_ref).max() <= 1e-6 and S_new.min() >= 9.9e-8 and S_new.max() <= 1.0


def test_simgcl_fused_step_full_size_equals_autograd_route(cfg2):
    """BASELINE config 4's training step at the cfg2 size: the fused SimGCL step (engine.step_simgcl: shared first hop, row-subset last hops, ONE
    backward pass for the three forwards, Adam in the last hop's epilogue) against the autograd route through the class surface (SimGCL_Encoder:
    three full forwards via _Propagate, util.loss BPR / L2 / InfoNCE, torch.optim.Adam) with the SAME injected noise tables -- losses and the
    whole table after the step (recommender/SimGCL.py:51-70,198-219); then two fused steps with in-kernel noise from one seed: bit-identical."""
    from types import SimpleNamespace
    from arlib_amd import engine
    from arlib_amd.recommender._base import SparseNormAdj
    from arlib_amd.recommender.SimGCL import SimGCL_Encoder
    from arlib_amd.util.loss import bpr_l2_loss
    from arlib_amd.util.sampler import MTState
    ops, Ab, U, I = cfg2['ops'], cfg2['Ab'], cfg2['U'], cfg2['I']
    N, d, L, B = U + I, 64, 2, 2048
    g = torch.Generator().manual_seed(4)
    E0 = torch.cat([torch.nn.init.xavier_uniform_(torch.empty(U, d), generator=g), torch.nn.init.xavier_uniform_(torch.empty(I, d), generator=g)], 0).to(DEV)
    gd = torch.Generator(device=DEV).manual_seed(5)
    noises = [[torch.rand(N, d, generator=gd, device=DEV) for _ in range(L)] for _ in range(2)]
    mt = MTState.from_seed(2018)
    s = cfg2['data'].pair_sampler
    s.shuffle(mt)
    b = torch.from_numpy(s.batch(mt, 0, B)).to(DEV)
    # fused
    # Adam with eps = 1e-3 on both routes: the FIRST Adam step is lr g / (|g| + eps), and two hops away from the batch |g| ~ 1e-8 = the default eps, where
    # d(update)/dg = lr / (4 eps) ~ 1e5 turns the routes' 1e-11 rounding differences in g into 7e-6 in the table (measured: 5.7e-4 of the table's
    # maximum).  With eps >> |g| the update is linear in g, so comparing the UPDATES at 1e-4 compares the gradients at 1e-4.
    AEPS = 1e-3
    eng = engine.PropagationEngine(Ab, U, I, d, L, 1e-4, 0.005, DEV, skip_layer0=True, table=E0.clone(), eps=AEPS)
    lo, cl = eng.step_simgcl(b[0], b[1], b[2], cl_rate=0.2, tau=0.2, eps=0.1, noises=noises)
    fused = eng.E0.clone(); lo = lo.cpu().numpy(); cl = float(cl)
    del eng
    # autograd route
    enc = SimGCL_Encoder.__new__(SimGCL_Encoder)
    torch.nn.Module.__init__(enc)
    enc.data = SimpleNamespace(user_num=U, item_num=I)
    enc.latent_size = enc.emb_size = d
    enc.eps, enc.n_layers, enc.n_prop_layers, enc._eng = 0.1, L, L, None
    packed = E0.clone()
    enc.embedding_dict = torch.nn.ParameterDict({'user_emb': torch.nn.Parameter(packed[:U]), 'item_emb': torch.nn.Parameter(packed[U:])})
    adj = SparseNormAdj.__new__(SparseNormAdj)
    adj.shape, adj.indptr, adj.indices, adj.values, adj._graph = (N, N), None, None, Ab.val, Ab
    enc.sparse_norm_adj = adj
    opt = torch.optim.Adam(enc.parameters(), lr=0.005, eps=AEPS)
    ue, ie = enc()
    ul, pl, nl = b[0].long(), b[1].long(), b[2].long()
    rec = bpr_l2_loss(ue[ul], ie[pl], ie[nl], 1e-4)
    cl2 = 0.2 * enc.cal_cl_loss([ul, pl], noises=noises)
    opt.zero_grad()
    (rec + cl2).backward()
    opt.step()
    ref = torch.cat([enc.embedding_dict['user_emb'].detach(), enc.embedding_dict['item_emb'].detach()], 0)
    assert abs(float(lo[0] + lo[1]) - float(rec.detach())) <= 1e-4 * abs(float(rec.detach())) and abs(cl - float(cl2.detach())) <= 1e-4 * abs(float(cl2.detach()))
    uf, ur = fused - E0, ref - E0                                                      # the step's updates
    assert ((uf - ur).abs().max() / ur.abs().max()).item() < 1e-4
    assert ((fused - ref).abs().max() / ref.abs().max()).item() < 1e-5                 # (and the tables themselves)
    rows = torch.from_numpy(np.random.default_rng(1).choice(N, 200_000, replace=False)).to(DEV)
    rn = ur[rows].norm(dim=1)
    assert (((uf[rows] - ur[rows]).norm(dim=1)) / torch.clamp(rn, min=1e-3 * float(rn.max()))).max().item() < 5e-4       # sampled rows, each UPDATE at its own magnitude (measured 1.2e-4: rows whose update is 1 % of the largest; the tables agree to 1e-8 there)
    moved = ((ref - E0).abs().max(dim=1)[0] > 0).float().mean().item()
    assert moved > 0.1                                                                 # dense Adam: two hops from the 6 K batch rows reach a large part of the graph, all of it moves
    del enc, opt, ref, fused
    outs = []
    for _ in range(2):
        e2 = engine.PropagationEngine(Ab, U, I, d, L, 1e-4, 0.005, DEV, skip_layer0=True, table=E0.clone())
        e2._noise_seed, e2._noise_stream = 1234567, 0
        for k in range(2):
            bk = torch.from_numpy(s.batch(MTState.from_seed(99 + k), k * B, B)).to(DEV)
            e2.step_simgcl(bk[0], bk[1], bk[2])
        outs.append(e2.E0.clone())
        del e2
    assert torch.equal(outs[0], outs[1])
