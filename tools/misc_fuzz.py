"""Randomised cross-check of the remaining kernels against float64 torch: BPR+L2 forward/backward (ragged sizes, duplicates,
saturated scores), InfoNCE forward/backward, Adam/SGD, gather/scatter, SimGCL perturbation, SDDMM rows, top-n projection,
SFA, NGCF glue, adjacency normalisation.   python3 tools/misc_fuzz.py [n_cases]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(4242)
dev = 'cuda'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
bad = 0


def report(case, name, errs, tol=2e-5):
    global bad
    w = max(errs.values())
    if not w < tol:
        bad += 1
        print('MISMATCH case %d %s: %s' % (case, name, {k: '%.2e' % v for k, v in errs.items()}), flush=True)


for case in range(n_cases):
    d = int(rng.choice([4, 8, 16, 20, 64, 100, 128, 256]))
    # ---- BPR + L2
    U, I, B = int(rng.integers(1, 500)), int(rng.integers(1, 300)), int(rng.integers(1, 3000))
    scale = float(rng.choice([0.05, 0.5, 3.0]))
    emb = torch.from_numpy((rng.standard_normal((U + I, d)) * scale).astype(np.float32)).to(dev)
    u = T(rng.integers(0, U, B).astype(np.int32)); p = T(rng.integers(0, I, B).astype(np.int32)); n = T(rng.integers(0, I, B).astype(np.int32))
    e = emb.double().requires_grad_(True)
    ue, pe, ne = e[u.long()], e[U + p.long()], e[U + n.long()]
    lb = -torch.log(1e-7 + torch.sigmoid((ue * pe).sum(1) - (ue * ne).sum(1))).mean()
    lr_ = 1e-3 * (ue.norm() + pe.norm())
    (lb + lr_).backward()
    G = torch.zeros_like(emb)
    lo = ops.bpr_l2_fwd_bwd(emb, U, u, p, n, 1e-3, G)
    report(case, 'bpr_l2 d=%d B=%d scale=%g' % (d, B, scale), {'bpr': abs(lo[0].item() - lb.item()) / max(abs(lb.item()), 1e-30),
                                                                'reg': abs(lo[1].item() - lr_.item()) / max(abs(lr_.item()), 1e-30), 'grad': rel(G, e.grad)}, 5e-5)
    # ---- InfoNCE
    nn_ = int(rng.integers(1, 700)); tau = float(rng.choice([0.1, 0.2, 1.0]))
    v1 = T(rng.standard_normal((nn_, d)).astype(np.float32)); v2 = T((rng.standard_normal((nn_, d)) + 0.5 * v1.cpu().numpy()).astype(np.float32))
    a = v1.double().requires_grad_(True); b = v2.double().requires_grad_(True)
    an, bn = F.normalize(a, dim=1), F.normalize(b, dim=1)
    pos = torch.exp((an * bn).sum(1) / tau); ttl = torch.exp(an @ bn.T / tau).sum(1)
    l = -torch.log(pos / ttl).mean(); l.backward()
    if d % 4 == 0 and d <= 256:
        lo, d1, d2 = ops.infonce_fwd_bwd(v1, v2, tau)
        report(case, 'infonce n=%d d=%d tau=%g' % (nn_, d, tau), {'loss': abs(lo.item() - l.item()) / abs(l.item()), 'd1': rel(d1, a.grad), 'd2': rel(d2, b.grad)}, 5e-5)
    # ---- Adam / SGD / gather / scatter / perturb
    N = int(rng.integers(1, 4000))
    P = T(rng.standard_normal((N, d)).astype(np.float32)); g = T(rng.standard_normal((N, d)).astype(np.float32))
    M = T((rng.standard_normal((N, d)) * 0.1).astype(np.float32)); V = T((rng.random((N, d)) * 0.01).astype(np.float32))
    t = int(rng.integers(1, 100))
    m2 = 0.9 * M.double() + 0.1 * g.double(); v2_ = 0.999 * V.double() + 0.001 * g.double() ** 2
    p2 = P.double() - (0.005 / (1 - 0.9 ** t)) * m2 / (v2_.sqrt() / (1 - 0.999 ** t) ** 0.5 + 1e-8)
    Pc = P.clone(); ops.adam_dense(Pc, g, M, V, 0.005, t)
    Ps = P.clone(); ops.sgd_dense(Ps, g, 0.01)
    idx = T(rng.integers(0, N, int(rng.integers(1, 500))).astype(np.int32))
    src = T(rng.standard_normal((idx.numel(), d)).astype(np.float32))
    dst = P.clone(); ops.scatter_add_rows(dst, idx, src, 0.5)
    dref = P.double().index_add(0, idx.long(), 0.5 * src.double())
    noise = T(rng.random((N, d)).astype(np.float32)); Ep = P.clone(); ops.simgcl_perturb_(Ep, noise, 0.1)
    pref = P.double() + torch.sign(P.double()) * F.normalize(noise.double(), dim=-1) * 0.1
    report(case, 'dense d=%d N=%d' % (d, N), {'adam': max(rel(Pc, p2), rel(M, m2), rel(V, v2_)), 'sgd': rel(Ps, P.double() - 0.01 * g.double()),
                                               'gather': rel(ops.gather_rows(P, idx), P[idx.long()]), 'scatter': rel(dst, dref), 'perturb': rel(Ep, pref)})
    # ---- SDDMM rows + top-n projection + PGA update
    if d % 4 == 0:
        Fr, Ic = int(rng.integers(1, 9)), int(rng.integers(1, 900))
        dY = T(rng.standard_normal((N, d)).astype(np.float32)); X = T(rng.standard_normal((N + Ic, d)).astype(np.float32))
        rows = T(rng.integers(0, N, Fr).astype(np.int32))
        out = ops.sddmm_rows_dense(dY, X, rows, N, Ic)
        report(case, 'sddmm d=%d' % d, {'sddmm': rel(out, dY.double()[rows.long()] @ X.double()[N:].T)})
        Mx = T(rng.standard_normal((Fr, Ic)).astype(np.float32)); nsel = int(rng.integers(0, Ic + 1))
        po, pi_ = ops.topn_project_rows(Mx, nsel)
        ref = torch.zeros_like(Mx)
        if nsel:
            ref.scatter_(1, torch.topk(Mx, nsel, dim=1)[1], 1.0)
        report(case, 'topn n=%d/%d' % (nsel, Ic), {'proj': float((po != ref).float().sum().item())}, 0.5)
    # ---- SFA (weighted rows) against the closed form in float64
    w = T(rng.integers(0, 4, N).astype(np.float32)); w[0] = 1.0
    r0 = T(rng.standard_normal(d).astype(np.float32))
    numel = int(w.sum().item()) * d
    loss, Gs = ops.sfa_l1(P, w, r0, numel)
    Xd, wd, r0d = P.double(), w.double(), r0.double()
    q = Xd @ r0d; r = Xd.T @ (wd * q); s_ = Xd @ r
    S, A_, Q = (wd * s_.abs()).sum(), r.abs().sum(), r @ r
    a_ = Xd.T @ (wd * torch.sign(s_))
    g_r = ((A_ / Q) * a_ + (S / Q) * torch.sign(r) - (2 * S * A_ / Q ** 2) * r) / numel
    Gd = wd[:, None] * ((A_ / (numel * Q)) * torch.sign(s_)[:, None] * r[None, :] + q[:, None] * g_r[None, :] + (Xd @ g_r)[:, None] * r0d[None, :])
    report(case, 'sfa d=%d N=%d' % (d, N), {'loss': abs(loss.item() - (S * A_ / (numel * Q)).item()) / abs((S * A_ / (numel * Q)).item()), 'grad': rel(Gs, Gd)}, 1e-4)
    # ---- NGCF glue
    if d % 4 == 0:
        Pn = T(rng.standard_normal((N, d)).astype(np.float32)); En = T(rng.standard_normal((N, d)).astype(np.float32))
        ST = ops.ngcf_combine(Pn, En)
        Zc = P.clone(); acc = g.clone(); ops.ngcf_act_(Zc, acc, 0.01)
        gz = ops.ngcf_act_bwd(g, Zc, 0.01)
        gST = T(rng.standard_normal((N, 2 * d)).astype(np.float32))
        gP, gE = ops.ngcf_combine_bwd(gST, Pn, En)
        report(case, 'ngcf d=%d' % d, {'combine': rel(ST, torch.cat([Pn + En, Pn * En], 1)), 'act': rel(Zc, F.leaky_relu(P, 0.01)), 'acc': rel(acc, g + F.leaky_relu(P, 0.01)),
                                        'act_bwd': rel(gz, g * torch.where(Zc > 0, 1.0, 0.01)), 'gP': rel(gP, gST[:, :d] + gST[:, d:] * En), 'gE': rel(gE, gST[:, :d] + gST[:, d:] * Pn)})
    # ---- CW term from top-k lists (arl_cw_topk_term_f32): loss, gradient on every row, SFA multiplicities vs float64; two runs bit-identical
    if d % 4 == 0:
        Uc, Fc, Ic = int(rng.integers(1, 1500)), int(rng.integers(0, 5)), int(rng.integers(8, 3000))
        kc = int(rng.integers(1, min(Ic, 64) + 1)); Tc = int(rng.integers(1, min(kc, 8) + 1))
        Upc = Uc + Fc
        Xc = (rng.standard_normal((Upc + Ic, d)) * float(rng.choice([1e-3, 0.1, 5.0]))).astype(np.float32)
        hot = int(rng.integers(0, Ic))
        top = rng.integers(0, Ic, (Upc, kc)).astype(np.int32)
        if rng.random() < 0.5:
            top[: max(1, Uc // 2), kc - 1] = hot                        # most negatives on one item
        tgc = rng.choice(Ic, size=Tc, replace=False).astype(np.int64)
        cc = 1.0 / (Uc * Tc)
        Xd = Xc.astype(np.float64)
        neg = top[:Uc][:, [kc - 1 - t_ for t_ in range(Tc)]].astype(np.int64)
        Gr = np.zeros_like(Xd); lr = 0.0
        for t_ in range(Tc):
            Gr[:Uc] += cc * (Xd[Upc + neg[:, t_]] - Xd[Upc + tgc[t_]])
            np.add.at(Gr, Upc + neg[:, t_], cc * Xd[:Uc])
            Gr[Upc + tgc[t_]] -= cc * Xd[:Uc].sum(0)
            lr += cc * ((Xd[:Uc] * Xd[Upc + neg[:, t_]]).sum() - (Xd[:Uc] * Xd[Upc + tgc[t_]]).sum())
        wr = np.zeros(Upc + Ic); wr[:Uc] = Tc
        np.add.at(wr, Upc + neg.reshape(-1), 1.0); np.add.at(wr, Upc + tgc, float(Uc))
        l1, G1, w1 = ops.cw_topk_term(T(Xc), Upc, Uc, T(top), T(tgc))
        l2, G2, w2 = ops.cw_topk_term(T(Xc), Upc, Uc, T(top), T(tgc))
        same = torch.equal(l1, l2) and torch.equal(G1, G2) and torch.equal(w1, w2)
        report(case, 'cw_topk_term U=%d I=%d d=%d k=%d T=%d' % (Uc, Ic, d, kc, Tc),
               {'loss': abs(l1.item() - lr) / max(abs(lr), float(np.abs(Gr).max()), 1e-30), 'G': rel(G1, torch.from_numpy(Gr).to(dev)),
                'w': float(np.abs(w1.cpu().numpy() - wr).max()), 'rerun': 0.0 if same else 1.0}, 2e-5)
print('%d cases: %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)
