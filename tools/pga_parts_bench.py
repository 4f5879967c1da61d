"""Element-wise and dense pieces of the factored PGA hop at cfg2 sizes.   python3 tools/pga_parts_bench.py"""
import sys, torch, time
sys.path.insert(0, '.')
N, d, F, I, U = 1100064, 64, 64, 100000, 1000000
X = torch.randn(N, d, device='cuda'); dc = torch.rand(N, 1, device='cuda'); S = torch.rand(F, I, device='cuda'); Z = torch.randn(N, d, device='cuda')
Y = torch.empty_like(X)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print('X * dinv          : %.3f ms' % t(lambda: X * dc))
print('Y.mul_(dinv)      : %.3f ms' % t(lambda: Y.mul_(dc)))
print('Y.add_(Z, alpha)  : %.3f ms' % t(lambda: Y.add_(Z, alpha=0.5)))
Up = U + F
print('fake rows addmm   : %.3f ms' % t(lambda: Y[U:Up].addmm_(S, X[Up:])))
print('item rows addmm   : %.3f ms' % t(lambda: Y[Up:].addmm_(S.t(), X[U:Up])))
c = 50
print('fake rows split-K : %.3f ms' % t(lambda: Y[U:Up].add_(torch.bmm(S.view(F, c, I // c).permute(1, 0, 2), X[Up:].view(c, I // c, d)).sum(0))))
print('addcmul           : %.3f ms' % t(lambda: torch.addcmul(Z, Y, dc, out=Y)))
