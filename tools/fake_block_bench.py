"""The two dense products of PGA's F x I fake block at cfg2 sizes: hand-written kernels against the library route they replace.
python3 tools/fake_block_bench.py"""
import sys, time, torch
sys.path.insert(0, '.')
from arlib_amd import ops
F, I, d = 64, 100000, 64
S = torch.rand(F, I, device='cuda'); X = torch.randn(I, d, device='cuda'); Xf = torch.randn(F, d, device='cuda')
rf = torch.rand(F, device='cuda'); ri = torch.rand(I, device='cuda')
Yf = torch.zeros(F, d, device='cuda'); Yi = torch.zeros(I, d, device='cuda')
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
c = 50
print('rows: kernel %.3f ms | bmm panels + addcmul %.3f ms | S @ X + addcmul %.3f ms' % (
    t(lambda: ops.fake_block_rows_(S, X, Yf, rscale=rf, alpha=0.5)),
    t(lambda: Yf.addcmul_(torch.bmm(S.view(F, c, I // c).permute(1, 0, 2), X.view(c, I // c, d)).sum(0), rf[:, None], value=0.5)),
    t(lambda: Yf.addcmul_(S @ X, rf[:, None], value=0.5))))
print('cols: kernel %.3f ms | S.t() @ Xf + addcmul %.3f ms' % (
    t(lambda: ops.fake_block_cols_(S, Xf, Yi, rscale=ri, alpha=0.5)),
    t(lambda: Yi.addcmul_(S.t() @ Xf, ri[:, None], value=0.5))))
