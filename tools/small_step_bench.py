"""Per-step time of the fused engine step in the launch-bound regime (ml-100k- to 50 K x 10 K-sized graphs).   python3 tools/small_step_bench.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops, engine
from arlib_amd.util import synthetic
for U, I, deg in ((943, 1412, 85.0), (6040, 3700, 130.0), (50000, 10000, 32.0)):
    data = synthetic.syn_v1(U, I, mean_deg=deg)
    rowptr, col = data.adjacency_pattern()
    val, _ = ops.norm_adj_values(torch.from_numpy(rowptr.astype(np.int32)).cuda(), torch.from_numpy(col).cuda(), torch.ones(len(col), device='cuda'), U + I)
    d, L, B = 64, 3, 2048
    eng = engine.PropagationEngine(ops.CSRGraph(rowptr, col, val, 'cuda'), U, I, d, L, 1e-4, 0.005, 'cuda', table=torch.randn(U + I, d, device='cuda') * 0.05)
    bs = [(torch.randint(0, U, (B,), dtype=torch.int32, device='cuda'), torch.randint(0, I, (B,), dtype=torch.int32, device='cuda'),
           torch.randint(0, I, (B,), dtype=torch.int32, device='cuda')) for _ in range(8)]
    for k in range(5): eng.step(*bs[k % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300
    for k in range(n): eng.step(*bs[k % 8])
    torch.cuda.synchronize()
    print('%d x %d (nnz %d) d=%d L=%d B=%d: %.3f ms/step' % (U, I, len(col) // 2, d, L, B, 1e3 * (time.perf_counter() - t0) / n))
    # A/B in the same process: the three separate launches the fused set / clear kernels replace
    fused = (ops.batch_rows_set_, ops.batch_rows_clear_)
    def set3(G, flags, bits, idx, src, scale=1.0, check_range=True):
        ops.scatter_add_rows(G, idx, src, scale, check_range=False); ops.mark_rows_(flags, idx, 1, check_range=False); ops.mark_bits_(bits, idx, True, G.shape[0], check_range=False)
    def clear3(G, flags, bits, idx, check_range=True):
        ops.zero_rows_(G, idx, check_range=False); ops.mark_rows_(flags, idx, 0, check_range=False); ops.mark_bits_(bits, idx, False, G.shape[0], check_range=False)
    ops.batch_rows_set_, ops.batch_rows_clear_ = set3, clear3
    for k in range(5): eng.step(*bs[k % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n): eng.step(*bs[k % 8])
    torch.cuda.synchronize()
    print('    with the six separate launches: %.3f ms/step' % (1e3 * (time.perf_counter() - t0) / n))
    ops.batch_rows_set_, ops.batch_rows_clear_ = fused
