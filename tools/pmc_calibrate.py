"""Calibration workload for FETCH_SIZE on the register-blocked SpMM's access pattern (one 256-B operand row per load instruction,
4 B per lane; MI355X_MICROARCH.md: widths other than 16 B/lane must be calibrated on a known byte count).  Every operand row is
gathered exactly once from a table far larger than the Infinity Cache, so one launch must fetch
    rows * 256 B (operand) + records * 8 B + wave tables
from memory.   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/pmc_calibrate.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arlib_amd import ops

dev = 'cuda:0'
n_out, deg, d = 131072, 32, 64
n_cols = n_out * deg                                                   # 4.19 M rows x 256 B = 1.07 GB
rng = np.random.default_rng(0)
col = rng.permutation(n_cols).astype(np.int32)
rowptr = np.arange(n_out + 1, dtype=np.int64) * deg
A = ops.CSRGraph(rowptr, col, np.ones(n_cols, np.float32), dev, n_cols=n_cols, validate=False).enable_blocked(col_block=4096)
X = torch.randn(n_cols, d, device=dev)
Y = torch.empty(n_out, d, device=dev)
for _ in range(5):
    ops.spmm(A, X, out=Y)
torch.cuda.synchronize()
st = A.blocked.sets[0]
known = n_cols * d * 4 + st['rec_col'].numel() * 8 + st['wave_rows'].numel() * 4 + st['wave_ptr'].numel() * 4
print('known fetch bytes per launch: %d (operand %d); write bytes: %d' % (known, n_cols * d * 4, n_out * d * 4))
