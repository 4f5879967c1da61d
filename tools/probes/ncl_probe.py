"""Probe: NCL golden comparison errors with the panel form vs the fused all-rows kernels (prints every close() pair)."""
import sys, os, tempfile
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import test_gpu_api as T
from arlib_amd.recommender.NCL import _AllRowsNCE
from conftest import rel_err, row_err


def noisy_close(a, b, tol=None, row_tol=None):
    print('   close: max-norm %.3e row-wise %.3e' % (rel_err(a, b), row_err(a, b)))
    return True


class MP:
    def chdir(self, p):
        os.chdir(str(p))


T.close = noisy_close
for fused in (False, True):
    _AllRowsNCE.FUSED = fused
    print('FUSED =', fused)
    torch.manual_seed(20260)
    T.test_ncl_prototype_phase_step_matches_reference(tempfile.mkdtemp(), MP())
