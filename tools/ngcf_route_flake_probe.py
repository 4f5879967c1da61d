import sys, io, contextlib, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_api as T
from arlib_amd.util.tool import seedSet
from arlib_amd.recommender.NGCF import NGCF
def run(fused):
    seedSet(2018)
    rec = NGCF(T.rec_args(emb_size=32, n_layers=2, model_name='NGCF'), T.make_data())
    with contextlib.redirect_stdout(io.StringIO()):
        rec.train(Epoch=1, evalNum=1, requires_embgrad=not fused)
    return [rec.model.embedding_dict[k].detach().cpu().numpy().copy() for k in ('user_emb', 'item_emb')] + [rec.model.W['w1_1'].detach().cpu().numpy().copy()]
# the predecessors of the route test in tests/test_gpu_api.py (same process, same allocator state as under pytest)
for g, e, l in (('g9_ngcf.npz', 32, 2), ('g9_ngcf128.npz', 128, 3)):
    T.test_ngcf_forward_and_steps_match_reference(g, e, l)
for g, e, l in (('g9_ngcf.npz', 32, 2), ('g9_ngcf128.npz', 128, 3)):
    T.test_ngcf_fused_engine_step_matches_reference(g, e, l)
base_f = run(True); base_a = run(False)
for it in range(2):
    f = run(True); a = run(False)
    out = []
    for x, y, bf, ba in zip(f, a, base_f, base_a):
        e = np.abs(x - y); m = np.abs(y).max()
        ea = np.abs(y - ba)
        out.append('fa %.1e (n>1e-4: %d)  ff %.1e  aa %.1e (n>1e-4: %d of %d, max abs %.1e, rows hit %d)' % (e.max() / m, int((e > 1e-4 * m).sum()), np.abs(x - bf).max() / m, ea.max() / m, int((ea > 1e-4 * m).sum()), ea.size, ea.max(), int((ea.max(1) > 1e-4 * m).sum())))
    print(it, ' | '.join(out), flush=True)
