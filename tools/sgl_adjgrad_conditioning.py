"""How well-conditioned is the quantity g26 pins?  SGL.train(requires_adjgrad=True), one epoch (22 Adam steps) on ml-100k, run twice by THIS library:
once from the seeded tables, once from the same tables with every entry moved by one part in 2^23 at random (one fp32 rounding).  The distance between the
two returned blocks / trained tables is what ANY fp32 implementation's rounding differences are amplified to by 22 Adam steps (g / sqrt(v) is scale-free:
an entry whose gradient sits at rounding level moves by up to lr either way), i.e. the floor under a comparison with the reference's own CPU run.
    python3 tools/sgl_adjgrad_conditioning.py"""
import contextlib, io, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import rel_err, row_err
from test_gpu_api import rec_args, make_data
from arlib_amd.util.tool import seedSet
from arlib_amd.recommender.SGL import SGL


def run(jitter, steps=None):
    seedSet(2018)
    rec = SGL(rec_args(emb_size=16, n_layers=2, model_name='SGL'), make_data())
    if jitter:
        gen = torch.Generator().manual_seed(jitter)
        with torch.no_grad():
            for p in rec.model.embedding_dict.values():
                sign = (torch.randint(0, 2, p.shape, generator=gen) * 2 - 1).to(p.device, p.dtype)
                p.mul_(1 + sign * 2.0 ** -23)
    rec.max_steps_per_epoch = steps
    with contextlib.redirect_stdout(io.StringIO()):
        blk = rec.train(requires_adjgrad=True, Epoch=1, gradIterationNum=10, evalNum=1)
    return blk.cpu().numpy(), rec.model.embedding_dict['user_emb'].detach().cpu().numpy(), rec.model.embedding_dict['item_emb'].detach().cpu().numpy()


for steps in (1, 5, 22):
    a = run(0, steps)
    for j in (1, 2):
        b = run(j, steps)
        print('steps %2d jitter %d: block max-norm %.2e row-wise %.2e | user table %.2e / %.2e | item table %.2e / %.2e' % (
            steps, j, rel_err(b[0], a[0]), row_err(b[0], a[0]), rel_err(b[1], a[1]), row_err(b[1], a[1]), rel_err(b[2], a[2]), row_err(b[2], a[2])), flush=True)
