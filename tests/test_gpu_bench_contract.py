"""bench.py's output contract on a small instance: one JSON line with the driver's fields, the roofline and cpu_baseline objects,
and the attack legs.  Keeps the contract from regressing while the benchmark itself runs at cfg2 size."""
import json
import os
import subprocess
import sys
import pytest
import torch
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract_small_instance():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--users', '30000', '--items', '3000', '--steps', '3', '--warmup', '1',
           '--attack-steps', '2', '--fake-users', '8', '--cpu-seconds', '0.5', '--api-steps', '5']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for key, typ in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int), ('ms_per_step', float),
                     ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str), ('config', dict)):
        assert isinstance(r[key], typ), key
    assert r['n_gpus'] == 1 and r['steps'] == 3 and r['warmup'] == 1 and r['vs_baseline'] is None and 'workload' in r['config']
    roof = r['roofline']
    assert roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and roof['peak'] == 8000.0
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-9 and 'traffic' in roof
    cb = r['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and 'sample' in cb
    assert cb['parity_vs_gpu']['table_rel_err'] < 1e-4                      # product vs oracle on the same steps
    assert r['attack']['value'] > 0 and r['attack']['cpu_baseline']['value'] > 0
    assert abs(r['attack']['cw_loss'] - r['attack']['cpu_baseline']['cw_loss']) <= 1e-4 * abs(r['attack']['cpu_baseline']['cw_loss'])
    assert r['attack_clear']['value'] > 0 and r['attack_dlattack_inner']['value'] > 0
    rep = r['repeat_ms_per_step']
    assert len(rep['regions']) == 3 and abs(rep['regions'][0] - r['ms_per_step']) < 1e-9 and rep['spread'] >= 0
    assert r['cpu_baseline_torch']['value'] > 0 and r['cpu_baseline_torch']['steps_timed'] >= 1      # reference-shaped stock-PyTorch step, on by default
    assert 'traffic_source' in roof
    assert r['class_api']['steps'] == 5 and r['class_api']['fused_engine'] and r['class_api']['ms_per_step'] > 0
    assert r['class_api']['epoch_wall_interactions_per_s'] > 0 and r['class_api']['epoch_wall']['serial_sampler_seconds'] > 0
    for leg in ('simgcl_step', 'ngcf_step'):                      # BASELINE configs 4 and 5 on one GPU
        assert r[leg]['value'] > 0 and r[leg]['algorithmic_bytes_per_step'] > 0 and 0 < r[leg]['hbm_frac'] < 1
    assert r['projected_ceiling_8gpu'] > 0 and r['rank_share_n8']['launches_per_step'] is None or r['rank_share_n8']['launches_per_step'] > 0      # one rank's share of the N = 8 step, alone
    sk = r['attack']['score_topk_pass']['stages_skipped_frac']
    assert all(0.0 <= x <= 1.0 for x in sk['trained_propagated_tables'] + sk['random_tables'])


def test_bench_gpus2_plain_invocation_self_launches():
    """`python bench.py --gpus 2` started the way the driver starts `--gpus 1` (no launcher, no WORLD_SIZE): bench.py starts its two ranks itself
    (child processes, before the parent touches the GPU) and prints the one JSON line with n_gpus 2 and the communication evidence.  On a 1-GPU
    box both ranks share cuda:0 over gloo (the documented test hooks); on a multi-GPU node the same invocation runs RCCL, one rank per GPU."""
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU')
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    single = torch.cuda.device_count() < 2
    if single:
        env['ARL_BENCH_SINGLE_DEVICE'] = '1'; env['ARL_BENCH_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--users', '30000', '--items', '3000', '--steps', '3', '--warmup', '1']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['steps'] == 3 and r['value'] > 0 and r['scaling'] == 'strong'
    c = r['comm']
    assert c['world_size'] == 2 and len(c['ranks']) == 2 and {x['rank'] for x in c['ranks']} == {0, 1}
    assert len({x['pid'] for x in c['ranks']}) == 2                                           # two processes
    assert c['allreduce_probe']['sum_of_rank_plus_1'] == c['allreduce_probe']['expected'] == 3.0
    assert c['backend'] == ('gloo' if single else 'nccl') and c['is_rccl'] == (not single)
    assert ('RCCL' in r['config']['parallelism']) == (not single)                             # the label follows the backend that ran
    assert c['distinct_devices'] == (1 if single else 2)
    assert 'projected_ceiling_8gpu' not in r or r['projected_ceiling_8gpu'] is None or r['projected_ceiling_8gpu'] > 0
