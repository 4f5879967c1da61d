# the rank-share leg alone + the sharded engine's GPU tests:  gpurun -- 'bash tools/probes/share_check.sh TAG'
R=$GRAFT_REPO_ROOT; T=${1:-share}
cd $R
python -m pytest tests/test_gpu_dist.py tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --cpu-baseline 0 --api-steps 0 --model-steps 0 --attack-steps 0 --clear-steps 0 --share-steps 30 > gpurun_out/${T}_share.json 2>/dev/null
python - <<PY
import json
r = json.load(open('gpurun_out/${T}_share.json')); s = r['rank_share_n8']
print('share busy %.4f ms, %s launches, span %.4f ms; t_1gpu %.3f ms; ceiling %.3f' % (s['gpu_busy_ms_per_step'], s['launches_per_step'], s['event_span_ms_per_step'], r['ms_per_step'], r['projected_ceiling_8gpu']))
PY
