# per-kernel time of the CLeaR leg alone (short bench run under rocprofv3):  gpurun -- 'bash tools/profile_clear_r04.sh TAG'
R=$GRAFT_REPO_ROOT; TAG=${1:-r04_i}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-baseline 0 --api-steps 0 --model-steps 0 --share-steps 0 --attack-steps 1 --clear-steps 15 > $O/bench.json 2> $O/stats.err
cd $R
find gpurun_out/prof_$TAG -name "*.db" -delete; find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/prof_$TAG/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if any(t in n for t in ('cw_', 'sfa_', 'score_mask', 'stage_', 'adam', 'spmm_blocked')) or float(r['Percentage']) > 0.5:
        print('%5s x %9.1f us = %8.2f ms  %s' % (r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, n[:100]))
PY
