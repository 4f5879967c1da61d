# round-4 evidence run: rocprofv3 kernel statistics of the default bench workload (training steps, attack legs, model legs, rank share).
#   gpurun -- 'bash tools/profile_r04.sh r04_g'   ->  gpurun_out/prof_r04_g/stats/*kernel_stats.csv, stats_bench.json
set -x
R=$GRAFT_REPO_ROOT
TAG=${1:-r04_g}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline 0 --api-steps 0 --model-steps 5 --share-steps 10 > $O/stats_bench.json 2> $O/stats.err
cd $R
find gpurun_out/prof_$TAG -name "*.db" -delete
# the CLeaR leg's launch sequence (which kernels run between the scoring pass and the SFA kernels): names in start order, first 3 steps
python3 - <<PY
import csv, glob
fs = glob.glob('gpurun_out/prof_$TAG/stats/**/*kernel_trace.csv', recursive=True)
rows = []
for f in fs:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
out = open('gpurun_out/prof_$TAG/clear_leg_sequence.txt', 'w')
seen = 0
for i, n in enumerate(names):
    if 'topk2_warm_kernel' in n:      # a warm-started scoring pass (second form of the stream) = one CLeaR step
        seen += 1
        if seen > 6:
            break
        out.write('---- step (warm scoring pass at launch %d)\n' % i)
        for m in names[i:i + 48]:
            out.write('  ' + m[:150] + '\n')
            if 'sfa_grad_kernel' in m:
                break
out.close()
PY
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head
