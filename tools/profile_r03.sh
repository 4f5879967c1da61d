# round-3 evidence run: rocprofv3 kernel statistics of the default bench workload (training steps + attack legs)
set -x
R=$GRAFT_REPO_ROOT
TAG=${1:-r03_b}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline 0 --api-steps 0 > $O/stats_bench.json 2> $O/stats.err
cd $R
find gpurun_out/prof_$TAG -name "*.db" -delete
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head
