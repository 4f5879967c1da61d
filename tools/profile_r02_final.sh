# end-of-round-2 evidence: GPU suite, the default bench line, and the rocprofv3 kernel statistics of the same workload
set -x
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r02_final
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline 0 --api-steps 0 > $O/stats_bench.json 2> $O/stats.err
cd $R
find gpurun_out/prof_r02_final -name "*.db" -delete
find gpurun_out/prof_r02_final -name "*kernel_stats.csv" | head
du -sh gpurun_out/prof_r02_final
