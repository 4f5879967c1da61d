set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r02
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline 0 --api-steps 0 > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-baseline 0 --attack-steps 0 --api-steps 0 --no-kernel-events --repeats 1 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-baseline 0 --attack-steps 0 --api-steps 0 --no-kernel-events --repeats 1 > /dev/null 2> $O/write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/l2 -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-baseline 0 --attack-steps 0 --api-steps 0 --no-kernel-events --repeats 1 > /dev/null 2> $O/l2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib -- python3 $R/tools/pmc_calibrate.py > $O/calib.out 2> $O/calib.err
cd $R
find gpurun_out/prof_r02 -name "*.csv" | head -30
find gpurun_out/prof_r02 -name "*.db" -delete
du -sh gpurun_out/prof_r02
