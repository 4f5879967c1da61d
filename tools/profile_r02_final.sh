# end-of-round-2 evidence: GPU suite, the default bench line, the rocprofv3 kernel statistics of the same workload, the SimGCL step,
# and the `--gpus 2` code path end to end on one GPU (gloo, both ranks on cuda:0: a functional check, not a measurement)
set -x
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err
python3 tools/simgcl_bench.py 2>&1 | tail -1
ARL_BENCH_BACKEND=gloo ARL_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/r02_final_gloo2.json 2> gpurun_out/r02_final_gloo2.err; echo "gloo2 rc=$?"; tail -c 600 gpurun_out/r02_final_gloo2.json
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r02_final
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline 0 --api-steps 0 > $O/stats_bench.json 2> $O/stats.err
cd $R
find gpurun_out/prof_r02_final -name "*.db" -delete
find gpurun_out/prof_r02_final -name "*kernel_trace.csv" -delete
find gpurun_out/prof_r02_final -name "*kernel_stats.csv" | head
