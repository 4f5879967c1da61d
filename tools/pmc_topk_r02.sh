set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_topk_r02
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O -- python3 $R/tools/topk_bench.py > $O/out.txt 2> $O/err.txt
cd $R
find gpurun_out/pmc_topk_r02 -name "*.db" -delete
find gpurun_out/pmc_topk_r02 -name "*counter_collection.csv" | head -3
tail -2 $O/out.txt
