# per-kernel times of the all-rows InfoNCE (tools/ncl_bench.py) under rocprofv3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_nce
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/ncl_bench.py > $O/out.log 2> $O/err.log
cd $R
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
tail -2 $O/out.log
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
