# per-kernel time of the CLeaR surrogate step (bench.py's clear_leg): rocprofv3 kernel statistics of a run that is almost only that leg
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_clear
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --api-steps 0 --l2-ceiling 0 --repeats 1 --attack-steps 1 --clear-steps 10 > $O/bench.json 2> $O/err.log
cd $R
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", tot / 1e6)
for r in rows[:40]:
    print("%-100s %5s calls %9.1f us avg %8.2f ms total" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
