# round-4 refresh of roofline.traffic: FETCH_SIZE / WRITE_SIZE / L2 hit passes (each alone with --kernel-trace) + the calibration pass
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_traffic_r04
rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --steps 4 --warmup 1 --cpu-baseline 0 --attack-steps 0 --api-steps 0 --l2-ceiling 0 --no-kernel-events --repeats 1 --model-steps 0 --share-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > /dev/null 2> $O/write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/l2 -- $B > /dev/null 2> $O/l2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib -- python3 $R/tools/pmc_calibrate.py > $O/calib.out 2> $O/calib.err
cd $R
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
known=$(grep -o "known fetch bytes per launch: [0-9]*" $O/calib.out | grep -o "[0-9]*$")
python3 tools/pmc_traffic.py $O/fetch $O/write $O/calib $known $O/r04_pmc_traffic_blocked | tail -5
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob('gpurun_out/pmc_traffic_r04/l2/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spmm' in r['Kernel_Name']:
            a = acc[r['Kernel_Name']]
            if r['Counter_Name'] == 'TCC_HIT_sum': a[0] += float(r['Counter_Value']); a[2] += 1
            if r['Counter_Name'] == 'TCC_MISS_sum': a[1] += float(r['Counter_Value'])
with open('gpurun_out/pmc_traffic_r04/r04_pmc_l2_hit_per_kernel.csv', 'w') as fh:
    fh.write('kernel,launches,TCC_HIT_sum_avg,TCC_MISS_sum_avg,hit_rate,miss_x128B_GB\n')
    for k, (h, m, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if n: fh.write('"%s",%d,%.0f,%.0f,%.4f,%.3f\n' % (k, n, h / n, m / n, h / max(h + m, 1.0), m / n * 128 / 1e9))
print(open('gpurun_out/pmc_traffic_r04/r04_pmc_l2_hit_per_kernel.csv').read()[:1500])
PY
