# per-kernel times of one cfg2-sized score_mask_topk bench (second form) under rocprofv3:  gpurun -- 'bash tools/probes/run_t2.sh [variant ...]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf /tmp/prof_$v
  if [ "$v" = default ]; then unset ARLIB_AMD_LIB; else export ARLIB_AMD_LIB=$R/arlib_amd/lib/libarlib_amd_$v.so; fi
  KINDS=${KINDS:-random} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/tools/topk_exit_bench.py > /tmp/prof_$v.txt 2>&1
  echo "== $v: $(grep -v amdgpu /tmp/prof_$v.txt | grep -m1 'cold')"
  V=$v python3 - <<'PY'
import csv, glob, os
f = glob.glob('/tmp/prof_%s/**/*kernel_trace.csv' % os.environ['V'], recursive=True)
rows = list(csv.DictReader(open(f[0])))
import collections
d = collections.defaultdict(list)
for r in rows:
    nm = r['Kernel_Name']
    if "topk" in nm:
        d[nm.split('(')[0][-60:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in d.items():
    print('   %-62s %s' % (k, ' '.join('%.2f' % x for x in v)))
PY
done
