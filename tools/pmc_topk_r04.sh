#!/bin/bash
# round 4 (second form of the stream, topk2_main_kernel): rocprofv3 counter passes over one cfg2-sized score_mask_topk pass (tools/topk_bench.py); counters only with --kernel-trace (gpurun rule)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_topk_r04
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/counters.txt | sort -u > $O/sq_names.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  i=$((i+1))
  U=${U:-1000000} CMP=0 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/topk_bench.py > $O/out$i.txt 2> $O/err$i.txt
  tail -1 $O/out$i.txt
done
cd $R
find gpurun_out/pmc_topk_r04 -name "*.db" -delete
python3 - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob('gpurun_out/pmc_topk_r04/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(p)):
        if 'topk2_main' in r['Kernel_Name'] and int(r['Grid_Size']) > 2000000:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
    print(p.split('/')[2], dict(acc))
PY
