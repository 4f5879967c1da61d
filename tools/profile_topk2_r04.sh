# round-4 evidence for the second form of score_mask_topk's stream:  gpurun -- 'bash tools/profile_topk2_r04.sh TAG'
# needs the profiling variant:  make -C arlib_amd/csrc variant NAME=t2prof DEFS=-DARL_TOPK2_PROF
R=$GRAFT_REPO_ROOT; T=${1:-r04_w}; O=$R/gpurun_out
cd $R
ARL_TOPK_FORM2=0 python3 tools/topk_exit_bench.py 2>&1 | grep -v amdgpu > $O/${T}_topk_first_form.txt
python3 tools/topk_exit_bench.py 2>&1 | grep -v amdgpu > $O/${T}_topk_second_form.txt
ARLIB_AMD_LIB=$R/arlib_amd/lib/libarlib_amd_t2prof.so python3 tools/topk2_prof.py 2>&1 | grep -v amdgpu > $O/${T}_topk2_section_timers.txt
bash tools/pmc_topk_r04.sh 2>&1 | tail -6 > $O/${T}_pmc_topk2.txt
python3 tools/topk_fuzz.py 400 2>&1 | tail -1 > $O/${T}_fuzz.txt
python3 tools/misc_fuzz.py 150 2>&1 | tail -1 >> $O/${T}_fuzz.txt
python3 tools/engine_fuzz.py 40 2>&1 | tail -1 >> $O/${T}_fuzz.txt
python3 tools/spmm_fuzz.py 150 2>&1 | tail -1 >> $O/${T}_fuzz.txt
for f in $O/${T}_topk_first_form.txt $O/${T}_topk_second_form.txt $O/${T}_fuzz.txt; do tail -4 $f; done
