cd $GRAFT_REPO_ROOT
for c in 16384 32768; do for w in 0 2048 4096; do
  echo "cold sample $c warm sample $w: $(ARL_TOPK2_BOOT_COLD=$c ARL_TOPK2_BOOT_WARM=$w KINDS=random,propagated python3 tools/topk_exit_bench.py 2>&1 | grep -v amdgpu | grep cold | sed 's/item-norm.*: cold/cold/; s/\[digest.*//' | tr '\n' '|')"
done; done
