#!/bin/bash
cd /root/repo
timeout 300 python3 -m pytest tests -x -q -m gpu -k "topk or attack" 2>&1 | tail -3
K=50 U=400000 python3 tools/topk_bench.py 2>&1 | tail -1
K=20 U=400000 python3 tools/topk_bench.py 2>&1 | tail -1
cp arlib_amd/lib/libarlib_amd.so /tmp/orig.so; cp arlib_amd/lib/libarlib_amd_prof.so arlib_amd/lib/libarlib_amd.so
K=50 python3 tools/_prof_topk.py 2>&1 | tail -1
K=20 python3 tools/_prof_topk.py 2>&1 | tail -1
cp /tmp/orig.so arlib_amd/lib/libarlib_amd.so
