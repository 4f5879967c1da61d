#!/bin/bash
# A/B of score_mask_topk builds (make -C arlib_amd/csrc variant NAME=... DEFS=...): one cfg2-sized pass each, unmasked random tables
for v in "" $VARIANTS; do
  lib=arlib_amd/lib/libarlib_amd${v:+_$v}.so
  echo "== ${v:-default}"
  ARLIB_AMD_LIB=$lib U=${U:-1000000} I=${I:-100000} D=${D:-64} CMP=${CMP:-0} python3 tools/topk_bench.py
done
