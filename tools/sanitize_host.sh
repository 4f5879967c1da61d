#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side native code: the host part of the product library (bit-exact sampler, MT19937,
# random.sample) and the oracle's C restatement.  GPU ASan is not available on this pool; this covers everything that runs on
# the host.   bash tools/sanitize_host.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/arl_asan
mkdir -p "$OUT"
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -I"$ROOT/include" \
    -o "$OUT/libarl_host_asan.so" "$ROOT/arlib_amd/csrc/arl_host.cpp"
gcc -O1 -g -fPIC -shared -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -o "$OUT/liboracle_asan.so" "$ROOT/oracle/arl_oracle.c" -lm
ASAN=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 ARL_ASAN_DIR=$OUT python3 "$ROOT/tools/sanitize_host.py"
