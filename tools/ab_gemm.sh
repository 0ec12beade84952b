#!/bin/bash
# Same-box comparison of variants of one source file on the GEMM micro-benchmark (tools/gemm_bench.py):
#   gpurun -- 'bash tools/ab_gemm.sh paintmind_amd/csrc/gemm256.hip variantA variantB [variantC ...]'
# Two rounds over all variants; the original file is restored and rebuilt at the end.
set -u
target=$1; shift
cp "$target" /tmp/ab_original
for round in 1 2; do
  for src in "$@"; do
    cp "$src" "$target"
    bash paintmind_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed for $src"; continue; }
    echo "== $(basename $src)"; python tools/gemm_bench.py 2>/dev/null | grep "^gemm" | awk '{print $2, $3, $4, $5, $6, $7, $8, $9, $10}'
  done
done
cp /tmp/ab_original "$target"; bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
