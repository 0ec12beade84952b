#!/bin/bash
# Same-box comparison of variants of one source file on the WHOLE bench (box-to-box spread on the pool is +-3 %, larger than
# most kernel tweaks).  Run through gpurun so that all builds and measurements happen on ONE MI355X:
#   gpurun -- 'bash tools/ab_same_box.sh paintmind_amd/csrc/gemm_common.h variantA variantB [...]'
# Two rounds; rebuilds the library each time; prints images/s (3 lanes) and the GEMM / attention / LayerNorm family milliseconds.
set -u
target=$1; shift
cp "$target" /tmp/ab_original
for round in 1 2; do
  for src in "$@"; do
    cp "$src" "$target"
    bash paintmind_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed for $src"; continue; }
    python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['kernel_families']
print('$(basename $src)', d['value'], d['ms_per_step'], d['self_check'], 'gemm', f['gemm']['ms'], 'attention', f['attention']['ms'], 'ln', f['layernorm']['ms'])"
  done
done
cp /tmp/ab_original "$target"; bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
