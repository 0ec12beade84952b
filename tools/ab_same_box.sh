#!/bin/bash
# Same-box A/B of two variants of one source file (box-to-box spread on the pool is +-3 %, larger than most kernel
# tweaks).  Run through gpurun so that both builds and all measurements happen on ONE MI355X:
#   gpurun -- 'bash tools/ab_same_box.sh paintmind_amd/csrc/gemm_common.h /path/in/repo/variantA /path/in/repo/variantB'
# Alternates A B A B, rebuilds the library each time, prints images/s and the GEMM / attention family milliseconds of
# the single-stream default workload.
set -u
target=$1; a=$2; b=$3
cp "$target" /tmp/ab_original
run() {
  PM_BENCH_STREAMS=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['kernel_families']
print('$1', d['value'], 'gemm', f['gemm']['ms'], 'attention', f['attention']['ms'])"
}
for v in A B A B; do
  src=$a; [ $v = B ] && src=$b
  cp "$src" "$target"; touch "$target"
  bash paintmind_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed for $v"; break; }
  run $v
done
cp /tmp/ab_original "$target"; touch "$target"; bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
