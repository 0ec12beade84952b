#!/bin/bash
# Same-box comparison of variants of attention.hip on the attention micro-benchmark (tools/attn_only.py):
#   gpurun -- 'bash tools/ab_attn.sh variantA variantB [...]'      (two rounds; the original is restored at the end)
set -u
target=paintmind_amd/csrc/attention.hip
cp "$target" /tmp/ab_original
for round in 1 2; do
  for src in "$@"; do
    cp "$src" "$target"
    bash paintmind_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed for $src"; continue; }
    echo "== $(basename $src)"; python tools/attn_only.py 2>/dev/null | tail -2
  done
done
cp /tmp/ab_original "$target"; bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
