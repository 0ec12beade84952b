#!/bin/bash
# same-box A/B of attention_bf16.hip variants on the self- and cross-attention micro-benchmarks + the whole bench
#   gpurun -- 'bash tools/ab_attn_micro.sh variantA.hip variantB.hip'
set -u
target=paintmind_amd/csrc/attention_bf16.hip
cp "$target" /tmp/ab_original
for round in 1 2; do
  for src in "$@"; do
    cp "$src" "$target"
    bash paintmind_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed for $src"; continue; }
    echo "== $(basename $src)"
    python tools/xattn_bench.py 2>&1 | grep -E "Nkv=(1024|77)\b"
    python tools/attn_only.py 2>&1 | grep -E "^B=64"
    python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['kernel_families']
print('bench', d['value'], d['ms_per_step'], d['self_check'], 'attention', f['attention']['ms'])"
  done
done
cp /tmp/ab_original "$target"; bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
