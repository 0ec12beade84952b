#!/bin/bash
# Where does the folded consumers' overhead come from?  Three builds of the library on the GPU box, one timing script:
#   bash tools/ab_fold_consumer.sh
for v in "" "-DPM_FOLD_NOAPPLY" "-DPM_FOLD_DIRECT"; do
  PM_EXTRA_FLAGS="$v" bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
  echo "=== build flags: '$v'"
  python3 tools/fold_bench.py 2>&1 | grep -E "heads|swiglu|logits"
done
PM_EXTRA_FLAGS="" bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
