#!/bin/bash
# Same-box comparison of builds that differ in compile-time flags (ablations, tuning constants):
#   gpurun -- 'bash tools/ab_flags.sh "python tools/producer_bench.py" "" "-DPM_HILO_DEPTH=2" ...'
# Each variant rebuilds the library with PM_EXTRA_FLAGS=<variant> and runs the command; two rounds; the default build is restored.
set -u
cmd=$1; shift
for round in 1 2; do
  for flags in "$@"; do
    PM_EXTRA_FLAGS="$flags" bash paintmind_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed for [$flags]"; continue; }
    echo "== [$flags]"
    eval "$cmd" 2>&1 | grep -v amdgpu.ids
  done
done
bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
