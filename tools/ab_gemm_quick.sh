#!/bin/bash
# like ab_gemm.sh, one round, three shapes (ablations: results are wrong by design, only the time matters)
set -u
target=$1; shift
cp "$target" /tmp/ab_original
for src in "$@"; do
  cp "$src" "$target"
  bash paintmind_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed for $src"; continue; }
  echo "== $(basename $src)"; python - <<'PY'
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import gemm_bench as g, torch
g.gemm(8192, 8192, 8192); g.gemm(65536, 1536, 512); g.gemm(65536, 2816, 512)
PY
done
cp /tmp/ab_original "$target"; bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
