#!/bin/bash
# Build libpaintmind_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libpaintmind_hip.so"
objdir="$here/build"
mkdir -p "$objdir"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
for f in gemm gemm256 gemm2b attention rowops vq sample loss engine; do
  src="$here/$f.hip"; obj="$objdir/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$here/common.h" -nt "$obj" ] || [ "$here/gemm_common.h" -nt "$obj" ] || [ "$here/../../include/pmhip.h" -nt "$obj" ]; then
    hipcc $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$objdir"/{gemm,gemm256,gemm2b,attention,rowops,vq,sample,loss,engine}.o
echo "built $out"
