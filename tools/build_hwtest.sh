#!/bin/bash
# build a hardware test of tools/hwtests in place:  bash tools/build_hwtest.sh attn_w1 [output name] [extra hipcc flags]
set -eu
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
name=$1; out=${2:-$1}; shift; shift || true
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I "$root/paintmind_amd/csrc" "$@" -o "$root/tools/hwtests/$out" "$root/tools/hwtests/$name.hip" 2>&1 | grep -E "error" -A6 || true
ls -la "$root/tools/hwtests/$out" | awk '{print $5, $9}'
