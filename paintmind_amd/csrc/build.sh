#!/bin/bash
# Build libpaintmind_hip.so for gfx950 in-tree (cross-compiles without a GPU).
# An object is rebuilt when the CONTENT of its source, of any header, or the flags changed (sha256 kept beside the object;
# timestamps are not trusted: a checkout or a copy can make a stale object look newer than its source).  FORCE=1 rebuilds all.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libpaintmind_hip.so"
objdir="$here/build"
mkdir -p "$objdir"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${PM_EXTRA_FLAGS:-}"     # PM_EXTRA_FLAGS: ablation builds (tools/ab_*.sh)
units="gemm gemm256 gemm2b attention attention_bf16 attention_dh rowops vq sample loss engine"
hdrsum="$(cat "$here"/*.h "$here/../../include/pmhip.h" | sha256sum | cut -d' ' -f1)"
pids=(); built=0; reused=0
for f in $units; do
  src="$here/$f.hip"; obj="$objdir/$f.o"; stamp="$objdir/$f.sha"
  want="$(echo "$FLAGS $hdrsum $(sha256sum < "$src")" | sha256sum | cut -d' ' -f1)"
  if [ "${FORCE:-0}" != "0" ] || [ ! -f "$obj" ] || [ ! -f "$stamp" ] || [ "$(cat "$stamp")" != "$want" ]; then
    rm -f "$stamp"
    ( hipcc $FLAGS -c "$src" -o "$obj" && echo "$want" > "$stamp" ) &
    pids+=($!); built=$((built + 1))
  else
    reused=$((reused + 1))
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
[ $rc -eq 0 ] || { echo "build failed" >&2; exit 1; }
objs=""; for f in $units; do objs="$objs $objdir/$f.o"; done
# the link has its own stamp (hash of every object stamp), written only after a successful link: an interrupted or failed
# link is redone by the next run even though every object is up to date
linkwant="$(for f in $units; do cat "$objdir/$f.sha"; done | sha256sum | cut -d' ' -f1)"
if [ ! -f "$out" ] || [ ! -f "$objdir/link.sha" ] || [ "$(cat "$objdir/link.sha")" != "$linkwant" ]; then
  rm -f "$objdir/link.sha"
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs
  echo "$linkwant" > "$objdir/link.sha"
fi
echo "built $out (compiled $built, reused $reused objects)"
