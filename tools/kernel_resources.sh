#!/bin/bash
# Per-kernel register / scratch / LDS usage of one translation unit (compile-time check for spills after a kernel edit):
#   bash tools/kernel_resources.sh attention_bf16 [extra hipcc flags]
set -eu
unit=$1; shift || true
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
tmp=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" --cuda-device-only -S "$here/paintmind_amd/csrc/$unit.hip" -o $tmp/k.s 2>/dev/null
python3 - $tmp/k.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) or [None, "?"])[1]
    print(f"{name[:90]:90s} vgpr {g('next_free_vgpr'):>4s} accum_off {g('accum_offset'):>4s} sgpr {g('next_free_sgpr'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
PY
grep -c "scratch_" $tmp/k.s | sed 's/^/scratch instructions in the unit: /'
rm -rf $tmp
