#!/bin/bash
# like ab_gemm.sh for ablations / epilogue variants: two rounds, the bench-shaped QKV (head split), SwiGLU and plain GEMMs
#   bash tools/ab_gemm_quick.sh <target file> variant...
set -u
target=$1; shift
cp "$target" /tmp/ab_original
for round in 1 2; do
for src in "$@"; do
  cp "$src" "$target"
  bash paintmind_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed for $src"; continue; }
  echo "== $(basename $src)"; python - <<'PY'
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import gemm_bench as g, torch
from paintmind_amd import ops, _lib
import ctypes as C
g.gemm(8192, 8192, 8192); g.gemm(65536, 1536, 512); g.gemm(65536, 2816, 512)
dev = torch.device("cuda:0")
a = (torch.rand(65536, 512, device=dev) * 2 - 1).to(torch.bfloat16)
w = (torch.rand(1536, 512, device=dev) * 2 - 1).to(torch.bfloat16)
lib = _lib.load()
q = torch.empty(64, 8, 1024, 64, device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros_like(q)
kinds = (C.c_int * 3)(0, 1, 2); outs = (C.c_void_p * 3)(q.data_ptr(), k.data_ptr(), vt.data_ptr())
def heads():
    lib.pmhip_gemm_heads(1, a.data_ptr(), 512, w.data_ptr(), 512, 65536, 512, 8, 1024, 1024, 3, kinds, outs, 0.125, ops.stream_ptr(dev))
print(f"heads qkv 65536x1536x512: {g.timeit(heads)*1e3:.1f} us")
w2 = (torch.rand(2816, 512, device=dev) * 2 - 1).to(torch.bfloat16)
b2 = torch.rand(2816, device=dev)
print(f"swiglu 65536x2816x512: {g.timeit(lambda: ops.gemm_swiglu(a, w2, b2))*1e3:.1f} us")
PY
done
done
cp /tmp/ab_original "$target"; bash paintmind_amd/csrc/build.sh > /dev/null 2>&1
