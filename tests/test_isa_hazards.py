"""Build gate for a gfx950 hardware hazard (tools/hwtests/pkfma_mfma.hip).

`v_pk_fma_f32` / `v_pk_mul_f32` / `v_pk_add_f32` whose LOW half takes SRC1 from the high register of its pair
(`op_sel:[_,1,...]`) lose that operand in lanes 48-63 when the other wave of the SIMD issues the first bf16 MFMA of a block in
the same cycles.  hipcc emits that form on its own (to broadcast a scalar that sits in an odd register), so the shipped
library is disassembled and every kernel is checked for it: no kernel of this library may contain the form, because every
kernel can share a SIMD with an MFMA kernel of another stream.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "paintmind_amd", "libpaintmind_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
VULNERABLE = re.compile(r"\bv_pk_(fma|mul|add)_f32\b.*\bop_sel:\[[01],1")


def device_disassembly(lib):
    """yield (code object index, disassembly text) for every gfx950 code object bundled into the shared library"""
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM binutils not present")
    objcopy, bundler, objdump = tools
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        assert starts, "no offload bundle in .hip_fatbin"
        for i, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
            piece = os.path.join(tmp, f"bundle{i}.bin")
            open(piece, "wb").write(blob[a:b])
            co = os.path.join(tmp, f"dev{i}.co")
            subprocess.run([bundler, "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--output={co}"], check=True, capture_output=True)
            if os.path.getsize(co) == 0:
                continue
            yield i, subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout


def test_no_packed_f32_instruction_takes_its_low_half_src1_from_a_high_register():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    bad, kernels, packed = [], 0, 0
    for i, text in device_disassembly(LIB):
        sym = "?"
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                sym = m.group(1)
                kernels += 1
                continue
            if "v_pk_" in line and "_f32" in line:
                packed += 1
                if VULNERABLE.search(line):
                    bad.append(f"code object {i} {sym}: {line.strip()}")
    assert kernels > 20 and packed > 1000, (kernels, packed)          # the scan really saw the library's kernels
    assert not bad, "gfx950 packed-f32 op_sel hazard form present:\n" + "\n".join(bad[:20])


def test_the_pattern_recognises_the_vulnerable_and_the_safe_forms():
    assert VULNERABLE.search("v_pk_fma_f32 v[16:17], v[86:87], v[92:93], v[90:91] op_sel:[0,1,0]")
    assert VULNERABLE.search("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]")
    assert VULNERABLE.search("v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1] op_sel_hi:[0,1]")
    assert not VULNERABLE.search("v_pk_fma_f32 v[98:99], v[92:93], v[150:151], v[16:17] op_sel_hi:[0,1,1]")
    assert not VULNERABLE.search("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,0,0]")
    assert not VULNERABLE.search("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]")
    assert not VULNERABLE.search("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]")
