#!/bin/bash
# Correctness + timing + PMC passes of the one-wave-per-SIMD attention kernel against the fourth generation (tools/hwtests/attn_w1).
#   gpurun -- 'bash tools/attn_w1_pmc.sh [B H N]'
export TMPDIR=/tmp
root=$PWD
$root/tools/hwtests/attn_w1 ${1:-64} ${2:-8} ${3:-1024} 6
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAVES"
)
i=0
for s in "${sets[@]}"; do
  out=$root/gpurun_out/attn_w1_pmc_$i; rm -rf $out
  (cd /tmp && rocprofv3 --kernel-trace --pmc $s --output-format csv -d $out -o p -- $root/tools/hwtests/attn_w1 ${1:-64} ${2:-8} ${3:-1024} 1 > $out.log 2>&1)
  i=$((i+1))
done
python3 - <<PY
import csv, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for d in sorted(glob.glob("$root/gpurun_out/attn_w1_pmc_[0-9]")):
    for f in glob.glob(d + "/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(g4|g5b|g5)::.*?(w1_kernel|bf16_kernel)", r["Kernel_Name"])
            if not m: continue
            k = m.group(1) + ":" + m.group(2)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
names = sorted({c for k in agg for c in agg[k]})
for k in sorted(agg):
    v = {c: agg[k][c] / max(1, len(nd[(k, c)])) for c in agg[k]}
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print("==", k)
    print("  " + "  ".join(f"{c.replace('SQ_', '')}={v[c] / wc:.3f}" for c in names if c in v and c.startswith("SQ_") and "INSTS" not in c and c not in ("SQ_WAVE_CYCLES", "SQ_WAVES")))
    print("  " + "  ".join(f"{c.replace('SQ_', '')}={v[c]:.4g}" for c in names if c in v and ("INSTS" in c or c in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES"))))
    if "GRBM_GUI_ACTIVE" in v:
        print(f"  mfma_util={v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):.4f}")
    if "SQ_INSTS_VALU" in v:
        print(f"  valu_per_mfma(non-matrix)={(v['SQ_INSTS_VALU'] - v['SQ_INSTS_MFMA']) / v['SQ_INSTS_MFMA']:.3f}")
PY
