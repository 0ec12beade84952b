#!/bin/bash
# PMC passes over the attention ablation harness (tools/hwtests/attn_abl): per-variant counters, averaged per dispatch.
#   gpurun -- 'bash tools/attn_abl_pmc.sh'
export TMPDIR=/tmp
root=$PWD
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"
 "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES"
)
i=0
for s in "${sets[@]}"; do
  out=$root/gpurun_out/attn_abl_pmc_$i; rm -rf $out
  (cd /tmp && rocprofv3 --kernel-trace --pmc $s --output-format csv -d $out -o p -- $root/tools/hwtests/attn_abl "$@" > $out.log 2>&1)
  i=$((i+1))
done
python3 - <<PY
import csv, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for d in sorted(glob.glob("$root/gpurun_out/attn_abl_pmc_[0-9]")):
    for f in glob.glob(d + "/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(a\d+|old)::", r["Kernel_Name"]) or re.search(r"attention_kernel", r["Kernel_Name"])
            if not m: continue
            k = m.group(1) if m.lastindex else "old"
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
names = sorted({c for k in agg for c in agg[k]})
for k in sorted(agg, key=lambda x: (x != "old", int(x[1:]) if x != "old" else 0)):
    v = {c: agg[k][c] / max(1, len(nd[(k, c)])) for c in agg[k]}
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print("==", k)
    print("  " + "  ".join(f"{c.replace('SQ_', '')}={v[c] / wc:.3f}" for c in names if c in v and c.startswith("SQ_") and "INSTS" not in c and c != "SQ_WAVE_CYCLES"))
    print("  " + "  ".join(f"{c.replace('SQ_', '')}={v[c]:.3g}" for c in names if c in v and ("INSTS" in c or c in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"))))
    if "GRBM_GUI_ACTIVE" in v: print(f"  mfma_util={v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
PY
