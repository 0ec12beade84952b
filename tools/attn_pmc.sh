#!/bin/bash
# PMC pass over the attention micro-benchmark (v=1: the round-2 kernel, PMHIP_ATTN_OLD=1): bash tools/attn_pmc.sh  (on the GPU box)
export TMPDIR=/tmp
for v in 0 1; do
  out=gpurun_out/attn_pmc_$v; rm -rf $out
  PMHIP_ATTN_OLD=$v rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out -o p -- python3 tools/attn_only.py > $out.log 2>&1
  PMHIP_ATTN_OLD=$v rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d ${out}b -o p -- python3 tools/attn_only.py > ${out}b.log 2>&1
  python3 - <<PY
import csv, collections, glob
for d in ("$out", "${out}b"):
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    agg = collections.defaultdict(float); n = set()
    for r in csv.DictReader(open(f[0])):
        if "attention" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    print("ATTN_OLD=$v", len(n), "dispatches", {k: round(v / len(n)) for k, v in agg.items()})
    if "GRBM_GUI_ACTIVE" in agg:
        print("  mfma_util", agg["SQ_VALU_MFMA_BUSY_CYCLES"] / (agg["GRBM_GUI_ACTIVE"] / 8 * 1024), "valu/mfma", agg["SQ_INSTS_VALU"] / agg["SQ_INSTS_MFMA"],
              "wait_any", agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"], "wait_inst", agg["SQ_WAIT_INST_ANY"] / agg["SQ_WAVE_CYCLES"])
PY
done
