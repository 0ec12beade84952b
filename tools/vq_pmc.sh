#!/bin/bash
# PMC pass over tools/vq_bench.py (the vq_scan kernel): gpurun -- 'bash tools/vq_pmc.sh'
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/vq_pmc; rm -rf $out ${out}b
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out -o p -- python3 $root/tools/vq_bench.py > $out.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d ${out}b -o p -- python3 $root/tools/vq_bench.py > ${out}b.log 2>&1)
python3 - <<PY
import csv, collections, glob
for d in ("$out", "${out}b"):
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    agg = collections.defaultdict(float); n = set()
    for r in csv.DictReader(open(f[0])):
        if "vq_scan" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    print(len(n), "dispatches", {k: round(v / len(n)) for k, v in agg.items()})
    if "GRBM_GUI_ACTIVE" in agg:
        print("  mfma_busy/gui", agg["SQ_VALU_MFMA_BUSY_CYCLES"] / (agg["GRBM_GUI_ACTIVE"] / 8 * 1024), "valu/mfma", agg["SQ_INSTS_VALU"] / agg["SQ_INSTS_MFMA"])
    kt = glob.glob(d + "/**/p_kernel_trace.csv", recursive=True)
    du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0])) if "vq_scan" in r["Kernel_Name"]]
    print("  vq_scan median us", sorted(du)[len(du) // 2], "GUI cycles/8/us", agg.get("GRBM_GUI_ACTIVE", 0) / max(1, len(n)) / 8 / sorted(du)[len(du) // 2] / 1e3 if du else None)
PY
