#!/bin/bash
# GPU cycles (GRBM_GUI_ACTIVE / 8 XCDs) and MFMA busy per launch for every variant of an attn_w1 harness binary: wall time is bent by
# the power governor (an ablated kernel multiplies constants and clocks higher), cycles are not.
#   gpurun -- 'bash tools/attn_w1_cycles.sh attn_w1_abl [B H N]'
export TMPDIR=/tmp
root=$PWD
bin=$root/tools/hwtests/${1:-attn_w1}
out=$root/gpurun_out/attn_w1_cyc; rm -rf $out
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out -o p -- $bin ${2:-64} ${3:-8} ${4:-1024} 2 > $out.log 2>&1)
tail -12 $out.log
python3 - <<PY
import csv, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set); dur = collections.defaultdict(list)
for f in glob.glob("$out/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(g4|g5\w*)::.*?(w1_kernel|bf16_kernel)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(agg):
    v = {c: agg[k][c] / max(1, len(nd[(k, c)])) for c in agg[k]}
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    print(f"{k:14s} cycles/launch {cyc:9.0f}  mfma_util {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}  wait_inst/wave {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}  wait_any/wave {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}  valu_active/wave {v['SQ_ACTIVE_INST_VALU'] / v['SQ_WAVE_CYCLES']:.3f}  wave_cycles {v['SQ_WAVE_CYCLES']:.4g}")
PY
