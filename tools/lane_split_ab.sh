#!/bin/bash
# Same-box A/B of the micro-batch lane split and the decode-overlap row limit, through both timing harnesses of bench.py
# (gpurun -- 'bash tools/lane_split_ab.sh'): "main" = the paced headline loop (python bench.py --workload W), "extra" = the
# free-running loop the secondary workloads are timed with (bench.extra_workload).  PM_BENCH_LANE_SPLIT="" = the library's default.
main() { wl=$1; cfg=$2; ov=$3; PMHIP_DECODE_OVERLAP_MAX_ROWS=$ov PM_BENCH_STREAMS=2 PM_BENCH_LANE_SPLIT=$cfg python bench.py --workload $wl --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-extra --no-roofline > gpurun_out/ls.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/ls.json").read().strip().splitlines()[-1])
print("main  $wl split=[$cfg] overlap_rows=$ov", d["value"], d["ms_per_step"], d["self_check"])
PY
}
extra() { wl=$1; cfg=$2; ov=$3; PMHIP_DECODE_OVERLAP_MAX_ROWS=$ov PM_BENCH_STREAMS=2 PM_BENCH_LANE_SPLIT=$cfg python - <<PY 2>/dev/null
import torch, bench
r = bench.extra_workload("$wl", "bf16", ${STEPS:-6}, torch.device("cuda:0"))
print("extra $wl split=[$cfg] overlap_rows=$ov", r["images_per_s"], r["ms_per_step"], r["self_check"])
PY
}
for rep in 1 2; do
for wl in maskgit-text-24L-d768-T8 paintmindv1-T18 maskgit-text-24L-d768-T12-cfg3 maskgit-uncond-12L-d512-T8; do
b=17,15; [ $wl = maskgit-uncond-12L-d512-T8 ] && b=33,31
main $wl "" 16384; main $wl "$b" 16384; main $wl "" 0
done
done
