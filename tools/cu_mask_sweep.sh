#!/bin/bash
# EXPERIMENT: concurrent lanes on CU-masked streams (PMHIP_LANE_CU_MASK, tools/cu_mask_lanes.py) against the default lanes,
# same box, whole bench.   gpurun -- 'bash tools/cu_mask_sweep.sh'
set -u
run() {
  echo -n "$* : "
  env "$@" python tools/cu_mask_lanes.py --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['self_check'])"
}
for round in 1 2; do
  run PM_BENCH_STREAMS=2
  run PM_BENCH_STREAMS=2 PMHIP_LANE_CU_MASK=1 PM_BENCH_LANE_SPLIT=32,32
  run PM_BENCH_STREAMS=2 PMHIP_LANE_CU_MASK=1 PM_BENCH_LANE_SPLIT=32,32 PMHIP_PERSIST256=128
  run PM_BENCH_STREAMS=2 PMHIP_LANE_CU_MASK=1 PM_BENCH_LANE_SPLIT=33,31 PMHIP_PERSIST256=128
  run PM_BENCH_STREAMS=4 PMHIP_LANE_CU_MASK=1 PM_BENCH_LANE_SPLIT=16,16,16,16 PMHIP_PERSIST256=64
  run PM_BENCH_STREAMS=3 PM_BENCH_LANE_SPLIT=22,21,21
done
