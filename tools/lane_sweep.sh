#!/bin/bash
# development: throughput of the default workload for different micro-batch lane splits
for split in "22,21,21" "32,32" "24,20,20" "32,16,16" "16,16,16,16" "28,18,18" "40,24" "20,22,22"; do
  n=$(echo $split | tr ',' '\n' | wc -l)
  v=$(PM_BENCH_STREAMS=$n PM_BENCH_LANE_SPLIT=$split python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "split $split: $v img/s"
done
