#!/bin/bash
for c in 4 6 8 11 4; do
  echo -n "chunk=$c "
  PMHIP_CHUNK256=$c PM_BENCH_STREAMS=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_families']['gemm'])"
done
