#!/bin/bash
# development: same-box sweep of the run-time tile-walk knobs (single-stream default workload, GEMM family ms)
run() { PM_BENCH_STREAMS=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], 'gemm', d['kernel_families']['gemm']['ms'])"; }
run "default(chunk256=6,chunk2b=12)"
for c in 3 4 8 12; do PMHIP_CHUNK256=$c run "chunk256=$c"; done
for c in 4 8 16; do PMHIP_CHUNK2B=$c run "chunk2b=$c"; done
PMHIP_GEMM2B=0 run "gemm2b=off"
run "default-again"
