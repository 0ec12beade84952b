#!/bin/bash
# Re-collect every artefact under profiles/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh <tag>      e.g. r01_e
# Passes: (1) kernel-trace stats of the default command (3 micro-batch lanes) and of the single-stream run, (2) FETCH_SIZE, (3) WRITE_SIZE, (4) SQ utilisation counters, (5) plain bench line.
# PMC passes are separate runs with --kernel-trace only, as the pool requires.
set -u
tag=${1:-r01_x}
export TMPDIR=/tmp
out=gpurun_out/$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $out/stats.log 2>&1
# one lane AND no side-stream decode (PMHIP_DECODE_OVERLAP_MAX_ROWS=0): a single lane otherwise runs each step's ViT decode beside the next
# step's tower (round 5), and kernels that share the chip report inflated durations (r06_z: attention 168 instead of 132 us per launch)
PM_BENCH_STREAMS=1 PMHIP_DECODE_OVERLAP_MAX_ROWS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $out/stats1.log 2>&1
PM_BENCH_NO_GRAPH=1 PM_BENCH_STREAMS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $out/pmc_fetch.log 2>&1
PM_BENCH_NO_GRAPH=1 PM_BENCH_STREAMS=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $out/pmc_write.log 2>&1
PM_BENCH_NO_GRAPH=1 PM_BENCH_STREAMS=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_util -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $out/pmc_util.log 2>&1
find $out -name "*.csv" | head -20
# keep only what the aggregation needs (the raw traces are large)
for d in pmc_fetch pmc_write pmc_util; do f=$(find $out/$d -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && mv "$f" $out/$d/p_counter_collection.csv; done
f=$(find $out/stats -name "r_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats.csv
f=$(find $out/stats1 -name "r_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats_single_stream.csv
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json
python3 tools/pmc_util.py $out/pmc_util $out/pmc_mfma_util.json
du -sh $out
