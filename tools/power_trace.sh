#!/bin/bash
# Samples socket power / clocks while a command runs (is the chip power-limited during the bench?):
#   gpurun -- 'bash tools/power_trace.sh gpurun_out/power.txt python bench.py --steps 40 --no-extra --no-cpu-baseline --no-roofline'
out=$1; shift
( while true; do
    echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E 'Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)' | tr -s ' ' | tr '\n' ';')"
    sleep 0.2
  done ) > $out 2>&1 &
sampler=$!
"$@"
rc=$?
kill $sampler 2>/dev/null
exit $rc
