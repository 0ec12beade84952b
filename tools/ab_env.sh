#!/bin/bash
# Same-box A/B of ONE environment switch on the whole bench, arms alternating:
#   gpurun -- 'bash tools/ab_env.sh PMHIP_ATTN_PLANES "1 0" [rounds] [extra bench args]'
# Every arm differs in the thing the conclusion is about: name the switch, list its values.
set -u
# (a development knob needs a development build first: PM_EXTRA_FLAGS=-DPM_DEV_KNOBS bash paintmind_amd/csrc/build.sh)
var=$1; vals=$2; rounds=${3:-3}; shift; shift; shift || true
for r in $(seq 1 $rounds); do
  for v in $vals; do
    env $var=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d.get('kernel_families', {})
print('$var=$v', d['value'], d['ms_per_step'], d['self_check'], ' '.join(f'{k} {round(x[\"ms\"], 2)}' for k, x in f.items() if x['launches']))"
  done
done
