#!/bin/bash
# can the HIP-runtime helper thread that busy-polls while GPU work is outstanding be put to sleep?
# Each variant: the default bench (headline only), reading images/s and the host-CPU fields of the JSON line.
#   gpurun -- 'bash tools/host_poll_probe.sh'
run() {
  label=$1; shift
  env "$@" python bench.py --steps 10 --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
th = ', '.join(f\"{t['thread']} {t['cpu_ms_per_step']}\" for t in d.get('host_cpu_ms_per_step_by_thread', []))
print('$label |', d['value'], 'img/s |', d['ms_per_step'], 'ms/step | host cpu', d.get('host_cpu_ms_per_step'), 'ms/step =', d.get('host_cpu_fraction_of_one_core'), 'cores | threads:', th)"
}
run "default                         " PM_X=1
run "HSA_ENABLE_INTERRUPT=1          " HSA_ENABLE_INTERRUPT=1
run "HSA_ENABLE_INTERRUPT=0          " HSA_ENABLE_INTERRUPT=0
run "PM_BENCH_BLOCKING_SYNC=1        " PM_BENCH_BLOCKING_SYNC=1
run "GPU_MAX_HW_QUEUES=2             " GPU_MAX_HW_QUEUES=2
run "HIP_FORCE_DEV_KERNARG=1         " HIP_FORCE_DEV_KERNARG=1
run "AMD_DIRECT_DISPATCH=0           " AMD_DIRECT_DISPATCH=0
run "BLOCKING_SYNC + DIRECT_DISPATCH=0" PM_BENCH_BLOCKING_SYNC=1 AMD_DIRECT_DISPATCH=0
run "HSA_ENABLE_SDMA=0               " HSA_ENABLE_SDMA=0
run "default again                   " PM_X=1
