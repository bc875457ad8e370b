#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in prev new; do
    if [ $lib = prev ]; then export OC_LIB_PATH=$PWD/tools/ab_libs/liboc_hip_prev.so; else unset OC_LIB_PATH; fi
    python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'value %.3f G' % (d['value']/1e9), 'ms/step %.5f' % d['ms_per_step'], 'kernel_ms_total %.4f' % d['kernel_ms_total'], 'host_fixed_us %.1f' % d['host_fixed_us'], 'us_per_launch %.3f' % d['roofline']['us_per_launch'])"
  done
done
