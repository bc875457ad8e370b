#!/bin/bash
# Round 4's side measurements in one GPU call: logs under gpurun_out/ (copy what is to be judged into profiles/).
cd "$(dirname "$0")/.."
O=gpurun_out
f() { grep -v amdgpu.ids; }
python tools/ab_time_major.py cramped_room 65536 16,4,8,0,4 16,2,12,0,4 16,4,8,1,4 32,2,12,1,4 32,2,8,1,4 32,2,16,1,4 32,2,12,1,1 32,2,12,1,2 32,4,8,1,4 16,4,6,0,4 16,4,12,0,4 2>&1 | f > $O/r04_ab_fresh_sweep.log
python tools/trace_policy.py 2>&1 | f > $O/r04_trace_policy.log
python tools/bench_policy_layouts.py 2>&1 | f > $O/r04_bench_policy_layouts.log
python tools/exp_sampling_cost.py 2>&1 | f > $O/r04_sampling_cost.log
python tools/policy_accuracy.py 2>&1 | f > $O/r04_policy_accuracy.log
tools/probes/split_f16 2>&1 | f > $O/r04_probe_split_f16.log
python tools/bench_mixed_play.py 2>&1 | f > $O/r04_bench_mixed_play.log
python tools/bench_configs.py 2>&1 | f > $O/r04_bench_configs.log
echo done
