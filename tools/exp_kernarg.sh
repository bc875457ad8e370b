# Where do kernel arguments live?  HIP_FORCE_DEV_KERNARG=0 (host memory) against 1 (device memory: this ROCm's default): bench.py and the
# policy kernel's phase trace under both.  Result (second session of round 3): device memory already is the default here -- host-side
# arguments cost the env launch 1.8 us and the policy prologue 3 us; nothing to gain from the variable.
#   gpurun -- 'bash tools/exp_kernarg.sh'   -> gpurun_out/r03b_kernarg_{0,1}.log, r03b_trace_kernarg_{0,1}.log
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do
  echo "=== HIP_FORCE_DEV_KERNARG=$v" 
  HIP_FORCE_DEV_KERNARG=$v python3 bench.py --no-cpu-baseline > gpurun_out/r03b_kernarg_$v.log 2>gpurun_out/r03b_kernarg_$v.err
  tail -1 gpurun_out/r03b_kernarg_$v.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['roofline']['frac'], {k:(v.get('us_per_step') or v.get('us_per_launch') if isinstance(v,dict) else v) for k,v in d.items() if isinstance(v,dict)})
"
  HIP_FORCE_DEV_KERNARG=$v python3 tools/trace_policy.py 5 4 245760 1024 3 > gpurun_out/r03b_trace_kernarg_$v.log 2>&1
  head -4 gpurun_out/r03b_trace_kernarg_$v.log
done
