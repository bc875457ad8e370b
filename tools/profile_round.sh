#!/bin/bash
# The round's rocprofv3 evidence in one GPU call (run through gpurun from the repo root):
#   tools/profile_round.sh r03
# writes gpurun_out/<round>_*.csv / *.log; copy what is to be judged into profiles/.  Counters in their own passes
# (--pmc never together with tracing other than --kernel-trace), the program itself after `--`, from /tmp with TMPDIR=/tmp.
set -e -o pipefail
R=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pick() { find "$1" -name "*_$2.csv" | head -1; }

rm -rf $OUT/prof_${R}_ns $OUT/prof_${R}_ws $OUT/pmc_${R}_w $OUT/pmc_${R}_f $OUT/pmc_${R}_pol $OUT/pmc_${R}_pol0
echo "[1/6] kernel stats, headline workload only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${R}_ns -- python3 $REPO/bench.py --steps 1000 --warmup 200 --no-cpu-baseline --no-secondary > $OUT/${R}_bench_under_rocprof.log 2>&1
cp "$(pick $OUT/prof_${R}_ns kernel_stats)" $OUT/${R}_kernel_stats.csv
echo "[2/6] kernel stats with the secondaries"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${R}_ws -- python3 $REPO/bench.py --steps 1000 --warmup 200 --no-cpu-baseline > $OUT/${R}_bench_under_rocprof_with_secondaries.log 2>&1
cp "$(pick $OUT/prof_${R}_ws kernel_stats)" $OUT/${R}_kernel_stats_with_secondaries.csv
echo "[3/6] WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${R}_w -- python3 $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary > $OUT/${R}_pmc_w.log 2>&1
cp "$(pick $OUT/pmc_${R}_w counter_collection)" $OUT/${R}_pmc_write_size.csv
echo "[4/6] FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${R}_f -- python3 $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary > $OUT/${R}_pmc_f.log 2>&1
cp "$(pick $OUT/pmc_${R}_f counter_collection)" $OUT/${R}_pmc_fetch_size.csv
echo "[5/6] policy kernel counters"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_${R}_pol -- python3 $REPO/tools/pmc_policy.py > $OUT/${R}_pmc_pol.log 2>&1
cp "$(pick $OUT/pmc_${R}_pol counter_collection)" $OUT/${R}_pmc_fused_policy_all.csv
cp "$(pick $OUT/pmc_${R}_pol kernel_trace)" $OUT/${R}_pmc_fused_policy_trace.csv
echo "[5b] policy kernel counters, one block per wavefront (OC_POLICY_DUO=0)"
export OC_POLICY_DUO=0
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_${R}_pol0 -- python3 $REPO/tools/pmc_policy.py > $OUT/${R}_pmc_pol0.log 2>&1
unset OC_POLICY_DUO
cp "$(pick $OUT/pmc_${R}_pol0 counter_collection)" $OUT/${R}_pmc_fused_policy_noduo_all.csv
cp "$(pick $OUT/pmc_${R}_pol0 kernel_trace)" $OUT/${R}_pmc_fused_policy_noduo_trace.csv
echo "[6/6] the plain bench line"
cd $REPO && python3 bench.py > $OUT/${R}_bench_n1.log 2> $OUT/${R}_bench_n1.err
echo done
