#!/bin/bash
# Round 6: launch geometry of the per-step kernel at the batch sizes an 8-GPU job runs per GPU (config 3-5: 8192 kitchens per layout),
# including multi-wavefront workgroups (OC_WG_WAVES: diagnostics build).  Writes gpurun_out/sweep_small_<layout>_<n>_wg<w>.log
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
for lay in cramped_room asymmetric_advantages coordination_ring counter_circuit; do
  for n in 4096 8192 16384; do
    for w in 1 2 4 8; do
      OC_WG_WAVES=$w python tools/sweep_tuning.py --layout $lay --kitchens $n --kpw 2,4,8,16 --passes 2 --wpc 8 --steps 300 2>&1 | grep us/launch | sed "s/^/$lay $n wg=$w /"
    done
  done
done
