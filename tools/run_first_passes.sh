#!/bin/bash
# A/B of the shorter first flush group (OC_FIRST_PASSES, diagnostics build): in place and into time-major slots, four layouts
set -e
for l in cramped_room coordination_ring counter_circuit asymmetric_advantages; do
  python tools/ab_time_major.py $l 65536 16,4,8,0,4 16,4,8,0,4,1 16,4,8,0,4,2 16,4,8,0,4,3
done
python tools/ab_geometry.py cramped_room 65536 32,2,12,1 32,2,12,1,1 32,4,12,1,2 32,3,12,1,1 16,2,12,1,1 16,2,12,1
python tools/ab_geometry.py coordination_ring 65536 16,2,12 16,2,12,0,1 16,4,12,0,2
python tools/ab_geometry.py asymmetric_advantages 65536 16,2,12 16,2,12,0,1 16,4,12,0,2
