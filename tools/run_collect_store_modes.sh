#!/bin/bash
# the store mode of fresh slots inside a rollout with the policies in the loop, re-measured with round 4's policy kernel
echo "## default (the hint: plain stores into fresh slots during policy rollouts)"
python tools/exp_collect_store_mode.py 2>&1 | grep -v amdgpu.ids
for m in 2,4 2,2 2,1 2,3; do
  echo "## OC_PLAIN_FRESH=0 OC_STORE_MODE=$m"
  OC_PLAIN_FRESH=0 OC_STORE_MODE=$m python tools/exp_collect_store_mode.py 2>&1 | grep -v amdgpu.ids
done
