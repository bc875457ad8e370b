#!/bin/bash
# Timing-only ablations of the duo policy kernel's position loop (OC_DUO_ABLATE, policy_duo_body.h): builds diagnostics libraries
# ab_libs/liboc_duo_<mask>.so for the given masks (HERE, no GPU), to be timed side by side on one box with
#   tools/ab_builds.sh duo "a0=ab_libs/liboc_duo_0.so a1=ab_libs/liboc_duo_1.so ..." tools/exp_policy_stagger.py   (OC_DIAGNOSTICS=1)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
C=$REPO/diverse-conventions_amd/csrc
mkdir -p $REPO/ab_libs $REPO/build/duo
FLAGS="--offload-arch=gfx950 -O3 -DOC_DIAGNOSTICS -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -fPIC -I $REPO/include"
( hipcc $FLAGS -c $C/overcooked_engine.hip -o $REPO/build/duo/engine.o ) &
( hipcc $FLAGS -c $C/rollout_ops.hip -o $REPO/build/duo/rollout.o ) &
for m in "$@"; do
  ( hipcc $FLAGS -DOC_DUO_ABLATE=$m -c $C/policy_ops.hip -o $REPO/build/duo/policy_$m.o ) &
done
wait
for m in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/ab_libs/liboc_duo_$m.so $REPO/build/duo/engine.o $REPO/build/duo/rollout.o $REPO/build/duo/policy_$m.o -lhiprtc
done
ls -la $REPO/ab_libs/liboc_duo_*.so
