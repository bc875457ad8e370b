#!/bin/bash
# A/B of several builds of the library in ONE GPU call (same box, same clocks), interleaved, twice each:
#   tools/ab_builds.sh <tag> "<name>=<lib.so> ..." <script> [args...]      (name "new" = the in-tree build: give it an empty path)
# writes gpurun_out/<tag>_<name>_{1,2}.log
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; LIBS=$2; shift 2
mkdir -p $REPO/gpurun_out
for i in 1 2; do
  for nl in $LIBS; do
    name=${nl%%=*}; lib=${nl#*=}
    if [ -z "$lib" ]; then python3 "$@" > $REPO/gpurun_out/${TAG}_${name}_$i.log 2>&1
    else OC_LIB_PATH=$REPO/$lib python3 "$@" > $REPO/gpurun_out/${TAG}_${name}_$i.log 2>&1; fi
  done
done
