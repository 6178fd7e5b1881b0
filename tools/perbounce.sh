#!/bin/bash
# usage: tools/perbounce.sh <tag> VAR=value ...   -- on the GPU box: kernel trace of two 400-spp steps of final_scene.yml with the given
# environment, then per-launch k_trace / shading-kernel durations of the last step (tools/perbounce.py).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $R/tools/profile_driver.py ${SCENE:-final_scene.yml} ${SPP:-400} 2 > $OUT/run.log 2>&1 || exit 1
f=$(ls $OUT/kt/*/*kernel_trace.csv | tail -1)
python3 $R/tools/perbounce.py $f 50 > $OUT/perbounce.txt
rm -rf $OUT/kt
cat $OUT/perbounce.txt
