set -e
for c in final_scene.yml cornell.yml; do bash tools/sweep.sh $c 64 RT_TRACE_WIDE_STACK "0"; RT_TRACE_WIDE_STACK=1 bash tools/sweep.sh $c 64 RT_REFILL_LANES "16";  bash tools/sweep.sh $c 64 RT_REFILL_LANES "8 24"; done
