set -e
for c in final_scene.yml cornell.yml; do bash tools/sweep.sh $c 64 RT_TRACE_BLOCK "512 1024"; done
