set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for c in final_scene.yml cornell.yml cfg.yml; do bash tools/sweep.sh $c 64 RT_SHADE_BLOCKS_PER_CU "4 32"; done
