set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3
timeout -k 60 120 ./ray_tracing_amd/rt_cli -c configs/showcase.yml -o gpurun_out/showcase.png --spp 256 2>&1 | tail -1
