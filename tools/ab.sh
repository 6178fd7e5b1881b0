set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3
timeout -k 10 300 python3 tools/profile_driver.py Synthetic:1000000 4 2 0 4 2>&1 | tail -3
