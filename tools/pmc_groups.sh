#!/bin/bash
# usage: tools/pmc_groups.sh <outdir> <config.yml> <spp> "<counters of pass 1>" "<counters of pass 2>" ...
# One rocprofv3 --pmc run per group (the hardware cannot split a group over passes); stops at the first killed pass.
set -u
OUT=$1; CFG=$2; SPP=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp
i=0
for GROUP in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pass$i" -- python3 "$R/tools/profile_driver.py" "$CFG" "$SPP" 1 > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i ($GROUP) rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/pass$i.log"; fi
  if [ $rc -eq 137 ] || [ $rc -eq 124 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "stopping after a killed pass"; exit 1; fi
done
python3 "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt"
