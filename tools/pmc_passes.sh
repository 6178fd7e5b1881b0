#!/bin/bash
# Collects rocprofv3 PMC counters for the render kernels, one run per counter group (the hardware cannot split
# groups over passes).  usage: tools/pmc_passes.sh <outdir> <config.yml> <spp> <repeats>
set -u
OUT=$1; CFG=$2; SPP=$3; REPS=$4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp
i=0
for GROUP in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pass$i" -- python3 "$R/tools/profile_driver.py" "$CFG" "$SPP" "$REPS" > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i ($GROUP) rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/pass$i.log"; fi
  if [ $rc -eq 137 ] || [ $rc -eq 124 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "stopping after a killed pass"; exit 1; fi
done
