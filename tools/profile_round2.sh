#!/bin/bash
# usage: tools/profile_round2.sh <tag>   -- on the GPU box: everything tools/profile_round.sh collects (bench, kernel stats, HBM
# traffic of final_scene.yml) plus the PMC passes behind profiles/valu_final_scene.json and profiles/memory_synthetic1m.json.
# Everything lands in gpurun_out/<tag>/ (copy what matters into profiles/).
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
bash $R/tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1 || { tail -5 $OUT/profile_round.log; exit 1; }
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
GR="GRBM_GUI_ACTIVE GRBM_COUNT"
bash $R/tools/pmc_groups.sh $OUT/pmc_final final_scene.yml 64 "$SQ1" "$SQ2" "$GR" > $OUT/pmc_final.log 2>&1 || { tail -5 $OUT/pmc_final.log; exit 1; }
bash $R/tools/pmc_groups.sh $OUT/pmc_cornell cornell.yml 64 "$SQ1" "$SQ2" "$GR" > $OUT/pmc_cornell.log 2>&1 || { tail -5 $OUT/pmc_cornell.log; exit 1; }
bash $R/tools/pmc_groups.sh $OUT/pmc_synth Synthetic:1000000 8 "$SQ1" "$SQ2" "$GR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "FETCH_SIZE" "WRITE_SIZE" \
  > $OUT/pmc_synth.log 2>&1 || { tail -5 $OUT/pmc_synth.log; exit 1; }
cd $R
python3 tools/pmc_to_json.py valu gpurun_out/$TAG/pmc_final/summary.txt "final_scene.yml 800x800, one 64-spp pass (50 bounces)" $OUT/valu_final_scene.json > /dev/null
python3 tools/pmc_to_json.py valu gpurun_out/$TAG/pmc_cornell/summary.txt "cornell.yml 600x600, one 64-spp pass (50 bounces)" $OUT/valu_cornell.json > /dev/null
python3 tools/pmc_to_json.py memory gpurun_out/$TAG/pmc_synth/summary.txt "Synthetic:1000000 3840x2160, one 8-spp pass (50 bounces)" $OUT/memory_synthetic1m.json > /dev/null
for d in pmc_final pmc_cornell pmc_synth; do rm -rf $OUT/$d/pass*/; done
echo "profile_round2 done"
