#!/bin/bash
# usage: tools/profile_round.sh <tag>   -- on the GPU box: bench.py alone, bench.py under rocprofv3 --kernel-trace --stats, and the
# FETCH_SIZE / WRITE_SIZE passes of the bench workload; everything lands in gpurun_out/<tag>/ (copy what matters into profiles/).
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(ls $OUT/kt/*/*kernel_stats.csv | tail -1) $OUT/kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/profile_driver.py final_scene.yml 400 1 > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/profile_driver.py final_scene.yml 400 1 > $OUT/write.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
tot = {}
for name in ("fetch", "write"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (out, name)):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            d = tot.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": {}})
            d[row["Counter_Name"]] += float(row["Counter_Value"])
            d["n"].setdefault(row["Counter_Name"], set()).add(row["Dispatch_Id"])
res = {}
for k, d in tot.items():
    n = max(len(d["n"].get("FETCH_SIZE", [])), 1)
    # counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes: doubled
    res[k] = {"launches": n, "fetch_kib_per_launch": d["FETCH_SIZE"] / n, "write_kib_per_launch": d["WRITE_SIZE"] / n,
              "hbm_bytes_per_launch": (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0 / n}
json.dump({"workload": "final_scene.yml 800x800, 400 spp, one pass (the bench step)", "kernels": res,
           "hbm_bytes_per_launch": res.get("k_trace", {}).get("hbm_bytes_per_launch")}, open(out + "/traffic_final_scene.json", "w"), indent=1)
PY
rm -rf $OUT/fetch $OUT/write $OUT/kt
