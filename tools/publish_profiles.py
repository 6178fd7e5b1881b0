"""Copies what tools/profile_round2.sh left under gpurun_out/<tag>/ into profiles/<name>/ and refreshes the three JSON files at the
top of profiles/ that bench.py quotes (each says where it came from).  usage: publish_profiles.py <tag> <name>"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", name)
os.makedirs(dst, exist_ok=True)
for f in ("bench.json", "bench_under_rocprof.json", "kernel_stats.csv", "traffic_final_scene.json", "valu_final_scene.json", "valu_cornell.json",
          "memory_synthetic1m.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
for d, out in (("pmc_final", "pmc_summary_final_scene_64spp.txt"), ("pmc_cornell", "pmc_summary_cornell_64spp.txt"), ("pmc_synth", "pmc_summary_synthetic1M_8spp.txt")):
    shutil.copy(os.path.join(src, d, "summary.txt"), os.path.join(dst, out))
notes = {"traffic_final_scene.json": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes; FETCH_SIZE doubled as the gfx950 correction for 16-byte-per-lane reads",
         "valu_final_scene.json": "pmc_summary_final_scene_64spp.txt; tools/pmc_groups.sh: one rocprofv3 --pmc run per counter group",
         "valu_cornell.json": "pmc_summary_cornell_64spp.txt; tools/pmc_groups.sh: one rocprofv3 --pmc run per counter group",
         "memory_synthetic1m.json": "pmc_summary_synthetic1M_8spp.txt; tools/pmc_groups.sh: one rocprofv3 --pmc run per counter group"}
for f, how in notes.items():
    d = json.load(open(os.path.join(dst, f)))
    d["source"] = "profiles/%s (%s)" % (name, how)
    for p in (os.path.join(dst, f), os.path.join(ROOT, "profiles", f)):
        json.dump(d, open(p, "w"), indent=1)
print("published", dst)
