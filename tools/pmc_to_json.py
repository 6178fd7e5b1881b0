"""Turns the per-kernel PMC summary of tools/pmc_groups.sh (summary.txt) into the small JSON files bench.py quotes:
  valu   <summary.txt> <workload text> <out.json>   VALU issue figures of k_trace and k_shade_lds
  memory <summary.txt> <workload text> <out.json>   L2 / fabric figures of k_trace (the 1M-sphere scene)"""
import json
import re
import sys


def parse(path):
    kernels, cur = {}, None
    for line in open(path):
        if line.startswith("=="):
            cur = kernels.setdefault(line[2:].strip(), {})
        else:
            m = re.match(r"\s+(\S+)\s+total (\S+)\s+per dispatch (\S+)\s+\((\d+) dispatches\)", line)
            if m and cur is not None:
                cur[m.group(1)] = dict(total=float(m.group(2)), per_dispatch=float(m.group(3)), dispatches=int(m.group(4)))
    return kernels


def pick(kernels, prefix):
    """The un-instrumented instantiation of a kernel (the one with the most dispatches)."""
    cands = [(max(c["dispatches"] for c in v.values()), k) for k, v in kernels.items() if k.replace("void ", "").startswith(prefix)]
    return kernels[max(cands)[1]], max(cands)[1]


def main():
    mode, path, workload, out = sys.argv[1:5]
    ks = parse(path)
    res = {"workload": workload, "source": path + " (tools/pmc_groups.sh: one rocprofv3 --pmc run per counter group)", "kernels": {}}
    names = ("k_trace", "k_shade_lds") if mode == "valu" else ("k_trace",)
    for name in names:
        try:
            c, full = pick(ks, name)
        except ValueError:
            continue
        t = lambda n: c[n]["total"] if n in c else None
        d = {"kernel": full}
        if mode == "valu":
            insts, active, thread = t("SQ_INSTS_VALU"), t("SQ_ACTIVE_INST_VALU"), t("SQ_THREAD_CYCLES_VALU")
            d.update(valu_wave_instructions=insts, salu_instructions=t("SQ_INSTS_SALU"), lds_instructions=t("SQ_INSTS_LDS"))
            if insts and thread:
                d["lanes_live_per_valu_instruction"] = thread / (insts * 64.0)
            waves_cycles, wait = t("SQ_WAVE_CYCLES"), t("SQ_WAIT_ANY")
            if waves_cycles and wait:
                d["wave_wait_frac"] = wait / waves_cycles
            grbm = t("GRBM_GUI_ACTIVE")
            if active and grbm:      # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                d["valu_busy_frac"] = active * 4.0 / (1024.0 * grbm / 8.0)
            if t("SQ_LDS_IDX_ACTIVE") and grbm:
                d["lds_busy_frac"] = t("SQ_LDS_IDX_ACTIVE") / (256.0 * grbm / 8.0)
                d["lds_bank_conflict_frac"] = (t("SQ_LDS_BANK_CONFLICT") or 0.0) / t("SQ_LDS_IDX_ACTIVE")
        else:
            hit, miss, ea, grbm = t("TCC_HIT_sum"), t("TCC_MISS_sum"), t("TCC_EA0_RDREQ_sum"), t("GRBM_GUI_ACTIVE")
            lat, req = t("TCP_TCC_READ_REQ_LATENCY_sum"), t("TCP_TCC_READ_REQ_sum")
            n = max(v["dispatches"] for v in c.values())
            d.update(dispatches=n, l2_hit_frac=hit / (hit + miss) if hit is not None and miss else None,
                     l2_requests_per_dispatch=(hit + miss) / n if hit is not None and miss is not None else None,
                     fabric_read_requests_per_dispatch=ea / n if ea is not None else None,
                     l1_to_l2_read_latency_cycles=lat / req if lat and req else None)
            if grbm:
                seconds = grbm / 8.0 / 2.4e9
                d["kernel_seconds_total_at_2.4GHz"] = seconds
                if hit is not None and miss is not None:
                    d["l2_request_rate_per_s"] = (hit + miss) / seconds
                    d["l2_bytes_per_s_at_64B"] = 64.0 * (hit + miss) / seconds
                if ea is not None:
                    d["fabric_read_bytes_per_s_at_64B"] = 64.0 * ea / seconds
            insts, thread, active = t("SQ_INSTS_VALU"), t("SQ_THREAD_CYCLES_VALU"), t("SQ_ACTIVE_INST_VALU")
            if insts and thread:
                d["lanes_live_per_valu_instruction"] = thread / (insts * 64.0)
            if active and grbm:
                d["valu_busy_frac"] = active * 4.0 / (1024.0 * grbm / 8.0)
            if t("SQ_WAVE_CYCLES") and t("SQ_WAIT_ANY"):
                d["wave_wait_frac"] = t("SQ_WAIT_ANY") / t("SQ_WAVE_CYCLES")
        res["kernels"][name] = {k: v for k, v in d.items() if v is not None}
    if mode == "memory" and "k_trace" in res["kernels"]:
        k = res["kernels"]["k_trace"]
        res.update(bound="valu_issue after the 4-wide tree (was L2 latency)" if k.get("valu_busy_frac", 0) > 0.7 else "l2_latency")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
