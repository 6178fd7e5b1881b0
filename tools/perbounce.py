"""Per-bounce ray counts and k_trace / k_shade durations from a rocprofv3 kernel trace.  usage: perbounce.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tr = [r for r in rows if r["Kernel_Name"].startswith("void k_trace")]
sh = [r for r in rows if (r["Kernel_Name"].startswith("void k_shade") or r["Kernel_Name"].startswith("k_shade_lds"))]
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
wk = [r for r in rows if r["Kernel_Name"].startswith("k_walkstep")][-(n - 1):]
tr, sh = tr[-n:], sh[-n:]
if wk:
    print("walk us: - " + " ".join("%.0f" % d(r) for r in wk) + "   sum %.0f, launches 11..: %.0f" % (sum(map(d, wk)), sum(map(d, wk[9:]))))
print("trace us:", " ".join("%.0f" % d(r) for r in tr))
print("shade us:", " ".join("%.0f" % d(r) for r in sh))
print("trace sum %.0f  shade sum %.0f   launches 11..: trace %.0f  shade %.0f" % (sum(map(d, tr)), sum(map(d, sh)), sum(map(d, tr[10:])), sum(map(d, sh[10:]))))
