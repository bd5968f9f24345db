"""Summarise rocprofv3 CSV output of tools/prof_bench.sh: per-kernel mean duration and PMC means."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]

def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))

def short(n):
    n = n.split("(")[0]
    return n.replace("void sdfs::", "").replace("sdfs::", "")[:60]

# kernel trace
for f in find("kt", "*kernel_trace.csv"):
    dur = defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    print("== kernel trace (us): name, calls, mean, min, max | vgpr sgpr lds scratch grid wg")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:60s} {len(v):5d} {sum(v)/len(v):10.1f} {min(v):10.1f} {max(v):10.1f} | {meta[k]}")
for f in find("kt", "*kernel_stats.csv"):
    print("== kernel_stats.csv")
    print(open(f).read()[:3000])

# pmc
for sub in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"== {sub}: per-dispatch means")
        for k, d in acc.items():
            if "pass_kernel" not in k and "k_" not in k:
                continue
            print(" ", k)
            for c, v in d.items():
                print(f"     {c:28s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
