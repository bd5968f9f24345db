"""Summarise a rocprofv3 run of any command of this repo by kernel symbol: calls, mean / total duration (kernel trace),
HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half
the bytes of 16-byte-per-lane streaming reads, WRITE_SIZE is exact; both in KiB).  Directory layout as written by
tools/newton_profile.sh: <root>/kt, <root>/pmc2 (FETCH_SIZE), <root>/pmc3 (WRITE_SIZE), <root>/kt.log (the command's
own JSON line).  Writes <root>/summary.json."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


def short(n):
    return n.split("(")[0].replace("void sdfs::", "").replace("sdfs::", "")[:70]


out = {"kernels": {}}
tot = 0.0
for f in find("kt", "*kernel_trace.csv"):
    dur = defaultdict(list)
    first = {}
    for r in csv.DictReader(open(f)):
        nm = short(r["Kernel_Name"])
        dur[nm].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        first.setdefault(nm, r)
    tot = sum(sum(v) for v in dur.values())
    for nm, v in dur.items():
        r = first[nm]
        out["kernels"][nm] = {"calls": len(v), "mean_us": sum(v) / len(v), "total_ms": sum(v) / 1e3, "share": sum(v) / tot,
                              "vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size")}
for sub, ctr in (("pmc2", "FETCH_SIZE"), ("pmc3", "WRITE_SIZE")):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for nm, v in acc.items():
            out["kernels"].setdefault(nm, {})[ctr] = sum(v) / len(v)
try:
    for ln in open(os.path.join(root, "kt.log")):
        if ln.startswith("{"):
            out["command_line_json"] = json.loads(ln)
except (OSError, ValueError):
    pass
print(f"== kernels by total time (kernel trace; {tot / 1e3:.1f} ms of kernels)")
print(f"{'kernel':70s} {'calls':>6s} {'mean us':>9s} {'total ms':>9s} {'share':>6s} {'HBM MB/launch':>14s} {'TB/s':>6s}  vgpr lds scratch")
for nm, d in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("total_ms", 0)):
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_traffic_bytes"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    hb = d.get("hbm_traffic_bytes")
    if "calls" not in d:
        continue
    rate = f"{hb / (d['mean_us'] * 1e-6) / 1e12:6.2f}" if hb else "     -"
    print(f"{nm:70s} {d['calls']:6d} {d['mean_us']:9.1f} {d['total_ms']:9.2f} {d['share']:6.3f} {(hb or 0) / 1e6:14.1f} {rate}  "
          f"{d.get('vgpr')} {d.get('lds')} {d.get('scratch')}")
if "command_line_json" in out:
    print("== the command's own line:", json.dumps({k: v for k, v in out["command_line_json"].items() if k != "plan"}))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
