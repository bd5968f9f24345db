"""Summarise rocprofv3 CSV output of tools/prof_bench.sh.

Round 2: the operator's passes are distinct kernel symbols (slice_kernel / line_kernel<N, MODE>) on the pair
plan, so they are keyed by symbol ("kernels" in summary.json, in first-launch order); the generic pass_kernel
variants are still split by launch order modulo the number of passes ("passes").

Kernel trace: mean duration per kernel; the pass kernels of one operator application share a
symbol per template variant, so they are also split by launch order (pass index = dispatch
order modulo the number of passes).  PMC passes: per-pass means; HBM traffic per launch is
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half
of the bytes of 16-B-per-lane streaming reads, WRITE_SIZE is exact; both are in KiB)."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


def short(n):
    return n.split("(")[0].replace("void sdfs::", "").replace("sdfs::", "")[:60]


out = {"passes": {}, "kernels": {}, "launch_order": []}
FAST = ("slice_kernel", "line_kernel", "line_stream_kernel")


def is_fast(name):
    return any(k in name for k in FAST)

for f in find("kt", "*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    dur = defaultdict(list)
    for r in rows:
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("== kernel trace (us): name, calls, mean, min, max")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:60s} {len(v):5d} {sum(v)/len(v):10.1f} {min(v):10.1f} {max(v):10.1f}")
    fk = sorted([r for r in rows if is_fast(r["Kernel_Name"])], key=lambda r: int(r["Start_Timestamp"]))
    for r in fk:
        nm = short(r["Kernel_Name"])
        if nm not in out["launch_order"]:
            out["launch_order"].append(nm)
            out["kernels"][nm] = {"vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size"),
                                  "grid": r.get("Grid_Size_X"), "wg": r.get("Workgroup_Size_X")}
    for nm in out["launch_order"]:
        v = dur[nm]
        out["kernels"][nm].update(n=len(v), mean_us=sum(v) / len(v), min_us=min(v), max_us=max(v))
    pk = sorted([r for r in rows if "pass_kernel" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
    byp = defaultdict(list)
    for i, r in enumerate(pk):
        byp[i % npass].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"== pass kernels by launch order (mod {npass}): pass, n, mean us, min, max, variant, VGPR, LDS, scratch")
    for p_, v in sorted(byp.items()):
        r = pk[p_]
        print(f"pass {p_}: n={len(v)} mean={sum(v)/len(v):.1f} min={min(v):.1f} max={max(v):.1f} "
              f"{short(r['Kernel_Name'])} vgpr={r.get('VGPR_Count')} agpr={r.get('Accum_VGPR_Count')} "
              f"lds={r.get('LDS_Block_Size')} scratch={r.get('Scratch_Size')} grid={r.get('Grid_Size_X')} wg={r.get('Workgroup_Size_X')}")
        out["passes"].setdefault(str(p_), {})["mean_us"] = sum(v) / len(v)
for f in find("kt", "*kernel_stats.csv"):
    print("== kernel_stats.csv (rocprofv3 --stats)")
    print(open(f).read()[:2500])

for sub in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in find(sub, "*counter_collection.csv"):
        allrows = list(csv.DictReader(open(f)))
        facc = defaultdict(lambda: defaultdict(list))
        for r in allrows:
            if is_fast(r["Kernel_Name"]):
                facc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if facc:
            print(f"== {sub}: per-launch means by kernel")
            for nm, cs in facc.items():
                line = {c: sum(v) / len(v) for c, v in cs.items()}
                print(f"  {nm}: " + "  ".join(f"{c}={v:.4g}" for c, v in line.items()))
                out["kernels"].setdefault(nm, {}).update(line)
        rows = [r for r in allrows if "pass_kernel" in r["Kernel_Name"]]
        byd = defaultdict(dict)
        for r in rows:
            byd[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        acc = defaultdict(lambda: defaultdict(list))
        for i, di in enumerate(sorted(byd)):
            for c, v in byd[di].items():
                acc[i % npass][c].append(v)
        print(f"== {sub}: per-launch means by pass")
        for p_ in sorted(acc):
            line = {c: sum(v) / len(v) for c, v in acc[p_].items()}
            print(f"  pass {p_}: " + "  ".join(f"{c}={v:.4g}" for c, v in line.items()))
            out["passes"].setdefault(str(p_), {}).update(line)
for p_, d in out["passes"].items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_traffic_bytes"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        print(f"pass {p_}: HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 = {d['hbm_traffic_bytes']/1e9:.3f} GB")
# the plan the profiled bench ran on (bench.py compares it with its own before quoting these traffic figures)
try:
    for ln in open(os.path.join(root, "kt.log")):
        if ln.startswith('{"metric"'):
            out["plan"] = json.loads(ln)["config"]["plan"]
            out["bench_line_of_kernel_trace_run"] = {k: v for k, v in json.loads(ln).items() if k in ("value", "ms_per_step", "kernels")}
except OSError:
    pass
for nm, d in out["kernels"].items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_traffic_bytes"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        print(f"{nm}: HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 = {d['hbm_traffic_bytes']/1e9:.3f} GB")
    if "SQ_INSTS_VALU" in d and "SQ_INSTS_MFMA" in d and "mean_us" in d:
        # SIMD issue model (tools/probes/coissue_probe.hip): VALU ~4.9 cycles, fp64 MFMA 64 / 20 cycles, 1024 SIMDs
        print(f"{nm}: VALU {d['SQ_INSTS_VALU']/1e6:.1f} M, MFMA {d['SQ_INSTS_MFMA']/1e6:.1f} M wave-instructions per launch; "
              f"SQ_VALU_MFMA_BUSY_CYCLES {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)/1e6:.0f} M")
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
