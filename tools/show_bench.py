import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(l)
print("value", round(d["value"], 2), d["unit"], "ms/step", round(d["ms_per_step"], 4))
for k in d["kernels"]:
    print("  ", k["name"], "avg_ms", round(k["avg_ms"], 4), "GB/s", round(k["GBps"], 1), "frac", round(k["frac_hbm"], 3))
for k in ("cpu_baseline", "gpu_over_cpu", "secondary"):
    if k in d:
        print(k, d[k])
