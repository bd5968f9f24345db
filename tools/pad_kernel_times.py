"""Per-kernel times of one application of T and of J.v on grids of the padded pair plan (HIP-event counters of the
library), next to the generic tiles:   python tools/pad_kernel_times.py [shape ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdfs_via_autodiff_amd as S  # noqa: E402


def run(shapes, pad):
    os.environ["SDFS_PAD_PLAN"] = "2" if pad else "0"
    model = "gcy" if len(shapes) == 6 else "ssy"
    g = S.GCY() if model == "gcy" else S.SSY()
    arr = (S.discretize_gcy if model == "gcy" else S.discretize_ssy)(g, shapes)
    T = S.KoopmansOperator(model, shapes, g.params, arr)
    w = np.full(shapes, 800.0)
    v = np.random.default_rng(1).standard_normal(shapes)
    for _ in range(3):
        T(w); T.jvp(w, v)
    T.set_profiling(True); T.reset_counters()
    for _ in range(20):
        T(w); T.jvp(w, v)
    n = float(np.prod(shapes))
    for k in T.counters():
        if k["launches"]:
            ms = k["total_ms"] / k["launches"]
            print(f"   {'padded ' if pad else 'generic'} {k['name']:44s} {ms * 1e3:8.1f} us  {k.get('bytes', 0) / max(ms, 1e-9) / 1e6:7.0f} GB/s" if "bytes" in k else
                  f"   {'padded ' if pad else 'generic'} {k['name']:44s} {ms * 1e3:8.1f} us  ({8 * n / ms / 1e6:6.0f} GB/s per grid stream)")
    T.close()


for a in (sys.argv[1:] or ["15,15,15,15,15,15", "12,12,12,12,12,12"]):
    shapes = tuple(int(x) for x in a.split(","))
    print(shapes, flush=True)
    run(shapes, True)
    if not os.environ.get("SDFS_PAD_TIMES_PADDED_ONLY"):      # (tools/pad_profile.sh profiles the padded kernels alone)
        run(shapes, False)
