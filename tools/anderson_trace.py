"""One Anderson solve per small grid for a rocprofv3 kernel trace (per-kernel durations of the fused passes):
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/and_trace -- python3 tools/anderson_trace.py [fused 0|1]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
os.environ["SDFS_AND_FUSED"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import sdfs_via_autodiff_amd as S  # noqa: E402

for model, shapes in (("ssy", (15,) * 4),):
    m = S.SSY() if model == "ssy" else S.GCY()
    arr = (S.discretize_ssy if model == "ssy" else S.discretize_gcy)(m, shapes)
    T = S.KoopmansOperator(model, shapes, m.params, arr)
    x, n, info = T.solve(np.full(shapes, 800.0), "anderson", tol=1e-8, max_iter=20000)
    print(model, shapes, n, info["final_err"])
