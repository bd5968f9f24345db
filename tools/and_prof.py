"""SSY 15^4 Anderson once (for rocprofv3 --kernel-trace --stats -- python3 tools/and_prof.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdfs_via_autodiff_amd as S  # noqa: E402

m = S.SSY(); shp = (15,) * 4
T = S.ssy_operator(shp, m.params, S.discretize_ssy(m, shp))
w0 = np.full(shp, 800.0)
algo = sys.argv[1] if len(sys.argv) > 1 else "anderson"
kw = dict(tol=1e-8) if algo != "newton" else dict(tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)
x, n, info = T.solve(w0, algo, **kw)
print(algo, n, info["n_apply"], info["final_err"])
