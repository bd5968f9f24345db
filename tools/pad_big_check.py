import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S
from oracle.c_oracle import COperator
g = S.GCY()
for shapes in ((31, 31, 17, 17, 19, 19), (25,) * 6):
    arr = S.discretize_gcy(g, shapes)
    T = S.KoopmansOperator("gcy", shapes, g.params, arr)
    print(shapes, int(np.prod(shapes)), T.describe_plan().splitlines()[:3], flush=True)
    oc = COperator("gcy", shapes, g.params, arr)
    rng = np.random.default_rng(1)
    w = 300 + 600 * rng.random(shapes)
    t0 = time.time(); want = oc(w); print("  oracle", round(time.time() - t0, 1), "s", flush=True)
    got = T(w)
    print("  T rel err", float(np.max(np.abs(got - want) / want)), "resid", T.residual(), float(np.max(np.abs(want - w))), flush=True)
    del got
    v = rng.standard_normal(shapes)
    jo = oc.jvp(w, v); jv = T.jvp(w, v)
    print("  jvp rel err", float(np.max(np.abs(jv - jo)) / np.max(np.abs(jo))), flush=True)
    del jo, jv, v, want, w
    T.close()
