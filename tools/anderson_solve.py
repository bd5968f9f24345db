"""One Anderson solve at GCY n^6 on the device loop (code/solvers.py:98-124 with the opt-in relative ridge)
from w = 800 to 1e-8 -- a timing command for tools/ab_libs.sh.  argv: [n] [ridge]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdfs_via_autodiff_amd as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ridge = float(sys.argv[2]) if len(sys.argv) > 2 else -1e-6
g = S.GCY(); shp = (n,) * 6
op = S.KoopmansOperator("gcy", shp, g.params, S.discretize_gcy(g, shp))
op.set_stream(torch.cuda.current_stream().cuda_stream)
ws = torch.full(shp, 800.0, dtype=torch.float64, device="cuda")
op.solve_dev(ws.data_ptr(), "anderson", tol=1e-8, max_iter=8, ridge=ridge)
for rep in range(2):
    ws.fill_(800.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it, info = op.solve_dev(ws.data_ptr(), "anderson", tol=1e-8, max_iter=2000, ridge=ridge)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({"grid": f"GCY {n}^6", "ridge": ridge, "passes": it, "seconds": dt, "ms_per_pass": dt / max(it, 1) * 1e3, "final_err": info["final_err"]}))
