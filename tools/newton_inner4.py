"""tools/newton_solve.py at inner tolerance 1e-4 (the row bench.py carries for config 5) -- a timing command for tools/ab_libs.sh.
argv: [krylov_f32] [n]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdfs_via_autodiff_amd as S

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = S.GCY(); shp = (n,) * 6
op = S.KoopmansOperator("gcy", shp, g.params, S.discretize_gcy(g, shp))
op.set_stream(torch.cuda.current_stream().cuda_stream)
ws = torch.full(shp, 800.0, dtype=torch.float64, device="cuda")
op.solve_dev(ws.data_ptr(), "newton", tol=1e-8, inner_rtol=1e-4, inner_atol=0.0, max_iter=1, krylov_f32=mode)
best = 1e9
for rep in range(2):
    ws.fill_(800.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it, info = op.solve_dev(ws.data_ptr(), "newton", tol=1e-8, inner_rtol=1e-4, inner_atol=0.0, krylov_f32=mode)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(json.dumps({"grid": f"GCY {n}^6", "krylov_f32": mode, "inner": 1e-4, "newton_steps": it, "operator_applies": info["n_apply"], "seconds": best, "final_step": info["final_err"]}))
