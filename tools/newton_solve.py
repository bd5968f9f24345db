"""One Newton-Krylov solve at GCY 20^6 (BASELINE configs[3]'s algorithm: code/solvers.py:51-95, inner BiCGSTAB) from
w = 800 to 1e-8, resident on the device -- the command tools/newton_profile.sh traces.  argv: [krylov_f32 (0 / 1 / 3)] [n]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (rocprofv3 runs it from /tmp)
import torch
import sdfs_via_autodiff_amd as S

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = S.GCY(); shp = (n,) * 6
op = S.KoopmansOperator("gcy", shp, g.params, S.discretize_gcy(g, shp))
op.set_stream(torch.cuda.current_stream().cuda_stream)
ws = torch.full(shp, 800.0, dtype=torch.float64, device="cuda")
op.solve_dev(ws.data_ptr(), "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, max_iter=1, krylov_f32=mode)    # warm-up: one step
ws.fill_(800.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
it, info = op.solve_dev(ws.data_ptr(), "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, krylov_f32=mode)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"grid": f"GCY {n}^6", "krylov_f32": mode, "newton_steps": it, "operator_applies": info["n_apply"], "seconds": dt,
                  "final_step": info["final_err"], "plan": op.describe_plan().strip().split("\n")}))
