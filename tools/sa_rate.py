"""Successive-approximation rate on the device loop: K iterations (tol = 0: never converges), one host sync.

    python tools/sa_rate.py [gcy20|gcy16] [K]      prints ms per iteration for the default choice, SDFS_SA_FUSED = 1 and 0"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sdfs_via_autodiff_amd as S  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "gcy20"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    shapes = {"gcy20": (20,) * 6, "gcy16": (16,) * 6, "gcy24": (24,) * 6, "gcy24x16": (24, 24, 24, 24, 16, 16),
              "gcy32": (32, 32, 32, 32, 16, 16)}[wl]
    m = S.GCY()
    arr = S.discretize_gcy(m, shapes)
    ops = {}
    for name, env in (("default", {}), ("fused", {"SDFS_SA_FUSED": "1"}), ("one launch per pass", {"SDFS_SA_FUSED": "0"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        ops[name] = S.KoopmansOperator("gcy", shapes, m.params, arr)
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    w = torch.full(shapes, 800.0, dtype=torch.float64, device="cuda")
    for rnd in range(3):
        for name, op in ops.items():
            x = w.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n, info = op.solve_dev(x.data_ptr(), "successive_approx", tol=0.0, max_iter=K, check_every=K)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"round {rnd} {name:22s} {dt / K * 1e3:.4f} ms per iteration ({K} iterations, err {info['final_err']:.6g})", flush=True)
    for name, op in ops.items():
        op.set_profiling(True); op.reset_counters()
        x = w.clone()
        op.solve_dev(x.data_ptr(), "successive_approx", tol=0.0, max_iter=20, check_every=20)
        for c in op.counters():
            print(f"  {name:22s} {c['name']:48s} launches {c['launches']:4d} avg {c['total_ms'] / max(c['launches'], 1):.4f} ms")
        op.set_profiling(False)


if __name__ == "__main__":
    main()
