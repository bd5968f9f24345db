"""Interleaved A/B timing of planner knobs in ONE process (per-call GPU boxes differ by several per cent, so
separate bench.py runs cannot rank variants that are a few per cent apart).

    python tools/ab_plan.py [workload] -- times one SA step (T + fused residual) per variant, `rounds` rounds of
    `reps` back-to-back steps each, variants interleaved; prints median / min ms per step and per kernel."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sdfs_via_autodiff_amd as S  # noqa: E402

VARIANTS = {
    "pair o1 p0 (default)": {},
    "a3 gathers (SDFS_A3_TABLES=0)": {"SDFS_A3_TABLES": "0"},
    "stream 0 (plain line kernels)": {"SDFS_LINE_STREAM": "0"},
    "stream 1 (middle only)": {"SDFS_LINE_STREAM": "1"},
    "stream 2 (last only)": {"SDFS_LINE_STREAM": "2"},
    "stream 7 (both, every extent)": {"SDFS_LINE_STREAM": "7"},
    "stream 5 (middle, every extent)": {"SDFS_LINE_STREAM": "5"},
    "stream 6 (last, every extent)": {"SDFS_LINE_STREAM": "6"},
    "pair o1 p2": {"SDFS_LINE_PERSIST": "2"},
    "pair o1 p3": {"SDFS_LINE_PERSIST": "3"},
    "pair o0 p2": {"SDFS_PAIR_ORDER": "0"},
    "pair o0 p0": {"SDFS_PAIR_ORDER": "0", "SDFS_LINE_PERSIST": "0"},
    "classic": {"SDFS_PLAN": "classic"},
}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "gcy20"
    shapes = {"gcy20": (20,) * 6, "gcy16": (16,) * 6, "gcy24": (24,) * 6, "gcy32": (32, 32, 32, 32, 16, 16),
              "gcy24x16": (24, 24, 24, 24, 16, 16)}[wl]
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    m = S.GCY()
    arr = S.discretize_gcy(m, shapes)
    ops = {}
    for name, env in VARIANTS.items():
        if only and not any(o in name for o in only):
            continue
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        ops[name] = S.KoopmansOperator("gcy", shapes, m.params, arr)
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    w = torch.from_numpy(400 + 500 * np.random.default_rng(0).random(shapes)).cuda()
    out = torch.empty_like(w)
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    for op in ops.values():
        op.set_stream(torch.cuda.current_stream().cuda_stream)
    rounds, reps = 7, 60
    times = {k: [] for k in ops}
    for r in range(rounds + 1):
        for name, op in ops.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                op.apply_dev(w.data_ptr(), out.data_ptr(), res.data_ptr())
            torch.cuda.synchronize()
            if r > 0:
                times[name].append((time.perf_counter() - t0) / reps * 1e3)
    for name, op in ops.items():
        op.set_profiling(True)
        for _ in range(40):
            op.apply_dev(w.data_ptr(), out.data_ptr(), res.data_ptr())
        per = "  ".join(f"{c['total_ms'] / max(c['launches'], 1):.4f}" for c in op.counters())
        op.set_profiling(False)
        t = sorted(times[name])
        print(f"{name:32s} median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f}  max {t[-1]:.4f}   kernels: {per}", flush=True)


if __name__ == "__main__":
    main()
