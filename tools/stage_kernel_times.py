"""Times of the sharded stage kernels on ONE GPU: rank r's handle of a world-G split of GCY n^6 (default 20^6 over 8),
stage 0 and stage 1 of T, of the linearising T and of J.v, HIP-event timed; bytes = 16 per local point and pass.
    python tools/stage_kernel_times.py [n] [G]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S  # noqa: E402
from sdfs_via_autodiff_amd import distributed as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
shapes = (n,) * 6
g = S.GCY()
arr = S.discretize_gcy(g, shapes)
A, B = D.SHARD_AXES["gcy"]
a_sz, b_sz = D.block_sizes(n, G), D.block_sizes(n, G)
a_off, b_off = D.block_offsets(a_sz), D.block_offsets(b_sz)
dev = torch.device("cuda", 0)


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


import os  # noqa: E402
for plan in ("default", "classic"):
    # default: the stages on the pair plan's kernels where the grid admits them (round 4); classic: the generic tiles
    if plan == "classic":
        os.environ["SDFS_PLAN"] = "classic"
    else:
        os.environ.pop("SDFS_PLAN", None)
    for r in (0, G - 1):
        for orient in (0, 1):
            if orient == 0:
                be = D.HipStages("gcy", shapes, g.params, arr, A, a_off[r], a_sz[r], B, b_off[r], b_sz[r], 0)
                ax0, n0, ax1, n1 = A, a_sz[r], B, b_sz[r]
            else:       # the mirror schedule's handle: the roles of the two axes swapped
                be = D.HipStages("gcy", shapes, g.params, arr, B, b_off[r], b_sz[r], A, a_off[r], a_sz[r], 0)
                ax0, n0, ax1, n1 = B, b_sz[r], A, a_sz[r]
            print(f"[{plan}] rank {r} of {G}, {'mirror ' if orient else ''}orientation: stage 0 on a block of {n0} along axis {ax0}, stage 1 on {n1} along axis {ax1}")
            print("  " + "\n  ".join(l for l in be.describe_plan().strip().splitlines()))
            sh0 = list(shapes); sh0[ax0] = n0
            sh1 = list(shapes); sh1[ax1] = n1
            x0 = torch.full(sh0, 800.0, dtype=torch.float64, device=dev) + torch.rand(sh0, dtype=torch.float64, device=dev)
            old = torch.full(sh1, 800.0, dtype=torch.float64, device=dev)
            res = torch.zeros(1, dtype=torch.float64, device=dev)
            z1 = be.run(0, D.MODE_T, x0)
            if tuple(z1.shape) != tuple(sh1):          # (same point count when the blocks are equal; otherwise a stand-in of stage 1's shape)
                z1 = torch.rand(sh1, dtype=torch.float64, device=dev) * 1e-40 + 1e-42
            v0 = torch.rand(sh0, dtype=torch.float64, device=dev)
            v1 = torch.rand(sh1, dtype=torch.float64, device=dev)
            tot = {}
            for name, mode in (("T", D.MODE_T), ("T lin", D.MODE_T_LIN), ("J.v", D.MODE_JVP)):
                in0, in1 = (v0, v1) if mode == D.MODE_JVP else (x0, z1)
                kw = dict(old=old, resid=res) if mode == D.MODE_T else {}
                t0 = timed(lambda: be.run(0, mode, in0))
                t1 = timed(lambda: be.run(1, mode, in1, **kw))
                n0p, n1p = float(np.prod(sh0)), float(np.prod(sh1))
                print(f"  {name:6s} stage 0 {t0 * 1e3:.3f} ms ({n0p:.3g} points), stage 1 {t1 * 1e3:.3f} ms ({n1p:.3g} points), both {(t0 + t1) * 1e3:.3f} ms", flush=True)
            be.close()
