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
for r in (0, G - 1):
    be = D.HipStages("gcy", shapes, g.params, arr, A, a_off[r], a_sz[r], B, b_off[r], b_sz[r], 0)
    print(f"rank {r} of {G}: A-block {a_sz[r]}, B-block {b_sz[r]}")
    print("  " + "\n  ".join(be.describe_plan().strip().splitlines()))
    sh0 = list(shapes); sh0[A] = a_sz[r]
    sh1 = list(shapes); sh1[B] = b_sz[r]
    x0 = torch.full(sh0, 800.0, dtype=torch.float64, device=dev)
    z1 = torch.rand(sh1, dtype=torch.float64, device=dev) * 1e-40 + 1e-42
    old = torch.full(sh1, 800.0, dtype=torch.float64, device=dev)
    res = torch.zeros(1, dtype=torch.float64, device=dev)
    for stage, mode, xin, kw in ((0, D.MODE_T, x0, {}), (1, D.MODE_T, z1, dict(old=old, resid=res))):
        for _ in range(3):
            be.run(stage, mode, xin, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        for _ in range(20):
            be.run(stage, mode, xin, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        npts = float(np.prod(xin.shape))
        print(f"  stage {stage} T: {dt * 1e3:.3f} ms for {npts:.3g} local points  ({npts * 16 / dt / 1e9:.0f} GB/s per pass-equivalent of 16 B/point)", flush=True)
    be.close()
