"""Small-grid plans side by side: successive approximation (hipGraph chunks, check_every=128) on several
unconditional grids with extents <= 16, once per variant of the create-time knobs; prints us per iteration.

    python tools/ab_small.py [shape ...]      e.g.  15,15,15,15  8,8,8,8,8,8"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdfs_via_autodiff_amd as S  # noqa: E402

VARIANTS = {
    "small (auto run)": {},
    "small run 1": {"SDFS_SMALL_R": "1"},
    "small run 4": {"SDFS_SMALL_R": "4"},
    "generic tiles": {"SDFS_SMALL_PLAN": "0", "SDFS_PAD_PLAN": "0"},
    # beyond 400 k points the planner leaves the small-grid kernels; SDFS_PLAN=pair keeps them wherever they are legal
    "small forced": {"SDFS_PLAN": "pair", "SDFS_PAD_PLAN": "0"},
    # 6-D grids beyond the small-grid plan's size: the padded pair plan (pad_kernels.hpp), the default there
    "padded pair plan": {"SDFS_PAD_PLAN": "2"},      # (2: also shapes that mix extents above and below 16, which the default leaves to the generic tiles)
}
DEFAULT = ["5,5,5,5", "10,10,10,10", "15,15,15,15", "16,16,16,16", "6,6,6,6,6,6", "8,8,8,8,8,8", "10,10,10,10,10,10",
           "12,12,12,12,12,12", "14,14,14,14,14,14"]


def main():
    shapes_l = [tuple(int(x) for x in a.split(",")) for a in (sys.argv[1:] or DEFAULT)]
    for shapes in shapes_l:
        model = "ssy" if len(shapes) == 4 else "gcy"
        m = S.SSY() if model == "ssy" else S.GCY()
        arr = (S.discretize_ssy if model == "ssy" else S.discretize_gcy)(m, shapes)
        w0 = np.full(shapes, 800.0)
        n_it = 2048 if np.prod(shapes) < 1e6 else 256
        row = []
        for name, env in VARIANTS.items():
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            T = S.KoopmansOperator(model, shapes, m.params, arr)
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
            T.solve(w0, "successive_approx", tol=0.0, max_iter=256, check_every=128)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                T.solve(w0, "successive_approx", tol=0.0, max_iter=n_it, check_every=128)
                best = min(best, time.perf_counter() - t0)
            row.append(f"{name}: {best / n_it * 1e6:8.2f} us")
            if os.environ.get("AB_SMALL_PLANS"):
                print("   ", name, "|", T.describe_plan().splitlines()[0][:150], flush=True)
            del T
        print(f"{'x'.join(map(str, shapes)):>20s}  " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
