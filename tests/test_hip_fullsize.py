"""
GPU parity tests at BASELINE.json's full GCY 20^6 grid on the paths that produce the published numbers
("parity-test gaps"):

 * the device successive-approximation loop solver() runs at 20^6 (slice_kernel<20, S_MID> + line_kernel<20,
   L_TFUSED> alternating over both line pairs) against k applications of the C oracle on the full grid
   (code/solvers.py:34-36 on code/gcy/discrete/gcy_wc_ratio.py:134-238), plus a small twin with a 20-extent
   slice pair and both line pairs of extent 20;
 * Newton-Krylov at 20^6: distance to the polished fixed point, |x - x*| < 1e-8 (north_star's tolerance), fp64
   and fp32 Krylov storage;
 * the conditional-tensor kernels at full size against the C oracle;
 * the sharded stage kernels at the real 20-over-8 split (3,3,3,3,2,2,2,2): the eight ranks' handles driven from
   ONE process (a GPU box admits at most six processes on its card), exchanges done by slicing on the host.

The contraction modulus of T is ~0.9988, so a residual r bounds the distance to the fixed point by r / 0.0012.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = (20,) * 6


@pytest.fixture(scope="module")
def S():
    import sdfs_via_autodiff_amd as S
    return S


@pytest.fixture(scope="module")
def gcy20(S):
    g = S.GCY()
    arr = S.discretize_gcy(g, SHAPES)
    return g.params, arr


@pytest.fixture(scope="module")
def c_oracle(gcy20):
    from oracle.c_oracle import COperator
    params, arr = gcy20
    return COperator("gcy", SHAPES, params, arr)


@pytest.mark.parametrize("fused", [None, "1"])
def test_device_sa_loop_gcy20_vs_c_oracle(S, gcy20, c_oracle, fused):
    """max_iter = k, tol = 0: the device loop's k-th iterate and its last error against k applications of the C
    oracle on the full grid.  Default at 20^6: three streamed launches per iteration (what solver() runs);
    SDFS_SA_FUSED=1: slices + fused line kernel (end of one application and start of the next)."""
    params, arr = gcy20
    if fused is not None:
        os.environ["SDFS_SA_FUSED"] = fused
    try:
        T = S.KoopmansOperator("gcy", SHAPES, params, arr)
    finally:
        os.environ.pop("SDFS_SA_FUSED", None)
    assert "pair plan pass" in T.describe_plan()
    w0 = np.full(SHAPES, 800.0)
    want, prev = w0, None
    for k in (1, 2, 3):
        prev, want = want, c_oracle(want)
        x, n, info = T.solve(w0, "successive_approx", tol=0.0, max_iter=k)
        assert n == k and info["n_apply"] == k
        rel = np.max(np.abs(x - want) / want)
        assert rel < 1e-11, (k, rel)
        err = float(np.max(np.abs(want - prev)))
        assert abs(info["final_err"] - err) <= 1e-9 * err
        del x
    # a start that is not constant along any axis
    w1 = 400 + 500 * np.random.default_rng(7).random(SHAPES)
    want = c_oracle(c_oracle(w1))
    x, n, _ = T.solve(w1, "successive_approx", tol=0.0, max_iter=2)
    assert np.max(np.abs(x - want) / want) < 1e-11
    T.close()


@pytest.mark.parametrize("fused", [None, "1"])
@pytest.mark.parametrize("shapes", [(20, 20, 16, 16, 20, 20), (20, 20, 20, 20, 16, 16)])
def test_device_sa_loop_small_twin_vs_oracle(S, shapes, fused):
    """20-extent slice pair and 20-extent line pairs on smaller grids (2.6e7 / 1.6e7 points, C oracle), both forms
    of the device loop."""
    from oracle.c_oracle import COperator
    g = S.GCY(); arr = S.discretize_gcy(g, shapes)
    if fused is not None:
        os.environ["SDFS_SA_FUSED"] = fused
    try:
        T = S.KoopmansOperator("gcy", shapes, g.params, arr)
    finally:
        os.environ.pop("SDFS_SA_FUSED", None)
    assert "pair plan pass" in T.describe_plan()
    oc = COperator("gcy", shapes, g.params, arr)
    w0 = np.full(shapes, 800.0)
    want = w0
    for k in (1, 2, 3, 4):
        want = oc(want)
        x, n, _ = T.solve(w0, "successive_approx", tol=0.0, max_iter=k)
        assert n == k
        assert np.max(np.abs(x - want) / want) < 1e-11, k
    T.close()


def test_newton_krylov_gcy20_distance_to_fixed_point(S, gcy20, c_oracle):
    """configs[3] on one GPU.  x* = Newton polished to a step of 1e-11, its residual confirmed by the C oracle below
    1e-12 (so x* is within 1e-9 of the fixed point); the 1e-8 solves, fp64 and fp32 Krylov storage, must lie within
    1e-8 of it."""
    params, arr = gcy20
    T = S.KoopmansOperator("gcy", SHAPES, params, arr)
    w0 = np.full(SHAPES, 800.0)
    x, n, info = T.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)
    assert info["status"] == 0 and info["final_err"] <= 1e-8 and n < 25
    xs, _, i2 = T.solve(x, "newton", tol=1e-11, inner_rtol=1e-9, inner_atol=0.0, max_iter=10)
    assert i2["status"] == 0
    r = float(np.max(np.abs(c_oracle(xs) - xs)))
    assert r < 1e-12, r
    d64 = float(np.max(np.abs(x - xs)))
    assert d64 < 5e-9, d64
    x32, n32, i32 = T.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, krylov_f32=1)
    assert i32["status"] == 0
    d32 = float(np.max(np.abs(x32 - xs)))
    assert d32 < 5e-9, d32
    del x32
    # config 5 "on MFMA" (round 4): fp32 LDS tiles + v_mfma_f32 for the J.v passes, BiCGSTAB's vector updates on their
    # first pass, the persistent fp32 middle pass -- the same distance to the same fixed point; and the relative-ridge
    # Anderson loop (opt-in) to the metric's tolerance
    xm, nm, im = T.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, krylov_f32=3)
    assert im["status"] == 0 and nm < 25
    dm = float(np.max(np.abs(xm - xs)))
    assert dm < 5e-9, dm
    del xm
    xa, na, ia = T.solve(w0, "anderson", tol=1e-8, max_iter=2000, ridge=-1e-6)
    assert ia["status"] == 0 and na < 542, (na, ia)          # (542 = successive approximation's count from the same start)
    da = float(np.max(np.abs(xa - xs)))
    assert da < 2e-5, da                                     # |T x - x|_2 <= 1e-8 at modulus 0.9988: within tol / (1 - modulus) of x*
    assert float(np.max(np.abs(c_oracle(xa) - xa))) < 1e-7
    T.close()


def test_conditional_kernels_gcy20_vs_c_oracle(S, gcy20, c_oracle):
    """The conditional-tensor kernels (slice-dependent z_Q, 25.6 MB at 20^6; Rouwenhorst tensors are slice-identical,
    so SDFS_NO_SLICE_MERGE forces them) against the C oracle on the full grid, T and its residual, and J.v."""
    params, arr = gcy20
    os.environ["SDFS_NO_SLICE_MERGE"] = "1"
    try:
        Tc = S.KoopmansOperator("gcy", SHAPES, params, arr)
    finally:
        del os.environ["SDFS_NO_SLICE_MERGE"]
    assert "pair plan pass" not in Tc.describe_plan()
    w = 400 + 500 * np.random.default_rng(3).random(SHAPES)
    want = c_oracle(w)
    got = Tc(w)
    assert np.max(np.abs(got - want) / want) < 1e-12
    assert Tc.residual() == pytest.approx(float(np.max(np.abs(want - w))), rel=1e-12)
    del got
    v = np.random.default_rng(4).standard_normal(SHAPES)
    jw = c_oracle.jvp(w, v)
    jv = Tc.jvp(w, v)
    assert np.max(np.abs(jv - jw)) <= 1e-11 * np.max(np.abs(jw))
    Tc.close()


def test_sharded_stages_gcy20_eight_way_split(S, gcy20, c_oracle):
    """The stage kernels of sdfs_create_sharded at 20^6 with the real 20-over-8 split: every rank's stage 0 on its
    z-block, the A->B exchange as host slicing, every rank's stage 1 on its h_c-block -- T with residual, linearise +
    J.v, and the mirror orientation -- against the single-GPU handle.  Also sdfs_pack_blocks at these block sizes."""
    import torch
    from sdfs_via_autodiff_amd import distributed as D
    params, arr = gcy20
    world = 8
    A, B = D.SHARD_AXES["gcy"]
    a_sz, b_sz = D.block_sizes(20, world), D.block_sizes(20, world)
    assert a_sz == [3, 3, 3, 3, 2, 2, 2, 2]
    a_off, b_off = D.block_offsets(a_sz), D.block_offsets(b_sz)
    T = S.KoopmansOperator("gcy", SHAPES, params, arr)
    w = 400 + 500 * np.random.default_rng(11).random(SHAPES)
    v = np.random.default_rng(12).standard_normal(SHAPES)
    want_T = T(w)
    want_res = T.residual()
    want_J = T.jvp(w, v)
    T.close()
    # the reference values themselves against the C oracle on the full grid: the test stands on its own
    ref_T = c_oracle(w)
    assert np.max(np.abs(want_T - ref_T) / ref_T) < 1e-12
    ref_J = c_oracle.jvp(w, v)
    assert np.max(np.abs(want_J - ref_J)) < 1e-11 * np.max(np.abs(ref_J))
    dev = torch.device("cuda", 0)

    def sl(axis, lo, n):
        s = [slice(None)] * 6
        s[axis] = slice(lo, lo + n)
        return tuple(s)

    def two_stage(be_of, ax0, off0, sz0, ax1, off1, sz1, mode, x_full, old_full=None):
        """stage 0 on the ax0-blocks, exchange, stage 1 on the ax1-blocks; returns the full result and max resid"""
        mid = np.empty(SHAPES)
        for r in range(world):
            xin = torch.from_numpy(np.ascontiguousarray(x_full[sl(ax0, off0[r], sz0[r])])).to(dev)
            mid[sl(ax0, off0[r], sz0[r])] = be_of(r).run(0, mode, xin).cpu().numpy()
        out = np.empty(SHAPES)
        res_max = 0.0
        for r in range(world):
            zin = torch.from_numpy(np.ascontiguousarray(mid[sl(ax1, off1[r], sz1[r])])).to(dev)
            old = res = None
            if old_full is not None:
                old = torch.from_numpy(np.ascontiguousarray(old_full[sl(ax1, off1[r], sz1[r])])).to(dev)
                if mode != D.MODE_JVP:
                    res = torch.zeros(1, dtype=torch.float64, device=dev)
            y = be_of(r).run(1, mode, zin, old=old, resid=res)
            torch.cuda.synchronize()
            out[sl(ax1, off1[r], sz1[r])] = y.cpu().numpy()
            if res is not None:
                res_max = max(res_max, float(res.item()))
        return out, res_max

    bes = [D.HipStages("gcy", SHAPES, params, arr, A, a_off[r], a_sz[r], B, b_off[r], b_sz[r], 0) for r in range(world)]
    try:
        got, res = two_stage(lambda r: bes[r], A, a_off, a_sz, B, b_off, b_sz, D.MODE_T, w, old_full=w)
        assert np.max(np.abs(got - want_T) / want_T) < 1e-12
        assert res == pytest.approx(want_res, rel=1e-12)
        got, _ = two_stage(lambda r: bes[r], A, a_off, a_sz, B, b_off, b_sz, D.MODE_T_LIN, w)
        assert np.max(np.abs(got - want_T) / want_T) < 1e-12
        got, _ = two_stage(lambda r: bes[r], A, a_off, a_sz, B, b_off, b_sz, D.MODE_JVP, v)
        assert np.max(np.abs(got - want_J)) <= 1e-11 * np.max(np.abs(want_J))
        # pack / unpack at the real block sizes: rank 0's z-block (3 x 20^5) cut along h_c into 3,3,3,3,2,2,2,2
        xin = torch.from_numpy(np.ascontiguousarray(w[sl(A, 0, a_sz[0])])).to(dev)
        packed = torch.empty(xin.numel(), dtype=torch.float64, device=dev)
        bes[0].pack_blocks(xin, packed, B, b_off + [20])
        ref = torch.cat([xin.narrow(B, b_off[j], b_sz[j]).contiguous().reshape(-1) for j in range(world)])
        torch.cuda.synchronize()
        assert torch.equal(packed, ref)
        back = torch.empty_like(xin)
        bes[0].pack_blocks(back, packed, B, b_off + [20], unpack=True)
        torch.cuda.synchronize()
        assert torch.equal(back, xin)
    finally:
        for be in bes:
            be.close()
    # mirror orientation: input sharded on h_c, result sharded on z
    bes = [D.HipStages("gcy", SHAPES, params, arr, B, b_off[r], b_sz[r], A, a_off[r], a_sz[r], 0) for r in range(world)]
    try:
        got, res = two_stage(lambda r: bes[r], B, b_off, b_sz, A, a_off, a_sz, D.MODE_T, w, old_full=w)
        assert np.max(np.abs(got - want_T) / want_T) < 1e-12
        assert res == pytest.approx(want_res, rel=1e-12)
    finally:
        for be in bes:
            be.close()
