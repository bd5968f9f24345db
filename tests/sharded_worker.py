"""
Worker for tests/test_distributed_cpu.py: run under torch.distributed.run with
    sharded_worker.py <model> <shape,comma,separated> <hip|oracle>
Rank 0 prints one line "RESULT {json}".
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

LETTERS = "abcdef"


class OracleStages:
    """numpy twin of HipStages (same interface): stage 0 contracts every axis but A on the
    rank's A-block, stage 1 contracts A on the rank's B-block and applies the aggregator."""

    def __init__(self, model, shapes, params, arrays, axis_a, a_lo, a_len, axis_b, b_lo, b_len):
        from oracle.c_oracle import COperator   # reuse its table/stride bookkeeping (numpy only here)
        meta = COperator.__new__(COperator)
        COperator.__init__(meta, model, shapes, params, arrays)
        self.D = len(shapes)
        self.shapes = tuple(shapes)
        self.theta, self.beta = meta.theta, meta.beta
        self.Q = [np.asarray(q) for q in meta.Q]
        self.qs = meta.qs
        self.order = [int(g) for g in meta.order]
        self.A, self.B = axis_a, axis_b
        self.a_sl, self.b_sl = slice(a_lo, a_lo + a_len), slice(b_lo, b_lo + b_len)
        self.shape0 = list(self.shapes); self.shape0[axis_a] = a_len
        self.shape1 = list(self.shapes); self.shape1[axis_b] = b_len
        # dense per-point tables on the full grid, sliced per stage (tiny test grids only)
        idx = np.indices(self.shapes)
        self.a1_full = meta.a1.ravel()[sum(idx[a] * meta.a1s[a] for a in range(self.D))]
        self.K_full = (meta.a2.ravel()[sum(idx[a] * meta.a2s[a] for a in range(self.D))]
                       * meta.a3.ravel()[sum(idx[a] * meta.a3s[a] for a in range(self.D))])
        self.c1 = self.c2 = None

    def _contract(self, x, g):
        """y[.., i_g, ..] = sum_I Q_g[cond.., i_g, I] x[.., I, ..] with cond = current indices."""
        cond = [c for c in range(self.D) if self.qs[g, c] != 0]
        cond.sort(key=lambda c: -self.qs[g, c])                    # slowest conditioning axis first
        q = self.Q[g].reshape([self.shapes[c] for c in cond] + [self.shapes[g]] * 2)
        sub_x = LETTERS[:self.D].replace(LETTERS[g], "Z")
        sub_q = "".join(LETTERS[c] for c in cond) + LETTERS[g] + "Z"
        return np.einsum(f"{sub_q},{sub_x}->{LETTERS[:self.D]}", q, x)

    def run(self, stage, mode, x, old=None, resid=None, out=None, gate=None, gate_tol=0.0):
        if gate is not None and not (float(gate[0]) > gate_tol):
            # closed gate (sdfs_apply_stage_gated_dev): nothing is written, the residual word stays 0
            if resid is not None:
                resid[0] = 0.0
            assert out is not None
            return out
        res = self._run(stage, mode, x, old, resid)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def _run(self, stage, mode, x, old=None, resid=None):
        x = x.numpy()
        sl0 = [slice(None)] * self.D; sl0[self.A] = self.a_sl
        sl1 = [slice(None)] * self.D; sl1[self.B] = self.b_sl
        if stage == 0:
            a1 = self.a1_full[tuple(sl0)]
            if mode == 1:
                y = self.c1 * x
            else:
                y = a1 * x ** self.theta
                if mode == 2:
                    self.c1 = a1 * x ** (self.theta - 1)
            for g in self.order:
                if g != self.A:
                    y = self._contract(y, g)
            return torch.from_numpy(np.ascontiguousarray(y))
        S = self._contract(x, self.A)
        K = self.K_full[tuple(sl1)]
        if mode == 1:
            out = self.c2 * S
        else:
            out = 1 + self.beta * (K * S) ** (1 / self.theta)
            if mode == 2:
                self.c2 = self.beta * (K * S) ** (1 / self.theta - 1) * K
        if resid is not None and old is not None:
            resid[0] = float(np.max(np.abs(out - old.numpy())))
        return torch.from_numpy(np.ascontiguousarray(out))



def main():
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(int(os.environ.get("SHARDED_WORKER_WATCHDOG", "110")), exit=True)
    model = sys.argv[1]
    shapes = tuple(int(x) for x in sys.argv[2].split(","))
    use_hip = sys.argv[3] == "hip"
    plain = len(sys.argv) > 4 and sys.argv[4] == "plain"      # unperturbed Rouwenhorst tensors: slice-merged plans, mirror schedule
    # SHARDED_WORKER_PG=nccl: the RCCL code path (list-form all_to_all on views of the pack buffers, device-tensor
    # all-reduces, ordering against the handle's stream) -- only with one rank per GPU, i.e. world size 1 on a 1-GPU box
    pg = os.environ.get("SHARDED_WORKER_PG", "gloo")
    if pg == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank = dist.get_rank()
    out = {}
    try:
        big = len(sys.argv) > 4 and sys.argv[4] == "big"
        out = body_big(rank, model, shapes) if big else body(rank, model, shapes, use_hip, plain)
    except Exception as e:
        import traceback
        out = {"error": f"rank {rank}: {e}\n{traceback.format_exc()}"}
    if rank == 0:
        print("RESULT " + json.dumps(out), flush=True)
    dist.destroy_process_group()


def body(rank, model, shapes, use_hip, plain=False):
    if True:
        from sdfs_via_autodiff_amd import distributed as D
        from oracle import models, ssy, gcy, solvers as osol
        if model == "ssy":
            p = models.ssy_params(); arr = ssy.discretize_ssy(p, shapes)
            T, J = (lambda w: ssy.T_ssy_factorised(w, shapes, p, arr)), (lambda w, v: ssy.jvp_ssy(w, v, shapes, p, arr))
        else:
            p = models.gcy_params(); arr = gcy.discretize_gcy(p, shapes)
            T, J = (lambda w: gcy.T_gcy_factorised(w, shapes, p, arr)), (lambda w, v: gcy.jvp_gcy(w, v, shapes, p, arr))
        arr = list(arr)
        rng = np.random.default_rng(7)
        for i in (() if plain else ((7,) if model == "ssy" else (1, 3))):   # make conditional tensors differ per slice
            qq = rng.random(arr[i].shape) + 0.05
            arr[i] = qq / qq.sum(axis=-1, keepdims=True)
        if use_hip:
            torch.cuda.set_device(0)
        op = D.ShardedKoopmans(model, shapes, p, arr, backend_factory=None if use_hip else OracleStages)
        dev = "cuda" if use_hip else "cpu"
        w = 400 + 500 * np.random.default_rng(0).random(shapes)
        v = np.random.default_rng(1).standard_normal(shapes)
        w_loc = op.scatter_from_full(torch.from_numpy(w)).to(dev)
        v_loc = op.scatter_from_full(torch.from_numpy(v)).to(dev)
        out = {}
        Tw = op.gather_full(op.apply_T(w_loc)).cpu().numpy()
        out["T"] = float(np.max(np.abs(Tw - T(w)) / np.abs(T(w))))
        Tl = op.gather_full(op.linearize(w_loc)).cpu().numpy()
        out["Tlin"] = float(np.max(np.abs(Tl - T(w)) / np.abs(T(w))))
        jv = op.gather_full(op.jvp(v_loc)).cpu().numpy()
        ref = J(w, v)
        out["jvp"] = float(np.max(np.abs(jv - ref)) / np.max(np.abs(ref)))
        out["resid"] = abs(op.sup_norm_diff(op.apply_T(w_loc), w_loc) - np.max(np.abs(T(w) - w)))
        out["sizes"] = (op.a_sizes, op.b_sizes)
        if max(out["T"], out["Tlin"], out["jvp"]) > 1e-6:      # wrong operator: do not start the solver loops
            return out
        # distributed Newton (tight inner tolerance) vs the oracle's polished fixed point
        nst = {}
        x_loc, n = D.newton_sharded(op, op.scatter_from_full(torch.full(shapes, 800.0, dtype=torch.float64)).to(dev),
                                    tol=1e-10, max_iter=30, inner_rtol=1e-9, inner_atol=0.0, stats=nst)
        x = op.gather_full(x_loc).cpu().numpy()
        if use_hip:
            # device-gated BiCGSTAB: one read of the scalar block per chunk of iterations; chunk = 1 gives the same iterates
            out["newton_krylov_syncs"] = nst.get("krylov_host_syncs", -1)
            out["newton_krylov_iters"] = nst.get("krylov_iters", -1)
            op._krylov.chunk = 1
            nst1 = {}
            x1_loc, n1 = D.newton_sharded(op, op.scatter_from_full(torch.full(shapes, 800.0, dtype=torch.float64)).to(dev),
                                          tol=1e-10, max_iter=30, inner_rtol=1e-9, inner_atol=0.0, stats=nst1)
            op._krylov.chunk = 8
            out["newton_chunk1_diff"] = float((x1_loc - x_loc).abs().max().item())
            out["newton_chunk1_iters"] = (n1, nst1.get("krylov_iters", -1))
        xs = osol.newton_polish(T, J, x.copy())
        out["newton_err"] = float(np.max(np.abs(x - xs)))
        out["newton_iters"] = n
        if use_hip:
            # BASELINE config 5 sharded: fp32 Krylov vectors, J.v streams and exchange buffers, fp64 outer residual
            x_loc32, n32 = D.newton_sharded(op, op.scatter_from_full(torch.full(shapes, 800.0, dtype=torch.float64)).to(dev),
                                            tol=1e-10, max_iter=30, inner_rtol=1e-6, inner_atol=0.0, krylov_f32=True)
            out["newton_f32_err"] = float(np.max(np.abs(op.gather_full(x_loc32).cpu().numpy() - xs)))
            out["newton_f32_iters"] = n32
        # distributed SA: same iteration count as the single-process oracle loop
        out["mirror_ok"] = bool(op.mirror_ok)
        errs, stats = [], {}
        x0 = op.n_exchanges
        sa_tol, sa_max = (2e-2, 4000) if plain else (1e-3, 120)
        ce = 7
        w0 = op.scatter_from_full(torch.full(shapes, 800.0, dtype=torch.float64)).to(dev)
        w0_before = w0.clone()
        xa_loc, na = D.successive_approx_sharded(op, w0, tol=sa_tol, max_iter=sa_max, errors=errs, stats=stats, check_every=ce)
        out["sa_exchanges"] = op.n_exchanges - x0
        # ownership: the caller's input is untouched, and the result is no buffer the operator (or a later solve) writes --
        # it survives an application, a J.v and a solve from another start (with and without the mirror phase)
        out["sa_input_kept"] = bool(torch.equal(w0, w0_before))
        xa_snapshot = xa_loc.clone()
        op.apply_T(w_loc); op.jvp(v_loc)
        for mir in (True, False):
            D.successive_approx_sharded(op, op.scatter_from_full(torch.full(shapes, 650.0, dtype=torch.float64)).to(dev),
                                        tol=sa_tol, max_iter=min(sa_max, 9), check_every=4, mirror=mir)
        w1 = op.scatter_from_full(torch.full(shapes, 800.0, dtype=torch.float64)).to(dev)
        w1_before = w1.clone()
        D.successive_approx_sharded(op, w1, tol=sa_tol, max_iter=5, check_every=2, mirror=False)
        out["sa_input_kept"] = out["sa_input_kept"] and bool(torch.equal(w1, w1_before))
        out["sa_result_kept"] = bool(torch.equal(xa_loc, xa_snapshot))
        out["sa_mirror_iters"] = stats.get("mirror_iters", 0)
        out["sa_host_syncs"] = stats.get("host_syncs", -1)
        out["sa_check_every"] = ce
        out["sa_n_errors"] = len(errs)
        xo, no = osol.successive_approx(T, np.full(shapes, 800.0), tol=sa_tol, max_iter=sa_max, verbose=False)
        out["sa_iters"] = (na, no)
        out["sa_err"] = float(np.max(np.abs(op.gather_full(xa_loc).cpu().numpy() - xo)))
        # one-iteration-per-read form of the same loop: identical iterates and iteration count
        xb_loc, nb = D.successive_approx_sharded(op, op.scatter_from_full(torch.full(shapes, 800.0, dtype=torch.float64)).to(dev),
                                                 tol=sa_tol, max_iter=sa_max, check_every=1)
        out["sa_check1"] = (nb, float(np.max(np.abs(op.gather_full(xb_loc).cpu().numpy() - op.gather_full(xa_loc).cpu().numpy()))))
        # Anderson on the sharded grid against the single-process oracle loop (jaxopt semantics restated, unpinned):
        # same iteration count and iterate on these small, well-conditioned histories
        a_tol = 1e-6
        st = {}
        a_ce = 5
        xa2, n_and = D.anderson_sharded(op, op.scatter_from_full(torch.full(shapes, 800.0, dtype=torch.float64)).to(dev),
                                        tol=a_tol, max_iter=3000, stats=st, check_every=a_ce)
        if use_hip:
            out["anderson_host_syncs"] = st.get("host_syncs", -1)
            out["anderson_check_every"] = a_ce
            xa3, n_and1 = D.anderson_sharded(op, op.scatter_from_full(torch.full(shapes, 800.0, dtype=torch.float64)).to(dev),
                                             tol=a_tol, max_iter=3000, check_every=1)
            out["anderson_check1"] = (n_and1, float((xa3 - xa2).abs().max().item()))
        xo2, n_ando = osol.anderson_solver(T, np.full(shapes, 800.0), tol=a_tol, max_iter=3000, verbose=False)
        out["anderson_iters"] = (n_and, n_ando)
        out["anderson_err"] = float(np.max(np.abs(op.gather_full(xa2).cpu().numpy() - xo2)))
        out["anderson_resid"] = float(np.max(np.abs(T(op.gather_full(xa2).cpu().numpy()) - op.gather_full(xa2).cpu().numpy())))
        out["anderson_rejected"] = st.get("rejected_mixes", -1)
        return out


def body_big(rank, model, shapes):
    """Grids of the compile-time pair plan (GCY, extents 16 / 20 / 24 / 32 in pairs): the stages run the pair plan's
    kernels on their blocks (sdfs_api.hip, build_stage_fast_plans).  Single applications against the C oracle on the
    full grid; the sharded loops against the single-GPU library on the same start (the oracle's loops would take
    minutes at this size)."""
    import sdfs_via_autodiff_amd as S
    from sdfs_via_autodiff_amd import distributed as D
    from oracle.c_oracle import COperator
    assert model == "gcy"
    torch.cuda.set_device(0)
    m = S.GCY()
    arr = S.discretize_gcy(m, shapes)
    op = D.ShardedKoopmans(model, shapes, m.params, arr)
    out = {"plan": op.backend.describe_plan(), "mirror_ok": bool(op.mirror_ok), "sizes": (op.a_sizes, op.b_sizes)}
    if op.backend_m is not None:
        out["plan_mirror"] = op.backend_m.describe_plan()
    co = COperator(model, shapes, m.params, arr)
    w = 400 + 500 * np.random.default_rng(0).random(shapes)
    v = np.random.default_rng(1).standard_normal(shapes)
    w_loc = op.scatter_from_full(torch.from_numpy(w)).cuda()
    v_loc = op.scatter_from_full(torch.from_numpy(v)).cuda()
    want = co(w)
    out["T"] = float(np.max(np.abs(op.gather_full(op.apply_T(w_loc)).cpu().numpy() - want) / want))
    out["Tlin"] = float(np.max(np.abs(op.gather_full(op.linearize(w_loc)).cpu().numpy() - want) / want))
    ref = co.jvp(w, v)
    out["jvp"] = float(np.max(np.abs(op.gather_full(op.jvp(v_loc)).cpu().numpy() - ref)) / np.max(np.abs(ref)))
    out["resid"] = abs(op.sup_norm_diff(op.apply_T(w_loc), w_loc) - np.max(np.abs(want - w))) / np.max(np.abs(want - w))
    if max(out["T"], out["Tlin"], out["jvp"]) > 1e-6:
        return out
    del want, ref
    # the fixed point, from the single-GPU library (every rank computes its own copy: a third of a second at 16^6)
    one = S.KoopmansOperator(model, shapes, m.params, arr)
    xs, ns, _ = one.solve(np.full(shapes, 800.0), "newton", tol=1e-10, inner_rtol=1e-6, inner_atol=0.0, max_iter=30)
    near = xs + 3e-5 * (np.random.default_rng(5).random(shapes) - 0.5)            # ~25 SA iterations from tol 1e-6
    xa, na, _ = one.solve(near, "successive_approx", tol=1e-6, max_iter=500)
    x24, _, _ = one.solve(np.full(shapes, 800.0), "successive_approx", tol=0.0, max_iter=24)
    one.close()
    scat = lambda a: op.scatter_from_full(torch.from_numpy(np.ascontiguousarray(a))).cuda()   # noqa: E731
    full = lambda x_loc: op.gather_full(x_loc).cpu().numpy()                      # noqa: E731
    # (the exchanges of this test go through gloo and host memory, 0.1-0.2 s each at this size: short loops)
    # successive approximation with the mirror schedule (both orientations of the stage plans), gated on the device
    st = {}
    x_loc, n = D.successive_approx_sharded(op, scat(near), tol=1e-6, max_iter=500, stats=st, check_every=8)
    out["sa"] = (n, na, float(np.max(np.abs(full(x_loc) - xa))), st.get("mirror_iters", 0), st.get("host_syncs", -1))
    x_loc, n = D.successive_approx_sharded(op, scat(near), tol=1e-6, max_iter=500, check_every=8, mirror=False)
    out["sa_exact"] = (n, na, float(np.max(np.abs(full(x_loc) - xa))))
    # opts.t_f32 on the sharded grid: fp32 intermediates between the stages (half the bytes per exchange) while the step is
    # far above their resolution -- 24 iterations from the reference's start, against the single-GPU fp64 iterate
    st = {}
    x_loc, n = D.successive_approx_sharded(op, scat(np.full(shapes, 800.0)), tol=1e-8, max_iter=24, check_every=8, stats=st, t_f32=True)
    out["sa_t32"] = (n, st.get("t32_iters", -1), float(np.max(np.abs(full(x_loc) - x24))))
    del x24
    if dist.get_world_size() > 2:
        return out
    # Newton-Krylov from the reference's start, device-gated BiCGSTAB chunks (loose inner solves: ~100 J.v applications)
    nst = {}
    x_loc, n = D.newton_sharded(op, scat(np.full(shapes, 800.0)), tol=1e-8, max_iter=30, inner_rtol=1e-2, inner_atol=0.0, stats=nst)
    out["newton"] = (n, float(np.max(np.abs(full(x_loc) - xs))), nst.get("krylov_host_syncs", -1), nst.get("krylov_iters", -1))
    # Anderson, device-resident state: the residual of what it returns
    st = {}
    x_loc, n = D.anderson_sharded(op, scat(near), tol=1e-7, max_iter=500, stats=st, check_every=8)
    out["anderson"] = (n, float(op.sup_norm_diff(op.apply_T(x_loc), x_loc)), float(np.max(np.abs(full(x_loc) - xs))), st.get("host_syncs", -1))
    return out


if __name__ == "__main__":
    main()
