"""
world_size-2 gloo tests of the multi-GPU layer: the sharding schedule
(stage 0 -> re-shard -> stage 1 -> re-shard), uneven blocks, the all-reduced residual and
the distributed Newton-Krylov / successive-approximation loops.

CPU tests: the local stages are computed by a numpy ORACLE backend (tests/sharded_worker.py),
so what is checked is the host logic of sdfs_via_autodiff_amd/distributed.py.
GPU test (-m gpu): two ranks on one MI355X run the real HIP stage kernels through sharded
C-ABI handles (exchanges staged through gloo).
Workers run under torch.distributed.run in a subprocess with a hard timeout.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def run_world2(model, shapes, backend, timeout=240, world=2, plain=False, pg=None, big=False):
    env = dict(os.environ)
    if pg:
        env["SHARDED_WORKER_PG"] = pg
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(REPO, "tests", "sharded_worker.py"), model, ",".join(map(str, shapes)), backend] + \
          (["plain"] if plain else []) + (["big"] if big else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=REPO, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert lines, f"no result (rc {r.returncode})\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    out = json.loads(lines[-1][7:])
    assert "error" not in out, out.get("error")
    return out


@pytest.mark.parametrize("model,shapes", [("ssy", (3, 5, 4, 3)), ("gcy", (3, 3, 2, 3, 2, 4))])
def test_sharded_operator_world2_gloo(model, shapes):
    out = run_world2(model, shapes, "oracle")
    assert out["T"] < 1e-13 and out["Tlin"] < 1e-13 and out["jvp"] < 1e-12, out
    assert out["resid"] < 1e-9
    a_sizes, b_sizes = out["sizes"]
    assert sum(a_sizes) == shapes[0] and max(a_sizes) - min(a_sizes) <= 1
    assert out["newton_err"] < 1e-8 and out["newton_iters"] < 20, out
    assert out["sa_iters"][0] == out["sa_iters"][1] and out["sa_err"] < 1e-9, out
    check_sa_gating_and_anderson(out)


def check_device_gated_newton_and_anderson(out):
    """HIP stage backend: BiCGSTAB reads its scalar block once per chunk of iterations (HipKrylov.chunk = 8), not per
    iteration, with the iterates of the one-iteration-per-read loop; Anderson reads its device state every check_every
    passes."""
    ks, ki, ns = out["newton_krylov_syncs"], out["newton_krylov_iters"], out["newton_iters"]
    assert ki > 0 and ks <= ki // 8 + 2 * ns, out            # per solve: one read behind INIT_FIN + one per chunk
    assert out["newton_chunk1_diff"] == 0.0 and out["newton_chunk1_iters"] == [ns, ki], out
    na = out["anderson_iters"][0]
    assert out["anderson_host_syncs"] <= na // out["anderson_check_every"] + 2, out
    assert out["anderson_check1"] == [na, 0.0], out          # chunked loop = one-pass-per-read loop, bit for bit


def check_sa_gating_and_anderson(out):
    """Device-gated loop: one host read per check_every iterations (+ one per phase), the same iterates as the
    one-read-per-iteration form; Anderson: the oracle's fixed point, and its iteration count within a quarter."""
    na = out["sa_iters"][0]
    assert out["sa_host_syncs"] <= na // out["sa_check_every"] + 3, out
    assert out["sa_n_errors"] == na, out
    assert out["sa_check1"][0] == na and out["sa_check1"][1] == 0.0, out
    assert out["sa_input_kept"] and out["sa_result_kept"], out
    n_and, n_ando = out["anderson_iters"]
    assert out["anderson_resid"] < 1e-5, out
    # (the iteration path follows the last bits of the ill-conditioned Gram matrix, which the sharded loop adds in an
    # order of its own -- one sweep per solve: the count agrees to a quarter, the fixed point to the tolerance)
    assert abs(n_and - n_ando) <= max(2, n_ando // 4) and out["anderson_err"] < 1e-4, out


def test_sharded_operator_world4_uneven_blocks():
    """20-over-8 style uneven blocks in miniature: 5 and 6 states over 4 ranks."""
    out = run_world2("gcy", (5, 2, 2, 6, 2, 3), "oracle", world=4)
    assert out["T"] < 1e-13 and out["Tlin"] < 1e-13 and out["jvp"] < 1e-12, out
    assert out["sizes"] == [[2, 1, 1, 1], [2, 2, 1, 1]]
    assert out["newton_err"] < 1e-8 and out["sa_iters"][0] == out["sa_iters"][1]
    check_sa_gating_and_anderson(out)


@pytest.mark.parametrize("model,shapes,world", [("ssy", (4, 5, 3, 3), 2), ("gcy", (4, 2, 2, 5, 2, 3), 2),
                                                ("gcy", (5, 2, 2, 6, 2, 3), 4)])
def test_mirror_schedule_one_exchange_per_iteration(model, shapes, world):
    """Unperturbed Rouwenhorst tensors: the mirror schedule (iterate alternating between the A- and the B-sharded
    layout) runs successive approximation with ONE exchange per iteration, switches to the fixed-layout form for
    the last iterations and stops on the reference's iteration with the reference's iterate."""
    out = run_world2(model, shapes, "oracle", world=world, plain=True, timeout=400)
    assert out["T"] < 1e-13 and out["Tlin"] < 1e-13 and out["jvp"] < 1e-12, out
    assert out["mirror_ok"]
    na, no = out["sa_iters"]
    assert na == no and out["sa_err"] < 1e-9, out
    m = out["sa_mirror_iters"]
    assert m > 0.5 * na, out                                   # most iterations ran in mirror form
    # mirror iterations: one exchange each; the switch back to layout A: at most one; exact iterations: two each
    # (the first exact iteration re-shards w as well)
    # (the gated loop enqueues whole chunks: up to check_every - 1 no-op iterations, with their exchanges, per phase)
    ce = out["sa_check_every"]
    assert out["sa_exchanges"] <= m + (ce - 1) + 1 + 2 * (na - m + ce - 1) + 1, out
    check_sa_gating_and_anderson(out)


def test_block_sizes():
    from sdfs_via_autodiff_amd.distributed import block_sizes, block_offsets
    assert block_sizes(20, 8) == [3, 3, 3, 3, 2, 2, 2, 2]
    assert block_offsets(block_sizes(20, 8)) == [0, 3, 6, 9, 12, 14, 16, 18]
    assert block_sizes(16, 8) == [2] * 8 and block_sizes(5, 2) == [3, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("model,shapes", [("ssy", (5, 7, 6, 4)), ("gcy", (5, 3, 2, 5, 3, 6))])
def test_sharded_hip_stages_world2(model, shapes):
    """Two ranks on one GPU (gloo for the exchanges): the real HIP stage kernels with
    sharded handles, offsets into the scale tables and uneven blocks."""
    out = run_world2(model, shapes, "hip", timeout=150)
    assert "newton_err" in out, out
    assert out["T"] < 1e-12 and out["Tlin"] < 1e-12 and out["jvp"] < 1e-11, out
    assert out["resid"] < 1e-8
    assert out["newton_err"] < 1e-8 and out["newton_iters"] < 20, out
    assert out["newton_f32_err"] < 1e-8 and out["newton_f32_iters"] < 25, out      # fp32 Krylov storage, fp64 fixed point
    assert out["sa_iters"][0] == out["sa_iters"][1] and out["sa_err"] < 1e-8, out
    check_device_gated_newton_and_anderson(out)
    check_sa_gating_and_anderson(out)


@pytest.mark.gpu
@pytest.mark.parametrize("model,shapes", [("ssy", (5, 7, 6, 4)), ("gcy", (4, 3, 2, 5, 3, 6))])
def test_sharded_hip_stages_plain_tensors_mirror(model, shapes):
    """Default (slice-identical) tensors: the slice-merged stage plans (a3 as a table with block offsets) and the
    mirror schedule on the real HIP stage kernels."""
    out = run_world2(model, shapes, "hip", timeout=200, plain=True)
    assert "newton_err" in out, out
    assert out["T"] < 1e-12 and out["Tlin"] < 1e-12 and out["jvp"] < 1e-11, out
    assert out["resid"] < 1e-8 and out["newton_err"] < 1e-8
    assert out["newton_f32_err"] < 1e-8, out
    assert out["mirror_ok"]
    na, no = out["sa_iters"]
    assert na == no and out["sa_err"] < 1e-8, out
    assert out["sa_mirror_iters"] > 0.5 * na, out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_stages_on_the_pair_plan_kernels(world):
    """GCY 16^6 over two and three ranks (blocks of 8, and of 6 / 5 / 5) on one GPU: both stages of both orientations run
    the pair plan's slice / line kernels on their blocks.  Single applications against the C oracle; Newton-Krylov,
    successive approximation (mirror schedule and exact) and Anderson against the single-GPU library's fixed point and
    iterates (/root/reference/code/solvers.py:19-124 are the loops; code/gcy/discrete/gcy_wc_ratio.py:230-236 the operator)."""
    shapes = (16,) * 6
    os.environ["SHARDED_WORKER_WATCHDOG"] = "280"
    try:
        out = run_world2("gcy", shapes, "hip", timeout=300, world=world, big=True)
    finally:
        os.environ.pop("SHARDED_WORKER_WATCHDOG", None)
    assert "stage 0 pair-plan pass 0: slices[" in out["plan"] and "stage 1 pair-plan pass 0: lines[" in out["plan"], out["plan"]
    assert out["mirror_ok"] and "stage 1 pair-plan pass 0: lines[" in out["plan_mirror"], out
    assert sum(out["sizes"][0]) == 16
    assert out["T"] < 1e-12 and out["Tlin"] < 1e-12 and out["jvp"] < 1e-11 and out["resid"] < 1e-10, out
    n, na, err, mirror_iters, syncs = out["sa"]
    assert abs(n - na) <= 1 and err < 1e-6 and mirror_iters > 0.5 * n and syncs <= n / 8 + 4, out["sa"]   # (the mirror phase stops on a two-step difference)
    n, na, err = out["sa_exact"]
    assert n == na and err < 1e-9, out["sa_exact"]
    # fp32 intermediates between the stages (sdfs_set_t_f32): all 24 iterations ran in that form; one stored float carries
    # 2^-24 relative, ~ w 2^-24 / |theta| = 1.3e-6 on T w per application
    n, n32, d = out["sa_t32"]
    assert n == 24 and n32 == 24 and 0.0 < d < 1e-4, out["sa_t32"]
    if world > 2:
        return
    n, err, syncs, its = out["newton"]
    assert n < 20 and err < 1e-7, out["newton"]
    assert 0 < syncs <= its / 8 + 2 * n + 2, out["newton"]
    n, r, err, syncs = out["anderson"]
    # (with jaxopt's absolute ridge of 1e-5 the mixing stalls once the residuals are this small -- 348 iterations here,
    # DESIGN 4.2; what is held is that the loop ends at the fixed point with one state read per check_every passes)
    assert r < 2e-7 and err < 1e-5 and syncs <= n / 8 + 3, out["anderson"]


@pytest.mark.gpu
def test_sharded_hip_stages_world4_uneven():
    out = run_world2("gcy", (5, 3, 2, 6, 3, 4), "hip", timeout=200, world=4)
    assert out["T"] < 1e-12 and out["Tlin"] < 1e-12 and out["jvp"] < 1e-11, out
    assert out["sizes"] == [[2, 1, 1, 1], [2, 2, 1, 1]]
    assert out["newton_err"] < 1e-8, out


@pytest.mark.gpu
@pytest.mark.parametrize("model,shapes", [("gcy", (4, 3, 2, 5, 3, 6)), ("ssy", (5, 7, 6, 4))])
def test_sharded_layer_on_rccl_world1(model, shapes):
    """The "nccl" (= RCCL) branch of the multi-GPU layer has never had more than one GPU to run on.  With a
    world of ONE rank it still executes on the real backend: init with device_id, the list-form all_to_all on views
    of the pack / receive buffers, device-tensor all-reduces and their ordering against the handle's stream -- the
    whole operator, Newton, the gated SA loop (mirror + exact phase) and Anderson.  What it cannot show is a link."""
    out = run_world2(model, shapes, "hip", timeout=200, world=1, plain=True, pg="nccl")
    assert out["T"] < 1e-12 and out["Tlin"] < 1e-12 and out["jvp"] < 1e-11, out
    assert out["resid"] < 1e-8 and out["newton_err"] < 1e-8, out
    assert out["mirror_ok"]
    na, no = out["sa_iters"]
    assert na == no and out["sa_err"] < 1e-8, out
    check_sa_gating_and_anderson(out)
    check_device_gated_newton_and_anderson(out)


@pytest.mark.gpu
def test_pack_blocks_kernel():
    """sdfs_pack_blocks (one launch per re-shard) against torch's narrow + contiguous: fp64 and fp32, uneven
    blocks, inner runs that are and are not multiples of 16 bytes, first / middle / last axis, and the way back."""
    import torch
    import sdfs_via_autodiff_amd as S
    from sdfs_via_autodiff_amd.distributed import HipStages, block_sizes, block_offsets
    m = S.GCY(); shapes = (4, 3, 2, 5, 3, 6)
    be = HipStages("gcy", shapes, m.params, S.discretize_gcy(m, shapes), 0, 0, 2, 3, 0, 3, 0)
    gen = torch.Generator().manual_seed(0)
    for shp, axis, world, dt in (((3, 4, 5, 7, 2, 6), 3, 4, torch.float64), ((6, 20, 5), 1, 8, torch.float64),
                                 ((2, 3, 9, 3), 2, 2, torch.float32), ((5, 4, 3), 2, 3, torch.float64),
                                 ((7, 4, 4), 0, 3, torch.float32), ((2, 16, 3, 5), 1, 16, torch.float64)):
        x = torch.rand(shp, generator=gen, dtype=torch.float64).to(dt).cuda()
        sizes = block_sizes(shp[axis], world); offs = block_offsets(sizes) + [shp[axis]]
        packed = torch.empty(x.numel(), dtype=dt, device="cuda")
        be.pack_blocks(x, packed, axis, offs)
        want = torch.cat([x.narrow(axis, offs[j], sizes[j]).contiguous().reshape(-1) for j in range(world)])
        torch.cuda.synchronize()
        assert torch.equal(packed, want), (shp, axis, world)
        back = torch.zeros_like(x)
        be.pack_blocks(back, packed, axis, offs, unpack=True)
        torch.cuda.synchronize()
        assert torch.equal(back, x), (shp, axis, world)
        # unpack with the Krylov operator's "- v" folded in (sdfs_unpack_blocks_sub)
        sub = torch.rand(shp, generator=gen, dtype=torch.float64).to(dt).cuda()
        back2 = torch.zeros_like(x)
        assert be.unpack_blocks_sub(back2, packed, sub, axis, offs)
        torch.cuda.synchronize()
        assert torch.equal(back2, x - sub), (shp, axis, world)
    be.close()
