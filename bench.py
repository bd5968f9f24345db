#!/usr/bin/env python3
"""
bench.py -- fixed-point iterations/sec of the wealth-consumption-ratio operator on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload gcy20|gcy16|ssy15] [--no-cpu]

A "step" is one successive-approximation iteration: one application of the Koopmans
operator T (expectation passes + fused Epstein-Zin aggregator) with the sup-norm
residual max|Tw - w| fused into the last kernel -- the body of the reference's hot loop
(code/solvers.py:34-36).  Inputs are resident in HBM when the timed region starts.
Default workload: GCY (20,)*6 fp64 (BASELINE.json configs[3] grid; the grid the
north_star roofline/CPU targets are quoted on), synthetic w = 400 + 500*U(0,1).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel: algorithmic bytes per launch / mean launch duration, measured with HIP events on
               the library's stream in a second loop of the same steps right behind the timed one (the timed loop
               itself runs without the two hipEventRecord per launch); also the in-run rate of a
               plain device copy of the same grid, so that the fraction of spec AND of this box's ceiling show
  cpu_baseline the oracle's C/OpenMP port of the same operator, timed on this box's cores
  secondary    time-to-converge runs, each GPU solve with its CPU twin (oracle solver through the C operator;
               bounded sample + stated extrapolation where a full CPU solve would take minutes)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "gcy20": ("gcy", (20,) * 6),
    "gcy16": ("gcy", (16,) * 6),
    "gcy12": ("gcy", (12,) * 6),
    "ssy15": ("ssy", (15,) * 4),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 vector = matrix peak (2.4 GHz x 256 CUs x 128 flop/clk)


def build_model(S, model, shapes):
    if model == "ssy":
        m = S.SSY()
        return m.params, S.discretize_ssy(m, shapes)
    m = S.GCY()
    return m.params, S.discretize_gcy(m, shapes)


def cpu_baseline(model, shapes, params, arrays, w_host, budget_s=12.0):
    """Oracle C/OpenMP port, bounded sample: as many full applications as fit ~budget_s."""
    from oracle.c_oracle import COperator, num_threads
    op = COperator(model, shapes, params, arrays)
    op(w_host)                                   # warm-up (page faults, thread pool)
    n, t0 = 0, time.perf_counter()
    while True:
        op(w_host)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 50:
            break
    # the port makes D axis sweeps and two power sweeps, each reading and writing the grid once (16 B / point)
    sweep_bytes = 16.0 * (len(shapes) + 2) * float(np.prod(shapes))
    return {"value": n / dt, "unit": "iterations/s", "cores": num_threads(), "host_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"{n} applications of T on the same {'x'.join(map(str, shapes))} grid "
                      f"(oracle/c/wc_oracle.c, factorised, OpenMP), {dt:.1f} s",
            "dram_GBps_of_its_own_sweeps": sweep_bytes * n / dt / 1e9}


def cpu_time_to_converge(model, shapes, params, arrays, gpu_iters, gpu_applies, full, budget_s=10.0):
    """The CPU twin of a Newton-Krylov time-to-converge entry: the oracle's newton_solver (oracle/solvers.py, which
    follows code/solvers.py:51-95) through the C/OpenMP operator.  full = True runs the whole solve; otherwise a
    bounded sample times T and J.v on the grid and the solve is EXTRAPOLATED with the application counts of the
    GPU solve (one T + one residual per Newton step, two J.v per BiCGSTAB iteration; the C oracle's J.v
    re-linearises at every call, as jax.jvp does in the reference)."""
    from oracle.c_oracle import COperator, num_threads
    from oracle import solvers as osol
    op = COperator(model, shapes, params, arrays)
    w0 = np.full(shapes, 800.0)
    if full:
        stats = {}
        t0 = time.perf_counter()
        x, n = osol.newton_solver(op, w0, tol=1e-8, bicgstab_tol=1e-6, bicgstab_atol=0.0, verbose=False, jvp=op.jvp, stats=stats)
        t = time.perf_counter() - t0
        return {"kind": "measured", "seconds": t, "iterations": n, "matvecs": stats.get("matvecs"), "cores": num_threads(),
                "final_step": float(np.max(np.abs(op(x) - x)))}
    w = 400 + 500 * np.random.default_rng(0).random(shapes)
    v = np.random.default_rng(1).standard_normal(shapes)
    op(w); op.jvp(w, v)
    nT = nJ = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s / 2 and nT < 20:
        op(w); nT += 1
    tT = (time.perf_counter() - t0) / nT
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s / 2 and nJ < 20:
        op.jvp(w, v); nJ += 1
    tJ = (time.perf_counter() - t0) / nJ
    n_jv = max(gpu_applies - gpu_iters, 0)
    return {"kind": "extrapolated", "seconds": gpu_iters * tT + n_jv * tJ, "cores": num_threads(),
            "sample": f"{nT} T and {nJ} J.v applications timed ({tT:.3f} s, {tJ:.3f} s each); solve = {gpu_iters} T + "
                      f"{n_jv} J.v, the counts of the GPU solve beside it"}


def cpu_literal_apply():
    """The reference's own formulation (8-D broadcast product summed over the next-state axes,
    code/ssy/discrete/ssy_wc_ratio.py:143-145) restated in numpy (oracle/ssy.py: T_ssy), one application at SSY 10^4 --
    where it still fits in memory (10^8 doubles per temporary); O(N^2), so it does not exist at the bench grid."""
    from oracle import models, ssy
    shp = (10,) * 4
    p = models.ssy_params(); arr = ssy.discretize_ssy(p, shp)
    w = 400 + 500 * np.random.default_rng(0).random(shp)
    t0 = time.perf_counter()
    ssy.T_ssy(w, shp, p, arr)
    t_lit = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(20):
        ssy.T_ssy_factorised(w, shp, p, arr)
    t_fac = (time.perf_counter() - t0) / 20
    return {"grid": "SSY 10x10x10x10", "literal_numpy_s_per_apply": t_lit, "factorised_numpy_s_per_apply": t_fac,
            "note": "literal = the reference's O(N^2) broadcast formulation, single numpy thread pool"}


def newton_roofline(op, w_host, N, krylov_f32, inner=1e-6):
    """Per-kernel algorithmic bytes / HIP-event time of one more Newton-Krylov solve with events around every launch
    (code/solvers.py:51-95; configs[3]'s algorithm): the three passes of J.v, the fused BLAS-1 group of BiCGSTAB
    (stream counts below, over the group's summed kernel time),
    the linearising applications of T, and one BiCGSTAB iteration as a whole against the HBM peak."""
    op.set_profiling(True)
    op.reset_counters()
    x, n, info = op.solve(w_host, "newton", tol=1e-8, inner_rtol=inner, inner_atol=0.0, krylov_f32=krylov_f32)
    cs = [c for c in op.counters() if c["launches"]]
    op.set_profiling(False)
    del x
    # J.v applications = launches of the FIRST pass of each J.v kernel family that ran (fp64 "jvp:", fp32 storage "jvp32:",
    # fp32 MFMA "jvpm32:"; a solve with reduced storage redoes the odd step in fp64); two per BiCGSTAB iteration
    fam = {}
    for c in cs:
        tag = c["name"].split(":", 1)[0]
        if tag in ("jvp", "jvp32", "jvpm32"):
            fam.setdefault(tag, c["launches"])
    iters = sum(fam.values()) / 2.0
    it64 = fam.get("jvp", 0) / 2.0                 # iterations on fp64 vectors (all of them, or the odd steps redone)
    fused = any(c["name"].startswith("jvp+") for c in cs)
    # grid streams of one iteration: 2 J.v of 9 (3 + 2 + 4) and 16 of BLAS-1 (p update 4, <rhat, q> 2,
    # s update 3, x / r update 7; <t, s> and <t, t> come out of J.v's last pass) = 34; with the p and s updates in J.v's
    # first pass and <rhat, q> in its last (csrc/krylov_kernels.hpp) 6 + 2 + 5, 5 + 2 + 4 and the x / r update's 7 = 31
    s64, b64 = (31, 7) if fused else (34, 16)
    fused32 = any(c["name"].startswith("jvpm32+") for c in cs)       # (the same on the fp32-MFMA first pass)
    s32, b32 = (31, 7) if fused32 else (34, 16)
    it_bytes_sum = (it64 * s64 * 8.0 + (iters - it64) * s32 * 4.0) * N
    b1_bytes_sum = (it64 * b64 * 8.0 + (iters - it64) * b32 * 4.0) * N
    out = {"kernels": [], "newton_steps": n, "bicgstab_iterations": iters, "bicgstab_iterations_fp64": it64,
           "updates_fused_into_jvp": fused}
    t_jv = t_b1 = 0.0
    for c in cs:
        avg = c["total_ms"] / c["launches"]
        row = {"name": c["name"], "launches": c["launches"], "avg_ms": avg, "total_ms": c["total_ms"]}
        if c["alg_bytes"] > 0:
            row["alg_GB"] = c["alg_bytes"] / 1e9
            row["GBps"] = c["alg_bytes"] / (avg * 1e-3) / 1e9
            row["frac_hbm"] = row["GBps"] / HBM_PEAK_GBS
        if c["name"].startswith(("jvp", "jvp32", "jvpm32")):
            t_jv += c["total_ms"]
        if c["name"] == "bicgstab_blas1":
            t_b1 = c["total_ms"]
            row["alg_GB_per_iteration"] = b1_bytes_sum / max(iters, 1) / 1e9
            row["ms_per_iteration"] = t_b1 / max(iters, 1)
            row["GBps"] = b1_bytes_sum / (t_b1 * 1e-3) / 1e9
            row["frac_hbm"] = row["GBps"] / HBM_PEAK_GBS
        out["kernels"].append(row)
    if iters > 0:
        it_ms = (t_jv + t_b1) / iters
        out["iteration"] = {"alg_GB": it_bytes_sum / iters / 1e9, "ms": it_ms, "GBps": it_bytes_sum / ((t_jv + t_b1) * 1e-3) / 1e9,
                            "frac_hbm": it_bytes_sum / ((t_jv + t_b1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "bytes_per_point": it_bytes_sum / iters / N}
    return out


def spawn_ranks(n, backend):
    """`python bench.py --gpus N` without a launcher: run torch.distributed.run as a child (one rank per GPU,
    rendezvous on 127.0.0.1) with this command line; rank 0 of the child prints the JSON line."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < n:
        print(f"bench.py: --gpus {n} needs {n} devices, {ndev} visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 0.2 s of timed work; the first ~50 steps after an
    ap.add_argument("--warmup", type=int, default=50)     # idle period run ~5 % slower (clock ramp)
    ap.add_argument("--workload", default="gcy20", choices=sorted(WORKLOADS))
    ap.add_argument("--allow-knobs", action="store_true", help="run although SDFS_* kernel-plan variables are set")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    # the library's test knobs change the kernels that run (SDFS_PLAN=classic is the round-1 path): a measurement must not
    # pick one up by accident
    knobs = sorted(k for k in os.environ if k.startswith("SDFS_") and k not in ("SDFS_BENCH_BACKEND", "SDFS_BENCH_SHARDED", "SDFS_BENCH_MIRROR"))
    if knobs and not args.allow_knobs:
        raise SystemExit(f"bench.py: kernel-plan variables set in the environment ({', '.join(knobs)}); unset them or pass --allow-knobs")
    # SDFS_BENCH_BACKEND=gloo rehearses the N > 1 path on a single GPU (all ranks on one device,
    # exchanges staged through the host); the real run uses RCCL ("nccl"), one rank per GPU.
    backend = os.environ.get("SDFS_BENCH_BACKEND", "nccl")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: start one rank per GPU ourselves, as a CHILD process and before anything in
        # this process touches the GPU (device_count() does not initialise it), and pass its exit code on
        sys.exit(spawn_ranks(args.gpus, backend))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # never fall through to a one-GPU measurement labelled as something else
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices, {ndev} visible")
    local_rank = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    # SDFS_BENCH_SHARDED=1 with --gpus 1: the N > 1 code path (sharded operator, exchanges, all-reduces) on a process
    # group of one rank -- the only way to run that path on RCCL with one GPU; a rehearsal, labelled as such
    force_sharded = os.environ.get("SDFS_BENCH_SHARDED", "0") == "1" and world == 1
    if force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    out_fd = None
    if world > 1 or force_sharded:
        # RCCL prints its version banner on stdout when the communicator is created (and any library under us may print
        # there too): everything but the JSON line goes to stderr, so that stdout is the one line the contract asks for
        sys.stdout.flush()
        out_fd = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import sdfs_via_autodiff_amd as S
    model, shapes = WORKLOADS[args.workload]
    params, arrays = build_model(S, model, shapes)
    N = int(np.prod(shapes))

    if world > 1 or force_sharded:
        from sdfs_via_autodiff_amd.distributed import bench_sharded
        line = bench_sharded(S, model, shapes, params, arrays, args, rank, local_rank, world)
        if rank == 0:
            if not args.no_cpu:
                # same bounded CPU sample as at N = 1 (rank 0's host cores; the other ranks wait at the barrier)
                w_host = 400 + 500 * np.random.default_rng(0).random(shapes)
                line["cpu_baseline"] = cpu_baseline(model, shapes, params, arrays, w_host)
                line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
            sys.stdout.flush()
            os.write(out_fd, (json.dumps(line) + "\n").encode())
        dist.barrier()
        dist.destroy_process_group()
        return

    op = S.KoopmansOperator(model, shapes, params, arrays, device=local_rank)
    stream = torch.cuda.current_stream()
    op.set_stream(stream.cuda_stream)
    w_host = 400 + 500 * np.random.default_rng(0).random(shapes)
    bufs = [torch.from_numpy(w_host).cuda(), torch.empty(shapes, dtype=torch.float64, device="cuda")]
    resid = torch.zeros(1, dtype=torch.float64, device="cuda")

    def step(i):
        op.apply_dev(bufs[i & 1].data_ptr(), bufs[(i + 1) & 1].data_ptr(), resid.data_ptr())

    # clock spin-up (setup, untimed, disclosed in config.spinup_steps): the first ~100 steps after an idle period
    # run ~5 % slower while the device ramps its clock; the driver's default of 5 warm-up steps would time the ramp
    SPINUP = 100
    for i in range(SPINUP):
        step(i)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last_resid = float(resid.item())
    # per-kernel durations: the same steps once more with HIP events around every launch (on the launch stream)
    op.set_profiling(True)
    for i in range(8):            # (creates the event pool)
        step(i)
    torch.cuda.synchronize()
    op.reset_counters()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    dt_prof = time.perf_counter() - t0
    counters = op.counters()
    op.set_profiling(False)
    # this box's copy ceiling, in the same run: the library's own streaming copy of the grid (read N + write N doubles;
    # 16 bytes per lane, eight loads in flight per lane, non-temporal loads and stores -- the fastest copy form
    # tools/probes/kernel_bench.hip found, profiles/round4_kernel_bench.txt), timed with HIP events on the launch
    # stream; torch's copy_ (what rounds 1-3 printed here) beside it
    def timed_copies(fn, reps=100):
        for _ in range(60):              # (the clock ramps for a few milliseconds after the host-side pause before this block)
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        e1.synchronize()
        return reps * 16.0 * N / (e0.elapsed_time(e1) * 1e-3) / 1e9
    scratch = torch.empty(shapes, dtype=torch.float64, device="cuda")
    torch_copy_gbs = timed_copies(lambda: scratch.copy_(bufs[0]))
    copy_gbs = timed_copies(lambda: op.stream_copy_dev(bufs[0].data_ptr(), scratch.data_ptr(), N))
    torch_copy_gbs = max(torch_copy_gbs, timed_copies(lambda: scratch.copy_(bufs[0])))
    del scratch

    dom = max(counters, key=lambda c: c["total_ms"])
    avg_ms = dom["total_ms"] / max(dom["launches"], 1)
    achieved = dom["alg_bytes"] / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch cannot be counted from inside this process (PMC counters need rocprofv3): the figure
    # is taken from the committed PMC passes of this same command (profiles/: (2*FETCH_SIZE + WRITE_SIZE)*1024,
    # gfx950 correction of MI355X_MICROARCH.md) and only if that profile was taken on the plan that ran here
    traffic, traffic_source = None, "no committed PMC profile of this workload"
    plan_now = op.describe_plan().strip().split("\n")
    try:
        pfile = os.path.join("profiles", f"round4_{args.workload}_pmc.json")
        pmc = json.load(open(os.path.join(ROOT, pfile)))
        if pmc.get("plan") != plan_now:
            traffic_source = f"{pfile} was taken on a different kernel plan: dropped"
        else:
            traffic = pmc["kernels"][pmc["launch_order"][counters.index(dom)]]["hbm_traffic_bytes"]
            traffic_source = f"{pfile} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, same plan)"
    except (OSError, KeyError, ValueError, IndexError):
        pass
    kernels = [{"name": c["name"], "launches": c["launches"],
                "avg_ms": c["total_ms"] / max(c["launches"], 1),
                "alg_GB": c["alg_bytes"] / 1e9,
                "GBps": c["alg_bytes"] / (c["total_ms"] / max(c["launches"], 1) * 1e-3) / 1e9,
                "frac_hbm": c["alg_bytes"] / (c["total_ms"] / max(c["launches"], 1) * 1e-3) / 1e9 / HBM_PEAK_GBS}
               for c in counters]
    line = {
        "metric": "fixed-point iterations/sec",
        "value": args.steps / dt,
        "unit": "iterations/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{model.upper()} {'x'.join(map(str, shapes))} grid, successive-approximation "
                               f"step (T apply + fused sup-norm residual), default calibration, Rouwenhorst",
                   "grid_points": N, "spinup_steps": SPINUP, "plan": plan_now, "env_knobs": knobs},
        "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_source,
                     "avg_launch_ms": avg_ms, "alg_bytes_per_launch": dom["alg_bytes"],
                     "events_from": f"a second loop of the same {args.steps} steps with HIP events around every launch "
                                    f"({dt_prof / args.steps * 1e3:.4f} ms per step against {dt / args.steps * 1e3:.4f} without)",
                     "copy_ceiling_GBps": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs,
                     "copy_ceiling_kind": "sdfs_stream_copy_dev: 16 B per lane, eight loads in flight per lane, non-temporal; "
                                          "HIP events on the launch stream, 100 copies of the grid",
                     "torch_copy_GBps": torch_copy_gbs, "guide_copy_GBps": 6290.0,
                     "step_alg_bytes": sum(c["alg_bytes"] for c in counters),
                     "step_frac": sum(c["alg_bytes"] for c in counters) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                     "step_frac_of_copy_ceiling": sum(c["alg_bytes"] for c in counters) / (dt / args.steps) / 1e9 / copy_gbs},
        # secondary bound: an fp64 MFMA holds its SIMD for its full 64 / 20 cycles and every VALU
        # instruction of the same SIMD shares that issue slot, so a pass costs MFMA + VALU issue cycles; the
        # nominal fp64 peak is 78.6 TFLOP/s.  Algorithmic flops = contraction MACs x 2 plus the two powers of
        # the operator at the 31 fp64 instructions (~50 flops) the kernels spend on each.
        "fp64_pipe": {"peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "achieved": (sum(c["alg_flops"] for c in counters) + 2 * 50.0 * N) / (dt / args.steps) / 1e12,
                      "contraction_only": sum(c["alg_flops"] for c in counters) / (dt / args.steps) / 1e12},
        "kernels": kernels,
        "ideal_single_pass_frac": 16.0 * N / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
        "last_residual": last_resid,
    }
    if not args.no_cpu:
        line["cpu_baseline"] = cpu_baseline(model, shapes, params, arrays, w_host)
        line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
    if not args.no_secondary:
        sec = {}
        if args.workload == "gcy20":
            # time-to-converge (sup-norm 1e-8) of the bench grid itself, device-resident Newton-Krylov
            # from the reference's start w = 800 (includes one 512 MB upload and download)
            del bufs
            torch.cuda.empty_cache()
            w800 = np.full(shapes, 800.0)
            t0 = time.perf_counter()
            x, n, info = op.solve(w800, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)
            t = time.perf_counter() - t0
            sec["gcy20_newton_1e-8"] = {"iterations": n, "operator_applies": info["n_apply"], "seconds": t,
                                        "applies_per_s": info["n_apply"] / t, "final_err": info["final_err"]}
            if not args.no_cpu:
                sec["gcy20_newton_1e-8"]["cpu_twin"] = cpu_time_to_converge(model, shapes, params, arrays, n, info["n_apply"], full=False)
            sec["gcy20_newton_1e-8"]["roofline"] = newton_roofline(op, w800, N, 0)
            # BASELINE config 5: the same solve with fp32 Krylov storage (fp64 arithmetic and outer residual)
            t0 = time.perf_counter()
            x, n, info = op.solve(w800, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, krylov_f32=1)
            t = time.perf_counter() - t0
            sec["gcy20_newton_1e-8_krylov_f32"] = {"iterations": n, "operator_applies": info["n_apply"], "seconds": t,
                                                   "applies_per_s": info["n_apply"] / t, "final_err": info["final_err"]}
            # ... and the fastest setting of the config-5 sweep that still reaches 1e-8 (profiles/round3_mixed_precision_sweep.txt):
            # fp32 Krylov storage with the inner solves at 1e-4 (inexact Newton: one more outer step, fewer matvecs)
            t0 = time.perf_counter()
            x, n, info = op.solve(w800, "newton", tol=1e-8, inner_rtol=1e-4, inner_atol=0.0, krylov_f32=1)
            t = time.perf_counter() - t0
            sec["gcy20_newton_1e-8_krylov_f32_inner_1e-4"] = {"iterations": n, "operator_applies": info["n_apply"], "seconds": t,
                                                              "applies_per_s": info["n_apply"] / t, "final_err": info["final_err"]}
            # config 5 "on MFMA": fp32 storage AND the J.v passes on an fp32 LDS tile with v_mfma_f32 (opts.krylov_f32 = 3)
            t0 = time.perf_counter()
            x, n, info = op.solve(w800, "newton", tol=1e-8, inner_rtol=1e-4, inner_atol=0.0, krylov_f32=3)
            t = time.perf_counter() - t0
            sec["gcy20_newton_1e-8_f32_mfma_inner_1e-4"] = {"iterations": n, "operator_applies": info["n_apply"], "seconds": t,
                                                            "applies_per_s": info["n_apply"] / t, "final_err": info["final_err"],
                                                            "roofline": newton_roofline(op, w800, N, 3, inner=1e-4)}
            del x, w800
            # the device-resident successive-approximation loop on the bench grid (what solver(...) runs): at 20^6 the
            # same three launches per iteration as the headline step (slices, streamed middle lines, streamed last
            # lines), gated on the previous iteration's error; the fused end + start form is SDFS_SA_FUSED=1
            ws = torch.full(shapes, 800.0, dtype=torch.float64, device="cuda")
            op.solve_dev(ws.data_ptr(), "successive_approx", tol=0.0, max_iter=4, check_every=4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_sa, info = op.solve_dev(ws.data_ptr(), "successive_approx", tol=0.0, max_iter=100, check_every=100)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            sec["gcy20_sa_device_loop"] = {"iterations": n_sa, "seconds": t, "ms_per_iteration": t / n_sa * 1e3,
                                           "iterations_per_s": n_sa / t, "final_err": info["final_err"]}
            # Anderson on the device-resident grid (batched-Gram loop, vec_kernels.hpp): to 1e-6, the tolerance the
            # accelerated iteration reaches at this grid before its history turns collinear
            ws.fill_(800.0)
            op.solve_dev(ws.data_ptr(), "anderson", tol=0.0, max_iter=40)
            ws.fill_(800.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_a, info = op.solve_dev(ws.data_ptr(), "anderson", tol=1e-6, max_iter=5000)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            sec["gcy20_anderson_1e-6_device"] = {"iterations": n_a, "seconds": t, "ms_per_iteration": t / max(n_a, 1) * 1e3,
                                                 "final_err": info["final_err"]}
            # ... and to the metric's tolerance, 1e-8 (on Anderson's own error, |T x - x|_2 -- stricter than the sup-norm step
            # of successive approximation): the reference's absolute ridge 1e-6 (code/solvers.py:113) and the opt-in relative
            # ridge (sdfs_opts.ridge < 0: |ridge| trace(G) / m; profiles/round4_anderson_ridge.txt), successive approximation
            # to a sup-norm step of 1e-8 beside them
            for key, kw in (("gcy20_anderson_1e-8_device", dict(ridge=1e-6)), ("gcy20_anderson_1e-8_relative_ridge_1e-6", dict(ridge=-1e-6))):
                ws.fill_(800.0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n_a, info = op.solve_dev(ws.data_ptr(), "anderson", tol=1e-8, max_iter=6000, **kw)
                torch.cuda.synchronize()
                t = time.perf_counter() - t0
                sec[key] = {"iterations": n_a, "seconds": t, "ms_per_iteration": t / max(n_a, 1) * 1e3, "final_err_l2": info["final_err"],
                            "status": info["status"]}
            ws.fill_(800.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_sa, info = op.solve_dev(ws.data_ptr(), "successive_approx", tol=1e-8, max_iter=100000)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            sec["gcy20_successive_approx_1e-8_device"] = {"iterations": n_sa, "seconds": t, "final_err_sup": info["final_err"]}
            del ws
            # the conditional-tensor kernels at full size: Rouwenhorst tensors are slice-identical, so the headline
            # runs the merged (unconditional) plan; SDFS_NO_SLICE_MERGE keeps z_Q (25.6 MB) / z_pi_Q conditional
            os.environ["SDFS_NO_SLICE_MERGE"] = "1"
            try:
                opc = S.KoopmansOperator(model, shapes, params, arrays, device=local_rank)
            finally:
                del os.environ["SDFS_NO_SLICE_MERGE"]
            opc.set_stream(stream.cuda_stream)
            bc = [torch.from_numpy(w_host).cuda(), torch.empty(shapes, dtype=torch.float64, device="cuda")]
            for i in range(20):
                opc.apply_dev(bc[i & 1].data_ptr(), bc[(i + 1) & 1].data_ptr(), resid.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(100):
                opc.apply_dev(bc[i & 1].data_ptr(), bc[(i + 1) & 1].data_ptr(), resid.data_ptr())
            torch.cuda.synchronize()
            tc = (time.perf_counter() - t0) / 100
            sec["gcy20_conditional_tensor_path"] = {"ms_per_step": tc * 1e3, "iterations_per_s": 1.0 / tc,
                                                    "plan": opc.describe_plan().strip().split("\n")}
            del bc
            opc.close()
            torch.cuda.empty_cache()
        m = S.SSY(); shp = (15,) * 4
        T = S.ssy_operator(shp, m.params, S.discretize_ssy(m, shp))
        sec["ssy15_plan"] = T.describe_plan().strip().split("\n")
        for algo, kw in (("successive_approx", dict(tol=1e-8)),
                         ("anderson", dict(tol=1e-8)),
                         ("newton", dict(tol=1e-8, inner_rtol=1e-6, inner_atol=0.0))):
            T.solve(np.full(shp, 800.0), algo, max_iter=64)      # warm-up (graph capture, buffers)
            t0 = time.perf_counter()
            x, n, info = T.solve(np.full(shp, 800.0), algo, **kw)
            t = time.perf_counter() - t0
            sec[f"ssy15_{algo}"] = {"iterations": n, "operator_applies": info["n_apply"], "seconds": t,
                                    "iterations_per_s": n / t, "applies_per_s": info["n_apply"] / t,
                                    "final_err": info["final_err"]}
            if algo == "newton" and not args.no_cpu:
                sec["ssy15_newton"]["cpu_twin"] = cpu_time_to_converge("ssy", shp, m.params, S.discretize_ssy(m, shp), n, info["n_apply"], full=True)
        T.close()
        # a grid between the plans (every extent <= 16, more points than the small-grid plan takes): GCY 15^6 on the padded
        # pair plan (csrc/pad_kernels.hpp) -- resident-HBM step of T with its residual, as the headline's
        g15 = S.GCY(); shp15 = (15,) * 6
        T15 = S.gcy_operator(shp15, g15.params, S.discretize_gcy(g15, shp15))
        dev = torch.device("cuda", 0)
        a15 = torch.full(shp15, 800.0, dtype=torch.float64, device=dev)
        # (the device-resident SA loop, replayed from a hipGraph: a host-launched loop of three short kernels per step is at
        # the mercy of the host's scheduling -- one run of it read 0.64 instead of 0.17 ms)
        T15.solve_dev(a15.data_ptr(), "successive_approx", tol=0.0, max_iter=100, check_every=100)      # (captures the graph)
        a15.fill_(800.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n15, _ = T15.solve_dev(a15.data_ptr(), "successive_approx", tol=0.0, max_iter=200, check_every=100)
        torch.cuda.synchronize()
        t15 = (time.perf_counter() - t0) / n15
        sec["gcy15_padded_pair_plan"] = {"ms_per_step": t15 * 1e3, "iterations_per_s": 1.0 / t15, "points": 15 ** 6,
                                         "alg_GBps": 56.0 * 15 ** 6 / t15 / 1e9, "plan": T15.describe_plan().strip().split("\n")[:3]}
        T15.close(); del a15
        # BASELINE config 4's second grid, GCY 16^6, device-resident from w = 800: the step, and time-to-converge (1e-8) of
        # successive approximation, Newton-Krylov (fp64; fp32-MFMA J.v at inner 1e-4: config 5) and Anderson (reference's
        # ridge; the opt-in relative one)
        g16 = S.GCY(); shp16 = (16,) * 6
        T16 = S.gcy_operator(shp16, g16.params, S.discretize_gcy(g16, shp16))
        a16 = torch.empty(shp16, dtype=torch.float64, device=dev)
        sec16 = {}
        for key, algo, kw in (("step_sa_device_loop", "successive_approx", dict(tol=0.0, max_iter=200, check_every=100)),
                              ("successive_approx_1e-8", "successive_approx", dict(tol=1e-8)),
                              ("newton_1e-8", "newton", dict(tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)),
                              ("newton_1e-8_f32_mfma_inner_1e-4", "newton", dict(tol=1e-8, inner_rtol=1e-4, inner_atol=0.0, krylov_f32=3)),
                              ("anderson_1e-8", "anderson", dict(tol=1e-8, max_iter=5000)),
                              ("anderson_1e-8_relative_ridge_1e-6", "anderson", dict(tol=1e-8, max_iter=5000, ridge=-1e-6))):
            best = None
            for rep in range(2):                      # (the first run of a loop form captures its graph / sizes its buffers)
                a16.fill_(800.0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n16, info16 = T16.solve_dev(a16.data_ptr(), algo, **kw)
                torch.cuda.synchronize()
                t = time.perf_counter() - t0
                if best is None or t < best[0]:
                    best = (t, n16, info16)
            t, n16, info16 = best
            sec16[key] = {"iterations": n16, "operator_applies": info16["n_apply"], "seconds": t, "status": info16["status"]}
            if key == "step_sa_device_loop":
                sec16[key] = {"ms_per_step": t / n16 * 1e3, "iterations_per_s": n16 / t, "alg_GBps": 56.0 * 16 ** 6 / (t / n16) / 1e9}
        sec["gcy16"] = sec16
        T16.close(); del a16
        if not args.no_cpu:
            sec["cpu_literal_formulation"] = cpu_literal_apply()
        # continuous-state SSY at the reference's default size (10, 10, 10, 20; Gauss-Hermite d = 5)
        grids = S.build_grid(m, 10, 10, 10, 20)
        nodes, weights = S.qnwnorm([5] * 4)
        Tc = S.T_fun_factory((np.array(m.params), grids, np.ascontiguousarray(nodes.T), weights), "quadrature", 20000)
        w1 = np.ones(Tc.shapes)
        for algo, kw in (("successive_approx", dict(tol=1e-5)),
                         ("newton", dict(tol=1e-7, inner_rtol=1e-6, inner_atol=0.0))):
            Tc.solve(w1, algo, max_iter=8)
            t0 = time.perf_counter()
            x, n, info = Tc.solve(w1, algo, **kw)
            t = time.perf_counter() - t0
            sec[f"ssy_continuous_10x10x10x20_d5_{algo}"] = {
                "iterations": n, "operator_applies": info["n_apply"], "seconds": t,
                "applies_per_s": info["n_apply"] / t, "final_err": info["final_err"]}
        line["secondary"] = sec
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
