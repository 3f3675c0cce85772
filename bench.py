#!/usr/bin/env python
"""Benchmark of the hot path: GMRES JVPs/sec of the py_driver_2d iage Krylov solve.

    python bench.py --gpus N --steps K --warmup W [--grid 416]

One "step" is one Krylov (GMRES) iteration = one finite-difference Jacobian-vector
product (a perturbed forward model year on the GPU) + one preconditioner apply + the
Arnoldi / modified Gram-Schmidt update + the least-squares solve + the two lin_combs +
the checkpoint trail (NetCDF3 vector files, Krylov_state.json), exactly the body of
`KrylovSolver.solve`'s loop (reference nk_ooc/krylov_solver.py:112-163).  Inputs are
resident in HBM before the timed region (iterate, F(iterate), preconditioner factors).

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling, every rank
owns one iage tracer module on its own GPU; the only collective is the per-iteration
all-reduce of the convergence flag (RCCL).  value = (N * K) / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import shutil
import sys
import tempfile
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(grid_n, gpu_attempts_per_year, budget_s):
    """time the oracle (CPU restatement of the reference path: NumPy stencil + SciPy
    SuperLU + restated Radau controller, single thread) on a bounded sample: the first
    Radau step attempts of the same forward year, until `budget_s` of wall time."""
    from oracle import radau
    from oracle.grid import default_axes
    from oracle.model import Iage, Py2dModel, gen_init_iterate

    depth, ypos = default_axes(grid_n, grid_n)
    model = Py2dModel(depth, ypos)
    tm = Iage(model)
    y0 = gen_init_iterate(model).reshape(-1)
    year = 365.0 * 86400.0
    t0 = time.perf_counter()
    solver = radau.RadauOracle(tm.comp_tend, tm.comp_jacobian, 0.0, y0, year, max_step=0.01 * year)
    steps = 0
    while time.perf_counter() - t0 < budget_s and solver.t < year:
        solver.step()
        steps += 1
    wall = time.perf_counter() - t0
    attempts = steps + solver.stats.nrejected
    sec_per_attempt = wall / max(attempts, 1)
    finished = solver.t >= year
    sec_per_jvp = wall if finished else sec_per_attempt * gpu_attempts_per_year
    return {
        "value": 1.0 / sec_per_jvp,
        "unit": "JVPs/s",
        "cores": 1,
        "kind": "port",
        "sample": (f"oracle (NumPy+SciPy SuperLU Radau restatement) on iage {grid_n}x{grid_n}: "
                   + (f"one full forward year in {wall:.1f} s" if finished else
                      f"first {attempts} Radau step attempts in {wall:.1f} s "
                      f"({sec_per_attempt:.2f} s each), scaled to the {gpu_attempts_per_year} "
                      f"attempts the GPU run needed per forward year"))
        + "; one JVP = one forward year; preconditioner and Arnoldi cost ignored",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=416, help="depth and ypos levels")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=25.0)
    ap.add_argument("--no-files", action="store_true", help="skip the NetCDF trail (not the default)")
    args = ap.parse_args()

    import torch

    from nk_ooc_amd import dist as nkdist
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    # NK2D_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the ranks
    # share the visible GPUs and the (4-byte) collectives run on CPU tensors; never used by the driver
    backend = os.environ.get("NK2D_BENCH_BACKEND", "nccl")
    rank, local_rank, world = nkdist.init_process_group_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    n = args.grid
    workdir = tempfile.mkdtemp(prefix=f"nk2d_bench_r{rank}_")
    try:
        cfg = make_config(workdir, n, n, extra_solverinfo={"krylov_rel_tol": "0.0"})
        gen_grid_vars_file(cfg["modelinfo"])
        ModelState.reset_class()
        ModelState.device_map = {"iage": local_rank}
        ModelState.write_files = not args.no_files
        ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])

        # inputs: gen_init_iterate profile + one fixed-point year (as the reference CI sets
        # up), F(iterate), and the preconditioner factors -- all resident before timing
        t_setup = time.perf_counter()
        iterate = ModelState("gen_init_iterate")
        iterate += iterate.comp_fcn(os.path.join(workdir, "fcn_init.nc"), None)
        fcn = iterate.comp_fcn(os.path.join(workdir, "fcn_00.nc"), None)
        fwd_stats = ModelState.last_stats[0]
        eng = iterate.tracer_modules[0].eng
        t_pc = time.perf_counter()
        eng.precond_setup()
        eng.sync()
        precond_setup_s = time.perf_counter() - t_pc
        setup_s = time.perf_counter() - t_setup

        def run(k_iters, tag):
            solverinfo = dict(cfg["solverinfo"])
            solverinfo["krylov_workdir"] = os.path.join(workdir, tag)
            solverinfo["krylov_max_iter"] = str(k_iters)
            solver = nkdist.DistributedKrylovSolver(iterate, solverinfo, resume=False, rewind=False,
                                                    hist_fname=None, device=device)
            solver.solve(os.path.join(workdir, f"increment_{tag}.nc"), fcn)
            return solver

        if args.warmup > 0:
            run(args.warmup, "krylov_warm")
        eng.profile_reset(1)
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        solver = run(args.steps, "krylov_timed")
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(el.item())
        prof = eng.profile_read()
        jvp_stats = ModelState.last_stats[0]

        if rank == 0:
            total_jvps = args.steps * world
            # HIP-event windows around back-to-back launches of the kernel on the context's stream
            # (nk2d_profile_reset/read): per-launch time net of the event machinery, and the
            # algorithmic bytes of exactly those launches
            bytes_per_launch = prof["bytes"] / max(prof["samples"], 1)
            kernel_us = max(prof["avg_us"], 1e-3)
            achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
            traffic = None
            pmc_fname = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_{n}.json")
            if os.path.exists(pmc_fname):
                # HBM-side bytes per launch of the same kernel from separate rocprofv3 --pmc
                # passes (FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE); see the file
                traffic = json.load(open(pmc_fname))["traffic_bytes_per_launch_upper"]
            out = {
                "metric": "GMRES JVPs/sec, py_driver_2d iage",
                "value": total_jvps / elapsed,
                "unit": "JVPs/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": 1000.0 * elapsed / args.steps,
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "f64",
                "data": "synthetic",
                "config": {
                    "workload": f"py_driver_2d iage {n}x{n} (BASELINE.json configs[2]), one iage tracer "
                                "module per GPU, Krylov iterations of KrylovSolver.solve with the "
                                "NetCDF3/JSON checkpoint trail" + (" disabled" if args.no_files else ""),
                    "grid": [n, n],
                    "tracer_modules_per_gpu": 1,
                    "krylov_iterations": args.steps,
                    "parallelism": f"tracer-module-per-gpu x{world}",
                },
                "roofline": {
                    "bound": "hbm",
                    "kernel": f"k_newton_fused<{eng.nz // 64 + (1 if eng.nz % 64 else 0)}, 0, 0, 1>",
                    "achieved": achieved,
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic,
                    "avg_launch_us": kernel_us,
                    "event_windows": prof["windows"],
                    "event_pair_empty_us": prof["event_overhead_us"],
                    "event_samples": prof["samples"],
                    "launches": prof["launches"],
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                },
                "forward_year": {k: jvp_stats[k] for k in
                                 ("nfev", "njev", "nlu", "nsteps", "nrejected", "nnewton", "nsweeps",
                                  "nlaunch", "seconds")},
                "setup_seconds": {"total": setup_s, "precond_factorisation": precond_setup_s},
            }
            if world == 1 and args.cpu_baseline_seconds > 0:
                attempts = jvp_stats["nsteps"] + jvp_stats["nrejected"]
                out["cpu_baseline"] = cpu_baseline(n, attempts, args.cpu_baseline_seconds)
                out["cpu_baseline"]["host_cpus"] = os.cpu_count()
            print(json.dumps(out), flush=True)
    finally:
        ModelState.reset_class()
        shutil.rmtree(workdir, ignore_errors=True)
        if world > 1:
            torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
