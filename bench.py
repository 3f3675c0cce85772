#!/usr/bin/env python
"""Benchmark of the hot path: GMRES JVPs/sec of the py_driver_2d iage Krylov solve.

    python bench.py --gpus N --steps K --warmup W [--grid 416]

One "step" is one Krylov (GMRES) iteration = one finite-difference Jacobian-vector
product (a perturbed forward model year on the GPU) + one preconditioner apply + the
Arnoldi / modified Gram-Schmidt update + the least-squares solve + the two lin_combs +
the checkpoint trail (NetCDF3 vector files, Krylov_state.json), exactly the body of
`KrylovSolver.solve`'s loop (reference nk_ooc/krylov_solver.py:112-163).  Inputs are
resident in HBM before the timed region (iterate, F(iterate), preconditioner factors).

N > 1: one process per GPU over RCCL (torch.distributed, backend nccl).  Called as
`python bench.py --gpus N` without a launcher around it, this process starts the N ranks
itself (`python -m torch.distributed.run`, before anything here touches the GPU) and relays
rank 0's JSON line.  Weak scaling: every rank owns one iage tracer module on its own GPU;
the only collective of that layout is the per-iteration all-reduce of the convergence
flag.  value = (N * K) / max-over-ranks time.  With N >= 2 the line also carries `shard_e2`:
ONE iage module with its two tracers sharded over ranks 0 and 1 (SURVEY.md section 8(e) level
2, every Radau norm and every Krylov inner product an all-reduce) timed against the same
module on one GPU.

At N = 1 the line also carries `ladder` (the metric's 26 x 26 ... 416 x 416 grid ladder: JVPs/s
and roofline fraction per size, the CPU oracle's forward year beside it), `roofline_precond` (the
one genuinely HBM-streaming kernel) and `cpu_baseline`.

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LADDER_SIZES = (26, 52, 104, 208)
YEAR = 365.0 * 86400.0


# untimed Krylov iterations ahead of the timed ones of the auxiliary legs: the slab of a large schedule cache is allocated by
# a thread of the library while the first frozen years run launch by launch (warm_until_cached waits for it where it can)
WARM_ITERS = 4


def warm_until_cached(krylov_once, engines, max_extra=40, agree=None):
    """untimed Krylov iterations until the frozen years of `engines` run as ONE launch: the schedule cache of a large grid
    (above 8 GB) is allocated by a thread of the library's own -- hipMalloc of 120 GB has been seen to take 0.03 to 3 s -- and
    the years of the meantime run launch by launch.  One-off set-up, like the preconditioner's factorisation: not timed.
    `agree` (several ranks whose iterations hold a collective): any-of over the ranks, so that all of them run the same count.
    Returns the iterations it took (0 where the years already are one launch)."""
    agree = agree or (lambda flag: flag)
    extra = 0
    while agree(any(eng.cache_pending() for eng in engines)) and extra < max_extra:
        krylov_once()
        extra += 1
    if agree(bool(extra) or any(eng.counter("frozen_persistent_years") == 0 for eng in engines)):
        krylov_once()       # the year that adopts the slab and builds the cache (or: an engine that never takes that way)
        extra += 1
    return extra


def progress(msg):
    """a line on stderr per leg (a long run shows that it is alive; the JSON line on stdout stays the only stdout output)"""
    rank = os.environ.get("RANK", "0")
    sys.stderr.write(f"[bench rank {rank} +{time.perf_counter() - _T0:.1f}s] {msg}\n")
    sys.stderr.flush()


_T0 = time.perf_counter()


class AuxDeadline:
    """N > 1 only.  The auxiliary legs (shard_e2, config4_mix, shard_e3) run collectives that no box of this pool could
    rehearse over RCCL -- the one-GPU boxes run them over gloo -- and a rank that fails inside one of them leaves the others
    waiting in a collective that RCCL never gives up on: the headline measured before them would be lost with the job.  So
    the legs get a wall-clock budget (--aux-budget), started on all ranks behind a barrier: when it runs out rank 0 prints
    the line with what it has (the leg that was running named in `aux_legs_timed_out`) and every rank leaves with status 3:
    the headline is on stdout, and the run is still recorded as one that did not finish -- a rank stuck in a collective is a
    hang to be diagnosed from the per-leg progress lines on stderr, not a success."""

    def __init__(self, seconds, rank, out):
        import threading

        self.rank, self.out, self.leg, self.seconds = rank, out, None, seconds
        self._timer = threading.Timer(seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()

    def _fire(self):
        try:
            if self.rank == 0 and self.out is not None:
                line = dict(self.out)
                line["aux_legs_timed_out"] = {"running": self.leg, "budget_s": self.seconds}
                print(json.dumps(line), flush=True)
            progress(f"auxiliary legs over their budget of {self.seconds:.0f} s in {self.leg}: leaving with status 3")
            if self.rank != 0:
                time.sleep(2.0)     # (the launcher ends every rank when the first one fails: rank 0's line goes out first)
        finally:
            os._exit(3)

    def cancel(self):
        self._timer.cancel()


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as fresh child
    processes (nothing in this process has touched the GPU) and relay what rank 0 prints"""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.startswith("{"):
            print(line, end="", flush=True)
        else:
            sys.stderr.write(line)
    return proc.wait()


def cpu_oracle_year(grid_n, budget_s, attempts_hint=None):
    """the oracle (CPU restatement of the reference path: NumPy stencil + SciPy SuperLU + restated Radau
    controller) on one host thread, from the gen_init_iterate profile: a full forward year when it
    finishes within `budget_s`, otherwise its first step attempts scaled to `attempts_hint` attempts"""
    from threadpoolctl import threadpool_limits

    from oracle import radau
    from oracle.grid import default_axes
    from oracle.model import Iage, Py2dModel, gen_init_iterate

    depth, ypos = default_axes(grid_n, grid_n)
    model = Py2dModel(depth, ypos)
    tm = Iage(model)
    y0 = gen_init_iterate(model).reshape(-1)
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        solver = radau.RadauOracle(tm.comp_tend, tm.comp_jacobian, 0.0, y0, YEAR, max_step=0.01 * YEAR)
        steps = 0
        while time.perf_counter() - t0 < budget_s and solver.t < YEAR:
            solver.step()
            steps += 1
        wall = time.perf_counter() - t0
    attempts = steps + solver.stats.nrejected
    finished = bool(solver.t >= YEAR)
    res = {"grid": grid_n, "full_year": finished, "wall_s": wall, "attempts": attempts,
           "nfev": solver.stats.nfev, "nlu": solver.stats.nlu}
    if finished:
        res["seconds_per_year"] = wall
    elif attempts_hint:
        res["seconds_per_year"] = wall / max(attempts, 1) * attempts_hint
        res["extrapolated_from_attempts"] = attempts
    return res


def cpu_baseline(grid_n, gpu_attempts_per_year, budget_s):
    res = cpu_oracle_year(grid_n, budget_s, gpu_attempts_per_year)
    sec = res["seconds_per_year"]
    how = (f"one full forward year in {res['wall_s']:.1f} s" if res["full_year"] else
           f"first {res['attempts']} Radau step attempts in {res['wall_s']:.1f} s "
           f"({res['wall_s'] / max(res['attempts'], 1):.2f} s each), scaled to the {gpu_attempts_per_year} "
           f"attempts a forward year takes under SciPy's decisions (counted by a GPU year in that mode)")
    return {
        "value": 1.0 / sec,
        "unit": "JVPs/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle (NumPy+SciPy SuperLU Radau restatement) on iage {grid_n}x{grid_n}: {how}; "
                  "one JVP = one forward year; preconditioner and Arnoldi cost ignored; the cost of an attempt is the two "
                  "SuperLU factorisations of the year's Jacobian pattern (the same at any step size), so the first attempts "
                  "are representative and, Newton iterations being fewest on the short first steps, a LOWER bound",
    }


class Workload:
    """iage on an n x n grid set up as the reference CI does: gen_init_iterate + one fixed-point year,
    F(iterate), preconditioner factors -- all resident before any timing"""

    def __init__(self, n, device_ordinal, tag, write_files=True):
        from nk_ooc_amd.model_config import ModelConfig
        from nk_ooc_amd.model_state import ModelState
        from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

        self.n = n
        self.workdir = tempfile.mkdtemp(prefix=f"nk2d_bench_{tag}_")
        self.cfg = make_config(self.workdir, n, n, extra_solverinfo={"krylov_rel_tol": "0.0"})
        gen_grid_vars_file(self.cfg["modelinfo"])
        ModelState.reset_class()
        ModelState.device_map = {"iage": device_ordinal}
        ModelState.write_files = write_files
        ModelState.model_config_obj = ModelConfig(self.cfg["modelinfo"])
        t0 = time.perf_counter()
        self.iterate = ModelState("gen_init_iterate")
        self.iterate += self.iterate.comp_fcn(os.path.join(self.workdir, "fcn_init.nc"), None)
        self.fcn = self.iterate.comp_fcn(os.path.join(self.workdir, "fcn_00.nc"), None)
        self.fwd_stats = ModelState.last_stats[0]
        self.eng = self.iterate.tracer_modules[0].eng
        t1 = time.perf_counter()
        self.eng.precond_setup()
        self.eng.sync()
        self.precond_setup_s = time.perf_counter() - t1
        self.setup_s = time.perf_counter() - t0

    def krylov(self, k_iters, tag, device):
        from nk_ooc_amd import dist as nkdist

        solverinfo = dict(self.cfg["solverinfo"])
        solverinfo["krylov_workdir"] = os.path.join(self.workdir, tag)
        solverinfo["krylov_max_iter"] = str(k_iters)
        solver = nkdist.DistributedKrylovSolver(self.iterate, solverinfo, resume=False, rewind=False,
                                                hist_fname=None, device=device)
        solver.solve(os.path.join(self.workdir, f"increment_{tag}.nc"), self.fcn)
        return solver

    def close(self):
        from nk_ooc_amd.model_state import ModelState

        ModelState.reset_class()
        shutil.rmtree(self.workdir, ignore_errors=True)


MIX_NAMES = ["iage", "phosphorus", "forced_dye"]
MIX_CFG_NAMES = {"iage": "iage", "phosphorus": "phosphorus", "forced_dye": "forced_{suff}:dye"}
MIX_DECAY = {"forced_surf_restore_opt": "none", "forced_sms_opt": "decay", "forced_sms_decay_rate": "1.0e-8"}


class MixWorkload:
    """BASELINE.json configs[3]: the three tracer modules of py_driver_2d together -- iage, phosphorus and the decay variant
    of forced (py_driver_2d's analogue of dye_decay) -- or the subset of them a rank holds; set up as the Newton solver
    calls the Krylov solver: F(x) from a year with history (the phosphorus preconditioner is made from it), products on
    the accepted steps of that year, every module an engine (HIP stream) of its own on the rank's GPU"""

    def __init__(self, n, names, device_ordinal, tag):
        import numpy as np

        from nk_ooc_amd.model_config import ModelConfig
        from nk_ooc_amd.model_state import ModelState
        from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

        self.n, self.names = n, list(names)
        self.workdir = tempfile.mkdtemp(prefix=f"nk2d_mix_{tag}_")
        self.cfg = make_config(self.workdir, n, n, tracer_module_names=",".join(MIX_CFG_NAMES[m] for m in names),
                               extra_modelinfo=MIX_DECAY, extra_solverinfo={"krylov_rel_tol": "0.0"})
        gen_grid_vars_file(self.cfg["modelinfo"])
        ModelState.reset_class()
        ModelState.device_map = {m: device_ordinal for m in names}
        ModelState.write_files = False          # a 0.5 GB history file per year at 416 x 416: this leg times the path
        ModelState.model_config_obj = ModelConfig(self.cfg["modelinfo"])
        self.iterate = ModelState("gen_init_iterate")
        for tms in self.iterate.tracer_modules:
            if tms.name == "forced_dye":
                # the initial iterate of the module is exactly uniform; give it structure (docs/DESIGN_history_r1-r3.md section 5)
                rng = np.random.default_rng(3)
                bump = np.cumsum(np.cumsum(rng.standard_normal((1, n, n)), axis=1), axis=2)
                tms.eng.upload(1.0 + 0.3 * bump / np.max(np.abs(bump)), out=tms.vec)
        self.hist_fname = os.path.join(self.workdir, "hist_00.nc")
        self.fcn = self.iterate.comp_fcn(os.path.join(self.workdir, "fcn_00.nc"), None, self.hist_fname)
        self.base_stats = {tms.name: st for tms, st in zip(self.iterate.tracer_modules, ModelState.last_stats)}

    def krylov(self, k_iters, tag, device, group=None):
        """group None: all modules are here, no collective at all (rank 0's one-GPU baseline runs while the others wait)"""
        from nk_ooc_amd import dist as nkdist
        from nk_ooc_amd.krylov_solver import KrylovSolver

        solverinfo = dict(self.cfg["solverinfo"])
        solverinfo["krylov_workdir"] = os.path.join(self.workdir, tag)
        solverinfo["krylov_max_iter"] = str(k_iters)
        if group is None:
            solver = KrylovSolver(self.iterate, solverinfo, False, False, self.hist_fname)
        else:
            solver = nkdist.DistributedKrylovSolver(self.iterate, solverinfo, resume=False, rewind=False,
                                                    hist_fname=self.hist_fname, device=device, group=group)
        solver.solve(os.path.join(self.workdir, f"increment_{tag}.nc"), self.fcn)
        return solver

    def sync(self):
        for tms in self.iterate.tracer_modules:
            tms.eng.sync()

    def engines(self):
        return [tms.eng for tms in self.iterate.tracer_modules]

    def counters(self):
        return {tms.name: {"frozen_years_rejected": tms.eng.frozen_fallbacks(), "frozen_years_resumed": tms.eng.frozen_resumes()}
                for tms in self.iterate.tracer_modules}

    def close(self):
        from nk_ooc_amd.model_state import ModelState

        ModelState.reset_class()
        ModelState.device_map = None
        ModelState.write_files = True
        shutil.rmtree(self.workdir, ignore_errors=True)


def run_config4_mix(args, rank, local_rank, world, device):
    """the three-module mix timed (i) on ONE GPU, its three forward years running concurrently on three streams, and, for
    N >= 2, (ii) with the modules dealt round-robin to min(N, 3) ranks (dist.partition_modules; the one collective is the
    all-reduce of the convergence flag per Krylov iteration): the strong-scaling figure of the line -- the same work on
    more GPUs -- next to the weak one of `value`"""
    import torch

    from nk_ooc_amd import dist as nkdist

    n = args.mix_grid or args.grid
    k = args.mix_steps
    out = {"what": "BASELINE.json configs[3]: iage + phosphorus + forced (decay) through KrylovSolver, products on frozen years, "
                   "no file trail (ModelState.write_files False)",
           "grid": [n, n], "modules": MIX_NAMES, "krylov_iterations": k}

    def timed(wl, tag, group=None, nranks=1):
        wl.krylov(WARM_ITERS, f"{tag}_warm", device, group)
        if nranks == 1:      # (a collective per iteration for more ranks: everybody warms up the same fixed count)
            warm_until_cached(lambda: wl.krylov(1, f"{tag}_warm_more", device, group), wl.engines())
        wl.sync()
        # A solve starts with M^-1 F, and phosphorus makes its preconditioner anew for every solve (three factorisations and an
        # eigenvalue problem, ~1 s at 416 x 416: once per NEWTON iteration in a spin-up).  That is not a Krylov iteration: a solve
        # of one iteration and a solve of k + 1 are timed, the difference is k iterations, the rest the start of a solve.
        els = []
        for its in (1, k + 1):
            if nranks > 1:
                torch.distributed.barrier(group=group)
            t0 = time.perf_counter()
            wl.krylov(its, f"{tag}_timed{its}", device, group)
            wl.sync()
            if nranks > 1:
                torch.distributed.barrier(group=group)
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            if nranks > 1:
                torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX, group=group)
            els.append(float(el.item()))
        wl.solve_start_s = max(els[0] - (els[1] - els[0]) / k, 0.0)
        return els[1] - els[0]

    if rank == 0:
        wl = MixWorkload(n, MIX_NAMES, local_rank, "one")
        try:
            t_one = timed(wl, "one")
            from nk_ooc_amd.model_state import ModelState

            years = {tms.name: st["seconds"] for tms, st in zip(wl.iterate.tracer_modules, ModelState.last_stats)}
            out["one_gpu"] = {"ms_per_krylov_iteration": 1000.0 * t_one / k, "module_jvps_per_s": len(MIX_NAMES) * k / t_one,
                              "start_of_a_solve_ms": 1000.0 * wl.solve_start_s,
                              "perturbed_year_seconds": years,
                              "sum_of_the_perturbed_years_ms": 1000.0 * sum(years.values()),
                              "years_back_to_back": sum(t.eng.tc * t.eng.ny for t in wl.iterate.tracer_modules) > ModelState.CONCURRENT_MAX_COLUMNS,
                              "base_year_seconds": {m: st["seconds"] for m, st in wl.base_stats.items()},
                              "counters": wl.counters()}
        finally:
            wl.close()
    if world >= 2:
        used = min(world, len(MIX_NAMES))
        group = torch.distributed.new_group(list(range(used)))
        parts = nkdist.partition_modules(MIX_NAMES, used)
        torch.distributed.barrier()
        if rank < used:
            wl = MixWorkload(n, parts[rank], local_rank, f"r{rank}")
            try:
                t_dist = timed(wl, f"dist{rank}", group, used)
            finally:
                wl.close()
            if rank == 0:
                out["distributed"] = {"ranks_used": used, "layout": {f"rank{r}": parts[r] for r in range(used)},
                                      "ms_per_krylov_iteration": 1000.0 * t_dist / k,
                                      "module_jvps_per_s": len(MIX_NAMES) * k / t_dist,
                                      "speedup_over_one_gpu": out["one_gpu"]["ms_per_krylov_iteration"] / (1000.0 * t_dist / k),
                                      "collectives_per_krylov_iteration": 1}
        torch.distributed.barrier()
    return out if rank == 0 else None


def run_shard_e3(args, rank, local_rank, world, backend):
    """SURVEY.md section 8(e) level 3 / BASELINE.json configs[4], measured: the phosphorus module with the columns of its
    Krylov basis (and the preconditioned products) dealt round-robin to the ranks -- dist.column_sharded_gmres: the owner
    of a column computes the product and broadcasts it, Gram-Schmidt and the linear combinations are partial sums plus
    all-reduces of whole vectors -- against the same loop on one rank.  The layout buys HBM, not time: whatever the
    scaling is, it is what this leg reports."""
    import numpy as np
    import torch
    import torch.distributed as tdist

    from nk_ooc_amd import dist as nkdist
    from nk_ooc_amd.engine import phosphorus_engine
    from nk_ooc_amd.grid import Grid2d

    n = args.shard3_grid or args.grid
    k = args.mix_steps
    grid = Grid2d.default(n, n)
    eng = phosphorus_engine(grid, device_id=local_rank)
    try:
        eng.set_region(np.ones((n, n), dtype=np.int32), np.outer(grid.depth.delta, grid.ypos.delta))
        prof = [np.interp(grid.depth.mid, d, v) for d, v in (([1.3e2, 2.6e2], [5.5e-3, 4.1]), ([9.5e1, 1.4e2], [7.1e-2, 1.5e-4]),
                                                             ([1.7e2, 2.5e2], [1.8e-2, 7.9e-4]))]
        x0 = np.stack([np.broadcast_to(p[:, None], (n, n)) for p in prof]).copy()
        x = eng.upload(x0)
        fx, st, _ = eng.comp_fcn(x)
        sched = eng.last_schedule()
        eng.precond_setup_state((x0 + eng.download(fx))[0])
        device = torch.device("cuda", local_rank) if backend == "nccl" else "cpu"
        out = {"what": "BASELINE.json configs[4]: phosphorus, Krylov basis columns sharded over the ranks (level 3)",
               "grid": [n, n], "krylov_iterations": k, "backend": backend, "base_year_seconds": st["seconds"]}
        alone = nkdist.ColumnComm(0, 1, device)
        nkdist.column_sharded_gmres(eng, alone, x, fx, 0.0, 0, 1, sched=sched)          # warm-up
        eng.sync()
        t0 = time.perf_counter()
        nkdist.column_sharded_gmres(eng, alone, x, fx, 0.0, 0, k, sched=sched)
        eng.sync()
        t_one = time.perf_counter() - t0
        out["one_gpu"] = {"ms_per_jvp": 1000.0 * t_one / k, "jvps_per_s": k / t_one}
        if world >= 2:
            comm = nkdist.ColumnComm(rank, world, device)
            nkdist.column_sharded_gmres(eng, comm, x, fx, 0.0, 0, 1, sched=sched)
            eng.sync()
            calls0, vec0, bytes0 = comm.calls, comm.vec_calls, comm.vec_bytes
            tdist.barrier()
            t0 = time.perf_counter()
            _, info = nkdist.column_sharded_gmres(eng, comm, x, fx, 0.0, 0, k, sched=sched)
            eng.sync()
            tdist.barrier()
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            tdist.all_reduce(el, op=tdist.ReduceOp.MAX)
            t_n = float(el.item())
            out["sharded"] = {"ranks": world, "ms_per_jvp": 1000.0 * t_n / k, "jvps_per_s": k / t_n,
                              "speedup_over_one_gpu": t_one / t_n,
                              "small_allreduces_per_jvp": (comm.calls - calls0) / k,
                              "vector_collectives_per_jvp": (comm.vec_calls - vec0) / k,
                              "vector_bytes_per_jvp": (comm.vec_bytes - bytes0) / k,
                              "columns_on_rank0": info["columns_here"]}
        out["frozen_years_rejected"] = eng.frozen_fallbacks()
        return out if rank == 0 else None
    finally:
        eng.close()


def run_newton_spinup(n):
    """what a SOLVE gets (round-3 verdict, weak 4): the complete Newton-Krylov spin-up of iage through the driver mirror with the
    reference's newton_krylov.cfg (input/py_driver_2d/newton_krylov.cfg: two or three Krylov iterations per Newton iteration,
    a new schedule each) -- set-up of the schedule cache, its asynchronous allocation and the launch-by-launch years of the
    meantime all inside the time.  products_per_second = Krylov iterations / seconds of the whole Newton solve (forward years
    with history files, preconditioner factorisations and the file trail included)."""
    from nk_ooc_amd import nk_driver
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import make_config, setup

    work = tempfile.mkdtemp(prefix="nk2d_spinup_")
    try:
        cfg = make_config(work, n, n, tracer_module_names="iage")
        ModelState.reset_class()
        ModelState.write_files = True
        t0 = time.perf_counter()
        setup(cfg, fp_cnt=1)
        t1 = time.perf_counter()
        solver = nk_driver.run(cfg)
        t2 = time.perf_counter()
        kry, it = [], 0
        while os.path.isdir(os.path.join(work, f"krylov_{it:02}")):
            kry.append(json.load(open(os.path.join(work, f"krylov_{it:02}", "Krylov_state.json")))["iteration"])
            it += 1
        eng = next(iter(ModelState._engines.values()))
        res = {"what": "iage spin-up through nk_driver with the reference's newton_krylov.cfg; everything a solve pays is in the time",
               "grid": [n, n], "setup_seconds": t1 - t0, "newton_solve_seconds": t2 - t1,
               "converged": bool(solver.converged().all()), "newton_iterations": int(solver.get_iteration()),
               "krylov_iterations": kry, "products": int(sum(kry)),
               "products_per_second_inside_the_solve": sum(kry) / (t2 - t1),
               "one_launch_frozen_years": eng.counter("frozen_persistent_years"),
               "schedule_cache_builds": eng.counter("frozen_cache_builds"),
               "stream_years": eng.counter("stream_years_run"),
               "frozen_years_rejected": eng.frozen_fallbacks()}
        return res
    finally:
        ModelState.reset_class()
        shutil.rmtree(work, ignore_errors=True)


def roofline_of(eng, n):
    """dominant kernel of the forward year.  Host control: `achieved` = the algorithmic bytes of the kernel's launches
    in the timed region / their duration, the duration of each launch shape measured live by a back-to-back replay
    of 200 launches inside ONE HIP event pair on the context's stream (nk2d_profile_replay: no event cost to subtract,
    the hand-over between launches included) and weighted with the shape counts of the timed region.  The library's
    event windows over the timed region itself (one or two launches per window, so a quarter of each reading is the
    event pair) and the static rocprofv3 figure of the same command are printed beside it, labelled."""
    prof = eng.profile_read()
    samples = max(prof["samples"], 1)
    win_bytes_per_launch = prof["bytes"] / samples
    net_us = max(prof["avg_us"], 1e-3)
    raw_us = net_us + prof["event_overhead_us"] * prof["windows"] / samples
    if eng.counter("frozen_persistent_years") - getattr(eng, "_launch_years_base", 0) > 0:
        # small grids: every frozen year of the timed region was ONE launch on the schedule cache (k_frozen_persistent): its
        # phases' algorithmic bytes over the wall time of the year around that launch
        from nk_ooc_amd.model_state import ModelState

        totals = eng.profile_totals()
        years = max(totals["launches"], 1)
        nbytes = totals["bytes"] / years
        # the launches themselves: a HIP event pair on the context's stream around every one of them (counter
        # "frozen_launch_us", since profile_reset: bench.py zeroes its reading there)
        launch_us = (eng.counter("frozen_launch_us") - getattr(eng, "_launch_us_base", 0)) / max(
            eng.counter("frozen_persistent_years") - getattr(eng, "_launch_years_base", 0), 1)
        achieved = nbytes / (launch_us * 1e-6) / 1e9
        levels = (eng.nz + 63) // 64
        flavour = ("a four-wave team per column" if eng.counter("frozen_team_years") else
                   "a wave per column, a wave's own data of the year in LDS") + (
            ", all workgroups resident, workgroups hand over to their lateral neighbours")
        out = {"bound": "hbm", "kernel": f"k_frozen_persistent<{levels}, 0, ...> (a whole frozen year in ONE launch on the schedule "
                                         f"cache: one simplified-Newton iteration per phase; {flavour})",
               "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
               "avg_launch_us": launch_us,
               "timing": "HIP event pair on the context's stream around every one-launch year of the timed region "
                         "(ALGORITHMIC bytes of the year's phases -- the launches they replace -- over that time)",
               "launches": totals["launches"], "algorithmic_bytes_per_launch": nbytes,
               "phases_per_launch": ModelState.last_stats[0]["nsweeps"],
               "year_seconds_host_clock": ModelState.last_stats[0]["seconds"],
               "one_launch_years": eng.counter("frozen_persistent_years"), "team_years": eng.counter("frozen_team_years"),
               "schedule_cache_builds": eng.counter("frozen_cache_builds")}
        pmc_fname = next((f for f in (os.path.join(ROOT, "profiles", f"r{r:02}_pmc_traffic_one_launch_{n}.json") for r in (4, 3))
                          if os.path.exists(f)), "")
        if os.path.exists(pmc_fname):
            pmc = json.load(open(pmc_fname))
            # the counters were collected on a year of another schedule (tools/probe_traffic.py: 9 751 phases, here
            # `phases_per_launch`): their ratio to the algorithmic bytes of THAT year, applied to this one's
            out["traffic"] = pmc["traffic_over_algorithmic"] * nbytes
            out["traffic_over_algorithmic"] = pmc["traffic_over_algorithmic"]
            out["traffic_source"] = ("static profile file " + os.path.relpath(pmc_fname, ROOT) + " (separate rocprofv3 --pmc "
                                     "FETCH_SIZE / WRITE_SIZE passes over the same kernel on another year: "
                                     f"{pmc['traffic_bytes_per_launch_upper']:.4g} B of fabric traffic for "
                                     f"{pmc['algorithmic_bytes_per_launch']:.4g} algorithmic; not collected in this run)")
        stats_fname = next((f for f in (os.path.join(ROOT, "profiles", d, "kernel_stats.csv")
                                        for d in (f"r04_rocprof_one_launch_{n}", f"r03_rocprof_one_launch_{n}")) if os.path.exists(f)), "")
        if os.path.exists(stats_fname):
            for line in open(stats_fname):
                if line.startswith('"void k_frozen_persistent<' + str(levels)):
                    avg_ns = float(line.rsplit('",', 1)[1].split(",")[2])
                    out["rocprofv3"] = {"avg_launch_us": avg_ns / 1000.0, "frac": nbytes / (avg_ns * 1e-9) / 1e9 / HBM_PEAK_GBS,
                                        "source": "static profile file " + os.path.relpath(stats_fname, ROOT)}
                    break
        return out
    windows = {
        "what": "HIP event pairs around the launches of single Newton iterations inside the timed region",
        "avg_launch_us_event_cost_included": raw_us,
        "avg_launch_us_net_of_empty_event_pair": net_us,
        "event_pair_empty_us": prof["event_overhead_us"],
        "event_windows": prof["windows"],
        "event_samples": prof["samples"],
        "algorithmic_bytes_per_launch": win_bytes_per_launch,
        "frac_event_cost_included": win_bytes_per_launch / (raw_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
        "frac_net_of_empty_event_pair": win_bytes_per_launch / (net_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
    }
    if True:
        shapes = eng.profile_shapes()
        names = ["stage + sweep + update", "stage + first sweep", "last sweep + update"]
        shapes_out, tot_bytes, tot_us, tot_cnt = [], 0.0, 0.0, 0
        for shape in range(3):
            cnt = shapes["counts"][shape]
            rep = eng.profile_replay(shape, 200)
            shapes_out.append({"shape": names[shape], "launches_in_timed_region": cnt, "replay_us_per_launch": rep["avg_us"],
                               "algorithmic_bytes_per_launch": rep["bytes"],
                               "GBs": rep["bytes"] / (rep["avg_us"] * 1e-6) / 1e9})
            tot_bytes += shapes["bytes"][shape]
            tot_us += cnt * rep["avg_us"]
            tot_cnt += cnt
        tot_cnt = max(tot_cnt, 1)
        bytes_per_launch, launch_us = tot_bytes / tot_cnt, max(tot_us / tot_cnt, 1e-3)
        timing = ("per launch shape: 200 launches queued back to back inside ONE HIP event pair on the context's stream "
                  "(hand-over between launches included, no event cost subtracted), weighted with the shape counts of "
                  "the timed region")
    achieved = bytes_per_launch / (launch_us * 1e-6) / 1e9
    out = {
        "bound": "hbm",
        "kernel": f"k_newton_fused<{(eng.nz + 63) // 64}, 0, 0, 1>",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": None,
        "avg_launch_us": launch_us,
        "timing": timing,
        "launches": prof["launches"],
        "algorithmic_bytes_per_launch": bytes_per_launch,
    }
    if prof["windows"] > 0:     # none in a region of frozen years: nothing there waits for a norm
        out["event_windows_over_timed_region"] = windows
    if shapes_out:
        out["launch_shapes"] = shapes_out
    pmc_fname = next((f for f in (os.path.join(ROOT, "profiles", f"r{r:02}_pmc_traffic_{n}.json") for r in (3, 2, 1))
                      if os.path.exists(f)), "")
    if os.path.exists(pmc_fname):
        pmc = json.load(open(pmc_fname))
        out["traffic"] = pmc["traffic_bytes_per_launch_upper"]
        out["traffic_source"] = ("static profile file " + os.path.relpath(pmc_fname, ROOT) +
                                 " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; not collected in this run)")
    stats_fname = next((f for f in (os.path.join(ROOT, "profiles", f"r{r:02}_rocprof_bench{n}", "kernel_stats.csv") for r in (3, 2))
                        if os.path.exists(f)), "")
    if os.path.exists(stats_fname):
        for line in open(stats_fname):
            if line.startswith('"void ' + out["kernel"]):
                avg_ns = float(line.rsplit('",', 1)[1].split(",")[2])
                out["rocprofv3"] = {"avg_launch_us": avg_ns / 1000.0,
                                    "frac": bytes_per_launch / (avg_ns * 1e-9) / 1e9 / HBM_PEAK_GBS,
                                    "source": "static profile file " + os.path.relpath(stats_fname, ROOT)}
                break
    return out


def precond_roofline(eng, reps=5):
    """the preconditioner apply: 2 ny dense mat-vecs per tracer streamed from HBM (k_pc_gemv), timed with a
    HIP event pair on the context's stream around whole applies"""
    v = eng.upload(__import__("numpy").ones(eng.shape))
    out = eng.precond_apply(v)
    eng.sync()
    eng.timer_begin()
    for _ in range(reps):
        eng.precond_apply(v, out=out)
    ms = eng.timer_end() / reps
    m = 3 * eng.nz
    nbytes = 2.0 * eng.ny * eng.tc * m * m * 8.0
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_pc_gemv (nk2d_precond_apply: forward + backward block substitution)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "ms_per_apply": ms, "launches_per_apply": 2 * eng.ny + 2,
            "algorithmic_bytes_per_apply": nbytes,
            "timing": "HIP event pair on the context's stream around whole applies (launch gaps included)"}


def run_ladder(device_ordinal, device, args):
    """the metric's grid ladder below the headline size: JVPs/s of the same Krylov loop, the dominant
    kernel's roofline fraction, and the CPU oracle's forward year at the sizes it finishes"""
    rows = []
    cpu_full = {26: 60.0, 52: 240.0}       # budget for a full CPU year; beyond: bounded sample
    for n in LADDER_SIZES:
        wl = Workload(n, device_ordinal, f"ladder{n}")
        try:
            wl.krylov(WARM_ITERS, "warm", device)
            warm_until_cached(lambda: wl.krylov(1, "warm_more", device), [wl.eng])
            wl.eng.profile_reset(1)
            wl.eng._launch_us_base = wl.eng.counter("frozen_launch_us")
            wl.eng._launch_years_base = wl.eng.counter("frozen_persistent_years")
            wl.eng.sync()
            t0 = time.perf_counter()
            wl.krylov(args.ladder_steps, "timed", device)
            wl.eng.sync()
            elapsed = time.perf_counter() - t0
            roof = roofline_of(wl.eng, n)
            from nk_ooc_amd.model_state import ModelState

            st = ModelState.last_stats[0]
            row = {"grid": n, "jvps_per_s": args.ladder_steps / elapsed,
                   "ms_per_jvp": 1000.0 * elapsed / args.ladder_steps,
                   "forward_year_s": st["seconds"], "nsteps": st["nsteps"], "nlaunch": st["nlaunch"],
                   "base_year_free_running_s": wl.fwd_stats["seconds"],
                   "roofline_frac": roof["frac"], "avg_launch_us": roof["avg_launch_us"],
                   "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"]}
            # the same iterations as ONE C call with every vector in HBM and no file trail (nk2d_gmres_solve): what the Python
            # solver and the checkpoint files cost at this size
            x_vec, f_vec = wl.iterate.tracer_modules[0].vec, wl.fcn.tracer_modules[0].vec
            sched = (wl.fcn._sched or {}).get("iage") if os.environ.get("NK2D_JVP_FROZEN", "1") != "0" else None
            wl.eng.gmres_solve(x_vec, f_vec, 0.0, 0, 1, sched=sched)
            wl.eng.sync()
            t_g = time.perf_counter()
            wl.eng.gmres_solve(x_vec, f_vec, 0.0, 0, args.ladder_steps, sched=sched)
            wl.eng.sync()
            row["jvps_per_s_gmres_solve_in_hbm"] = args.ladder_steps / (time.perf_counter() - t_g)
            if args.cpu_baseline_seconds > 0:
                wl.eng.set_option("jac_fresh", 0)       # attempts of a year under SciPy's decisions, as the oracle takes them
                _, st_f, _ = wl.eng.comp_fcn(wl.iterate.tracer_modules[0].vec)
                wl.eng.set_option("jac_fresh", integrator_mode()["jac_fresh"])
                attempts = st_f["nsteps"] + st_f["nrejected"]
                cpu = cpu_oracle_year(n, cpu_full.get(n, 12.0), attempts)
                if "seconds_per_year" in cpu:
                    row["cpu_oracle"] = {"seconds_per_year": cpu["seconds_per_year"], "full_year": cpu["full_year"],
                                         "cores": 1, "attempts_timed": cpu["attempts"]}
                    row["gpu_over_cpu"] = cpu["seconds_per_year"] / (elapsed / args.ladder_steps)
            rows.append(row)
        finally:
            wl.close()
    return rows


def run_shard_e2(args, rank, local_rank, world, backend, one_gpu_ms, allreduces_hint):
    """SURVEY.md section 8(e) level 2, measured: ONE iage module, its two tracers on ranks 0 and 1.  Untimed: the
    all-reduce latency itself and the coupled year that gives F(x) (every Radau norm all-reduced; when latency x expected
    count exceeds --shard-budget seconds the leg is skipped and says so).  Timed: one Krylov iteration -- perturbed
    year frozen on the coupled year's steps (no exchange), preconditioner, CGS-2 with fused multi-dots, residual."""
    import numpy as np
    import torch
    import torch.distributed as tdist

    from nk_ooc_amd import dist as nkdist
    from nk_ooc_amd.grid import Grid2d

    group = tdist.new_group([0, 1])
    result = None
    if rank < 2:
        n = args.shard_grid or args.grid
        grid = Grid2d.default(n, n)
        comm = nkdist.ShardComm(device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu", group=group)
        for _ in range(20):
            comm.allreduce_scalar(1.0)
        t0 = time.perf_counter()
        for _ in range(200):
            comm.allreduce_scalar(1.0)
        latency = (time.perf_counter() - t0) / 200
        scale = (n / float(args.grid)) ** 0.5 if args.grid else 1.0       # steps grow like sqrt(n) on this ladder
        # untimed: F(x), a coupled year (one all-reduce per Newton iteration and error estimate); timed: one Krylov
        # iteration whose perturbed year repeats the steps of that year and exchanges nothing
        expected = allreduces_hint * scale * latency + 3.0 * one_gpu_ms / 1000.0
        verdict = torch.tensor([expected], dtype=torch.float64)
        result = {"layout": "ONE iage module, tracer per rank on ranks 0 and 1 (block-diagonal Jacobian); the year that gives "
                            "F(x) all-reduces every Radau norm (untimed), the perturbed years of the Krylov iterations repeat "
                            "its accepted steps and exchange nothing; every Krylov inner product is an all-reduce(SUM) of "
                            "1 .. (j+1) nreg doubles",
                  "backend": backend, "grid": [n, n], "allreduce_latency_us": 1.0e6 * latency,
                  "expected_seconds": expected, "one_gpu_ms_per_jvp": one_gpu_ms}
        if expected > args.shard_budget:
            result["skipped"] = f"expected {expected:.0f} s exceeds --shard-budget {args.shard_budget:.0f} s"
        else:
            eng = nkdist.iage_shard_engine(grid, rank, comm, device_id=local_rank)
            eng.set_region(np.ones((n, n), dtype=np.int32), np.outer(grid.depth.delta, grid.ypos.delta))
            col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
            x = eng.upload(np.broadcast_to(col[:, None], (1, n, n)).copy())
            calls_base = comm.calls
            fx, st_c, _ = eng.comp_fcn(x)                   # F(x): a coupled year, untimed
            sched = eng.last_schedule()
            result["allreduces_of_the_coupled_year"] = comm.calls - calls_base
            # (the vector norm hook pairs a Newton norm with the next one or with the error estimate: fewer collectives
            # than norms read)
            result["norms_read_by_the_controller_in_that_year"] = st_c["nnewton"] + st_c["nsteps"] + st_c["nrejected"] + 3
            eng.precond_setup()
            eng.sync()
            calls0 = comm.calls
            tdist.barrier(group=group)
            t0 = time.perf_counter()
            _, info = nkdist.sharded_gmres(eng, comm, x, fx, 0.0, 0, 1, sched=sched)
            eng.sync()
            tdist.barrier(group=group)
            elapsed = time.perf_counter() - t0
            result.update({"krylov_iterations": 1, "ms_per_jvp": 1000.0 * elapsed, "jvps_per_s": 1.0 / elapsed,
                           "allreduces_per_jvp": info["allreduces"] - calls0,
                           "speedup_over_one_gpu": (one_gpu_ms / (1000.0 * elapsed)) if (one_gpu_ms and n == args.grid) else None})
            eng.close()
        del verdict
    tdist.barrier()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=4,
                    help="untimed Krylov iterations (then as many more as the allocation of a large grid's schedule cache takes, "
                         "config.untimed_iterations_until_the_schedule_cache_was_allocated)")
    ap.add_argument("--grid", type=int, default=416, help="depth and ypos levels")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=25.0)
    ap.add_argument("--no-files", action="store_true", help="skip the NetCDF trail (not the default)")
    ap.add_argument("--no-ladder", action="store_true", help="skip the 26..208 grid ladder")
    ap.add_argument("--ladder-steps", type=int, default=3)
    ap.add_argument("--no-shard", action="store_true", help="skip the sharded-module leg of N >= 2 runs")
    ap.add_argument("--shard-grid", type=int, default=0, help="grid of the sharded-module leg (default: --grid)")
    ap.add_argument("--no-shard3", action="store_true", help="skip the basis-column-sharded phosphorus leg (shard_e3)")
    ap.add_argument("--shard3-grid", type=int, default=0, help="grid of that leg (default: --grid)")
    ap.add_argument("--shard3-first", action="store_true", help="N = 1: run that leg ahead of the ladder (diagnostic order)")
    ap.add_argument("--no-mix", action="store_true", help="skip the three-module leg (config4_mix)")
    ap.add_argument("--no-spinup", action="store_true", help="skip the whole Newton-Krylov solve (newton_spinup)")
    ap.add_argument("--mix-grid", type=int, default=0, help="grid of the three-module leg (default: --grid)")
    ap.add_argument("--mix-steps", type=int, default=3, help="Krylov iterations timed in the three-module leg")
    ap.add_argument("--shard-budget", type=float, default=120.0,
                    help="skip the sharded-module leg when its expected wall time exceeds this many seconds")
    ap.add_argument("--aux-budget", type=float, default=600.0,
                    help="N > 1: seconds the auxiliary legs may take before rank 0 prints the line without them (0: no limit)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, all-reduce their ranks over gloo and print the world size "
                         "(CPU check of the self-launch path, tests/test_dist.py)")
    args = ap.parse_args()
    if os.environ.get("NK2D_BENCH_DEBUG"):
        # stacks of every thread on stderr should a leg hang (seconds)
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["NK2D_BENCH_DEBUG"]), exit=False)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's invocation: no launcher around us.  Nothing above has touched the GPU.
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import torch

    from nk_ooc_amd import dist as nkdist

    if args.launch_check:
        rank, _, world = nkdist.init_process_group_from_env("gloo")
        total = torch.tensor([rank + 1], dtype=torch.int64)
        if world > 1:
            torch.distributed.all_reduce(total)
            torch.distributed.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": world, "rank_sum": int(total.item())}), flush=True)
        return

    # NK2D_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the ranks
    # share the visible GPUs and the (tiny) collectives run on CPU tensors; never used by the driver
    backend = os.environ.get("NK2D_BENCH_BACKEND", "nccl")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    if backend == "nccl" and world_env > 1:
        torch.cuda.set_device(local_rank)       # before the process group: RCCL binds to the current device
    rank, _, world = nkdist.init_process_group_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    n = args.grid
    wl = None
    deadline = None
    # the checkpoint trail as nk_driver.run keeps it: every file in program order on one writer thread, complete on disk
    # when a solve returns -- i.e. inside every timed region (NK2D_ASYNC_TRAIL=0: written inside the call that asks for it)
    from nk_ooc_amd import trail

    trail.set_enabled(os.environ.get("NK2D_ASYNC_TRAIL", "1") != "0")
    try:
        progress(f"set-up of iage {n}x{n}")
        wl = Workload(n, local_rank, f"r{rank}", write_files=not args.no_files)
        eng = wl.eng
        progress("warm-up and timed Krylov iterations")
        if args.warmup > 0:
            wl.krylov(args.warmup, "krylov_warm", device)
        def any_rank(flag):
            if world == 1:
                return flag
            buf = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
            torch.distributed.all_reduce(buf, op=torch.distributed.ReduceOp.MAX)
            return bool(buf.item())

        extra_warm = warm_until_cached(lambda: wl.krylov(1, "krylov_warm_more", device), [eng], agree=any_rank)
        eng.profile_reset(1)
        eng._launch_us_base = eng.counter("frozen_launch_us")
        eng._launch_years_base = eng.counter("frozen_persistent_years")
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        wl.krylov(args.steps, "krylov_timed", device)
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(el.item())
        from nk_ooc_amd.model_state import ModelState

        jvp_stats = ModelState.last_stats[0]
        out = None
        if rank == 0:
            total_jvps = args.steps * world
            year_keys = ("nfev", "njev", "nlu", "nsteps", "nrejected", "nnewton", "nsweeps", "nlaunch", "seconds")
            totals = eng.profile_totals()
            roof = roofline_of(eng, n)
            out = {
                "metric": "GMRES JVPs/sec, py_driver_2d iage (value = value_frozen: products on frozen years, the default; "
                          "value_reference_semantic: the reference's product, two free-running years, same solver, same run)",
                "value": total_jvps / elapsed,
                "value_frozen": total_jvps / elapsed,
                "value_reference_semantic": None,
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
                    "checkpoint_trail": ("disabled" if args.no_files else
                                         ("in program order on one writer thread, on disk when the solve returns (inside the "
                                          "timed region)" if trail.TRAIL.enabled else "written inside the call that asks for it")),
                    "tracer_modules_per_gpu": 1,
                    "krylov_iterations": args.steps,
                    "untimed_iterations_until_the_schedule_cache_was_allocated": extra_warm,
                    "parallelism": f"tracer-module-per-gpu x{world}",
                    "collective_backend": backend if world > 1 else None,
                    "integrator_mode": integrator_mode(),
                },
                "roofline": roof,
                "end_to_end": {
                    "what": "algorithmic bytes of ALL launches of the dominant kernel in the timed region / wall "
                            "time of the timed region (host round trips, other kernels and the file trail included "
                            "in the time, their bytes not counted): a lower bound of the whole path's rate",
                    "dominant_kernel_bytes_per_jvp": totals["bytes"] / args.steps,
                    "dominant_kernel_launches_per_jvp": totals["launches"] / args.steps,
                    "achieved_GBs_per_gpu": totals["bytes"] / elapsed / 1e9,
                    "frac_of_hbm_peak": totals["bytes"] / elapsed / 1e9 / HBM_PEAK_GBS,
                },
                "jvp": {
                    "mode": ("frozen controller (internal numerical differentiation): the perturbed year of every product "
                             "repeats the accepted steps, Newton iteration counts, Jacobian times and factorisations of "
                             "the free-running year that produced F(x), checked afterwards against SciPy's Newton "
                             "convergence test; NK2D_JVP_FROZEN=0 gives two free-running years as the reference has them"
                             if os.environ.get("NK2D_JVP_FROZEN", "1") != "0" else "free-running perturbed years"),
                    "perturbed_year": {k: jvp_stats[k] for k in year_keys},
                    "base_year_free_running": {k: wl.fwd_stats[k] for k in year_keys},
                    "frozen_years_rejected": eng.frozen_fallbacks(),
                    "frozen_years_resumed": eng.frozen_resumes(),
                    "error_estimates_checked_last_year": jvp_stats.get("nerr_checked"),
                    "largest_error_estimate_last_year": jvp_stats.get("max_err"),
                },
                "roofline_year": {
                    "what": "algorithmic bytes of ALL Newton-iteration launches of one perturbed (frozen) year / the seconds "
                            "of that year (every other kernel of the year in the time, its bytes not counted)",
                    "bytes_per_year": totals["bytes"] / args.steps,
                    "seconds_per_year": jvp_stats["seconds"],
                    "achieved": totals["bytes"] / args.steps / jvp_stats["seconds"] / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": totals["bytes"] / args.steps / jvp_stats["seconds"] / 1e9 / HBM_PEAK_GBS,
                },
                "setup_seconds": {"total": wl.setup_s, "precond_factorisation": wl.precond_setup_s},
            }
            out["config"]["schedule_cache_gb"] = 1.0e-9 * eng.counter("frozen_cache_bytes")
            if world == 1:
                if "one_launch_years" in roof:
                    # the same year launch by launch (option "frozen_persistent" 0), with the per-launch figures of its dominant
                    # kernel as rounds 1 and 2 reported them
                    eng.set_option("frozen_persistent", 0)
                    eng.profile_reset(1)
                    eng._launch_us_base = eng.counter("frozen_launch_us")
                    eng._launch_years_base = eng.counter("frozen_persistent_years")
                    t_l = time.perf_counter()
                    wl.krylov(2, "krylov_launch_path", device)
                    eng.sync()
                    t_l = time.perf_counter() - t_l
                    out["launch_per_phase_path"] = {"what": "the same Krylov iterations with every frozen year as a sequence of "
                                                            "launches (option frozen_persistent 0)",
                                                    "jvps_per_s": 2.0 / t_l, "ms_per_jvp": 500.0 * t_l,
                                                    "year_seconds": ModelState.last_stats[0]["seconds"],
                                                    "roofline": roofline_of(eng, n)}
                    eng.set_option("frozen_persistent", 1)
                    out["config"]["launch_per_phase_jvps_per_s"] = out["launch_per_phase_path"]["jvps_per_s"]
                    out["roofline"]["launch_per_phase_jvps_per_s"] = out["launch_per_phase_path"]["jvps_per_s"]
                out["roofline_precond"] = precond_roofline(eng)
                # the same Krylov iterations as ONE C call with every vector in HBM and no file trail (nk2d_gmres_solve),
                # for what the checkpoint trail and the Python mirror cost in the timed region above
                x_vec, f_vec = wl.iterate.tracer_modules[0].vec, wl.fcn.tracer_modules[0].vec
                sched = (wl.fcn._sched or {}).get("iage") if os.environ.get("NK2D_JVP_FROZEN", "1") != "0" else None
                eng.gmres_solve(x_vec, f_vec, 0.0, 0, 1, sched=sched)
                eng.sync()
                t_g = time.perf_counter()
                eng.gmres_solve(x_vec, f_vec, 0.0, 0, args.steps, sched=sched)
                eng.sync()
                t_g = time.perf_counter() - t_g
                out["gmres_solve_in_hbm"] = {"what": "nk2d_gmres_solve: the same Krylov iterations in one C call, no files",
                                             "jvps_per_s": args.steps / t_g, "ms_per_jvp": 1000.0 * t_g / args.steps}
        progress("reference-semantic products (two free-running years)")
        if os.environ.get("NK2D_JVP_FROZEN", "1") != "0":
            # the reference's own product, two free-running years, through the same solver on every rank: what `value`
            # is without the frozen controller (measured, two Krylov iterations, same barriers and MAX over ranks)
            os.environ["NK2D_JVP_FROZEN"] = "0"
            try:
                eng.sync()
                if world > 1:
                    torch.distributed.barrier()
                eng.profile_reset(1)      # (the bytes of the Newton commands of these years: roofline_free_year below)
                t_f = time.perf_counter()
                wl.krylov(2, "krylov_free", device)
                eng.sync()
                if world > 1:
                    torch.distributed.barrier()
                el = torch.tensor([time.perf_counter() - t_f], dtype=torch.float64, device=device)
                if world > 1:
                    torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
                t_f = float(el.item())
            finally:
                os.environ["NK2D_JVP_FROZEN"] = "1"
            if rank == 0:
                out["value_reference_semantic"] = 2.0 * world / t_f
                # (also where the driver's parser keeps them: inside `config` and `roofline`)
                out["config"]["value_reference_semantic"] = 2.0 * world / t_f
                out["roofline"]["value_reference_semantic"] = 2.0 * world / t_f
                free_st = ModelState.last_stats[0]
                free_totals = eng.profile_totals()
                # the free-running year (k_stream: one resident kernel executes the host controller's launches as commands): the
                # algorithmic bytes of EVERY Newton command the year executed -- the speculative ones that were dropped and the
                # iterations of attempts that failed included, they ran -- over the seconds of the year
                out["roofline_free_year"] = {
                    "what": "algorithmic bytes of all Newton-iteration commands of one free-running perturbed year (command stream; "
                            "dropped speculation and failed attempts included) / the seconds of that year (set-up, error-estimate and "
                            "step-boundary commands in the time, their bytes not counted)",
                    "bytes_per_year": free_totals["bytes"] / 2.0,
                    "newton_commands_per_year": free_totals["launches"] / 2.0,
                    "seconds_per_year": free_st["seconds"],
                    "achieved": free_totals["bytes"] / 2.0 / free_st["seconds"] / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": free_totals["bytes"] / 2.0 / free_st["seconds"] / 1e9 / HBM_PEAK_GBS,
                    "years_as_command_streams": eng.counter("stream_years_run") > 0,
                }
                out["roofline"]["free_year_frac"] = out["roofline_free_year"]["frac"]
                out["jvp"]["free_running_products"] = {"jvps_per_s": 2.0 * world / t_f, "ms_per_jvp": 1000.0 * t_f / 2.0,
                                                       "krylov_iterations": 2,
                                                       "perturbed_year": {k: free_st[k] for k in year_keys},
                                                       "years_as_command_streams": eng.counter("stream_years_run")}
        elif rank == 0:
            out["value_reference_semantic"] = out["value"]
            out["value_frozen"] = None
        one_gpu_ms = 1000.0 * elapsed / args.steps
        wl_base_stats = dict(wl.fwd_stats)
        faithful_attempts = None
        if rank == 0 and world == 1 and args.cpu_baseline_seconds > 0:
            # the CPU baseline restates the reference's integrator (SciPy's decisions): what one of ITS years needs is
            # counted by a year of this engine in that mode (counters within 10 % of solve_ivp's, tests/test_gpu_comp_fcn.py)
            eng.set_option("jac_fresh", 0)
            _, st_f, _ = eng.comp_fcn(wl.iterate.tracer_modules[0].vec)
            eng.set_option("jac_fresh", integrator_mode()["jac_fresh"])
            faithful_attempts = st_f["nsteps"] + st_f["nrejected"]
        wl.close()
        wl = None
        if world > 1 and args.aux_budget > 0:
            torch.distributed.barrier()
            deadline = AuxDeadline(args.aux_budget, rank, out)
        if world >= 2 and not args.no_shard:
            progress("shard_e2: one module, tracer per rank")
            base = wl_base_stats
            hint = base["nnewton"] + 2 * (base["nsteps"] + base["nrejected"]) + 10
            if deadline:
                deadline.leg = "shard_e2"
            try:
                shard = run_shard_e2(args, rank, local_rank, world, backend, one_gpu_ms, hint)
            except Exception as exc:           # an auxiliary leg must not cost the line
                shard = {"error": f"{type(exc).__name__}: {exc}"}
            if rank == 0:
                out["shard_e2"] = shard
        if not args.no_mix:
            progress("config4_mix: iage + phosphorus + forced")
            if deadline:
                deadline.leg = "config4_mix"
            try:
                mix = run_config4_mix(args, rank, local_rank, world, device)
            except Exception as exc:           # an auxiliary leg must not cost the line (same failure on every rank)
                mix = {"error": f"{type(exc).__name__}: {exc}"}
            if rank == 0:
                out["config4_mix"] = mix
                if "distributed" in mix:
                    out["strong_scaling"] = {
                        "what": "config4_mix: the same three-module Krylov iterations on one GPU and on "
                                f"{mix['distributed']['ranks_used']} GPUs (modules dealt round-robin)",
                        "n_gpus_used": mix["distributed"]["ranks_used"],
                        "speedup": mix["distributed"]["speedup_over_one_gpu"],
                        "one_gpu_ms_per_krylov_iteration": mix["one_gpu"]["ms_per_krylov_iteration"],
                        "ms_per_krylov_iteration": mix["distributed"]["ms_per_krylov_iteration"]}
        if not args.no_shard3 and world >= 2:
            progress("shard_e3: phosphorus, Krylov basis columns over the ranks")
            if deadline:
                deadline.leg = "shard_e3"
            try:
                shard3 = run_shard_e3(args, rank, local_rank, world, backend)
            except Exception as exc:
                shard3 = {"error": f"{type(exc).__name__}: {exc}"}
            if rank == 0:
                out["shard_e3"] = shard3
        if deadline:
            deadline.cancel()
        if rank == 0:
            if world == 1 and not args.no_shard3 and args.shard3_first:
                # (diagnostic order, round-3 verdict weak 7: in round 3 the ladder ran 2.7 times slower behind this leg)
                progress("shard_e3 (one rank) ahead of the ladder")
                try:
                    out["shard_e3"] = run_shard_e3(args, rank, local_rank, world, backend)
                except Exception as exc:
                    out["shard_e3"] = {"error": f"{type(exc).__name__}: {exc}"}
            if world == 1 and not args.no_ladder:
                progress("ladder 26 .. 208")
                out["ladder"] = run_ladder(local_rank, device, args)
                out["ladder"].append({"grid": n, "jvps_per_s": out["value"], "ms_per_jvp": out["ms_per_step"],
                                      "forward_year_s": jvp_stats["seconds"], "nsteps": jvp_stats["nsteps"],
                                      "nlaunch": jvp_stats["nlaunch"], "roofline_frac": out["roofline"]["frac"],
                                      "avg_launch_us": out["roofline"]["avg_launch_us"],
                                      "algorithmic_bytes_per_launch": out["roofline"]["algorithmic_bytes_per_launch"]})
            if world == 1 and args.cpu_baseline_seconds > 0:
                progress("CPU baseline (oracle, one host thread)")
                out["cpu_baseline"] = cpu_baseline(n, faithful_attempts, args.cpu_baseline_seconds)
                out["cpu_baseline"]["host_cpus"] = os.cpu_count()
                if "ladder" in out:
                    out["ladder"][-1]["cpu_oracle"] = {
                        "seconds_per_year": 1.0 / out["cpu_baseline"]["value"], "full_year": False, "cores": 1}
            if world == 1 and not args.no_spinup:
                progress("newton_spinup: the whole Newton-Krylov solve")
                try:
                    out["newton_spinup"] = run_newton_spinup(n)
                    out["config"]["newton_spinup_products_per_s"] = out["newton_spinup"]["products_per_second_inside_the_solve"]
                except Exception as exc:       # an auxiliary leg must not cost the line
                    out["newton_spinup"] = {"error": f"{type(exc).__name__}: {exc}"}
            if world == 1 and not args.no_shard3 and not args.shard3_first:
                progress("shard_e3 (one rank): phosphorus through the column-sharded loop")
                try:
                    out["shard_e3"] = run_shard_e3(args, rank, local_rank, world, backend)
                except Exception as exc:       # an auxiliary leg must not cost the line
                    out["shard_e3"] = {"error": f"{type(exc).__name__}: {exc}"}
            print(json.dumps(out), flush=True)
    finally:
        if deadline:
            deadline.cancel()
        if wl is not None:
            wl.close()
        if world > 1:
            torch.distributed.destroy_process_group()


def integrator_mode():
    """the controller mode of the forward years in this run (engine defaults, overridable from the environment)"""
    from nk_ooc_amd.engine import DEFAULT_JAC_FRESH, DEFAULT_JAC_STAGE, DEFAULT_LIN_TOL

    return {"jac_fresh": int(float(os.environ.get("NK2D_JAC_FRESH", DEFAULT_JAC_FRESH))),
            "jac_stage": int(float(os.environ.get("NK2D_JAC_STAGE", DEFAULT_JAC_STAGE))),
            "jvp_frozen": os.environ.get("NK2D_JVP_FROZEN", "1") != "0",
            "lin_tol": float(os.environ.get("NK2D_LIN_TOL", DEFAULT_LIN_TOL)), "rtol": 1.0e-6, "atol": 1.0e-6}


if __name__ == "__main__":
    main()
