"""Multi-GPU layout of the Krylov hot path: tracer modules are partitioned over ranks.

The reference solves all tracer modules in one process; they are mathematically
independent (separate `solve_ivp` calls, separate Hessenbergs and preconditioners,
`nk_ooc/py_driver_2d/model_state.py:95-121`, `nk_ooc/krylov_solver.py:114-121,173-180`)
and only the stopping test couples them: `converged(...).all()` over every (module,
region) (`krylov_solver.py:159`).  Here each rank owns the modules assigned to its GPU and
runs the unchanged Krylov loop on them; the ONLY collective is an all-reduce (logical AND)
of the convergence flag once per Krylov iteration -- over RCCL/xGMI on GPUs
(`backend="nccl"`), over gloo in the CPU tests.  A module that has converged keeps
iterating until all have, exactly as in the reference.
"""

import os

import numpy as np

from .krylov_solver import KrylovSolver


def partition_modules(names, world_size):
    """round-robin assignment of tracer-module names to ranks"""
    return [list(names[rank::world_size]) for rank in range(world_size)]


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


class _AllFlag:
    """what `.all()` is called on: AND over the local (module, region) flags and over ranks"""

    def __init__(self, flags, buf):
        self._flags = np.asarray(flags)
        self._buf = buf

    def all(self):
        local = bool(self._flags.all())
        dist = _dist()
        if dist is None or dist.get_world_size() == 1:
            return local
        # the solver's one flag tensor, allocated once (on the GPU for nccl = RCCL, on the host for gloo)
        self._buf.fill_(1 if local else 0)
        dist.all_reduce(self._buf, op=dist.ReduceOp.MIN)
        return bool(self._buf.item())


class DistributedKrylovSolver(KrylovSolver):
    """KrylovSolver over the LOCAL tracer modules with a global stopping test"""

    def __init__(self, iterate, solverinfo, resume, rewind, hist_fname, device=None):
        super().__init__(iterate, solverinfo, resume, rewind, hist_fname)
        self._flag_buf = None
        if _dist() is not None:
            import torch

            self._flag_buf = torch.zeros(1, dtype=torch.int32, device=device if device is not None else "cpu")

    def converged(self, beta, precond_resid_norm):
        return _AllFlag(super().converged(beta, precond_resid_norm), self._flag_buf)


def init_process_group_from_env(backend):
    """one process per GPU, launched by torch.distributed.run"""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world


# ---------------------------------------------------------------------------------------------------
# Level 2 of SURVEY.md section 8(e): the tracers of ONE module sharded over ranks.
#
# The tracers of iage (and of any module whose Jacobian is block diagonal over its tracers:
# nk_ooc/py_driver_2d/advection.py:179, iage.py:64) never exchange field data.  What couples them is
# scalar: SciPy's Radau takes every decision from RMS norms over the whole module (radau.py:118,481,
# common.py:63-65), and the Krylov solver's inner products sum over the module's tracers
# (nk_ooc/tracer_module_state_base.py:379-388).  A rank therefore owns an engine for its tracers only
# (tc = 1 for iage) and the library's norm hook (nk2d_set_norm_hook) turns every norm the integrator
# reads into an all-reduce(SUM) of one double -- RCCL on GPUs, gloo in the CPU tests -- so that all
# ranks take bit-identical decisions.  The Arnoldi projections use the fused multi-dot of section 8(e):
# classical Gram-Schmidt with re-orthogonalisation (CGS-2), 2 all-reduces of (j+1) nreg doubles per
# Krylov iteration plus one for the norm, instead of the j+1 sequential ones modified Gram-Schmidt
# (model_state_base.py:365-377) would need.  These collectives are latency-bound (8 bytes ... a few
# hundred bytes): the xGMI bandwidth is irrelevant, their COUNT is what costs.
# ---------------------------------------------------------------------------------------------------
class ShardComm:
    """sum of small float64 arrays over the ranks of a process group; counts its calls"""

    def __init__(self, device="cpu", group=None, capacity=1024):
        import torch

        self._torch = torch
        self._group = group
        self._buf = torch.zeros(capacity, dtype=torch.float64, device=device)
        self._host = torch.zeros(capacity, dtype=torch.float64).pin_memory() if str(device) != "cpu" else None
        self.calls = 0

    def allreduce(self, arr):
        dist = _dist()
        arr = np.asarray(arr, dtype=np.float64)
        if dist is None or dist.get_world_size(self._group) == 1:
            return arr.copy()
        n = arr.size
        flat = self._torch.from_numpy(np.ascontiguousarray(arr).reshape(-1))
        view = self._buf[:n]
        view.copy_(flat)
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self._group)
        self.calls += 1
        return view.cpu().numpy().reshape(arr.shape).copy()

    def allreduce_scalar(self, val):
        return float(self.allreduce(np.array([val]))[0])


def iage_shard_engine(grid, shard, comm, device_id=0, **kwargs):
    """engine of ONE tracer of the iage module (shard 0: iage, 1: iage_slow_rest; iage.py:12-41) whose
    Radau controller is coupled with the other shard's through `comm`"""
    from .engine import ModuleEngine

    rate = 24.0 / 86400.0 * 10.0 / grid.depth.delta[0]
    rates = (rate, 0.01 * rate)
    eng = ModuleEngine(grid, tc=1, surf_rate=(rates[shard],), const_src=1.0 / (365.0 * 86400.0),
                       device_id=device_id, **kwargs)
    eng.set_option("device_ctl", 0)
    eng.set_norm_hook(comm.allreduce_scalar, 2.0 * len(grid.depth) * len(grid.ypos))
    return eng


class ShardedVectorSpace:
    """region-weighted inner products and Gram-Schmidt of a module whose tracers are sharded: local
    kernels + one all-reduce per reduction"""

    def __init__(self, eng, comm):
        self.eng, self.comm = eng, comm

    def dot(self, a, b):
        return self.comm.allreduce(self.eng.dot(a, b))

    def norm(self, a):
        return np.sqrt(self.dot(a, a))

    def cgs2(self, w, basis):
        """orthogonalise w against the (orthonormal) basis in place: classical Gram-Schmidt, twice.
        Returns the projection coefficients (n, nreg) -- what modified Gram-Schmidt returns, to rounding."""
        h_tot = np.zeros((len(basis), self.eng.nreg))
        for sweep in range(2):
            h_val = self.comm.allreduce(self.eng.multi_dot(w, basis))
            # cells outside every region: the reference's projections subtract each basis vector once there
            # (region broadcast fills 1.0, tracer_module_state_base.py:502-515) -- the first pass does that
            self.eng.multi_axpy(w, basis, h_val, fill=1.0 if sweep == 0 else 0.0)
            h_tot += h_val
        return h_tot


def sharded_gmres(eng, comm, x, fx, rel_tol, min_iter, max_iter, sched=None):
    """KrylovSolver.solve (nk_ooc/krylov_solver.py:85-165) for the local tracers of a sharded module:
    the same loop as nk2d_gmres_solve with every reduction all-reduced over the shards and CGS-2
    instead of sequential MGS.  The preconditioner of iage is block diagonal over its tracers
    (iage.py:66-93): each shard applies its own block.  `sched`: the accepted steps of the (coupled) year that
    produced fx, `eng.last_schedule()` -- identical on every shard, the controller saw module-wide norms; the
    perturbed years then repeat those steps (internal numerical differentiation) and need NO exchange at all: the
    only collectives left per Krylov iteration are its five reductions.  Returns (increment, info)."""
    from .krylov_solver import least_squares_coeffs

    vs = ShardedVectorSpace(eng, comm)
    r0 = eng.precond_apply(fx)
    beta = vs.norm(r0)
    basis = [eng.scale(r0, -(1.0 / beta))]
    prods = []
    hess = np.zeros((1, 1, 0, eng.nreg))
    resid_norms = []
    j = 0
    while True:
        sigma = 1.0e-4 * vs.norm(x)
        sigma = np.where(sigma == 0.0, 1.0, sigma)
        perturbed = eng.axpby(1.0, x, sigma, basis[j])
        fpert = None
        if sched is not None and len(sched) > 0:
            from .engine import Nk2dFrozenMismatch

            bad = 0.0
            try:
                fpert, _ = eng.comp_fcn_frozen(perturbed, sched)
            except Nk2dFrozenMismatch:
                bad = 1.0
            # a shard whose frozen year was rejected (recorded Newton counts not enough for its perturbed state) needs
            # a coupled free-running year -- which every shard must then run: one flag per product
            if comm.allreduce_scalar(bad) > 0.0:
                fpert = None
        if fpert is None:
            fpert, _, _ = eng.comp_fcn(perturbed)
        w = eng.precond_apply(eng.diff_scale(fpert, fx, 1.0 / sigma))
        prods.append(w.copy())
        grown = np.zeros((1, j + 2, j + 1, eng.nreg))
        grown[:, : j + 1, :j, :] = hess
        grown[0, : j + 1, j, :] = vs.cgs2(w, basis)
        grown[0, j + 1, j, :] = vs.norm(w)
        hess = grown
        coeff = least_squares_coeffs(beta[np.newaxis], hess)[0]
        approx = eng.lin_comb(basis, coeff)
        resid = eng.lin_comb(prods, coeff)
        eng.axpby(1.0, resid, 1.0, r0, out=resid)
        resid_norms.append(vs.norm(resid))
        j += 1
        if (j >= min_iter and (resid_norms[-1] < rel_tol * beta).all()) or j >= max_iter:
            break
        basis.append(eng.scale(w, 1.0 / hess[0, j, j - 1, :]))
    return approx, {"beta": beta, "h_mat": hess[0], "resid_norm": np.array(resid_norms), "iters": j,
                    "allreduces": comm.calls}
