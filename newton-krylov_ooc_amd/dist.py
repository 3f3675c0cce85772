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

    def __init__(self, flags, buf, group=None):
        self._flags = np.asarray(flags)
        self._buf = buf
        self._group = group

    def all(self):
        local = bool(self._flags.all())
        dist = _dist()
        if dist is None or dist.get_world_size(self._group) == 1:
            return local
        # the solver's one flag tensor, allocated once (on the GPU for nccl = RCCL, on the host for gloo)
        self._buf.fill_(1 if local else 0)
        dist.all_reduce(self._buf, op=dist.ReduceOp.MIN, group=self._group)
        return bool(self._buf.item())


class DistributedKrylovSolver(KrylovSolver):
    """KrylovSolver over the LOCAL tracer modules with a global stopping test"""

    def __init__(self, iterate, solverinfo, resume, rewind, hist_fname, device=None, group=None):
        """group: the ranks that hold the modules of this solve (default: all)"""
        super().__init__(iterate, solverinfo, resume, rewind, hist_fname)
        self._flag_buf = None
        self._group = group
        if _dist() is not None:
            import torch

            self._flag_buf = torch.zeros(1, dtype=torch.int32, device=device if device is not None else "cpu")

    def converged(self, beta, precond_resid_norm):
        return _AllFlag(super().converged(beta, precond_resid_norm), self._flag_buf, self._group)


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
        # everything a call needs exists beforehand: the tensor the collective works on (in HBM for nccl = RCCL, on the
        # host for gloo) and, for a device tensor, ONE pinned host staging buffer with a NumPy view -- no tensor is
        # created, no pageable copy made per call (these collectives are 8 ... a few hundred bytes: their fixed cost is all
        # there is)
        self._buf = torch.zeros(capacity, dtype=torch.float64, device=device)
        self._on_device = str(device) != "cpu"
        self._stage = torch.zeros(capacity, dtype=torch.float64).pin_memory() if self._on_device else self._buf
        self._stage_np = self._stage.numpy()
        self.calls = 0

    def allreduce(self, arr):
        dist = _dist()
        arr = np.asarray(arr, dtype=np.float64)
        if dist is None or dist.get_world_size(self._group) == 1:
            return arr.copy()
        n = arr.size
        self._stage_np[:n] = arr.reshape(-1)
        view = self._buf[:n]
        if self._on_device:
            view.copy_(self._stage[:n], non_blocking=True)
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self._group)
        if self._on_device:
            self._stage[:n].copy_(view)          # waits for the collective on torch's stream
        self.calls += 1
        return self._stage_np[:n].reshape(arr.shape).copy()

    def allreduce_scalar(self, val):
        dist = _dist()
        if dist is None or dist.get_world_size(self._group) == 1:
            return float(val)
        self._stage_np[0] = val
        view = self._buf[:1]
        if self._on_device:
            view.copy_(self._stage[:1], non_blocking=True)
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self._group)
        if self._on_device:
            self._stage[:1].copy_(view)
        self.calls += 1
        return float(self._stage_np[0])


def iage_shard_engine(grid, shard, comm, device_id=0, **kwargs):  # noqa: D401
    """engine of ONE tracer of the iage module (shard 0: iage, 1: iage_slow_rest; iage.py:12-41) whose
    Radau controller is coupled with the other shard's through `comm`"""
    from .engine import ModuleEngine

    rate = 24.0 / 86400.0 * 10.0 / grid.depth.delta[0]
    rates = (rate, 0.01 * rate)
    eng = ModuleEngine(grid, tc=1, surf_rate=(rates[shard],), const_src=1.0 / (365.0 * 86400.0),
                       device_id=device_id, **kwargs)
    eng.set_option("device_ctl", 0)
    # the vector hook: the controller pairs what it can into one all-reduce (norm + next norm, norm + error estimate)
    eng.set_norm_hook(comm.allreduce, 2.0 * len(grid.depth) * len(grid.ypos), vector=True)
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


# ---------------------------------------------------------------------------------------------------
# Level 3 of SURVEY.md section 8(e): the columns of the Krylov basis sharded over ranks (BASELINE configs[4]).
#
# The reference keeps every Arnoldi vector v_i and every preconditioned product W_i in a file of its own and walks
# through them in mod_gram_schmidt (nk_ooc/model_state_base.py:365-377) and lin_comb (:619-624).  Here column i lives in
# the HBM of rank i mod G; the state x, F(x) and the preconditioner are replicated.  Per Krylov iteration j:
#   * the owner of v_j computes w = M^-1 J v_j (one perturbed year) and BROADCASTS it (N doubles over xGMI);
#   * Gram-Schmidt, classical and twice (CGS-2): every rank forms the inner products with ITS columns in one fused
#     multi-dot, the (j + 1) nreg numbers are all-reduced (a gather by zero padding), every rank forms the partial sum
#     of its columns, the partial sums are all-reduced (N doubles) and subtracted everywhere;
#   * the preconditioned residual sum_i c_i W_i + M^-1 F: local partial sums + one all-reduce of N doubles; the iterate
#     x_j = sum_i c_i v_i likewise, once, when the solve ends.
# Nothing of the product is shared out: the layout buys HBM (a basis that does not fit one GPU), not time -- at these
# sizes (all vectors of a solve fit one GPU's Infinity Cache) the collectives are pure overhead, and bench.py reports the
# scaling it measures, which is flat to negative.
# ---------------------------------------------------------------------------------------------------
class ColumnComm(ShardComm):
    """ShardComm + whole-vector collectives on engine vectors: zero-copy on device tensors (RCCL) where the engine hands out
    a tensor view of its HBM (`vec_tensor`), through host arrays otherwise (gloo rehearsals, NumPy stand-in engines)"""

    def __init__(self, rank, size, device="cpu", group=None, capacity=4096):
        super().__init__(device=device, group=group, capacity=capacity)
        self.rank, self.size = rank, size
        self.device = device
        self.vec_calls = 0
        self.vec_bytes = 0

    def allreduce(self, arr):
        if self.size == 1:
            return np.asarray(arr, dtype=np.float64).copy()
        return super().allreduce(arr)

    def _global(self, rank):
        dist = _dist()
        return rank if self._group is None else dist.get_global_rank(self._group, rank)

    def _view(self, eng, vec):
        """(tensor, write_back): a tensor the collective can work on in place"""
        if str(self.device) != "cpu" and hasattr(eng, "vec_tensor"):
            eng.sync()
            return eng.vec_tensor(vec), None
        host = np.ascontiguousarray(eng.download(vec))
        return self._torch.from_numpy(host.reshape(-1)), host

    def _done(self, eng, vec, tensor, host):
        if host is None:
            self._torch.cuda.current_stream(tensor.device).synchronize()
        else:
            eng.upload(host, out=vec)
        self.vec_calls += 1
        self.vec_bytes += tensor.numel() * 8

    def bcast_vec(self, eng, vec, src):
        dist = _dist()
        if dist is None or self.size == 1:
            return vec
        tensor, host = self._view(eng, vec)
        dist.broadcast(tensor, src=self._global(src), group=self._group)
        self._done(eng, vec, tensor, host)
        return vec

    def allreduce_vec(self, eng, vec):
        dist = _dist()
        if dist is None or self.size == 1:
            return vec
        tensor, host = self._view(eng, vec)
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self._group)
        self._done(eng, vec, tensor, host)
        return vec


def column_sharded_gmres(eng, comm, x, fx, rel_tol, min_iter, max_iter, sched=None):
    """KrylovSolver.solve (nk_ooc/krylov_solver.py:85-165) with basis column i and product W_i on rank i mod comm.size
    (see above); every rank holds the whole module (x, fx, preconditioner) and returns the same increment and numbers.
    `sched`: the accepted steps of the year behind fx on THIS rank's engine (the products repeat them)."""
    from .krylov_solver import least_squares_coeffs

    rank, size = comm.rank, comm.size
    zero = eng.scale(x, 0.0)

    def product(direction):
        sigma = 1.0e-4 * np.sqrt(eng.dot(x, x))
        sigma = np.where(sigma == 0.0, 1.0, sigma)
        perturbed = eng.axpby(1.0, x, sigma, direction)
        fpert = None
        if sched is not None and len(sched) > 0:
            from .engine import Nk2dFrozenMismatch, Nk2dScheduleMismatch

            try:
                fpert, _ = eng.comp_fcn_frozen(perturbed, sched)
            except (Nk2dFrozenMismatch, Nk2dScheduleMismatch):
                fpert = None
        if fpert is None:
            fpert, _, _ = eng.comp_fcn(perturbed)
        return eng.precond_apply(eng.diff_scale(fpert, fx, 1.0 / sigma))

    def local_part(store, coeff_of, fill):
        """- sum over this rank's columns i of bcast(coeff_of(i)) store[i]  (one fused launch; zero without columns)"""
        part = eng.scale(zero, 1.0)
        cols = sorted(store)
        if cols:
            eng.multi_axpy(part, [store[i] for i in cols], np.stack([coeff_of(i) for i in cols]), fill=fill)
        return part

    r0 = eng.precond_apply(fx)
    beta = np.sqrt(eng.dot(r0, r0))
    basis, prods = {}, {}
    if rank == 0:
        basis[0] = eng.scale(r0, -(1.0 / beta))
    hess = np.zeros((1, 1, 0, eng.nreg))
    resid_norms = []
    coeff = None
    j = 0
    while True:
        owner = j % size
        w = product(basis[j]) if rank == owner else eng.scale(zero, 1.0)
        comm.bcast_vec(eng, w, owner)
        if rank == owner:
            prods[j] = w.copy()
        # CGS-2 against the j + 1 columns spread over the ranks
        h_tot = np.zeros((j + 1, eng.nreg))
        for sweep in range(2):
            h_loc = np.zeros((j + 1, eng.nreg))
            cols = sorted(basis)
            if cols:
                h_loc[cols] = eng.multi_dot(w, [basis[i] for i in cols])
            h_val = comm.allreduce(h_loc)
            # cells outside every region: the reference subtracts each basis vector once there (region broadcast fills 1.0,
            # tracer_module_state_base.py:502-515) -- the first pass does that
            part = local_part(basis, lambda i: h_val[i], 1.0 if sweep == 0 else 0.0)
            comm.allreduce_vec(eng, part)
            eng.axpby(1.0, w, 1.0, part, out=w)
            h_tot += h_val
        grown = np.zeros((1, j + 2, j + 1, eng.nreg))
        grown[:, : j + 1, :j, :] = hess
        grown[0, : j + 1, j, :] = h_tot
        grown[0, j + 1, j, :] = np.sqrt(eng.dot(w, w))
        hess = grown
        coeff = least_squares_coeffs(beta[np.newaxis], hess)[0]
        resid = local_part(prods, lambda i: -coeff[i], 1.0)
        comm.allreduce_vec(eng, resid)
        eng.axpby(1.0, resid, 1.0, r0, out=resid)
        resid_norms.append(np.sqrt(eng.dot(resid, resid)))
        j += 1
        if (j >= min_iter and (resid_norms[-1] < rel_tol * beta).all()) or j >= max_iter:
            break
        if rank == j % size:
            basis[j] = eng.scale(w, 1.0 / hess[0, j, j - 1, :])
    approx = local_part(basis, lambda i: -coeff[i], 1.0)
    comm.allreduce_vec(eng, approx)
    return approx, {"beta": beta, "h_mat": hess[0], "resid_norm": np.array(resid_norms), "iters": j,
                    "allreduces": comm.calls, "vector_collectives": comm.vec_calls, "vector_bytes": comm.vec_bytes,
                    "columns_here": sorted(basis)}
