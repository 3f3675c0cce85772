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

    def __init__(self, flags, device):
        self._flags = np.asarray(flags)
        self._device = device

    def all(self):
        local = bool(self._flags.all())
        dist = _dist()
        if dist is None or dist.get_world_size() == 1:
            return local
        import torch

        flag = torch.tensor([1 if local else 0], dtype=torch.int32, device=self._device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())


class DistributedKrylovSolver(KrylovSolver):
    """KrylovSolver over the LOCAL tracer modules with a global stopping test"""

    def __init__(self, iterate, solverinfo, resume, rewind, hist_fname, device=None):
        super().__init__(iterate, solverinfo, resume, rewind, hist_fname)
        self._flag_device = device if device is not None else "cpu"

    def converged(self, beta, precond_resid_norm):
        return _AllFlag(super().converged(beta, precond_resid_norm), self._flag_device)


def init_process_group_from_env(backend):
    """one process per GPU, launched by torch.distributed.run"""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world
