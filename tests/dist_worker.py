"""worker of tests/test_dist.py: one rank of a world_size-2 gloo run of the module-per-rank
Krylov loop.  The engine is a NumPy stand-in (the distributed logic under test lives in
nk_ooc_amd.dist / krylov_solver / model_state, not in the kernels)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class FakeVec:
    def __init__(self, eng, arr):
        self.eng = eng
        self.arr = arr
        self.ptr = id(self)

    def copy(self):
        return FakeVec(self.eng, self.arr.copy())


class FakeEngine:
    """linear model F(x) = A x - x + b per tracer module, identity-like preconditioner"""

    def __init__(self, nz, ny, tc, seed):
        self.nz, self.ny, self.tc = nz, ny, tc
        self.shape = (tc, nz, ny)
        self.nreg = 1
        self.module_kind = 0
        self.state_dependent_precond = False
        rng = np.random.default_rng(seed)
        n = tc * nz * ny
        self.A = 0.3 * rng.standard_normal((n, n)) / np.sqrt(n)
        self.b = rng.standard_normal(n)
        self.calls = 0

    def close(self):
        pass

    def set_region(self, mask, weight):
        self.mask = mask
        w = np.where(mask > 0, weight, 0.0)
        self.wn = (w / w.sum()).reshape(-1)
        self.nreg = int(mask.max())

    def new_vec(self):
        return FakeVec(self, np.zeros(self.shape))

    def upload(self, host, out=None):
        out = self.new_vec() if out is None else out
        out.arr[:] = np.asarray(host).reshape(self.shape)
        return out

    def download(self, vec):
        return vec.arr.copy()

    def sync(self):
        pass

    def last_schedule(self):
        """no Radau steps behind the CPU stand-in's years: products stay free-running"""
        return np.zeros((0, 6))

    def comp_fcn(self, x, out=None, **kw):
        self.calls += 1
        res = (self.A @ x.arr.reshape(-1) - x.arr.reshape(-1) + self.b).reshape(self.shape)
        return FakeVec(self, res), {"nsteps": 0}, None

    def precond_apply(self, v, out=None):
        return FakeVec(self, -v.arr)

    def dot(self, a, b):
        prod = (a.arr * b.arr).reshape(self.tc, -1)
        return np.array([sum(np.dot(self.wn, p) for p in prod)])

    def scale(self, x, s, out=None):
        res = x.arr * np.asarray(s).reshape(-1)[0]
        if out is not None:
            out.arr[:] = res
            return out
        return FakeVec(self, res)

    def axpby(self, a, x, b, y, out=None):
        res = np.asarray(a).reshape(-1)[0] * x.arr + np.asarray(b).reshape(-1)[0] * y.arr
        if out is not None:
            out.arr[:] = res
            return out
        return FakeVec(self, res)

    def diff_scale(self, x, y, s, out=None):
        res = (x.arr - y.arr) * np.asarray(s).reshape(-1)[0]
        if out is not None:
            out.arr[:] = res
            return out
        return FakeVec(self, res)

    def lin_comb(self, vecs, coef, out=None):
        res = sum(c[0] * v.arr for c, v in zip(np.asarray(coef), vecs))
        return FakeVec(self, res)

    def mgs(self, w, basis):
        h = np.empty((len(basis), 1))
        for i, v in enumerate(basis):
            h[i] = self.dot(w, v)
            w.arr -= h[i, 0] * v.arr
        return h

    def apply_region_mask(self, v):
        return v

    def multi_dot(self, w, basis):
        return np.stack([self.dot(w, v) for v in basis])

    def multi_axpy(self, w, basis, h, fill=1.0):
        for coef, v in zip(np.asarray(h), basis):
            w.arr -= coef[0] * v.arr
        return w


def shard_main(outdir):
    """tracers of one module on two ranks: all-reduced inner products, CGS-2 and the sharded GMRES loop"""
    import torch.distributed as dist

    from nk_ooc_amd import dist as nkdist

    rank, _, world = nkdist.init_process_group_from_env("gloo")
    assert world == 2
    nz, ny = 6, 5
    eng = FakeEngine(nz, ny, 1, seed=10 + rank)
    weight = np.outer(np.linspace(1.0, 2.0, nz), np.linspace(3.0, 1.0, ny))
    eng.set_region(np.ones((nz, ny), dtype=np.int32), weight)
    wn = eng.wn.reshape(nz, ny)
    rng = np.random.default_rng(123)                       # the same full vectors on both ranks
    full = rng.standard_normal((5, 2, nz, ny))

    def wdot(a, b):
        return float((wn[np.newaxis] * a * b).sum())

    comm = nkdist.ShardComm("cpu")
    vs = nkdist.ShardedVectorSpace(eng, comm)
    local = [eng.upload(v[rank:rank + 1]) for v in full]
    dot_err = abs(vs.dot(local[0], local[1])[0] - wdot(full[0], full[1])) / abs(wdot(full[0], full[1]))
    # orthonormal basis of three full vectors (redundantly on both ranks), then project a fourth
    basis_full = []
    for v in full[:3]:
        v = v.copy()
        for b in basis_full:
            v -= wdot(v, b) * b
        basis_full.append(v / np.sqrt(wdot(v, v)))
    w_full = full[3].copy()
    h_want = []
    for b in basis_full:
        h_want.append(wdot(w_full, b))
        w_full -= h_want[-1] * b
    basis = [eng.upload(b[rank:rank + 1]) for b in basis_full]
    w = eng.upload(full[3][rank:rank + 1])
    h_got = vs.cgs2(w, basis)
    h_err = float(np.max(np.abs(h_got[:, 0] - np.array(h_want))))
    ortho = float(np.max(np.abs(comm.allreduce(eng.multi_dot(w, basis)))))
    calls = comm.calls
    x, fx = local[4], eng.comp_fcn(local[4])[0]
    before = comm.calls
    _, info = nkdist.sharded_gmres(eng, comm, x, fx, 0.0, 0, 3)
    res = {"dot_err": dot_err, "h_err": h_err, "ortho": ortho, "allreduces": calls,
           "gmres_resid_drop": float(info["resid_norm"][-1][0] / info["beta"][0]),
           "gmres_allreduces_per_iter": (comm.calls - before - 1) / info["iters"]}
    with open(os.path.join(outdir, f"shard{rank}.json"), "w") as fptr:
        json.dump(res, fptr)
    dist.barrier()
    dist.destroy_process_group()


def columns_main(outdir):
    """SURVEY.md section 8(e) level 3: basis column i on rank i mod world; against the same solve on one rank"""
    import torch.distributed as dist

    from nk_ooc_amd import dist as nkdist

    rank, _, world = nkdist.init_process_group_from_env("gloo")
    nz, ny = 6, 5
    eng = FakeEngine(nz, ny, 2, seed=7)                      # the whole module, replicated
    weight = np.outer(np.linspace(1.0, 2.0, nz), np.linspace(3.0, 1.0, ny))
    eng.set_region(np.ones((nz, ny), dtype=np.int32), weight)
    rng = np.random.default_rng(99)
    x = eng.upload(rng.standard_normal(eng.shape))
    fx = eng.comp_fcn(x)[0]
    iters = 6
    comm = nkdist.ColumnComm(rank, world, "cpu")
    inc, info = nkdist.column_sharded_gmres(eng, comm, x, fx, 0.0, 0, iters)
    alone = nkdist.ColumnComm(0, 1, "cpu")
    inc1, info1 = nkdist.column_sharded_gmres(eng, alone, x, fx, 0.0, 0, iters)
    # ... and against the tracer-sharded loop's reference: the unsharded Krylov numbers of sequential MGS agree to rounding
    res = {"h_err": float(np.max(np.abs(info["h_mat"] - info1["h_mat"]))),
           "resid_err": float(np.max(np.abs(info["resid_norm"] - info1["resid_norm"]))),
           "inc_err": float(np.max(np.abs(inc.arr - inc1.arr)) / np.max(np.abs(inc1.arr))),
           "resid_drop": float(info["resid_norm"][-1][0] / info["beta"][0]),
           "columns_here": info["columns_here"], "iters": info["iters"],
           "small_allreduces_per_iter": info["allreduces"] / info["iters"],
           "vector_collectives_per_iter": (info["vector_collectives"] - 1) / info["iters"],
           "alone_collectives": info1["vector_collectives"] + info1["allreduces"]}
    with open(os.path.join(outdir, f"columns{rank}.json"), "w") as fptr:
        json.dump(res, fptr)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "shard":
        return shard_main(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "columns":
        return columns_main(sys.argv[1])
    import torch.distributed as dist

    from nk_ooc_amd import dist as nkdist
    from nk_ooc_amd import model_state
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    outdir = sys.argv[1]
    rank, _, world = nkdist.init_process_group_from_env("gloo")
    assert world == 2
    # global list of tracer modules, partitioned round-robin over the ranks
    all_modules = ["iage", "forced_a"]
    mine = nkdist.partition_modules(all_modules, world)[rank]
    assert mine == [all_modules[rank]]
    workdir = os.path.join(outdir, f"rank{rank}")
    cfg = make_config(workdir, 6, 5, tracer_module_names="iage" if rank == 0 else "forced_{suff}:a",
                      extra_solverinfo={"krylov_rel_tol": "1.0e-30"})
    gen_grid_vars_file(cfg["modelinfo"])
    model_state._module_engine = lambda name, module_def, grid, device_id, modelinfo: FakeEngine(
        6, 5, len(module_def["tracers"]), seed=rank)
    model_state.ModelState.reset_class()
    model_state.ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    iterate = model_state.ModelState("gen_init_iterate")
    fcn = iterate.comp_fcn(os.path.join(workdir, "fcn_00.nc"), None)

    # local stopping flags that differ between the ranks: rank 0 is satisfied after 1
    # iteration, rank 1 after 3 -> with the global AND both must run 3 iterations
    from nk_ooc_amd.krylov_solver import KrylovSolver

    want_local = 1 if rank == 0 else 3
    KrylovSolver.converged = lambda self, beta, resid: np.array([[self.get_iteration() >= want_local]])
    solverinfo = dict(cfg["solverinfo"])
    solverinfo["krylov_workdir"] = os.path.join(workdir, "krylov_00")
    solver = nkdist.DistributedKrylovSolver(iterate, solverinfo, False, False, None)
    inc = solver.solve(os.path.join(workdir, "increment_00.nc"), fcn)
    iters = solver.get_iteration()
    # the increment solves the local linear system better with more iterations
    eng = iterate.tracer_modules[0].eng
    jac = eng.A - np.eye(eng.A.shape[0])
    resid = np.linalg.norm(jac @ inc.tracer_modules[0].vec.arr.reshape(-1) + fcn.tracer_modules[0].vec.arr.reshape(-1))
    with open(os.path.join(outdir, f"result{rank}.json"), "w") as fptr:
        json.dump({"rank": rank, "iters": iters, "modules": mine, "resid": resid,
                   "fcn_norm": float(np.linalg.norm(fcn.tracer_modules[0].vec.arr))}, fptr)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
