"""worker of tests/test_gpu_dist.py: one rank of a world_size-2 run of the module-per-rank Krylov
loop with the real HIP engines (both ranks share the box's GPU; the 4-byte collectives of the
stopping test run on CPU tensors over gloo -- on a multi-GPU node the backend is nccl = RCCL)."""
import json
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODULES = ["iage", "phosphorus"]


def solve_one(workdir, module, solver_cls, **solver_kw):
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    cfg = make_config(workdir, 22, 9, tracer_module_names=module,
                      extra_solverinfo={"krylov_max_iter": "3", "krylov_rel_tol": "1.0e-30"})
    gen_grid_vars_file(cfg["modelinfo"])
    ModelState.reset_class()
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    ModelState.write_files = True
    iterate = ModelState("gen_init_iterate")
    hist = os.path.join(workdir, "hist_00.nc")
    fcn = iterate.comp_fcn(os.path.join(workdir, "fcn_00.nc"), None, hist)
    solverinfo = dict(cfg["solverinfo"], krylov_workdir=os.path.join(workdir, "krylov_00"))
    solver = solver_cls(iterate, solverinfo, False, False, hist, **solver_kw)
    inc = solver.solve(os.path.join(workdir, "increment_00.nc"), fcn)
    state = solver._solver_state
    res = {"module": module, "iters": solver.get_iteration(),
           "beta": state.get_value_saved_state("beta").tolist(),
           "h_mat": state.get_value_saved_state("h_mat").tolist(),
           "inc_norm": inc.norm().tolist()}
    ModelState.reset_class()
    return res


def main():
    import torch.distributed as dist

    from nk_ooc_amd import dist as nkdist

    outdir = sys.argv[1]
    rank, _, world = nkdist.init_process_group_from_env("gloo")
    mine = nkdist.partition_modules(MODULES, world)[rank]
    assert len(mine) == 1
    res = solve_one(os.path.join(outdir, f"rank{rank}"), mine[0], nkdist.DistributedKrylovSolver)
    with open(os.path.join(outdir, f"result{rank}.json"), "w") as fptr:
        json.dump(res, fptr)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
