"""worker of tests/test_gpu_trail.py::test_a_killed_solve_leaves_a_resumable_prefix: one Krylov solve of a 20 x 5 iage set-up with
the checkpoint trail on its writer thread.
    python trail_kill_worker.py WORKDIR full|victim|resume [KILL_AFTER_FILE]
victim: the process ends abruptly (os._exit, no flush, no atexit) at the first submit behind KILL_AFTER_FILE -- the job it was
about to queue and everything the solve would have written after it are lost.  (From the main thread, between two library
calls: no resident kernel is on the GPU at that moment; the writer's queue is drained first so that every run dies at the
same point of the trail.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd import trail  # noqa: E402
from nk_ooc_amd.krylov_solver import KrylovSolver  # noqa: E402
from nk_ooc_amd.model_config import ModelConfig  # noqa: E402
from nk_ooc_amd.model_state import ModelState  # noqa: E402
from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config  # noqa: E402

workdir, mode = sys.argv[1], sys.argv[2]
cfg = make_config(workdir, 20, 5, extra_solverinfo={"krylov_rel_tol": "0.0", "krylov_max_iter": "3"})
if mode != "resume":
    gen_grid_vars_file(cfg["modelinfo"])
ModelState.reset_class()
ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
ModelState.write_files = True
trail.set_enabled(True)
if mode == "victim":
    kill_after = sys.argv[3]
    serve_one = trail.TRAIL.submit

    def submit_and_watch(job):
        # the same point of the trail in every run: what is queued reaches the disk first, then the first submit behind the
        # chosen file (or step) is where the process dies -- with the job it was about to queue, and everything after it, lost
        trail.TRAIL.drain()
        if kill_after.endswith("@iteration1"):
            # the reference's own window (krylov_solver.py:167-181): the step log says iteration 1, the Arnoldi vector of that
            # iteration is not written yet (the first submit behind the step log's `inc_iteration` is the dump of basis_01)
            import json

            state_fname = os.path.join(os.path.dirname(kill_after), "Krylov_state.json")
            if os.path.exists(state_fname) and json.load(open(state_fname))["iteration"] >= 1:
                os._exit(9)
        elif os.path.exists(kill_after):
            os._exit(9)
        serve_one(job)

    trail.TRAIL.submit = submit_and_watch
    trail.submit = submit_and_watch
    import nk_ooc_amd.model_state as ms_mod
    import nk_ooc_amd.solver_state as ss_mod
    import nk_ooc_amd.stats_file as sf_mod

    for mod in (ms_mod, ss_mod, sf_mod):
        mod.trail.submit = submit_and_watch
iterate_fname = os.path.join(workdir, "iterate_00.nc")
fcn_fname = os.path.join(workdir, "fcn_00.nc")
if mode == "resume":
    iterate = ModelState(iterate_fname)
    fcn = ModelState(fcn_fname)         # (with the schedule side file the first process left: products on frozen years again)
else:
    iterate = ModelState("gen_init_iterate").dump(iterate_fname, "worker")
    fcn = iterate.comp_fcn(fcn_fname, None)
    trail.flush()
info = dict(cfg["solverinfo"], Krylov_workdir=os.path.join(workdir, "krylov_00"))
solver = KrylovSolver(iterate, info, resume=(mode == "resume"), rewind=False, hist_fname=None)
inc = solver.solve(os.path.join(workdir, "increment_00.nc"), fcn)
np.save(os.path.join(workdir, f"result_{mode}.npy"), inc.tracer_modules[0].get_tracer_vals_all())
print(f"{mode}: iteration {solver.get_iteration()}, jvp mode {ModelState.last_jvp_mode}", flush=True)
ModelState.reset_class()
