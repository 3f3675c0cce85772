"""The checkpoint trail on its writer thread (newton-krylov_ooc_amd/trail.py) under a real Krylov solve: the same files with
the same values as the synchronous trail (/root/reference/nk_ooc/krylov_solver.py:85-181 writes each of them inside the
call), complete on disk when `solve` returns; and the two-half download it rests on (nk2d_vec_download_begin / _end)."""
import json
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup_run(tmp_path, nz, ny, extra_solverinfo=None):
    from nk_ooc_amd.model_config import ModelConfig
    from nk_ooc_amd.model_state import ModelState
    from nk_ooc_amd.setup_solver import gen_grid_vars_file, make_config

    cfg = make_config(str(tmp_path), nz, ny, extra_solverinfo=extra_solverinfo)
    gen_grid_vars_file(cfg["modelinfo"])
    ModelState.reset_class()
    ModelState.model_config_obj = ModelConfig(cfg["modelinfo"])
    return cfg, ModelState


def test_download_in_two_halves_equals_download():
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    eng = iage_engine(Grid2d.default(70, 9, 0.1, 1000.0))
    rng = np.random.default_rng(5)
    hosts = [rng.standard_normal((2, 70, 9)) for _ in range(5)]
    vecs = [eng.upload(h) for h in hosts]
    pending = [eng.download_begin(v) for v in vecs]
    # what is queued on the stream afterwards may change the vector: the copy has read it by then, in stream order
    eng.upload(np.zeros((2, 70, 9)), out=vecs[0])
    got = [None] * 5

    def take():
        for ind in (3, 0, 4, 1):            # any thread, any order
            got[ind] = pending[ind].result()

    thread = threading.Thread(target=take)
    thread.start()
    thread.join()
    for ind in (3, 0, 4, 1):
        assert np.array_equal(got[ind], hosts[ind]), ind
    with pytest.raises(RuntimeError):
        pending[0].result()                 # taken once
    del pending                             # the one that was never taken is released with its ticket
    again = eng.download_begin(vecs[2])     # ... and its staging pair serves the next download
    assert np.array_equal(again.result(), hosts[2])
    assert np.array_equal(eng.download(vecs[0]), np.zeros((2, 70, 9)))
    eng.close()


def _solve(tmp_path, tag, asynchronous):
    from nk_ooc_amd import trail
    from nk_ooc_amd.krylov_solver import KrylovSolver

    root = os.path.join(str(tmp_path), tag)
    os.makedirs(root)
    cfg, ModelState = _setup_run(root, 26, 26, extra_solverinfo={"krylov_rel_tol": "0.0", "krylov_max_iter": "3"})
    ModelState.write_files = True
    was = trail.set_enabled(asynchronous)
    try:
        jobs0 = trail.TRAIL.jobs_run
        iterate = ModelState("gen_init_iterate")
        fcn = iterate.comp_fcn(os.path.join(root, "fcn_00.nc"), None)
        info = dict(cfg["solverinfo"], Krylov_workdir=os.path.join(root, "krylov_00"))
        solver = KrylovSolver(iterate, info, resume=False, rewind=False, hist_fname=None)
        inc = solver.solve(os.path.join(root, "increment_00.nc"), fcn)
        # on disk when solve returns: nothing of the trail is still queued
        assert trail.TRAIL.pending() == 0
        jobs = trail.TRAIL.jobs_run - jobs0
        values = inc.tracer_modules[0].get_tracer_vals_all()
    finally:
        trail.set_enabled(was)
        ModelState.reset_class()
    return root, values, jobs


def test_krylov_solve_async_trail_equals_sync_trail(tmp_path):
    from nk_ooc_amd import ncio

    root_s, inc_s, jobs_s = _solve(tmp_path, "sync", False)
    root_a, inc_a, jobs_a = _solve(tmp_path, "async", True)
    assert jobs_s == 0 and jobs_a > 20          # (the writer thread did the writing in the second run only)
    assert np.array_equal(inc_s, inc_a)
    files_s = sorted(os.listdir(os.path.join(root_s, "krylov_00")))
    files_a = sorted(os.listdir(os.path.join(root_a, "krylov_00")))
    assert files_s == files_a
    assert {"basis_00.nc", "basis_02.nc", "w_raw_02.nc", "w_02.nc", "perturb_fcn_w_raw_02.nc", "krylov_res_02.nc",
            "precond_fcn_00.nc", "Krylov_state.json", "Krylov_stats.nc"} <= set(files_a)
    for name in files_a + ["../increment_00.nc", "../fcn_00.nc"]:
        path_s, path_a = os.path.join(root_s, "krylov_00", name), os.path.join(root_a, "krylov_00", name)
        if name.endswith(".nc"):
            data_s, _ = ncio.read_file(path_s)
            data_a, _ = ncio.read_file(path_a)
            assert list(data_s) == list(data_a), name
            for key in data_s:
                assert np.array_equal(data_s[key], data_a[key]), (name, key)
        elif name.endswith(".json"):
            text_s = open(path_s).read().replace(root_s, "ROOT")
            text_a = open(path_a).read().replace(root_a, "ROOT")
            assert text_s == text_a, name
    state = json.load(open(os.path.join(root_a, "krylov_00", "Krylov_state.json")))
    assert state["iteration"] == 3
    # the schedule side file of the forward year travelled on the trail too
    assert os.path.exists(os.path.join(root_a, ".nk2d", "fcn_00.nc.sched.npz"))


def test_resume_reads_behind_the_writer_thread(tmp_path):
    """a file of the process' own trail that is opened by name is complete: every reader flushes the queue first"""
    from nk_ooc_amd import trail

    cfg, ModelState = _setup_run(tmp_path, 20, 5)
    ModelState.write_files = True
    was = trail.set_enabled(True)
    try:
        state = ModelState("gen_init_iterate")
        want = state.tracer_modules[0].get_tracer_vals_all()
        names = [os.path.join(str(tmp_path), f"v_{ind}.nc") for ind in range(12)]
        for ind, fname in enumerate(names):
            (state * float(ind + 1)).dump(fname, "test")
        ModelState._resident.clear()        # a new process has nothing in HBM: the files are read
        back = ModelState(names[-1]).tracer_modules[0].get_tracer_vals_all()
        assert np.array_equal(back, want * 12.0)
        assert trail.TRAIL.pending() == 0
    finally:
        trail.set_enabled(was)
        ModelState.reset_class()


@pytest.mark.parametrize("kill_after", ["w_raw_01.nc", "perturb_fcn_w_raw_02.nc", "krylov_res_00.nc", "krylov_res_01.nc", "w_02.nc",
                                        "@iteration1"])
def test_a_killed_solve_leaves_a_resumable_prefix(tmp_path, kill_after):
    """the claim trail.py makes: a run that dies finds a PREFIX of the synchronous trail on disk -- every step the step log
    names has its file, complete --, and `--resume` goes on from it to the result of the uninterrupted solve, bit for bit.
    A solve in a process of its own is ended abruptly (os._exit at the first submit behind a given file of the trail: between
    a file and the step that names it, between the Hessenberg matrix and `inc_iteration`, ... -- or, "@iteration1", right
    behind the step log's `inc_iteration`, before the next Arnoldi vector is written: the reference's own window); a third
    process resumes (out-of-core contract of
    /root/reference/nk_ooc/solver_state.py:13-157 and krylov_solver.py:85-181)"""
    import subprocess
    import sys

    from nk_ooc_amd import ncio

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trail_kill_worker.py")
    full_dir, part_dir = str(tmp_path / "full"), str(tmp_path / "part")
    os.makedirs(full_dir)
    os.makedirs(part_dir)
    done = subprocess.run([sys.executable, worker, full_dir, "full"], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    victim = subprocess.run([sys.executable, worker, part_dir, "victim", os.path.join(part_dir, "krylov_00", kill_after)],
                            capture_output=True, text=True, timeout=300)
    assert victim.returncode == 9, (victim.returncode, victim.stderr[-2000:])
    assert not os.path.exists(os.path.join(part_dir, "result_victim.npy"))
    # what is on disk: a step log that parses, and behind every step it names the file the step stands for
    kdir = os.path.join(part_dir, "krylov_00")
    state = json.load(open(os.path.join(kdir, "Krylov_state.json")))
    assert state["iteration"] < 3
    named = [entry.split(" complete for ")[1] for entry in state["step_log"] if " complete for " in entry]
    assert named, state["step_log"]
    for fname in named:
        assert os.path.exists(fname), fname
        data, _ = ncio.read_file(fname)
        assert all(np.all(np.isfinite(val)) for val in data.values()), fname
    if kill_after == "@iteration1":
        # the reference's own window: the step log is at iteration 1, the Arnoldi vector of that iteration was never written
        # (the resumed solver rebuilds it from w_00 and the saved Hessenberg matrix, KrylovSolver._rebuild_basis)
        assert state["iteration"] == 1 and not os.path.exists(os.path.join(kdir, "basis_01.nc"))
    else:
        assert os.path.exists(os.path.join(kdir, kill_after))
    resumed = subprocess.run([sys.executable, worker, part_dir, "resume"], capture_output=True, text=True, timeout=300)
    assert resumed.returncode == 0, resumed.stderr[-3000:]
    # (the products the resumed run still had to form ran on frozen years again -- the schedule side file of F(x) travelled on
    # the trail --; killed behind the last product, it forms none)
    assert "iteration 3" in resumed.stdout and "free_running" not in resumed.stdout, resumed.stdout
    assert "jvp mode frozen" in resumed.stdout or kill_after == "w_02.nc", resumed.stdout
    assert np.array_equal(np.load(os.path.join(part_dir, "result_resume.npy")), np.load(os.path.join(full_dir, "result_full.npy")))
    h_full = json.load(open(os.path.join(full_dir, "krylov_00", "Krylov_state.json")))["h_mat"]
    h_part = json.load(open(os.path.join(kdir, "Krylov_state.json")))["h_mat"]
    assert h_full == h_part
