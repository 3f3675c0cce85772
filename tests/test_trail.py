"""The checkpoint trail's writer thread (newton-krylov_ooc_amd/trail.py): program order on disk, a prefix after a failure,
the same bytes as the synchronous trail.  CPU only.

What it stands in for: the reference writes every file inside the call that asks for it
(/root/reference/nk_ooc/model_state_base.py:93-111, /root/reference/nk_ooc/solver_state.py:60-72,
/root/reference/nk_ooc/stats_file.py:100-140)."""

import os
import threading
import time

import numpy as np
import pytest

from nk_ooc_amd import ncio, trail
from nk_ooc_amd.grid import SpatialAxis
from nk_ooc_amd.solver_state import SolverState
from nk_ooc_amd.stats_file import StatsFile


@pytest.fixture
def async_trail():
    was = trail.set_enabled(True)
    yield trail.TRAIL
    try:
        trail.set_enabled(was)
    except Exception:       # noqa: BLE001 -- a test that queued a failing job has already seen it
        pass


def test_jobs_run_in_order_on_one_other_thread(async_trail):
    seen, threads = [], set()

    def job(i):
        def run():
            if i == 0:
                time.sleep(0.05)        # the first job is slow: the later ones still come behind it
            seen.append(i)
            threads.add(threading.get_ident())
        return run

    for i in range(20):
        trail.submit(job(i))
    trail.flush()
    assert seen == list(range(20))
    assert threads and threading.get_ident() not in threads and len(threads) == 1
    assert async_trail.pending() == 0


def test_a_failed_job_leaves_a_prefix_and_is_raised_at_the_next_call(async_trail, tmp_path):
    def write(name):
        return lambda: (tmp_path / name).write_text(name)

    def fail():
        raise OSError("disk full")

    trail.submit(write("a"))
    trail.submit(fail)
    trail.submit(write("b"))        # queued behind the failure: never written
    with pytest.raises(OSError, match="disk full"):
        trail.flush()
    assert (tmp_path / "a").exists() and not (tmp_path / "b").exists()
    trail.submit(write("c"))        # the error was delivered once; the trail goes on
    trail.flush()
    assert (tmp_path / "c").exists()


def test_disabled_trail_writes_inside_the_call(tmp_path):
    was = trail.set_enabled(False)
    try:
        trail.submit(lambda: (tmp_path / "x").write_text("x"))
        assert (tmp_path / "x").exists()
        with pytest.raises(ValueError):
            trail.submit(lambda: (_ for _ in ()).throw(ValueError("now")))
    finally:
        trail.set_enabled(was)


def _solver_state_session(workdir, name):
    st = SolverState(name, str(workdir))
    st.set_value_saved_state("beta", np.array([[1.5, 2.5]]))
    for it in range(3):
        st.log_step("comp_fcn")
        st.set_value_saved_state("h_mat", np.arange(2.0 * (it + 2) * (it + 1)).reshape(1, it + 2, it + 1, 2))
        st.inc_iteration()
    return st


def test_solver_state_file_is_byte_identical_and_resumes(async_trail, tmp_path):
    st = _solver_state_session(tmp_path, "Async")
    # what the solver goes on with is what a resumed run reads -- before the file is even on disk
    assert isinstance(st.get_value_saved_state("h_mat"), np.ndarray)
    trail.flush()
    trail.set_enabled(False)
    _solver_state_session(tmp_path, "Sync")
    a = (tmp_path / "Async_state.json").read_text().replace("Async", "X")
    b = (tmp_path / "Sync_state.json").read_text().replace("Sync", "X")
    assert a == b
    trail.set_enabled(True)
    st2 = _solver_state_session(tmp_path, "Again")
    resumed = SolverState("Again", str(tmp_path), resume=True)      # (_load flushes the queue first)
    assert resumed.get_iteration() == st2.get_iteration() == 3
    assert np.array_equal(resumed.get_value_saved_state("h_mat"), st2.get_value_saved_state("h_mat"))


def _stats_session(workdir, name):
    st = SolverState(name, str(workdir))
    sf = StatsFile(name, str(workdir), 2, st)
    sf.def_dimensions({"depth": 3})
    sf.def_vars({"resid": {"dimensions": ("iteration", "region"), "attrs": {"long_name": "r", "units": "1"}},
                 "col": {"dimensions": ("depth",), "attrs": {"long_name": "c"}}})
    vals = np.array([1.0, 2.0])
    sf.put_vars_iteration_invariant({"col": np.array([7.0, 8.0, 9.0])})
    for it in range(3):
        sf.put_vars(it, {"resid": vals})
        vals *= 0.5             # the caller goes on with its array: the queued call kept a copy
    return os.path.join(str(workdir), f"{name}_stats.nc")


def test_stats_file_async_equals_sync(async_trail, tmp_path):
    fa = _stats_session(tmp_path, "A")
    data_a, _ = ncio.read_file(fa)          # (read_file flushes)
    trail.set_enabled(False)
    fb = _stats_session(tmp_path, "B")
    data_b, _ = ncio.read_file(fb)
    assert set(data_a) == set(data_b)
    for key in data_a:
        assert np.array_equal(data_a[key], data_b[key]), key
    assert np.array_equal(data_a["resid"], [[1.0, 2.0], [0.5, 1.0], [0.25, 0.5]])


def test_vector_files_then_step_log_order_on_disk(async_trail, tmp_path):
    """a step is never logged on disk before the file it stands for: observed from the writer's own order"""
    order = []
    depth = SpatialAxis("depth", np.linspace(0.0, 4.0, 5), "m")
    ypos = SpatialAxis("ypos", np.linspace(0.0, 3.0, 4), "m")
    st = SolverState("Order", str(tmp_path))
    real_write = ncio.write_state_file

    def spy_write(fname, *args, **kwargs):
        real_write(fname, *args, **kwargs)
        order.append(os.path.basename(fname))

    for it in range(4):
        fname = str(tmp_path / f"vec_{it}.nc")
        vals = {"t": np.full((4, 3), float(it))}
        trail.submit(lambda f=fname, v=vals: spy_write(f, [depth, ypos], v, "h"))
        st.log_step(f"wrote {it}", per_iteration=False)
        trail.submit(lambda i=it: order.append(f"log {i}"))
    trail.flush()
    assert order == [x for it in range(4) for x in (f"vec_{it}.nc", f"log {it}")]
    on_disk = SolverState("Order", str(tmp_path), resume=True)
    assert all(on_disk.step_logged(f"wrote {it}", per_iteration=False) for it in range(4))
    data, _ = ncio.read_file(str(tmp_path / "vec_3.nc"), ["t"])
    assert np.array_equal(data["t"], np.full((4, 3), 3.0))
