#!/usr/bin/env python
"""G7 / G8 of SURVEY.md section 8(c): Krylov and Newton traces made by the reference's OWN solver classes.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/gen_ref_traces.py [krylov|newton|all] [n]      (n x n grid: 26 (default), 30 = the grid of
                                                                        scripts/ci_py_driver_2d_iage.sh:13-14, 52)

It runs, in this process and with the I/O stand-ins of tests/ref_harness/shims.py (netCDF4 / xarray / pint are
absent from the image; none of them is on the arithmetic path), exactly what the reference's CI scripts run:
`python -m nk_ooc.py_driver_2d.setup_solver --fp_cnt 1 ...` followed by `python -m nk_ooc.nk_driver
--persist ...` (scripts/ci_py_driver_2d_iage_column_regions.sh:24-48), i.e. nk_ooc/nk_driver.py:38-67 driving
nk_ooc/newton_solver.py:140-334 and nk_ooc/krylov_solver.py:85-165, on iage at 26 x 26, and collects from the
work directory what the solvers checkpoint: every vector file of the Krylov directories, `beta` / `h_mat` of
Krylov_state.json, the preconditioned residual norms of Krylov_stats.nc, the Newton iterates and the two step
logs (with the work directory spelled $workdir).

  krylov : krylov_rel_tol = 2e-4, newton_max_iter = 1 -> one Krylov solve with several iterations
           (the committed baselines of the reference hold iteration 0 only)  -> krylov_trace_26x26.npz
  newton : the reference's default tolerances, Newton run to convergence       -> newton_trace_26x26.npz

Fixtures are data (inputs + the reference's outputs); no reference source text is stored.

Run with OMP_NUM_THREADS=1 (the 30 x 30 and 52 x 52 fixtures were): SciPy's Radau is not bit-reproducible across BLAS thread
counts, and tests/test_ref_traces.py replays the traces bit for bit under one BLAS thread.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _read_vec(fname, names=("iage", "iage_slow_rest")):
    from scipy.io import netcdf_file

    with netcdf_file(fname, "r", mmap=False) as fptr:
        return np.stack([np.array(fptr.variables[name].data, dtype=np.float64) for name in names])


def _state(fname, workdir):
    text = open(fname).read().replace(workdir, "$workdir")
    return json.loads(text)


def _nd(obj):
    return np.asarray(obj["__ndarray__"], dtype=np.float64)


def run_reference(workdir, n, solverinfo_over):
    """setup_solver + nk_driver of the reference, in-process"""
    from ref_harness import shims

    shims.install()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from nk_ooc import nk_driver
    from nk_ooc.py_driver_2d import setup_solver

    os.environ.setdefault("USER", "nk2d")          # the reference's cfg reader interpolates HOME and USER
    os.environ.setdefault("HOME", os.path.expanduser("~"))
    os.makedirs(workdir, exist_ok=True)
    override = os.path.join(workdir, "override.cfg")
    with open(override, "w") as fptr:
        fptr.write(f"[modelinfo]\ndepth_nlevs = {n}\nypos_nlevs = {n}\n\n[solverinfo]\n")
        for key, val in solverinfo_over.items():
            fptr.write(f"{key} = {val}\n")
    input_dir = os.path.join(REF, "input", "py_driver_2d")
    cfgs = ",".join([os.path.join(input_dir, "newton_krylov.cfg"), os.path.join(input_dir, "model_params.cfg"), override])
    common = ["--model_name", "py_driver_2d", "--tracer_module_names", "iage", "--persist",
              "--cfg_fnames", cfgs, "--workdir", workdir]
    setup_solver.main(setup_solver.parse_args(["--fp_cnt", "1"] + common))
    try:
        nk_driver.main(nk_driver.parse_args(common))
        ended = "converged"
    except RuntimeError as msg:            # newton_max_iter reached (newton_solver.py raises)
        ended = f"RuntimeError: {msg}"
    return ended


def collect_krylov(kdir, workdir):
    state = _state(os.path.join(kdir, "Krylov_state.json"), workdir)
    iters = state["iteration"]
    out = {"iterations": iters, "beta": _nd(state["beta"]), "h_mat": _nd(state["h_mat"]),
           "precond_fcn": _read_vec(os.path.join(kdir, "precond_fcn_00.nc"))}
    for quantity in ("basis", "w_raw", "w", "perturb_fcn_w_raw", "krylov_res"):
        vecs = []
        for j in range(iters + 1):
            fname = os.path.join(kdir, f"{quantity}_{j:02}.nc")
            if os.path.exists(fname):
                vecs.append(_read_vec(fname))
        out[quantity] = np.stack(vecs)
    from scipy.io import netcdf_file

    with netcdf_file(os.path.join(kdir, "Krylov_stats.nc"), "r", mmap=False) as fptr:
        out["precond_resid_norm"] = np.array(fptr.variables["precond_resid_norm_iage"].data, dtype=np.float64)
        out["precond_rhs_norm"] = np.array(fptr.variables["precond_rhs_norm_iage"].data, dtype=np.float64)
    return out, state["step_log"]


def gen(kind, n=26):
    workdir = tempfile.mkdtemp(prefix=f"ref_{kind}_")
    over = {"krylov": {"krylov_rel_tol": "2.0e-4", "newton_max_iter": "1"}, "newton": {}}[kind]
    ended = run_reference(workdir, n, over)
    newton = _state(os.path.join(workdir, "Newton_state.json"), workdir)
    out = {"n": n, "ended": ended, "newton_iterations": newton["iteration"],
           "newton_step_log": json.dumps(newton["step_log"]),
           "init_iterate": _read_vec(os.path.join(workdir, "gen_init_iterate", "init_iterate.nc"))
           if os.path.exists(os.path.join(workdir, "gen_init_iterate", "init_iterate.nc"))
           else _read_vec(os.path.join(workdir, "init_iterate.nc"))}
    iterates, fcns, incs = [], [], []
    for it in range(newton["iteration"] + 1):
        for lst, quantity in ((iterates, "iterate"), (fcns, "fcn"), (incs, "increment")):
            fname = os.path.join(workdir, f"{quantity}_{it:02}.nc")
            if os.path.exists(fname):
                lst.append(_read_vec(fname))
    out["iterate"], out["fcn"] = np.stack(iterates), np.stack(fcns)
    if incs:
        out["increment"] = np.stack(incs)
    nk = 0
    while os.path.isdir(os.path.join(workdir, f"krylov_{nk:02}")):
        tr, log = collect_krylov(os.path.join(workdir, f"krylov_{nk:02}"), workdir)
        for key, val in tr.items():
            out[f"k{nk}_{key}"] = val
        out[f"k{nk}_step_log"] = json.dumps(log)
        nk += 1
    out["krylov_solves"] = nk
    np.savez_compressed(os.path.join(HERE, f"{kind}_trace_{n}x{n}.npz"), **out)
    print(f"wrote {kind}_trace_{n}x{n}.npz: ended={ended}, newton iterations {newton['iteration']}, "
          f"krylov iterations {[int(out[f'k{i}_iterations']) for i in range(nk)]}")
    shutil.rmtree(workdir, ignore_errors=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 26
    for kind in (("krylov", "newton") if what == "all" else (what,)):
        gen(kind, size)
