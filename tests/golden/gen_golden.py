#!/usr/bin/env python
"""Generate golden vectors by importing the genuine reference numerics.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/gen_golden.py

It imports `nk_ooc.py_driver_2d.{advection,horiz_mix,vert_mix,iage}`,
`nk_ooc.spatial_axis` and `nk_ooc.krylov_solver` from /root/reference.  Three
third-party modules the reference imports for FILE I/O only (netCDF4, xarray,
pint) are absent from this image; inert placeholder modules are registered for
them so that the `import` statements succeed -- none of them is touched by the
arithmetic exercised here.  Outputs are small `.npz` fixtures (inputs + expected
outputs) written next to this script; nothing of the reference's source text is
stored.
"""

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_placeholders():
    nc = types.ModuleType("netCDF4")

    class Dataset:  # never instantiated by the code paths used below
        def __init__(self, *a, **k):
            raise RuntimeError("netCDF4 placeholder: file I/O is not available")

    nc.Dataset = Dataset
    nc.default_fillvals = {"f8": 9.969209968386869e36, "i4": -2147483647}
    sys.modules["netCDF4"] = nc
    xr = types.ModuleType("xarray")
    xr.Dataset = dict
    xr.DataArray = object
    sys.modules["xarray"] = xr
    pint = types.ModuleType("pint")

    class UnitRegistry:
        def __init__(self, *a, **k):
            pass

    pint.UnitRegistry = UnitRegistry
    sys.modules["pint"] = pint


def ref_axes(nz, ny):
    from nk_ooc.spatial_axis import spatial_axis_defn_dict, spatial_axis_from_defn_dict

    depth = spatial_axis_from_defn_dict(spatial_axis_defn_dict(
        axisname="depth", units="m", nlevs=nz, edge_start=0.0, edge_end=4000.0,
        delta_ratio_max=19.0))
    ypos = spatial_axis_from_defn_dict(spatial_axis_defn_dict(
        axisname="ypos", units="m", nlevs=ny, edge_start=0.0, edge_end=50.0e5,
        delta_ratio_max=1.0))
    return depth, ypos


def ref_setup(nz, ny, max_abs_vvel, horiz_mix_coeff):
    """instantiate the reference processes and an iage tracer module on a grid"""
    from nk_ooc.py_driver_2d.advection import Advection
    from nk_ooc.py_driver_2d.horiz_mix import HorizMix
    from nk_ooc.py_driver_2d.iage import iage
    from nk_ooc.py_driver_2d.vert_mix import VertMix

    depth, ypos = ref_axes(nz, ny)
    modelinfo = {"max_abs_vvel": repr(max_abs_vvel),
                 "horiz_mix_coeff": repr(horiz_mix_coeff)}
    processes = {}
    processes["advection"] = Advection(depth, ypos, modelinfo)
    processes["horiz_mix"] = HorizMix(depth, ypos, modelinfo)
    processes["vert_mix"] = VertMix(depth, ypos)
    # the tracer-module constructor reads files; build the instance without it and
    # set exactly the attributes its arithmetic uses (iage.py:12-20)
    tm = object.__new__(iage)
    tm.name = "iage"
    tm.tracer_cnt = 2
    tm.depth = depth
    tm.ypos = ypos
    tm.surf_restore_rate = 24.0 / 86400.0 * 10.0 / depth.delta[0]
    tm.surf_slow_factor = 0.01
    return depth, ypos, processes, tm


def gen_static(tag, nz, ny, vv, kh, times, seed):
    from nk_ooc.py_driver_2d.advection import Advection

    depth, ypos, processes, tm = ref_setup(nz, ny, vv, kh)
    out = {"nz": nz, "ny": ny, "max_abs_vvel": vv, "horiz_mix_coeff": kh,
           "times": np.asarray(times)}
    for ax in (depth, ypos):
        for nm in ("edges", "mid", "delta", "delta_r", "delta_mid", "delta_mid_r"):
            out[f"{ax.axisname}_{nm}"] = getattr(ax, nm).copy()
    out["stream"] = Advection.stream.copy()
    out["vvel"] = Advection.vvel.copy()
    out["wvel"] = Advection.wvel.copy()
    out["hmix_coeff"] = processes["horiz_mix"]._mixing_coeff.copy()
    vm = processes["vert_mix"]
    out["bldepth"] = np.stack([vm.bldepth(t) for t in times])
    out["vmix_coeff"] = np.stack([vm.mixing_coeff(t).copy() for t in times])
    rng = np.random.default_rng(seed)
    y = rng.standard_normal(2 * nz * ny)
    out["y"] = y
    out["tend"] = np.stack([tm.comp_tend(t, y, processes).copy() for t in times])
    jacs = [tm.comp_jacobian(t, y, processes).tocsr() for t in times]
    for i, jac in enumerate(jacs):
        jac.sum_duplicates()
        jac.sort_indices()
        out[f"jac{i}_data"] = jac.data
        out[f"jac{i}_indices"] = jac.indices
        out[f"jac{i}_indptr"] = jac.indptr
    np.savez_compressed(os.path.join(HERE, f"static_{tag}.npz"), **out)
    print("wrote static", tag)
    return depth, ypos, processes, tm


def gen_comp_fcn(tag, nz, ny, vv, kh, init="profile"):
    """one forward year exactly as py_driver_2d/model_state.py:102-114 drives it"""
    from scipy import integrate

    depth, ypos, processes, tm = ref_setup(nz, ny, vv, kh)
    col = np.interp(depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = np.stack([np.broadcast_to(col[:, None], (nz, ny))] * 2).reshape(-1).copy()
    if init == "bumpy":
        rng = np.random.default_rng(7)
        y0 = y0 + 0.05 * rng.standard_normal(y0.size)
    time_range = (0.0, 365.0 * 86400.0)
    sol = integrate.solve_ivp(
        tm.comp_tend, time_range, y0, "Radau", np.array(time_range),
        max_step=(time_range[1] - time_range[0]) * 0.01, atol=1.0e-6, rtol=1.0e-6,
        args=(processes,), jac=tm.comp_jacobian,
        jac_sparsity=tm.comp_jacobian_sparsity(time_range[0], y0, processes))
    np.savez_compressed(
        os.path.join(HERE, f"comp_fcn_{tag}.npz"), nz=nz, ny=ny, max_abs_vvel=vv,
        horiz_mix_coeff=kh, y0=y0, yT=sol.y[:, -1].copy(),
        fcn=(sol.y[:, -1] - y0), nfev=sol.nfev, njev=sol.njev, nlu=sol.nlu)
    print("wrote comp_fcn", tag, sol.nfev, sol.njev, sol.nlu)


def gen_precond(tag, nz, ny, vv, kh, seed):
    """iage.apply_precond_jacobian arithmetic (iage.py:78-93) on a seeded vector"""
    from scipy import sparse
    from scipy.sparse import linalg as sp_linalg

    depth, ypos, processes, tm = ref_setup(nz, ny, vv, kh)
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(2 * nz * ny)
    time_range = (0.0, 365.0 * 86400.0)
    # same statements as the method body, with the file-backed containers replaced
    # by the flat vector (the method itself needs an xarray-backed instance)
    time_n = 3
    time_delta = (time_range[1] - time_range[0]) / time_n
    mat_id = sparse.identity(v.size)
    mat = sparse.identity(v.size)
    for time_ind in range(time_n):
        time = time_range[0] + (time_ind + 0.5) * time_delta
        mat_tmp = time_delta * tm.comp_jacobian(time, v, processes)
        mat *= mat_id - mat_tmp
    mat = mat_id - mat
    res = sp_linalg.spsolve(mat, v)
    np.savez_compressed(os.path.join(HERE, f"precond_{tag}.npz"), nz=nz, ny=ny,
                        max_abs_vvel=vv, horiz_mix_coeff=kh, v=v, res=res - v)
    print("wrote precond", tag)


def ref_forced(nz, ny, params):
    """reference `forced` tracer module (one tracer) with the given option dict"""
    from nk_ooc.py_driver_2d.advection import Advection
    from nk_ooc.py_driver_2d.forced import forced
    from nk_ooc.py_driver_2d.horiz_mix import HorizMix
    from nk_ooc.py_driver_2d.vert_mix import VertMix

    depth, ypos = ref_axes(nz, ny)
    modelinfo = {"max_abs_vvel": "0.1", "horiz_mix_coeff": "1000.0"}
    processes = {"advection": Advection(depth, ypos, modelinfo),
                 "horiz_mix": HorizMix(depth, ypos, modelinfo),
                 "vert_mix": VertMix(depth, ypos)}
    tm = object.__new__(forced)
    tm.name = "forced_x"
    tm.tracer_cnt = 1
    tm.depth = depth
    tm.ypos = ypos
    full = dict(params)
    if full["surf_restore_opt"] != "none":
        full["surf_restore_rate"] = 10.0 / depth.delta[0] * (24.0 / 86400.0)
    forced.params = full
    return depth, ypos, processes, tm


def gen_forced(tag, nz, ny, params, seed, with_fcn=False):
    from scipy import integrate

    depth, ypos, processes, tm = ref_forced(nz, ny, params)
    year = 365.0 * 86400.0
    times = [0.0, 0.3 * year, 0.66 * year]
    rng = np.random.default_rng(seed)
    y = rng.standard_normal(nz * ny)
    out = {"nz": nz, "ny": ny, "times": np.asarray(times), "y": y,
           "surf_restore_opt": params["surf_restore_opt"], "sms_opt": params["sms_opt"],
           "surf_restore_const": params.get("surf_restore_const", 0.0),
           "sms_decay_rate": params.get("sms_decay_rate", 0.0),
           "sms_const": params.get("sms_const", 0.0)}
    out["tend"] = np.stack([tm.comp_tend(t, y, processes).copy() for t in times])
    for i, t in enumerate(times):
        jac = tm.comp_jacobian(t, y, processes).tocsr()
        jac.sum_duplicates()
        jac.sort_indices()
        out[f"jac{i}_data"], out[f"jac{i}_indices"], out[f"jac{i}_indptr"] = jac.data, jac.indices, jac.indptr
    if with_fcn:
        # init_iterate_vals [1.0] of the forced_{suff} definition plus seeded structure: from
        # the exactly uniform state the reference's own map amplifies 1e-15 noise to 1e-3
        # (stale-Jacobian Newton on roundoff-level modes), which no parity test can use
        y0 = np.ones(nz * ny) + 0.2 * np.random.default_rng(seed + 100).standard_normal(nz * ny)
        time_range = (0.0, year)
        sol = integrate.solve_ivp(
            tm.comp_tend, time_range, y0, "Radau", np.array(time_range), max_step=year * 0.01,
            atol=1.0e-6, rtol=1.0e-6, args=(processes,), jac=tm.comp_jacobian)
        out.update(y0=y0, fcn=sol.y[:, -1] - y0, nfev=sol.nfev, njev=sol.njev, nlu=sol.nlu)
    np.savez_compressed(os.path.join(HERE, f"forced_{tag}.npz"), **out)
    print("wrote forced", tag)


def synthetic_forcing(nz, ny, seed):
    """seeded forcing records on the model axes: 12 records from day 15 to day 345 (both ends of the
    year are extrapolated), restoring targets around 1, sources of both signs around 1e-8"""
    rng = np.random.default_rng(seed)
    times = (15.0 + 30.0 * np.arange(12)) * 86400.0
    phase = 2.0 * np.pi * times / (365.0 * 86400.0)
    restore = 1.0 + 0.3 * np.sin(phase)[:, None] * np.linspace(0.5, 1.5, ny)[None, :] \
        + 0.05 * rng.standard_normal((12, ny))
    sms = 2.0e-8 * (np.cos(phase)[:, None, None] * np.exp(-np.arange(nz) / (0.3 * nz))[None, :, None]
                    + 0.5 * rng.standard_normal((12, nz, ny)))
    return times, restore, sms


def gen_forced_file(tag, nz, ny, restore_opt, sms_opt, sink_thres, seed, with_fcn=False):
    """reference forced module with file-driven options.  The reference builds its forcing functions
    with utils.gen_forcing_fcn, which reads a NetCDF file (netCDF4, absent here) and ends in
    `interpolate.interp1d(dim0_in, data, axis=0, fill_value="extrapolate", assume_sorted=True)`
    (utils.py:529-531); the generator makes that same call on seeded records that are already on the
    model axes and hands the functions to the genuine comp_tend / comp_jacobian."""
    from scipy import integrate, interpolate
    from scipy import sparse
    from scipy.sparse import linalg as sp_linalg

    from nk_ooc.py_driver_2d.forced import forced

    params = {"surf_restore_opt": restore_opt, "sms_opt": sms_opt}
    if restore_opt == "const":
        params["surf_restore_const"] = 1.5
    if sms_opt == "decay":
        params["sms_decay_rate"] = 1.0e-8
    if sms_opt == "file":
        params["sms_scalef"] = 1.0
        if sink_thres is not None:
            params["sink_thres"] = sink_thres
    depth, ypos, processes, tm = ref_forced(nz, ny, params)
    times_rec, restore, sms = synthetic_forcing(nz, ny, seed)
    if restore_opt == "file":
        forced.surf_restore_fcn = interpolate.interp1d(times_rec, restore, axis=0, fill_value="extrapolate",
                                                        assume_sorted=True)
    if sms_opt == "file":
        forced.sms_fcn = interpolate.interp1d(times_rec, sms, axis=0, fill_value="extrapolate", assume_sorted=True)
    year = 365.0 * 86400.0
    times = [0.0, 0.3 * year, 0.66 * year, year]
    rng = np.random.default_rng(seed + 1)
    y = 0.4 + 0.4 * rng.standard_normal(nz * ny)      # below and above the threshold, some negative
    out = {"nz": nz, "ny": ny, "times": np.asarray(times), "y": y, "surf_restore_opt": restore_opt,
           "sms_opt": sms_opt, "sink_thres": -1.0 if sink_thres is None else sink_thres,
           "surf_restore_const": params.get("surf_restore_const", 0.0),
           "sms_decay_rate": params.get("sms_decay_rate", 0.0),
           "rec_times": times_rec, "restore_vals": restore, "sms_vals": sms}
    out["tend"] = np.stack([tm.comp_tend(t, y, processes).copy() for t in times])
    for i, t in enumerate(times):
        jac = sparse.csr_matrix(tm.comp_jacobian(t, y, processes))
        jac.sum_duplicates()
        jac.sort_indices()
        out[f"jac{i}_data"], out[f"jac{i}_indices"], out[f"jac{i}_indptr"] = jac.data, jac.indices, jac.indptr
    # the preconditioner formula of forced.apply_precond_jacobian (forced.py:222-241) with the tracer
    # of the three time levels given directly (the reference reads them from the precond file)
    states = [0.4 + 0.4 * np.random.default_rng(seed + 10 + k).standard_normal(nz * ny) for k in range(3)]
    v = np.random.default_rng(seed + 20).standard_normal(nz * ny)
    time_delta = year / 3
    mat_id = sparse.identity(v.size)
    mat = sparse.identity(v.size)
    for k in range(3):
        mat_tmp = time_delta * tm.comp_jacobian((k + 0.5) * time_delta, states[k], processes)
        mat *= mat_id - mat_tmp
    mat = mat_id - mat
    out.update(precond_states=np.stack(states), precond_v=v, precond_res=sp_linalg.spsolve(sparse.csc_matrix(mat), v) - v)
    if with_fcn:
        y0 = 0.6 + 0.2 * np.random.default_rng(seed + 100).standard_normal(nz * ny)
        time_range = (0.0, year)
        sol = integrate.solve_ivp(
            tm.comp_tend, time_range, y0, "Radau", np.array(time_range), max_step=year * 0.01,
            atol=1.0e-6, rtol=1.0e-6, args=(processes,), jac=tm.comp_jacobian)
        out.update(y0=y0, fcn=sol.y[:, -1] - y0, nfev=sol.nfev, njev=sol.njev, nlu=sol.nlu)
    np.savez_compressed(os.path.join(HERE, f"forced_{tag}.npz"), **out)
    print("wrote forced", tag)


def gen_phosphorus(tag, nz, ny, seed, with_fcn=False):
    """reference phosphorus module: tendencies, Jacobian, one forward year"""
    from scipy import integrate

    from nk_ooc.py_driver_2d.advection import Advection
    from nk_ooc.py_driver_2d.horiz_mix import HorizMix
    from nk_ooc.py_driver_2d.phosphorus import phosphorus
    from nk_ooc.py_driver_2d.vert_mix import VertMix

    depth, ypos = ref_axes(nz, ny)
    modelinfo = {"max_abs_vvel": "0.1", "horiz_mix_coeff": "1000.0"}
    processes = {"advection": Advection(depth, ypos, modelinfo),
                 "horiz_mix": HorizMix(depth, ypos, modelinfo),
                 "vert_mix": VertMix(depth, ypos)}
    tm = object.__new__(phosphorus)
    tm.name = "phosphorus"
    tm.tracer_cnt = 3
    tm.depth = depth
    tm.ypos = ypos
    tm.light_lim = np.outer(np.exp((-1.0 / 25.0) * depth.mid),
                            np.exp(-1.0 * ((ypos.mid - 2.5e6) / 1.5e6) ** 2))
    tm.po4_ind, tm.dop_ind, tm.pop_ind = 0, 1, 2
    tm.params = phosphorus.gen_params({})
    tm.pop_sink_work = np.zeros((nz + 1, ny))
    year = 365.0 * 86400.0
    times = [0.0, 0.3 * year, 0.66 * year]
    rng = np.random.default_rng(seed)
    # positive, vertically structured state (the init_iterate profiles of tracer_module_defs.yaml)
    prof = [np.interp(depth.mid, zs, vs) for zs, vs in (([1.3e2, 2.6e2], [5.5e-3, 4.1e0]),
                                                       ([9.5e1, 1.4e2], [7.1e-2, 1.5e-4]),
                                                       ([1.7e2, 2.5e2], [1.8e-2, 7.9e-4]))]
    y0 = np.stack([np.broadcast_to(p[:, None], (nz, ny)) for p in prof]).copy()
    y = (y0 * (1.0 + 0.2 * rng.random(y0.shape))).reshape(-1)
    out = {"nz": nz, "ny": ny, "times": np.asarray(times), "y": y, "light_lim": tm.light_lim}
    out["tend"] = np.stack([tm.comp_tend(t, y, processes).copy() for t in times])
    for i, t in enumerate(times):
        jac = tm.comp_jacobian(t, y, processes).tocsr()
        jac.sum_duplicates()
        jac.sort_indices()
        out[f"jac{i}_data"], out[f"jac{i}_indices"], out[f"jac{i}_indptr"] = jac.data, jac.indices, jac.indptr
    if with_fcn:
        time_range = (0.0, year)
        sol = integrate.solve_ivp(
            tm.comp_tend, time_range, y, "Radau", np.array(time_range), max_step=year * 0.01,
            atol=1.0e-6, rtol=1.0e-6, args=(processes,), jac=tm.comp_jacobian)
        out.update(y0=y, fcn=sol.y[:, -1] - y, nfev=sol.nfev, njev=sol.njev, nlu=sol.nlu)
    np.savez_compressed(os.path.join(HERE, f"phosphorus_{tag}.npz"), **out)
    print("wrote phosphorus", tag)


def gen_lstsq(seed):
    """_comp_krylov_basis_coeffs known answers (krylov_solver.py:168-181)"""
    from nk_ooc.krylov_solver import _comp_krylov_basis_coeffs

    rng = np.random.default_rng(seed)
    out = {}
    for case, (ntm, j, nreg) in enumerate([(1, 0, 1), (1, 3, 1), (2, 5, 3)]):
        h = np.zeros((ntm, j + 2, j + 1, nreg))
        for c in range(j + 1):
            h[:, : c + 2, c, :] = rng.standard_normal((ntm, c + 2, nreg))
        beta = np.abs(rng.standard_normal((ntm, nreg))) + 0.1
        out[f"h{case}"] = h
        out[f"beta{case}"] = beta
        out[f"coeff{case}"] = _comp_krylov_basis_coeffs(beta, h)
    np.savez_compressed(os.path.join(HERE, "lstsq.npz"), **out)
    print("wrote lstsq")


def gen_forced_file_all():
    gen_forced_file("file_restore_sms_22x9", 22, 9, "file", "file", None, 11, with_fcn=True)
    gen_forced_file("file_sink_thres_22x9", 22, 9, "const", "file", 0.5, 12, with_fcn=True)
    gen_forced_file("file_restore_decay_70x5", 70, 5, "file", "decay", None, 13)


def gen_comp_fcn_large(n, leg):
    """forward year on an n x n grid (n = 52, 104: the production-mode parity cases).  Two legs that
    may run as separate processes (each takes ~15 min at 104 x 104):
      leg "ref":   solve_ivp driven with the genuine reference functions, exactly as
                   py_driver_2d/model_state.py:102-114 does -> y0, fcn, nfev/njev/nlu;
      leg "sched": the oracle's Radau restatement on the same y0 -> accepted-step schedule
                   (t, t_new, h, n_newton, t_jac, h_lu) for step-replay mode, and its result.
    `merge` joins them into comp_fcn_{n}x{n}.npz after checking that the two results are bit-identical
    (the oracle reproduces solve_ivp)."""
    tag = f"{n}x{n}"
    part = os.path.join(HERE, f"_part_{leg}_{tag}.npz")
    if leg == "ref":
        from scipy import integrate

        depth, ypos, processes, tm = ref_setup(n, n, 0.1, 1000.0)
        col = np.interp(depth.mid, [55.0, 200.0], [0.0, 2.0])
        y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).reshape(-1).copy()
        time_range = (0.0, 365.0 * 86400.0)
        sol = integrate.solve_ivp(
            tm.comp_tend, time_range, y0, "Radau", np.array(time_range),
            max_step=(time_range[1] - time_range[0]) * 0.01, atol=1.0e-6, rtol=1.0e-6,
            args=(processes,), jac=tm.comp_jacobian,
            jac_sparsity=tm.comp_jacobian_sparsity(time_range[0], y0, processes))
        np.savez_compressed(part, y0=y0, fcn=sol.y[:, -1] - y0, nfev=sol.nfev, njev=sol.njev, nlu=sol.nlu)
        print("ref leg", tag, sol.nfev, sol.njev, sol.nlu)
    elif leg == "sched":
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
        from oracle import radau
        from oracle.grid import default_axes
        from oracle.model import Iage, Py2dModel, gen_init_iterate

        model = Py2dModel(*default_axes(n, n), 0.1, 1000.0)
        tm = Iage(model)
        y0 = gen_init_iterate(model).reshape(-1)
        res, solver = radau.comp_fcn(tm, y0, return_solver=True)
        st = solver.stats
        np.savez_compressed(part, y0=y0, fcn=res, schedule=np.array(solver.schedule, dtype=np.float64),
                            nfev=st.nfev, njev=st.njev, nlu=st.nlu, nrejected=st.nrejected, nnewton=st.nnewton)
        print("sched leg", tag, st.nfev, st.njev, st.nlu, len(solver.schedule))
    else:
        ref = np.load(os.path.join(HERE, f"_part_ref_{tag}.npz"))
        sch = np.load(os.path.join(HERE, f"_part_sched_{tag}.npz"))
        assert np.array_equal(ref["y0"], sch["y0"]) and np.array_equal(ref["fcn"], sch["fcn"]), \
            "oracle and solve_ivp(reference functions) differ"
        for key in ("nfev", "njev", "nlu"):
            assert int(ref[key]) == int(sch[key]), key
        np.savez_compressed(os.path.join(HERE, f"comp_fcn_{tag}.npz"), nz=n, ny=n, max_abs_vvel=0.1,
                            horiz_mix_coeff=1000.0, y0=ref["y0"], fcn=ref["fcn"], nfev=ref["nfev"],
                            njev=ref["njev"], nlu=ref["nlu"], schedule=sch["schedule"],
                            nrejected=sch["nrejected"], nnewton=sch["nnewton"])
        print("wrote comp_fcn", tag, "(solve_ivp with reference functions == oracle, bit for bit)")


def main():
    _install_placeholders()
    sys.path.insert(0, REF)
    if len(sys.argv) == 4 and sys.argv[1] == "large":     # large <n> <ref|sched|merge>
        gen_comp_fcn_large(int(sys.argv[2]), sys.argv[3])
        return
    if sys.argv[1:] == ["forced_file"]:      # only the fixtures of the file-driven forced options
        gen_forced_file_all()
        return
    year = 365.0 * 86400.0
    times = [0.0, 0.3 * year, 0.5 * year, 0.7 * year, 0.2613 * year]
    gen_static("26x26", 26, 26, 0.1, 1000.0, times, 0)
    gen_static("30x30", 30, 30, 0.1, 1000.0, times[:3], 1)
    gen_static("70x40", 70, 40, 0.1, 1000.0, times[:3], 2)
    gen_static("20x3_columns", 20, 3, 0.0, 0.0, times[:3], 3)
    gen_precond("26x26", 26, 26, 0.1, 1000.0, 4)
    gen_precond("20x3_columns", 20, 3, 0.0, 0.0, 5)
    gen_lstsq(6)
    gen_phosphorus("22x9", 22, 9, 9, with_fcn=True)
    gen_phosphorus("70x12", 70, 12, 10)
    gen_forced("decay_22x9", 22, 9, {"surf_restore_opt": "none", "sms_opt": "decay",
                                     "sms_decay_rate": 1.0e-8}, 7, with_fcn=True)
    gen_forced("restore_const_22x9", 22, 9, {"surf_restore_opt": "const", "surf_restore_const": 1.5,
                                             "sms_opt": "const", "sms_const": -2.0e-9}, 8)
    gen_forced_file_all()
    gen_comp_fcn("20x3_columns", 20, 3, 0.0, 0.0)
    gen_comp_fcn("26x26", 26, 26, 0.1, 1000.0)
    gen_comp_fcn("26x26_bumpy", 26, 26, 0.1, 1000.0, init="bumpy")


if __name__ == "__main__":
    main()
