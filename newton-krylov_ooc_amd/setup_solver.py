"""Input generation for a run directory: grid_vars.nc and the initial iterate.

Counterpart of the reference's `nk_ooc/py_driver_2d/setup_solver.py:61-198` for the
pieces the Krylov hot path needs: `gen_grid_vars_file` (axes, `grid_weight` =
outer(dz, dy), `region_mask` = 1 or the column index when the lateral processes are
switched off) and `gen_init_iterate` (profile + `fp_cnt` forward years on the GPU).
"""

import os

import numpy as np

from . import ncio
from .grid import axis_from_modelinfo
from .model_config import ModelConfig, read_cfg_files
from .model_state import ModelState

_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUT_DIR = os.path.join(_REPO_ROOT, "input", "py_driver_2d")


def default_cfg_fnames():
    return ",".join(os.path.join(INPUT_DIR, name) for name in ("newton_krylov.cfg", "model_params.cfg"))


def make_config(workdir, nz=None, ny=None, tracer_module_names="iage", extra_modelinfo=None,
                extra_solverinfo=None, cfg_fnames=None):
    """read the reference cfg files with the overrides a CI script would put in
    `override.cfg` (grid size, lateral processes) and `--workdir --persist` on top"""
    modelinfo = {"tracer_module_names": tracer_module_names, "reinvoke": "False"}
    if nz is not None:
        modelinfo["depth_nlevs"] = str(nz)
    if ny is not None:
        modelinfo["ypos_nlevs"] = str(ny)
    modelinfo.update(extra_modelinfo or {})
    overrides = {"DEFAULT": {"workdir": workdir}, "modelinfo": modelinfo,
                 "solverinfo": dict(extra_solverinfo or {})}
    return read_cfg_files(cfg_fnames or default_cfg_fnames(), overrides=overrides)


def gen_grid_vars_file(modelinfo):
    """write grid_vars.nc (setup_solver.py:134-182)"""
    depth = axis_from_modelinfo(modelinfo["depth_axisname"], modelinfo)
    ypos = axis_from_modelinfo(modelinfo["ypos_axisname"], modelinfo)
    weight = np.outer(depth.delta, ypos.delta)
    if float(modelinfo["max_abs_vvel"]) == 0.0 and float(modelinfo["horiz_mix_coeff"]) == 0.0:
        mask = np.empty(weight.shape, dtype=np.int32)
        for ypos_i in range(weight.shape[1]):
            mask[:, ypos_i] = ypos_i + 1
    else:
        mask = np.ones(weight.shape, dtype=np.int32)
    dims = (depth.axisname, ypos.axisname)
    extra = {
        "grid_weight": (dims, ">f8", {"long_name": "grid-cell area", "units": "m^2"}, weight),
        "region_mask": (dims, ">i4", {"long_name": "Region Mask", "cell_measures": "area: grid_weight"}, mask),
    }
    fname = modelinfo["grid_vars_fname"]
    os.makedirs(os.path.dirname(fname), exist_ok=True)
    history = ncio.history_stamp(f"{__name__}.gen_grid_vars_file")
    ncio.write_state_file(fname, [depth, ypos], {}, history, extra_vars=extra)
    return depth, ypos


def setup(config, fp_cnt=1, init_iterate_opt="gen_init_iterate"):
    """grid_vars file, model configuration, initial iterate after fp_cnt forward years"""
    solverinfo, modelinfo = config["solverinfo"], config["modelinfo"]
    os.makedirs(solverinfo["workdir"], exist_ok=True)
    gen_grid_vars_file(modelinfo)
    ModelState.reset_class()
    ModelState.model_config_obj = ModelConfig(modelinfo)
    caller = f"{__name__}.setup"
    init_iterate = ModelState(init_iterate_opt)
    gen_dir = os.path.join(solverinfo["workdir"], "gen_init_iterate")
    os.makedirs(gen_dir, exist_ok=True)
    for fp_iter in range(fp_cnt):
        init_iterate.dump(os.path.join(gen_dir, f"init_iterate_{fp_iter:04}.nc"), caller)
        fcn = init_iterate.comp_fcn(os.path.join(gen_dir, f"fcn_{fp_iter:04}.nc"), None,
                                    os.path.join(gen_dir, f"hist_{fp_iter:04}.nc"))
        init_iterate += fcn
    init_iterate_fname = solverinfo["init_iterate_fname"]
    os.makedirs(os.path.dirname(init_iterate_fname), exist_ok=True)
    init_iterate.dump(init_iterate_fname, caller)
    return init_iterate
