"""`<Solver>_stats.nc`: per-iteration solver statistics (NetCDF3, unlimited `iteration`).

Same file the reference keeps (`nk_ooc/stats_file.py`), written with
`scipy.io.netcdf_file` instead of the netCDF4 package.
"""

import functools
import os

import numpy as np
from scipy.io import netcdf_file

from . import ncio
from . import trail
from .solver_state import action_step_log_wrap

_FILL = {"f8": 9.969209968386869e36, "i4": -2147483647}
_NC_TYPE = {"f8": ">f8", "i4": ">i4"}


def _snapshot(val):
    """what a queued call keeps of an argument: arrays and containers are copied, the caller may go on changing its own"""
    if isinstance(val, np.ndarray):
        return val.copy()
    if isinstance(val, dict):
        return {key: _snapshot(item) for key, item in val.items()}
    if isinstance(val, (list, tuple)):
        return type(val)(_snapshot(item) for item in val)
    return val


def _on_trail(method):
    """every change of the file is a job of the checkpoint trail (trail.py): run at once by default, behind the files
    written before it on the trail's writer thread where that is switched on"""

    @functools.wraps(method)
    def queued(self, *args, **kwargs):
        args = _snapshot(args)
        kwargs = {key: (val if key == "solver_state" else _snapshot(val)) for key, val in kwargs.items()}
        trail.submit(lambda: method(self, *args, **kwargs))

    return queued


class StatsFile:
    def __init__(self, name, workdir, region_cnt, solver_state):
        self._fname = os.path.join(workdir, f"{name}_stats.nc")
        self._create_stats_file(name=name, fname=self._fname, region_cnt=region_cnt,
                                solver_state=solver_state)

    @action_step_log_wrap("_create_stats_file {fname}", per_iteration=False)
    @_on_trail
    def _create_stats_file(self, name, fname, region_cnt, solver_state):
        with netcdf_file(fname, "w", version=2) as fptr:
            creator = f"{type(self).__module__}.{type(self).__name__}._create_stats_file"
            fptr.history = f"{ncio.history_stamp(creator)} for {name} solver"
            fptr.createDimension("iteration", None)
            fptr.createDimension("region", region_cnt)
            var = fptr.createVariable("iteration", ">i4", ("iteration",))
            var.long_name = f"{name} solver iteration"
            var = fptr.createVariable("region", ">i4", ("region",))
            var.long_name = "region index (0-based)"
            var.comment = "axis attribute is a work-around to enable pyferret to read stats files"
            var.axis = "T"
            fptr.variables["region"][:] = np.arange(region_cnt)

    @_on_trail
    def def_dimensions(self, dimensions):
        with netcdf_file(self._fname, "a") as fptr:
            for dimname, dimlen in dimensions.items():
                if dimname not in fptr.dimensions:
                    fptr.createDimension(dimname, dimlen)
                elif fptr.dimensions[dimname] != dimlen:
                    raise RuntimeError(f"dimension {dimname} length mismatch")

    @_on_trail
    def def_vars(self, vars_metadata):
        with netcdf_file(self._fname, "a") as fptr:
            for varname, metadata in vars_metadata.items():
                datatype = metadata.get("datatype", "f8")
                attrs = dict(metadata.get("attrs", {}))
                if "_FillValue" not in attrs and "iteration" in metadata["dimensions"]:
                    attrs["_FillValue"] = _FILL[datatype]
                var = fptr.createVariable(varname, _NC_TYPE[datatype], metadata["dimensions"])
                for key, val in attrs.items():
                    if val is not None:
                        setattr(var, key, val)
                # a record variable added after records exist must be as long as the others
                if metadata["dimensions"] and metadata["dimensions"][0] == "iteration":
                    for rec in range(fptr.variables["iteration"].shape[0]):
                        var[rec] = attrs["_FillValue"]

    @_on_trail
    def put_vars_iteration_invariant(self, name_vals_dict):
        if not name_vals_dict:
            return
        with netcdf_file(self._fname, "a") as fptr:
            for name, vals in name_vals_dict.items():
                var = fptr.variables[name]
                if "iteration" in var.dimensions:
                    raise RuntimeError(f"iteration is a dimension for {name}")
                var[:] = vals

    @_on_trail
    def put_vars(self, iteration, name_vals_dict):
        if not name_vals_dict:
            return
        with netcdf_file(self._fname, "a") as fptr:
            if iteration == fptr.variables["iteration"].shape[0]:
                for varname, var in fptr.variables.items():
                    if varname == "iteration":
                        var[iteration] = iteration
                    elif var.dimensions and var.dimensions[0] == "iteration":
                        var[iteration] = getattr(var, "_FillValue")
            for name, vals in name_vals_dict.items():
                var = fptr.variables[name]
                if "iteration" not in var.dimensions:
                    raise RuntimeError(f"iteration is not a dimension for {name}")
                var[iteration] = vals
