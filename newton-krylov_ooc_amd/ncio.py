"""NetCDF3 (64-bit offset) vector files of the solver's on-disk trail.

The reference keeps every solver vector in its own NetCDF3_64BIT_OFFSET file
(`nk_ooc/model_state_base.py:93-111`, `nk_ooc/py_driver_2d/tracer_module_state.py:71-96`)
written through the netCDF4 package.  That package is not part of this image;
the same on-disk format (CDF-2, big-endian f8) is produced and read here with
`scipy.io.netcdf_file(version=2)`, so files written by either side open in the other.
"""

import os
from datetime import datetime

import numpy as np
from scipy.io import netcdf_file

from . import trail


def _decode(val):
    return val.decode() if isinstance(val, bytes) else val


def read_file(fname, varnames=None):
    """return (vars: dict name -> ndarray copy, attrs: dict) of a NetCDF3 file"""
    trail.flush()       # (a file of this process' own trail may still be on its way to the disk)
    with netcdf_file(fname, "r", mmap=False) as fptr:
        names = list(fptr.variables) if varnames is None else list(varnames)
        data = {}
        for name in names:
            if name not in fptr.variables:
                raise KeyError(f"variable {name} not found in {fname}")
            var = fptr.variables[name]
            arr = np.array(var.data)
            # native byte order for downstream ctypes hand-off
            data[name] = arr.astype(arr.dtype.newbyteorder("="), copy=True)
        attrs = {key: _decode(val) for key, val in fptr._attributes.items()}
    return data, attrs


def read_var_attrs(fname, varname):
    trail.flush()       # (a file of this process' own trail may still be on its way to the disk)
    with netcdf_file(fname, "r", mmap=False) as fptr:
        return {k: _decode(v) for k, v in fptr.variables[varname]._attributes.items()}


def read_var_dims(fname, varnames):
    trail.flush()       # (a file of this process' own trail may still be on its way to the disk)
    with netcdf_file(fname, "r", mmap=False) as fptr:
        return {name: tuple(fptr.variables[name].dimensions) for name in varnames}


def history_stamp(creator, caller=None):
    datestamp = datetime.now().strftime("%Y-%m-%d %H:%M:%S")
    msg = f"{datestamp}: created by {creator}"
    if caller is not None:
        msg = f"{msg} called from {caller}"
    return msg


def _define_axis(fptr, axis):
    for dimname, dimlen in axis.dump_dimensions().items():
        if dimname not in fptr.dimensions:
            fptr.createDimension(dimname, dimlen)
        elif fptr.dimensions[dimname] != dimlen:
            raise RuntimeError(f"dimension {dimname} length mismatch")
    if axis.axisname in fptr.variables:
        return
    for varname, metadata in axis.dump_vars_metadata().items():
        var = fptr.createVariable(varname, ">f8", metadata["dimensions"])
        for key, val in metadata["attrs"].items():
            setattr(var, key, val)


def write_state_file(fname, axes, tracer_vals, history, extra_vars=None):
    """write one model-state file: axis variables + one f8 (depth, ypos) variable per
    tracer, in the order the reference defines them.

    axes: [depth_axis, ypos_axis]; tracer_vals: ordered dict name -> (nz, ny) array;
    extra_vars: optional dict name -> (dims, dtype, attrs, values)"""
    # (under a temporary name, renamed when complete: a run killed inside the write leaves no half-written vector file under a
    # name a resumed run would open)
    partial = fname + ".partial"
    with netcdf_file(partial, "w", version=2) as fptr:
        fptr.history = history
        for axis in axes:
            _define_axis(fptr, axis)
        dims = tuple(axis.axisname for axis in axes)
        for name in tracer_vals:
            fptr.createVariable(name, ">f8", dims)
        if extra_vars:
            for name, (vdims, dtype, attrs, _) in extra_vars.items():
                var = fptr.createVariable(name, dtype, vdims)
                for key, val in attrs.items():
                    setattr(var, key, val)
        for axis in axes:
            for name, vals in axis.dump_vals_dict().items():
                fptr.variables[name][:] = vals
        for name, vals in tracer_vals.items():
            fptr.variables[name][:] = vals
        if extra_vars:
            for name, (_, _, _, vals) in extra_vars.items():
                fptr.variables[name][:] = vals
    os.replace(partial, fname)


def write_vars_file(fname, dimensions, variables, history):
    """generic writer: dimensions dict name -> len, variables dict
    name -> (dims, dtype, attrs, values)"""
    with netcdf_file(fname, "w", version=2) as fptr:
        fptr.history = history
        for dimname, dimlen in dimensions.items():
            fptr.createDimension(dimname, dimlen)
        for name, (vdims, dtype, attrs, _) in variables.items():
            var = fptr.createVariable(name, dtype, vdims)
            for key, val in attrs.items():
                setattr(var, key, val)
        for name, (_, _, _, vals) in variables.items():
            fptr.variables[name][:] = vals
