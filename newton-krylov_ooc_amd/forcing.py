"""Forcing fields of the `forced` tracer module read from NetCDF files.

Reference: `gen_forcing_fcn` (`nk_ooc/utils.py:488-533`) reads a variable whose first dimension
is time, scales it, interpolates it linearly (with linear extrapolation) to the model's axes
along every other dimension whose coordinate differs from the model's, and returns a function of
time that interpolates linearly between the records (again extrapolating beyond the ends).

Here the spatial part runs once on the host (`load_forcing`); the records then live on the device
and the time interpolation happens inside the kernels that prepare the time-dependent planes of a
step attempt (`nk2d_desc.restore_* / sms_*`, include/nk2d.h).
"""

import numpy as np

from . import ncio


def interp_extrap(x_in, data, x_out, axis):
    """linear interpolation of `data` along `axis` from the increasing coordinate x_in to x_out;
    beyond the ends the first / last interval is continued (interp1d(fill_value="extrapolate")):
    slope = (y_hi - y_lo) / (x_hi - x_lo),  y = slope * (x - x_lo) + y_lo"""
    x_in = np.asarray(x_in, dtype=np.float64)
    x_out = np.asarray(x_out, dtype=np.float64)
    if x_in.ndim != 1 or len(x_in) < 2 or not np.all(np.diff(x_in) > 0.0):
        raise ValueError("coordinate must be 1-d, increasing, with at least 2 points")
    data = np.moveaxis(np.asarray(data, dtype=np.float64), axis, 0)
    if data.shape[0] != len(x_in):
        raise ValueError("coordinate length does not match the data")
    hi = np.clip(np.searchsorted(x_in, x_out), 1, len(x_in) - 1)
    lo = hi - 1
    shape = (-1,) + (1,) * (data.ndim - 1)
    slope = (data[hi] - data[lo]) / (x_in[hi] - x_in[lo]).reshape(shape)
    res = slope * (x_out - x_in[lo]).reshape(shape) + data[lo]
    return np.moveaxis(res, 0, axis)


def load_forcing(fname, varname, additional_dims_out, scalef=1.0):
    """(times, records on the model axes) of variable `varname` of NetCDF file `fname`;
    `additional_dims_out`: the model's coordinate of every non-time dimension, in file order"""
    dims = ncio.read_var_dims(fname, [varname])[varname]
    if len(dims) not in (1, 2, 3):
        raise ValueError(f"unexpected ndim={len(dims)}")
    if len(additional_dims_out) != len(dims) - 1:
        raise ValueError(f"len(additional_dims_out) = {len(additional_dims_out)} must be {len(dims) - 1}")
    data, _ = ncio.read_file(fname, [varname] + list(dims))
    vals = scalef * np.asarray(data[varname], dtype=np.float64)
    for axis in range(1, len(dims)):
        dim_in = np.asarray(data[dims[axis]], dtype=np.float64)
        dim_out = np.asarray(additional_dims_out[axis - 1], dtype=np.float64)
        if len(dim_in) != len(dim_out) or (dim_in != dim_out).any():
            vals = interp_extrap(dim_in, vals, dim_out, axis)
    return np.asarray(data[dims[0]], dtype=np.float64), np.ascontiguousarray(vals)
