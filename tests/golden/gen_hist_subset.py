#!/usr/bin/env python
"""Reduced copy of a reference history baseline (data fixture): every variable of
baselines/ci_py_driver_2d_iage/hist_0000.nc with at most two dimensions in full, the (time, depth, ypos)
variables at every 10th of the 61 time samples.  2.3 MB of NetCDF -> a compressed .npz the repository can carry.
Run in the build container only (reads /root/reference)."""
import os

import numpy as np
from scipy.io import netcdf_file

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/baselines/ci_py_driver_2d_iage/hist_0000.nc"
TIME_IDX = np.arange(0, 61, 10)

out = {"time_index": TIME_IDX}
with netcdf_file(SRC, "r", mmap=False) as fptr:
    for name, var in fptr.variables.items():
        data = np.array(var.data, dtype=np.float64)
        if data.ndim == 3:
            data = data[TIME_IDX]
        out["var_" + name] = data
        out["dims_" + name] = np.array(",".join(var.dimensions))
np.savez_compressed(os.path.join(HERE, "ref_baselines", "ci_py_driver_2d_iage", "hist_0000_subset.npz"), **out)
print("wrote hist_0000_subset.npz with", len([k for k in out if k.startswith("var_")]), "variables")
