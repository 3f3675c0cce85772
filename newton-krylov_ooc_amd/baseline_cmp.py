#!/usr/bin/env python
"""Compare a NetCDF file of an experiment directory with the same file of a baseline directory.

Counterpart of the reference's comparer (`nk_ooc/baseline_cmp.py:12-55` with
`utils.metadata_same` / `utils.isclose_all_vars`, `nk_ooc/utils.py:212-342`), which its CI scripts
call once per file with per-file tolerances.  Same command line and exit status:

    python -m nk_ooc_amd.baseline_cmp --fname iterate_01.nc --expr_dir WORK --baseline_dir BASE \\
        --rtol 1.9e-2 --atol 2e-9

Files are read with SciPy's NetCDF3 reader (the format both code bases write).  Differences
are reported through `logging`, one line per mismatch, values with their adjusted tolerances.
Unit conversion between differing (non time-like) units needs pint, which this image lacks:
differing units are reported as a metadata mismatch instead.
"""

import argparse
import logging
import os
import sys

import numpy as np
from scipy.io import netcdf_file

from .hist import units_conversion_factor

LOG = logging.getLogger(__name__)


def _decode(val):
    return val.decode() if isinstance(val, bytes) else val


def _attrs(var):
    res = {}
    for key, val in var._attributes.items():
        val = _decode(val)
        res[key] = val.tolist() if isinstance(val, np.ndarray) else val
    return res


def _summary(fname):
    """({dim: len}, {var: (dims, attrs)}) of a file; record dimensions report their current length"""
    with netcdf_file(fname, "r", mmap=False) as fptr:
        variables = {name: (tuple(var.dimensions), _attrs(var)) for name, var in fptr.variables.items()}
        dims = {}
        for name, length in fptr.dimensions.items():
            if length is None:
                length = fptr._recs
            dims[name] = length
    return dims, variables


def metadata_same(fname1, fname2):
    """True if dimension names and lengths, variable names, dimensions and attributes agree"""
    dims1, vars1 = _summary(fname1)
    dims2, vars2 = _summary(fname2)
    same = True
    if dims1.keys() != dims2.keys():
        LOG.info("    dimension name mismatch in %s and %s", fname1, fname2)
        same = False
    for name in dims1.keys() & dims2.keys():
        if dims1[name] != dims2[name]:
            LOG.info("    %s length mismatch in %s and %s", name, fname1, fname2)
            same = False
    if vars1.keys() != vars2.keys():
        LOG.info("    variable name mismatch in %s and %s", fname1, fname2)
        same = False
    for name in vars1.keys() & vars2.keys():
        if vars1[name][0] != vars2[name][0]:
            LOG.info("    %s dimension mismatch in %s and %s", name, fname1, fname2)
            same = False
        if vars1[name][1] != vars2[name][1]:
            LOG.info("    %s attribute mismatch in %s and %s", name, fname1, fname2)
            same = False
    return same


def _close_values(name, vals1, fill1, vals2, fill2, rtol, atol):
    if vals1.shape != vals2.shape:
        LOG.info("    var1.shape %s != var2.shape %s for %s", vals1.shape, vals2.shape, name)
        return False
    same = True
    missing1 = np.zeros(vals1.shape, dtype=bool) if fill1 is None else vals1 == fill1
    missing2 = np.zeros(vals2.shape, dtype=bool) if fill2 is None else vals2 == fill2
    if (missing1 != missing2).any():
        LOG.info("    _FillValue pattern mismatch for %s", name)
        same = False
    either = missing1 | missing2
    vals1 = np.where(either, np.nan, vals1.astype(np.float64))
    vals2 = np.where(either, np.nan, vals2.astype(np.float64))
    close = np.isclose(vals1, vals2, rtol=rtol, atol=atol, equal_nan=True)
    if not close.all():
        for val1, val2 in zip(vals1[~close].reshape(-1), vals2[~close].reshape(-1)):
            gap = abs(val1 - val2)
            with np.errstate(divide="ignore", invalid="ignore"):
                rtol_adj = (gap - atol) / abs(val2)
            LOG.info("    %.10e %.10e not close, atol_adj=%e, rtol_adj=%e", val1, val2,
                     gap - rtol * abs(val2), rtol_adj)
        LOG.info("    %s vals not close", name)
        same = False
    return same


def isclose_all_vars(fname1, fname2, rtol, atol):
    """True if every variable the two files share is close (numpy.isclose semantics, NaN == NaN,
    fill values must coincide)"""
    same = True
    with netcdf_file(fname1, "r", mmap=False) as fptr1, netcdf_file(fname2, "r", mmap=False) as fptr2:
        for name, var1 in fptr1.variables.items():
            if name not in fptr2.variables:
                continue
            var2 = fptr2.variables[name]
            if var1.data.dtype.kind in "SU" or var2.data.dtype.kind in "SU":
                continue
            units1, units2 = _decode(getattr(var1, "units", None)), _decode(getattr(var2, "units", None))
            vals1 = np.array(var1.data)
            if units1 is not None and units2 is not None and units1 != units2:
                if "since" in units1 or "since" in units2:
                    raise ValueError(f"time-like units disagree '{units1}'!='{units2}'")
                # values of the first file are expressed in the units of the second (utils.py:304-310)
                factor = units_conversion_factor(units1, units2)
                if factor is None:
                    LOG.info("    %s units differ ('%s', '%s') and cannot be converted", name, units1, units2)
                    same = False
                    continue
                if factor != 1.0:
                    vals1 = vals1 * factor
            if not _close_values(name, vals1, getattr(var1, "_FillValue", None),
                                 np.array(var2.data), getattr(var2, "_FillValue", None), rtol, atol):
                same = False
    return same


def compare(fname, expr_dir, baseline_dir, rtol=1.0e-7, atol=2.0e-9):
    expr_fname = os.path.join(expr_dir, fname)
    baseline_fname = os.path.join(baseline_dir, fname)
    LOG.info("expr_fname = %s", expr_fname)
    LOG.info("baseline_fname = %s", baseline_fname)
    meta_ok = metadata_same(expr_fname, baseline_fname)
    vals_ok = isclose_all_vars(expr_fname, baseline_fname, rtol=rtol, atol=atol)
    return meta_ok and vals_ok


def parse_args(args_list=None):
    parser = argparse.ArgumentParser(description="compare NetCDF file to baseline",
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--fname", help="name of file to be compared")
    parser.add_argument("--expr_dir", help="directory with file")
    parser.add_argument("--baseline_dir", help="directory with baseline file")
    parser.add_argument("--rtol", help="relative tolerance", type=float, default=1.0e-7)
    parser.add_argument("--atol", help="absolute tolerance", type=float, default=2.0e-9)
    return parser.parse_args([] if args_list is None else args_list)


def main(args):
    logging.basicConfig(format="%(filename)s:%(funcName)s:%(message)s", level="INFO", stream=sys.stdout)
    sys.exit(0 if compare(args.fname, args.expr_dir, args.baseline_dir, args.rtol, args.atol) else 1)


if __name__ == "__main__":
    main(parse_args(sys.argv[1:]))
