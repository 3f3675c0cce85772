"""History file of one forward model year (61 dense-output samples).

The Newton driver asks `comp_fcn` for a history file; the reference writes it in
`nk_ooc/py_driver_2d/model_state.py:141-233` with the per-process variables of
`advection.py:78-109`, `horiz_mix.py:73-98`, `vert_mix.py:103-138` and the tracer-like
variables of `py_driver_2d/tracer_module_state.py:110-260`.  The samples themselves come from
the device integrator's dense output (`nk2d_comp_fcn_hist`); the reductions below are the
reference's, on the host, once per file.
"""

import numpy as np
from scipy.io import netcdf_file

from . import ncio
from .grid import BLDEPTH_MIN, bldepth_time_knots


def hist_stamp():
    return ncio.history_stamp(f"{__name__}._gen_hist")


def time_mean_weights(ntime):
    """trapezoid in time: t = 0 and t = end are both in the file"""
    weights = np.full(ntime, 1.0 / (ntime - 1))
    weights[0] *= 0.5
    weights[-1] *= 0.5
    return weights


def bldepth(grid, time):
    tvals, fvals = bldepth_time_knots()
    frac = np.interp(time, tvals, fvals)
    return BLDEPTH_MIN + (grid.bldepth_max - BLDEPTH_MIN) * frac


_PINT_SYMBOL = {"years": "a", "year": "a"}     # pint abbreviates the year as "a" and sorts by symbol


def _parse_units(expr):
    """{unit: power} of a units expression: blanks and `*` multiply, `/` divides the NEXT term
    only, `^` raises, parentheses group (pint evaluates left to right)"""
    powers = {}

    def merge(other, sign):
        for name, power in other.items():
            powers[name] = powers.get(name, 0) + sign * power

    pos, sign, n = 0, 1, len(expr)
    while pos < n:
        ch = expr[pos]
        if ch.isspace() or ch == "*":
            pos += 1
        elif ch == "/":
            sign = -1
            pos += 1
        elif ch == "(":
            depth, end = 1, pos + 1
            while depth:
                depth += {"(": 1, ")": -1}.get(expr[end], 0)
                end += 1
            merge(_parse_units(expr[pos + 1:end - 1]), sign)
            pos, sign = end, 1
        else:
            end = pos
            while end < n and not expr[end].isspace() and expr[end] not in "*/()":
                end += 1
            name, _, exponent = expr[pos:end].partition("^")
            if name != "1":
                merge({name: int(exponent) if exponent else 1}, sign)
            pos, sign = end, 1
    return powers


def units_str_format(units_str):
    """units string in the reference's canonical format (`utils.units_str_format`,
    nk_ooc/utils.py:189-205, which lets pint evaluate the expression): powers of a unit merged,
    numerator terms separated by blanks, every denominator term after a " / ", terms ordered by
    pint's symbol; a time unit (d, s) that would come first of two denominator terms goes last.
    Pinned by the reference's tests/test_utils.py cases (tests/test_host.py)."""
    powers = _parse_units(units_str)
    order = sorted(powers, key=lambda name: _PINT_SYMBOL.get(name, name))

    def term(name):
        power = abs(powers[name])
        return name if power == 1 else f"{name}^{power}"

    numer = [term(name) for name in order if powers[name] > 0]
    denom = [term(name) for name in order if powers[name] < 0]
    parts = [" ".join(numer) if numer else "1"] + denom
    if len(parts) == 3 and parts[1] in ("d", "s"):
        parts[1], parts[2] = parts[2], parts[1]
    return " / ".join(parts)


def _units_product(*units):
    return units_str_format(" ".join(f"({unit})" for unit in units))


# SI value and dimension of the units that appear in the files of this code base (tracer_module_defs.yaml,
# history and stats files); prefixes are spelled out, nothing is guessed
_UNIT_TABLE = {
    "m": (1.0, "L"), "meter": (1.0, "L"), "meters": (1.0, "L"), "cm": (1.0e-2, "L"), "mm": (1.0e-3, "L"),
    "km": (1.0e3, "L"),
    "s": (1.0, "T"), "sec": (1.0, "T"), "second": (1.0, "T"), "seconds": (1.0, "T"), "h": (3600.0, "T"),
    "hr": (3600.0, "T"), "hour": (3600.0, "T"), "hours": (3600.0, "T"),
    "d": (86400.0, "T"), "day": (86400.0, "T"), "days": (86400.0, "T"),
    "a": (365.25 * 86400.0, "T"), "yr": (365.25 * 86400.0, "T"), "year": (365.25 * 86400.0, "T"),
    "years": (365.25 * 86400.0, "T"),
    "mol": (1.0, "N"), "mmol": (1.0e-3, "N"), "umol": (1.0e-6, "N"), "nmol": (1.0e-9, "N"),
    "g": (1.0e-3, "M"), "kg": (1.0, "M"), "mg": (1.0e-6, "M"),
}


def units_conversion_factor(units_from, units_to):
    """number to multiply values in `units_from` with to express them in `units_to` (the role of
    pint's Quantity.to in utils.isclose_all_vars, nk_ooc/utils.py:304-310); None when a unit is not in
    the table or the dimensions differ"""
    def reduce(expr):
        scale, dims = 1.0, {}
        for name, power in _parse_units(expr).items():
            if name not in _UNIT_TABLE:
                return None
            value, dim = _UNIT_TABLE[name]
            scale *= value ** power
            dims[dim] = dims.get(dim, 0) + power
        return scale, {dim: power for dim, power in dims.items() if power}

    src, dst = reduce(units_from), reduce(units_to)
    if src is None or dst is None or src[1] != dst[1]:
        return None
    return src[0] / dst[0]


def tracer_var_attrs(attrs, tname, label="", unit=None):
    """attributes of a tracer-like history variable, in the order the file holds them"""
    var_attrs = dict(attrs)
    var_attrs["long_name"] = attrs.get("long_name", tname) + label
    var_attrs["units"] = attrs.get("units", "1") if unit is None else unit
    return var_attrs


class HistWriter:
    """History files are written on a background thread: a file is 423 MB at 416 x 416 (the 61 samples of every tracer,
    their anomalies, the mixing coefficient), a second of reductions and big-endian conversion that nothing on the Newton
    level waits for -- what the solvers need from a history (its time axis for the preconditioner file, the tracer samples
    for the statistics) is served from the record kept in memory.  Everything that touches a history file BY NAME goes
    through here: `wait` before reading, `remove` / `rename` instead of os.remove / os.rename."""

    KEEP = 3      # records kept in memory (about 170 MB each at 416 x 416 with two tracers)

    def __init__(self):
        self._pool = None
        self._pending = {}
        self._records = {}

    def submit(self, fname, grid, time, module_hists, vmix_samples, stamp, background=True):
        import os
        from concurrent.futures import ThreadPoolExecutor

        key = os.path.abspath(fname)
        self.wait(fname)
        self._records.pop(key, None)
        self._records[key] = {"time": np.array(time), "module_hists": module_hists, "stamp": stamp}
        while len(self._records) > self.KEEP:
            self._records.pop(next(iter(self._records)))
        if not background:
            write_hist_file(fname, grid, time, module_hists, vmix_samples, stamp)
            return
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="nk2d-hist")
        # at most two files in flight: a third submit waits for the oldest (bounds the host memory held by the queue)
        while len(self._pending) >= 2:
            self.wait(next(iter(self._pending)))
        self._pending[key] = self._pool.submit(_write_then_rename, fname, grid, time, module_hists, vmix_samples, stamp)

    def wait(self, fname=None):
        import os

        keys = list(self._pending) if fname is None else [os.path.abspath(fname)]
        for key in keys:
            fut = self._pending.pop(key, None)
            if fut is not None:
                fut.result()

    def record(self, fname):
        import os

        return self._records.get(os.path.abspath(fname)) if fname is not None else None

    def tracer_samples(self, fname, names):
        """{tracer name: (samples [ntime, nz, ny], attributes as the file holds them)} of a history kept in memory, or None"""
        rec = self.record(fname)
        if rec is None:
            return None
        out = {}
        for tracers, hist in rec["module_hists"]:
            for ind, (tname, attrs) in enumerate(tracers.items()):
                if tname in names:
                    var_attrs = tracer_var_attrs(attrs, tname)
                    var_attrs["cell_methods"] = "time: point"
                    out[tname] = (hist[:, ind], var_attrs)
        return out if len(out) == len(names) else None

    def _queue(self, key, job):
        """a file operation behind whatever the writer's thread still has to do (one thread: jobs run in the order given)"""
        if self._pool is None or not self._pending:
            job()
        else:
            self._pending.pop(key, None)
            self._pending[key] = self._pool.submit(job)

    def remove(self, fname):
        import os

        key = os.path.abspath(fname)
        self._records.pop(key, None)

        def job():
            if os.path.exists(fname):
                os.remove(fname)

        self._queue(key, job)

    def rename(self, src, dst):
        """src -> dst once src is written (queued behind it: the caller does not wait for a file it only renames)"""
        import os

        ksrc, kdst = os.path.abspath(src), os.path.abspath(dst)
        rec = self._records.pop(ksrc, None)
        self._records.pop(kdst, None)
        if rec is not None:
            self._records[kdst] = rec
        was_pending = ksrc in self._pending

        def job():
            if os.path.exists(src):
                os.rename(src, dst)

        if was_pending:
            # whoever waits for src or dst now waits for the rename
            self._pending.pop(ksrc, None)
            self._pending.pop(kdst, None)
            self._pending[kdst] = self._pool.submit(job)
        else:
            self.wait(dst)
            job()

    def forget(self):
        self.wait()
        self._records.clear()


def _write_then_rename(fname, grid, time, module_hists, vmix_samples, stamp):
    """the background write: under a temporary name, then renamed -- the step log may already call the forward year complete
    while this is on its way, and a run killed in between must find either the whole file or none under the name (a resumed
    run then fails on the missing file and is rewound, as after any other interrupted step; never on half a file)"""
    import os

    tmp = fname + ".partial"
    write_hist_file(tmp, grid, time, module_hists, vmix_samples, stamp)
    os.replace(tmp, fname)


def write_hist_file(fname, grid, time, module_hists, vmix_coeff, stamp=None):
    """module_hists: list of (tracer metadata dict name -> attrs, hist [ntime, tc, nz, ny]);
    vmix_coeff: the (nz-1, ny) mixing coefficient / dz_mid at every time [ntime, nz-1, ny], or a function of t returning one;
    stamp: the history attribute (made by the caller when the file is written later than it is asked for)"""
    depth, ypos = grid.depth, grid.ypos
    dname, yname = depth.axisname, ypos.axisname
    dedge, yedge = depth.dump_names["edges"], ypos.dump_names["edges"]
    ntime = len(time)
    weights = time_mean_weights(ntime)
    with netcdf_file(fname, "w", version=2) as fptr:
        fptr.history = stamp if stamp is not None else hist_stamp()
        fptr.createDimension("time", None)
        for axis in (depth, ypos):
            for dimname, dimlen in axis.dump_dimensions().items():
                if dimname not in fptr.dimensions:
                    fptr.createDimension(dimname, dimlen)

        def defvar(name, dims, attrs):
            var = fptr.createVariable(name, ">f8", dims)
            for key, val in attrs.items():
                setattr(var, key, val)
            if name != "time" and "time" in dims:
                var.cell_methods = "time: point"
            return var

        defvar("time", ("time",), {"long_name": "time", "units": "seconds since 0001-01-01",
                                   "calendar": "noleap"})
        for axis in (depth, ypos):
            for name, metadata in axis.dump_vars_metadata().items():
                defvar(name, metadata["dimensions"], metadata["attrs"])
        defvar("stream", (dedge, yedge), {"long_name": "velocity streamfunction", "units": "m^2 / s"})
        defvar("vvel", (dname, yedge), {"long_name": "velocity in ypos direction", "units": "m / s"})
        defvar("wvel", (dedge, yname), {"long_name": "velocity in depth direction", "units": "m / s"})
        defvar("horiz_mixing_coeff", (dname, yedge),
               {"long_name": "horizontal mixing coefficient", "units": "m^2 / s"})
        defvar("bldepth", ("time", yname), {"long_name": "boundary layer depth", "units": "m"})
        defvar("vert_mixing_coeff", ("time", dedge, yname),
               {"long_name": "vertical mixing coefficient", "units": "m^2 / s"})
        for tracers, _ in module_hists:
            for tname, attrs in tracers.items():
                units = attrs.get("units", "1")
                for suffix, dims, label, unit in (
                    ("", ("time", dname, yname), "", units),
                    ("_time_mean", (dname, yname), ", time mean", units),
                    ("_time_anom", ("time", dname, yname), ", time anomaly", units),
                    ("_time_std", (dname, yname), ", time std dev", units),
                    ("_time_delta", (dname, yname), ", end state minus start state", units),
                    ("_depth_int", ("time", yname), ", depth integral", _units_product(units, depth.units)),
                    ("_ypos_mean", ("time", dname), ", ypos mean", units),
                    ("_depth_ypos_int", ("time",), ", depth-ypos integral",
                     _units_product(units, depth.units, ypos.units)),
                ):
                    defvar(tname + suffix, dims, tracer_var_attrs(attrs, tname, label, unit))

        # ---- values
        fptr.variables["time"][:] = time
        for axis in (depth, ypos):
            for name, vals in axis.dump_vals_dict().items():
                fptr.variables[name][:] = vals
        fptr.variables["stream"][:] = grid.stream
        fptr.variables["vvel"][:] = grid.vvel
        fptr.variables["wvel"][:] = grid.wvel
        hmix = np.empty((len(depth), len(ypos) + 1))
        hmix[:, 1:-1] = grid.hmix_coeff * ypos.delta_mid
        hmix[:, 0] = hmix[:, 1]      # edge values copied to avoid missing values
        hmix[:, -1] = hmix[:, -2]
        fptr.variables["horiz_mixing_coeff"][:] = hmix
        bld = np.stack([bldepth(grid, t) for t in time])
        fptr.variables["bldepth"][:] = bld
        vmix = np.empty((ntime, len(depth) + 1, len(ypos)))
        for ind, t in enumerate(time):
            coeff = vmix_coeff(t) if callable(vmix_coeff) else vmix_coeff[ind]
            vmix[ind, 1:-1, :] = coeff * depth.delta_mid[:, np.newaxis]
        vmix[:, 0, :] = vmix[:, 1, :]
        vmix[:, -1, :] = vmix[:, -2, :]
        fptr.variables["vert_mixing_coeff"][:] = vmix
        yspan = ypos.edges.max() - ypos.edges.min()
        for tracers, hist in module_hists:
            for ind, tname in enumerate(tracers):
                vals = hist[:, ind]                       # (time, depth, ypos)
                mean = np.einsum("i,i...", weights, vals)
                anom = vals - mean
                fptr.variables[tname][:] = vals
                fptr.variables[tname + "_time_mean"][:] = mean
                fptr.variables[tname + "_time_anom"][:] = anom
                fptr.variables[tname + "_time_std"][:] = np.sqrt(np.einsum("i,i...", weights, anom ** 2))
                fptr.variables[tname + "_time_delta"][:] = vals[-1] - vals[0]
                fptr.variables[tname + "_depth_int"][:] = (depth.delta[:, np.newaxis] * vals).sum(axis=-2)
                ypos_int = (ypos.delta * vals).sum(axis=-1)
                fptr.variables[tname + "_ypos_mean"][:] = ypos_int / yspan
                fptr.variables[tname + "_depth_ypos_int"][:] = (depth.delta * ypos_int).sum(axis=-1)
