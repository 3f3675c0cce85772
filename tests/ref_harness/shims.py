"""Stand-ins for the three I/O-only third-party modules the reference imports and this image lacks
(netCDF4, xarray, pint), so that the reference's OWN solver classes can run in the build container
("shim plan").  TEST INFRASTRUCTURE: used by tests/golden/gen_ref_traces.py (which
makes the G7 / G8 fixtures with nk_ooc.krylov_solver / nk_ooc.newton_solver themselves) and by
tests/test_ref_dropin.py (the real nk_ooc.nk_driver driving this repository's plugin).  Nothing here is
on the arithmetic path: the reference uses these modules for files, for the per-module container of
tracer arrays and for unit strings.

* netCDF4.Dataset: an in-memory model of a classic (NetCDF3, 64-bit offset) file, read from / written to
  disk with scipy.io.netcdf_file -- modes r / w / a, one unlimited dimension, variables with attribute
  access, `__dict__` = attributes (as netCDF4's Variable), `default_fillvals`.
* xarray.Dataset / DataArray: an ordered mapping of named arrays with elementwise arithmetic.
* pint.UnitRegistry: unit strings compared as strings, no conversion.
"""
import sys
import types

import numpy as np
from scipy.io import netcdf_file

DEFAULT_FILLVALS = {"S1": "\x00", "i1": -127, "u1": 255, "i2": -32767, "u2": 65535, "i4": -2147483647,
                    "u4": 4294967295, "i8": -9223372036854775806, "u8": 18446744073709551614,
                    "f4": 9.969209968386869e36, "f8": 9.969209968386869e36}


class Dimension:
    def __init__(self, name, size):
        self.name = name
        self._size = size          # None: unlimited
        self._cur = 0

    def isunlimited(self):
        return self._size is None

    @property
    def size(self):
        return self._cur if self._size is None else self._size

    def __len__(self):
        return self.size


class Variable:
    __slots__ = ("_ds", "_name", "_dtype", "_dims", "_attrs", "_data")

    def __init__(self, ds, name, dtype, dims, fill_value):
        object.__setattr__(self, "_ds", ds)
        object.__setattr__(self, "_name", name)
        object.__setattr__(self, "_dtype", np.dtype(dtype).newbyteorder("="))
        object.__setattr__(self, "_dims", tuple(dims))
        object.__setattr__(self, "_attrs", {})
        fill = fill_value
        if fill is None:
            key = self._dtype.str[1:]
            fill = DEFAULT_FILLVALS.get(key, 0)
        else:
            self._attrs["_FillValue"] = self._dtype.type(fill_value)
        shape = tuple(len(ds.dimensions[d]) for d in self._dims)
        object.__setattr__(self, "_data", np.full(shape, fill, dtype=self._dtype))

    # netCDF4's Variable.__dict__ is the attribute dictionary (the reference relies on it)
    @property
    def __dict__(self):
        return dict(self._attrs)

    name = property(lambda self: self._name)
    dimensions = property(lambda self: self._dims)
    datatype = property(lambda self: self._dtype)
    dtype = property(lambda self: self._dtype)
    shape = property(lambda self: self._data.shape)
    ndim = property(lambda self: self._data.ndim)

    def __len__(self):
        return self._data.shape[0]

    def ncattrs(self):
        return list(self._attrs)

    def getncattr(self, key):
        return self._attrs[key]

    def setncattr(self, key, val):
        self._attrs[key] = val

    def setncatts(self, attrs):
        self._attrs.update(attrs)

    def __getattr__(self, key):
        attrs = object.__getattribute__(self, "_attrs")
        if key in attrs:
            return attrs[key]
        raise AttributeError(key)

    def __setattr__(self, key, val):
        self._attrs[key] = val

    def _is_record(self):
        return bool(self._dims) and self._ds.dimensions[self._dims[0]].isunlimited()

    def __getitem__(self, idx):
        return np.array(self._data[idx], copy=True)

    def __setitem__(self, idx, val):
        if self._is_record():
            first = idx[0] if isinstance(idx, tuple) else idx
            need = None
            if isinstance(first, (int, np.integer)):
                need = int(first) + 1
            elif isinstance(first, slice) and first.stop is None and first.start is None:
                need = np.asarray(val).shape[0] if np.ndim(val) == self._data.ndim else None
            if need is not None and need > self._data.shape[0]:
                self._ds._grow_records(need)
        self._data[idx] = val

    def _resize_records(self, nrec):
        if nrec <= self._data.shape[0]:
            return
        fill = self._attrs.get("_FillValue", DEFAULT_FILLVALS.get(self._dtype.str[1:], 0))
        grown = np.full((nrec,) + self._data.shape[1:], fill, dtype=self._dtype)
        grown[: self._data.shape[0]] = self._data
        object.__setattr__(self, "_data", grown)


class Dataset:
    def __init__(self, fname, mode="r", format=None, **_):   # noqa: A002 (netCDF4's keyword)
        d = object.__getattribute__(self, "__dict__")
        d["_nc_fname"] = fname
        d["_nc_mode"] = mode
        d["_nc_attrs"] = {}
        d["_nc_open"] = True
        d["dimensions"] = {}
        d["variables"] = {}
        if mode in ("r", "a", "r+"):
            self._read()
        elif mode != "w":
            raise ValueError(f"mode {mode!r} not supported by the netCDF4 stand-in")

    # ---- attributes of the file ------------------------------------------------------------
    def __getattr__(self, key):
        attrs = object.__getattribute__(self, "__dict__").get("_nc_attrs", {})
        if key in attrs:
            return attrs[key]
        raise AttributeError(key)

    def __setattr__(self, key, val):
        self._nc_attrs[key] = val

    def ncattrs(self):
        return list(self._nc_attrs)

    def getncattr(self, key):
        return self._nc_attrs[key]

    def setncattr(self, key, val):
        self._nc_attrs[key] = val

    def setncatts(self, attrs):
        self._nc_attrs.update(attrs)

    def set_auto_mask(self, flag):
        pass

    def sync(self):
        pass

    def filepath(self):
        return self._nc_fname

    # ---- definitions -------------------------------------------------------------------------
    def createDimension(self, name, size=None):
        if name in self.dimensions:
            raise RuntimeError("NetCDF: String match to name in use")
        if size is None and any(dim.isunlimited() for dim in self.dimensions.values()):
            raise RuntimeError("NetCDF: NC_UNLIMITED size already in use")
        self.dimensions[name] = Dimension(name, size)
        return self.dimensions[name]

    def createVariable(self, name, datatype, dimensions=(), fill_value=None, **_):
        if name in self.variables:
            raise RuntimeError("NetCDF: String match to name in use")
        if isinstance(dimensions, str):
            dimensions = (dimensions,)
        var = Variable(self, name, datatype, dimensions, fill_value)
        self.variables[name] = var
        return var

    def _grow_records(self, nrec):
        for dim in self.dimensions.values():
            if dim.isunlimited():
                dim._cur = max(dim._cur, nrec)
        for var in self.variables.values():
            if var._is_record():
                var._resize_records(nrec)

    # ---- disk -----------------------------------------------------------------------------------
    def _read(self):
        with netcdf_file(self._nc_fname, "r", mmap=False, maskandscale=False) as src:
            for key, val in src._attributes.items():
                self._nc_attrs[key] = val.decode() if isinstance(val, bytes) else val
            for name, size in src.dimensions.items():
                self.dimensions[name] = Dimension(name, size)
            nrec = getattr(src, "_recs", 0)
            for dim in self.dimensions.values():
                if dim.isunlimited():
                    dim._cur = nrec
            for name, svar in src.variables.items():
                var = Variable(self, name, svar.data.dtype, svar.dimensions, None)
                data = np.array(svar.data, dtype=svar.data.dtype.newbyteorder("="), copy=True)
                object.__setattr__(var, "_data", data.reshape(tuple(
                    (nrec if self.dimensions[d].isunlimited() else len(self.dimensions[d])) for d in svar.dimensions)))
                for key, val in svar._attributes.items():
                    var._attrs[key] = val.decode() if isinstance(val, bytes) else val
                self.variables[name] = var

    def _write(self):
        with netcdf_file(self._nc_fname, "w", version=2) as dst:
            for key, val in self._nc_attrs.items():
                setattr(dst, key, val)
            for name, dim in self.dimensions.items():
                dst.createDimension(name, None if dim.isunlimited() else dim.size)
            for name, var in self.variables.items():
                dvar = dst.createVariable(name, var._dtype.newbyteorder(">"), var._dims)
                for key, val in var._attrs.items():
                    if val is None:
                        continue
                    if key == "_FillValue":
                        val = np.array(val, dtype=var._dtype.newbyteorder(">"))
                    setattr(dvar, key, val)
                if var._is_record():
                    if var._data.shape[0] > 0:
                        dvar[0: var._data.shape[0]] = var._data
                elif var._data.ndim == 0:
                    dvar.assignValue(var._data[()])
                else:
                    dvar[:] = var._data

    def close(self):
        if self._nc_open and self._nc_mode in ("w", "a", "r+"):
            self._write()
        object.__getattribute__(self, "__dict__")["_nc_open"] = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


# ---- xarray -----------------------------------------------------------------------------------------
class DataArray:
    def __init__(self, data, dims=None, **_):
        self.values = np.array(data, dtype=float, copy=True)
        self.dims = (dims,) if isinstance(dims, str) else (tuple(dims) if dims is not None else ())

    shape = property(lambda self: self.values.shape)

    def copy(self, deep=True):
        return DataArray(self.values, self.dims)


def _operand(other, name):
    return other[name].values if isinstance(other, XrDataset) else other


class XrDataset(dict):
    """ordered mapping name -> DataArray with the elementwise arithmetic the reference uses"""

    def _binary(self, other, fcn):
        res = XrDataset()
        for name, arr in self.items():
            res[name] = DataArray(fcn(arr.values, _operand(other, name)), arr.dims)
        return res

    def _inplace(self, other, fcn):
        for name, arr in self.items():
            arr.values = fcn(arr.values, _operand(other, name))
        return self

    def __neg__(self):
        return self._binary(None, lambda a, b: -a)

    def __add__(self, other):
        return self._binary(other, lambda a, b: a + b)

    def __sub__(self, other):
        return self._binary(other, lambda a, b: a - b)

    def __mul__(self, other):
        return self._binary(other, lambda a, b: a * b)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self._binary(other, lambda a, b: a / b)

    def __rtruediv__(self, other):
        return self._binary(other, lambda a, b: b / a)

    def __iadd__(self, other):
        return self._inplace(other, lambda a, b: a + b)

    def __isub__(self, other):
        return self._inplace(other, lambda a, b: a - b)

    def __imul__(self, other):
        return self._inplace(other, lambda a, b: a * b)

    def __itruediv__(self, other):
        return self._inplace(other, lambda a, b: a / b)

    def copy(self, deep=True):
        res = XrDataset()
        for name, arr in self.items():
            res[name] = arr.copy()
        return res


# ---- pint ---------------------------------------------------------------------------------------------
class _Quantity:
    def __init__(self, magnitude, units):
        self.magnitude = magnitude
        self.units = units

    def to(self, units):
        if units != self.units:
            raise ValueError(f"the pint stand-in cannot convert {self.units} to {units}")
        return self


class _Units:
    """what `ureg(units_str).units` is formatted from: `f"{...:~}"` gives pint's abbreviated expression
    (terms sorted by symbol, powers as `**`), built from this repository's own units parser"""

    def __init__(self, text):
        self.text = text

    def __format__(self, spec):
        import os

        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        if root not in sys.path:
            sys.path.insert(0, root)
        from nk_ooc_amd import hist

        powers = hist._parse_units(self.text)
        sym = {name: hist._PINT_SYMBOL.get(name, name) for name in powers}
        order = sorted(powers, key=lambda name: sym[name])

        def term(name):
            return sym[name] if abs(powers[name]) == 1 else f"{sym[name]} ** {abs(powers[name])}"

        numer = " * ".join(term(name) for name in order if powers[name] > 0) or "1"
        return " / ".join([numer] + [term(name) for name in order if powers[name] < 0])

    def __eq__(self, other):
        return isinstance(other, _Units) and other.text == self.text


class _Parsed(str):
    units = property(lambda self: _Units(str(self)))


class UnitRegistry:
    Quantity = _Quantity

    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, units):
        return _Parsed(units)


def install():
    """register the stand-ins under the names the reference imports (only where the real module is absent)"""
    if "netCDF4" not in sys.modules:
        nc = types.ModuleType("netCDF4")
        nc.Dataset = Dataset
        nc.default_fillvals = dict(DEFAULT_FILLVALS)
        nc.__nk2d_stand_in__ = True
        sys.modules["netCDF4"] = nc
    if "xarray" not in sys.modules:
        xr = types.ModuleType("xarray")
        xr.Dataset = XrDataset
        xr.DataArray = DataArray
        sys.modules["xarray"] = xr
    if "pint" not in sys.modules:
        pint = types.ModuleType("pint")
        pint.UnitRegistry = UnitRegistry
        sys.modules["pint"] = pint
