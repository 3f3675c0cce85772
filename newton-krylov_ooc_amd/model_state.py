"""Device-resident model state with the reference's `ModelState` plugin surface.

Mirrors, for the py_driver_2d model, what the solvers call on a model state
(`nk_ooc/model_state_base.py`, `nk_ooc/tracer_module_state_base.py`,
`nk_ooc/py_driver_2d/model_state.py`): construction from a file name (or the
pseudo-files "zeros" / "gen_init_iterate"), `comp_fcn`, `apply_precond_jacobian`,
`gen_precond_jacobian`, `comp_jacobian_fcn_state_prod`, `dump`, `dot_prod` / `norm` /
`mean`, `mod_gram_schmidt`, module-level `lin_comb`, and the arithmetic operators with
float / `ndarray[ntm, nreg]` / state operands, including the step-log idempotence
contract and the NetCDF3 file trail.

The values live in HBM (one `DevVec` per tracer module, one `ModuleEngine` = one HIP
stream per module, optionally on different GPUs); every operation is a kernel launch
through the C ABI.  Files are written at the same points as the reference; a
snapshot cache keyed by file name serves re-opens from HBM instead of re-reading.
"""

import logging
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import hist as hist_mod
from . import ncio
from . import trail
from .engine import (PHOSPHORUS_PARAM_NAMES, Nk2dFrozenMismatch, Nk2dScheduleMismatch, forced_engine, iage_engine,
                     phosphorus_engine)
from .limiter import scalef_for_bound
from .grid import Grid2d, SpatialAxis

YEAR = 365.0 * 86400.0


def _strtobool(val):
    val = str(val).lower()
    if val in ("y", "yes", "t", "true", "on", "1"):
        return True
    if val in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError(f"invalid truth value {val!r}")


def _class_name(obj):
    return f"{obj.__module__}.{type(obj).__name__}"


# side file of a comp_fcn result: the accepted Radau steps of the year that produced it, per tracer module (what the
# perturbed years of the finite-difference products around that result repeat); not part of the reference's file set
SCHED_SUFFIX = ".sched.npz"
_CRC_PREFIX = "__crc__"


def sched_path(fname):
    """where the schedule of the result file `fname` is kept for a resumed run: a hidden directory beside it, so that the
    work directory's own file set is the reference's"""
    return os.path.join(os.path.dirname(os.path.abspath(fname)), ".nk2d", os.path.basename(fname) + SCHED_SUFFIX)


def _crc(host):
    """checksum of a module's values as they are written to / read from the state file"""
    import zlib

    return zlib.crc32(np.ascontiguousarray(host, dtype=np.float64).tobytes())


class TracerModuleState:
    """the tracers of one module, resident on the module's GPU"""

    __array_priority__ = 100

    def __init__(self, name, module_def, engine, vec):
        self.name = name
        self._tracer_module_def = module_def
        self.tracer_names = list(module_def["tracers"])
        self.tracer_cnt = len(self.tracer_names)
        units = {meta.get("attrs", {}).get("units") for meta in module_def["tracers"].values()}
        self.units = units.pop() if len(units) == 1 else None
        self.eng = engine
        self.vec = vec

    def _like(self, vec):
        return TracerModuleState(self.name, self._tracer_module_def, self.eng, vec)

    def copy(self):
        return self._like(self.vec.copy())

    def get_tracer_vals_all(self):
        return self.eng.download(self.vec)

    def log_vals(self, msg, vals):
        logger = logging.getLogger(__name__)
        vals = np.asarray(vals)
        if vals.ndim >= 1 and vals.shape[-1] == 1:
            self.log_vals(msg, vals[..., 0])
            return
        if vals.ndim == 0:
            logger.info("%s[%s]=%e", msg, self.name, vals)
        elif vals.ndim == 1:
            for j in range(vals.shape[0]):
                logger.info("%s[%s,%d]=%e", msg, self.name, j, vals[j])
        elif vals.ndim == 2:
            for i in range(vals.shape[0]):
                for j in range(vals.shape[1]):
                    logger.info("%s[%s,%d,%d]=%e", msg, self.name, i, j, vals[i, j])
        else:
            raise ValueError(f"vals.ndim={vals.ndim} not handled")

    # ---- reductions ---------------------------------------------------------
    def dot_prod(self, other):
        return self.eng.dot(self.vec, other.vec)

    def mean(self):
        return self.eng.dot(self.vec, self.eng_ones())

    def eng_ones(self):
        ones = getattr(self.eng, "_ones_vec", None)
        if ones is None:
            ones = self.eng.upload(np.ones(self.eng.shape))
            self.eng._ones_vec = ones
        return ones

    # ---- arithmetic (new objects) -----------------------------------------------
    def _coef(self, other):
        """float or ndarray[nreg] -> region coefficient array, else None"""
        if isinstance(other, (int, float)):
            return np.full(self.eng.nreg, float(other))
        if isinstance(other, np.ndarray) and other.shape == (self.eng.nreg,):
            return other
        return None

    def __neg__(self):
        return self._like(self.eng.scale(self.vec, -1.0))

    def __add__(self, other):
        if not isinstance(other, TracerModuleState):
            return NotImplemented
        return self._like(self.eng.axpby(1.0, self.vec, 1.0, other.vec))

    def __sub__(self, other):
        if not isinstance(other, TracerModuleState):
            return NotImplemented
        return self._like(self.eng.diff_scale(self.vec, other.vec, 1.0))

    def __mul__(self, other):
        coef = self._coef(other)
        if coef is None:
            return NotImplemented
        return self._like(self.eng.scale(self.vec, coef))

    __rmul__ = __mul__

    def __truediv__(self, other):
        coef = self._coef(other)
        if coef is None:
            return NotImplemented
        return self._like(self.eng.scale(self.vec, 1.0 / coef))

    # ---- arithmetic (in place) -------------------------------------------------------
    def __iadd__(self, other):
        if not isinstance(other, TracerModuleState):
            return NotImplemented
        self.eng.axpby(1.0, self.vec, 1.0, other.vec, out=self.vec)
        return self

    def __isub__(self, other):
        if not isinstance(other, TracerModuleState):
            return NotImplemented
        self.eng.diff_scale(self.vec, other.vec, 1.0, out=self.vec)
        return self

    def __imul__(self, other):
        coef = self._coef(other)
        if coef is None:
            return NotImplemented
        self.eng.scale(self.vec, coef, out=self.vec)
        return self

    def __itruediv__(self, other):
        coef = self._coef(other)
        if coef is None:
            return NotImplemented
        self.eng.scale(self.vec, 1.0 / coef, out=self.vec)
        return self


def _module_engine(name, module_def, grid, device_id, modelinfo):
    """HIP engine for a tracer module of py_driver_2d"""
    py_mod_name = module_def.get("py_mod_name", name)
    kwargs = {}
    if "lin_tol" in modelinfo:
        kwargs["lin_tol"] = float(modelinfo["lin_tol"])
    if py_mod_name == "iage":
        return iage_engine(grid, device_id=device_id, **kwargs)
    if py_mod_name == "forced":
        return forced_engine(grid, modelinfo, device_id=device_id, **kwargs)
    if py_mod_name == "phosphorus":
        params = {key: modelinfo[key] for key in PHOSPHORUS_PARAM_NAMES if key in modelinfo}
        return phosphorus_engine(grid, params, device_id=device_id, **kwargs)
    raise NotImplementedError(
        f"tracer module {name} (py_mod_name={py_mod_name}) has no HIP engine yet")


class ModelState:
    """state of all tracer modules; one engine (GPU stream) per module"""

    __array_priority__ = 100

    model_config_obj = None
    time_range = (0.0, YEAR)
    write_files = True      # keep the reference's NetCDF trail on disk
    device_map = None       # optional {tracer_module_name: device ordinal}
    _engines = None
    _grid = None
    _resident = {}
    _sched_by_name = {}     # fcn file -> {module name: accepted Radau steps of the year that produced it}
    CONCURRENT_MAX_COLUMNS = 768   # (tracer, ypos) columns of all modules together up to which their years run side by side
    RESIDENT_MAX = 256      # device snapshots kept by name; the oldest are dropped beyond this, with or
                            # without the files on disk (write_files=False: a dropped name cannot be re-opened)
    _hist_end = {}          # hist file -> {module name: end-of-year state} of state dependent preconditioners
    _precond_state = {}     # precond file -> {module name: linearisation field}
    hist_writer = hist_mod.HistWriter()      # history files are written on a background thread (hist.HistWriter)
    # off: comp_fcn returns with its history file on disk (the reference's contract, for callers that open the file
    # themselves); the driver mirror and the set-up, which reach history files only through this class, switch it on
    async_hist = os.environ.get("NK2D_ASYNC_HIST", "0") == "1"
    last_stats = None       # stats of the most recent comp_fcn, per module
    last_jvp_mode = None    # "frozen" / "free_running": how the most recent finite-difference product ran its perturbed year

    # ---- class-level set-up (py_driver_2d/model_state.py:44-65) -----------------------
    @classmethod
    def flush_files(cls):
        """every file asked for so far -- history files, the checkpoint trail -- is on disk when this returns"""
        cls.hist_writer.wait()
        trail.flush()

    @classmethod
    def reset_class(cls):
        cls.hist_writer.forget()
        trail.flush()
        if cls._engines:
            for eng in cls._engines.values():
                eng.close()
        cls._engines = None
        cls._grid = None
        cls._resident = {}
        cls._sched_by_name = {}
        cls._hist_end = {}
        cls._precond_state = {}
        cls.model_config_obj = None

    @classmethod
    def _set_class_vars(cls):
        if cls._engines is not None:
            return
        cfg = cls.model_config_obj
        if cfg is None:
            raise RuntimeError("ModelState.model_config_obj is None")
        modelinfo = cfg.modelinfo
        data, _ = ncio.read_file(modelinfo["grid_vars_fname"])
        axes = []
        for key in ("depth_axisname", "ypos_axisname"):
            axisname = modelinfo[key]
            edges_name = f"{axisname}_edges"
            units = ncio.read_var_attrs(modelinfo["grid_vars_fname"], edges_name).get("units")
            axes.append(SpatialAxis(axisname, data[edges_name], units))
        cls.depth, cls.ypos = axes
        cls._grid = Grid2d(cls.depth, cls.ypos, float(modelinfo["max_abs_vvel"]),
                           float(modelinfo["horiz_mix_coeff"]))
        cls._engines = {}
        for name in modelinfo["tracer_module_names"].split(","):
            module_def = cfg.tracer_module_defs[name]
            device_id = (cls.device_map or {}).get(name, 0)
            eng = _module_engine(name, module_def, cls._grid, device_id, modelinfo)
            mask_name = next(iter(module_def["tracers"].values()))["region_mask_varname"]
            gv = cfg.grid_vars[mask_name]
            eng.set_region(gv["region_mask"], gv["grid_weight"])
            cls._engines[name] = eng

    # ---- construction ------------------------------------------------------------------
    def __init__(self, fname, _modules=None):
        self._set_class_vars()
        cfg = self.model_config_obj
        names = cfg.modelinfo["tracer_module_names"].split(",")
        self.tracer_modules = np.empty(len(names), dtype=object)
        if _modules is not None:
            for ind, tms in enumerate(_modules):
                self.tracer_modules[ind] = tms
            return
        cached = self._resident.get(os.path.abspath(fname)) if isinstance(fname, str) else None
        # the accepted steps of the forward year that produced this file, if it is the result of one (comp_fcn):
        # remembered by name in this process (dump() forgets them when the name is written again), and kept next to the
        # file for a resumed run -- with a checksum of the values they belong to, verified against what the file holds now
        side = None
        if isinstance(fname, str):
            if cached is None and fname not in ("zeros", "gen_init_iterate"):
                trail.flush()       # (a file of this process' own trail may still be on its way to the disk)
            self._sched = self._sched_by_name.get(os.path.abspath(fname))
            if self._sched is None and cached is None and os.path.exists(sched_path(fname)):
                with np.load(sched_path(fname)) as data:
                    side = {key: data[key] for key in data.files}
        for ind, name in enumerate(names):
            module_def = cfg.tracer_module_defs[name]
            eng = self._engines[name]
            if cached is not None:
                vec = cached[ind].copy()
            else:
                host = self._load_host(fname, module_def, eng)
                if side is not None and int(side.get(_CRC_PREFIX + name, -1)) != _crc(host):
                    logging.getLogger(__name__).warning(
                        "%s: the schedule side file does not belong to these values (checksum of %s) -- ignored",
                        fname, name)
                    side = None
                vec = eng.upload(host)
            self.tracer_modules[ind] = TracerModuleState(name, module_def, eng, vec)
        if side is not None:
            self._sched = {key: val for key, val in side.items() if not key.startswith(_CRC_PREFIX)}

    def _load_host(self, fname, module_def, eng):
        """(tc, nz, ny) host values of one module from a file or pseudo-file
        (py_driver_2d/tracer_module_state.py:30-69)"""
        shape = (len(self.depth), len(self.ypos))
        tracers = module_def["tracers"]
        if fname == "zeros":
            return np.zeros((len(tracers),) + shape)
        if fname == "gen_init_iterate":
            vals = []
            for tracer_name, metadata in tracers.items():
                src = metadata
                if "init_iterate_vals" not in metadata:
                    if "shadows" not in metadata:
                        raise ValueError(f"gen_init_iterate failure for {tracer_name}")
                    src = tracers[metadata["shadows"]]
                column = np.interp(self.depth.mid, src["init_iterate_val_depths"],
                                   src["init_iterate_vals"])
                vals.append(np.broadcast_to(column[:, np.newaxis], shape))
            return np.stack(vals)
        if not self.write_files and not os.path.exists(fname):
            raise FileNotFoundError(
                f"{fname}: not on disk (ModelState.write_files is False) and its device snapshot was "
                f"dropped from the resident cache (RESIDENT_MAX={self.RESIDENT_MAX})")
        data, _ = ncio.read_file(fname, list(tracers))
        for tracer_name in tracers:
            if data[tracer_name].shape != shape:
                raise ValueError(f"unexpected dimension lengths for {tracer_name} in {fname}")
        return np.stack([data[name] for name in tracers])

    def _new(self, modules):
        return type(self)(None, _modules=modules)

    def copy(self):
        return self._new([tms.copy() for tms in self.tracer_modules])

    # ---- files ----------------------------------------------------------------------------
    def dump(self, fname, caller=None):
        """write the state to a NetCDF3 file (model_state_base.py:93-111) and remember a
        device snapshot under that name"""
        if fname is None:
            return self
        if caller is None:
            raise ValueError("caller unknown")
        # device snapshot under the file's name: a later ModelState(fname) of this process skips the
        # file read.  Oldest snapshots are dropped beyond RESIDENT_MAX whether or not the files are
        # written: the cache is bounded (a Krylov solve re-opens only names of its own iteration range).
        cache = self._resident
        # whatever schedule was known under this name belonged to the values it held before
        self._sched_by_name.pop(os.path.abspath(fname), None)
        if self.write_files:
            side_file = sched_path(fname)

            def drop_side_file():
                if os.path.exists(side_file):
                    os.remove(side_file)

            trail.submit(drop_side_file)
        cache.pop(os.path.abspath(fname), None)
        cache[os.path.abspath(fname)] = [tms.vec.copy() for tms in self.tracer_modules]
        while len(cache) > self.RESIDENT_MAX:
            cache.pop(next(iter(cache)))
        if self.write_files:
            history = ncio.history_stamp(f"{_class_name(self)}.dump", caller)
            axes = [self.depth, self.ypos]
            names = [tms.tracer_names for tms in self.tracer_modules]
            if trail.TRAIL.enabled and all(hasattr(tms.eng, "download_begin") for tms in self.tracer_modules):
                # the trail's writer thread: the copies to the host are queued on the modules' streams behind what
                # produced the values (and ahead of whatever changes them next), the thread waits for them and writes
                pending = [tms.eng.download_begin(tms.vec) for tms in self.tracer_modules]

                def write():
                    tracer_vals = {}
                    for module_names, download in zip(names, pending):
                        host = download.result()
                        for ind, tracer_name in enumerate(module_names):
                            tracer_vals[tracer_name] = host[ind]
                    ncio.write_state_file(fname, axes, tracer_vals, history)

                trail.submit(write)
            else:
                tracer_vals = {}
                for tms in self.tracer_modules:
                    host = tms.get_tracer_vals_all()
                    for ind, tracer_name in enumerate(tms.tracer_names):
                        tracer_vals[tracer_name] = host[ind]
                trail.submit(lambda: ncio.write_state_file(fname, axes, tracer_vals, history))
        return self

    # ---- logging ----------------------------------------------------------------------------
    def log_vals(self, msg, vals):
        for ind, tms in enumerate(self.tracer_modules):
            if isinstance(msg, list):
                for msg_ind, submsg in enumerate(msg):
                    tms.log_vals(submsg, vals[msg_ind, ind, ...])
            else:
                tms.log_vals(msg, vals[ind, ...])

    def log(self, msg=None):
        msg_full = ["mean", "norm"] if msg is None else [f"{msg},mean", f"{msg},norm"]
        self.log_vals(msg_full, np.stack((self.mean(), self.norm())))

    # ---- reductions ---------------------------------------------------------------------------
    def _per_module(self, fcn):
        res = np.empty((len(self.tracer_modules), self.model_config_obj.region_cnt))
        for ind, tms in enumerate(self.tracer_modules):
            res[ind, :] = fcn(ind, tms)
        return res

    def mean(self):
        return self._per_module(lambda ind, tms: tms.mean())

    def dot_prod(self, other):
        return self._per_module(lambda ind, tms: tms.dot_prod(other.tracer_modules[ind]))

    def norm(self):
        return np.sqrt(self.dot_prod(self))

    def mgs_against(self, basis):
        """in-place modified Gram-Schmidt of self against the given (resident) states;
        returns the projection coefficients [ntm, len(basis), nreg].  Per tracer module the
        len(basis) dot / axpy pairs run back to back on the device (nk2d_mgs)."""
        h_val = np.empty((len(self.tracer_modules), len(basis), self.model_config_obj.region_cnt))
        for ind, tms in enumerate(self.tracer_modules):
            h_val[ind] = tms.eng.mgs(tms.vec, [b.tracer_modules[ind].vec for b in basis])
        return h_val

    def mod_gram_schmidt(self, basis_cnt, fname_fcn, quantity):
        """file-name flavour of `mgs_against` (reference signature, model_state_base.py:365-377)"""
        return self.mgs_against([type(self)(fname_fcn(quantity, i_val)) for i_val in range(basis_cnt)])

    @classmethod
    def lin_comb_of(cls, coeff, states):
        """sum_j coeff[:, j, :] * states[j], accumulated in order, one launch per module"""
        mods = []
        for ind, tms in enumerate(states[0].tracer_modules):
            vecs = [state.tracer_modules[ind].vec for state in states]
            mods.append(tms._like(tms.eng.lin_comb(vecs, coeff[ind])))
        return states[0]._new(mods)

    # ---- arithmetic -------------------------------------------------------------------------------
    def _binary(self, other, op):
        if isinstance(other, ModelState):
            mods = [op(a, b) for a, b in zip(self.tracer_modules, other.tracer_modules)]
        elif isinstance(other, float):
            mods = [op(a, other) for a in self.tracer_modules]
        elif isinstance(other, np.ndarray) and other.shape[0] == len(self.tracer_modules):
            mods = [op(a, other[ind, ...]) for ind, a in enumerate(self.tracer_modules)]
        else:
            return NotImplemented
        if any(m is NotImplemented for m in mods):
            return NotImplemented
        return self._new(mods)

    def _inplace(self, other, op):
        for ind, tms in enumerate(self.tracer_modules):
            if isinstance(other, ModelState):
                arg = other.tracer_modules[ind]
            elif isinstance(other, float):
                arg = other
            elif isinstance(other, np.ndarray) and other.shape[0] == len(self.tracer_modules):
                arg = other[ind, ...]
            else:
                return NotImplemented
            self.tracer_modules[ind] = op(tms, arg)
        return self

    def __neg__(self):
        return self._new([-tms for tms in self.tracer_modules])

    def __add__(self, other):
        return self._binary(other, lambda a, b: a + b) if isinstance(other, ModelState) else NotImplemented

    __radd__ = __add__

    def __sub__(self, other):
        return self._binary(other, lambda a, b: a - b) if isinstance(other, ModelState) else NotImplemented

    def __mul__(self, other):
        return self._binary(other, lambda a, b: a * b)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self._binary(other, lambda a, b: a / b)

    def __iadd__(self, other):
        if not isinstance(other, ModelState):
            return NotImplemented
        return self._inplace(other, lambda a, b: a.__iadd__(b))

    def __isub__(self, other):
        if not isinstance(other, ModelState):
            return NotImplemented
        return self._inplace(other, lambda a, b: a.__isub__(b))

    def __imul__(self, other):
        return self._inplace(other, lambda a, b: a.__imul__(b))

    def __itruediv__(self, other):
        return self._inplace(other, lambda a, b: a.__itruediv__(b))

    # ---- the function whose root is sought ------------------------------------------------------------
    def comp_fcn(self, res_fname, solver_state, hist_fname=None, frozen=None):
        """one forward model year per tracer module on its GPU: F(x) = y(T) - x
        (py_driver_2d/model_state.py:67-139).  The result carries the accepted Radau steps of every module's year
        (`_sched`); `frozen`: such schedules of another year, {module name: schedule} -- this year then repeats
        those steps instead of choosing its own (the perturbed year of a finite-difference product,
        comp_jacobian_fcn_state_prod)."""
        logger = logging.getLogger(__name__)
        fcn_complete_step = f"comp_fcn complete for {res_fname}"
        if solver_state is not None and solver_state.step_logged(fcn_complete_step):
            logger.debug('"%s" logged, returning result', fcn_complete_step)
            return type(self)(res_fname)
        mods, stats, hists, scheds = [], [], [], {}
        t_eval = np.linspace(self.time_range[0], self.time_range[1], 61)

        def forward_year(tms):
            if frozen is not None and hist_fname is None and len(frozen.get(tms.name, ())) > 0:
                try:
                    fx, st = tms.eng.comp_fcn_frozen(tms.vec, frozen[tms.name])
                    return fx, st, None
                except (Nk2dFrozenMismatch, Nk2dScheduleMismatch) as msg:
                    # the recorded Newton iteration counts are not enough for this state even after the resumes, its error
                    # estimates are out of bounds, or the schedule is not this engine's: a free-running year
                    logger.warning("%s: %s -- free-running year instead", tms.name, msg)
            if hist_fname is None:
                return tms.eng.comp_fcn(tms.vec)
            return tms.eng.comp_fcn_hist(tms.vec, t_eval)

        # the modules are independent (own context, own HIP stream, own host control loop): where their years
        # together leave the chip room -- small grids: a year is a chain of latency-bound phases on a few dozen
        # waves -- they run concurrently, one host thread each (the ctypes calls release the GIL).  Where every
        # module fills the chip on its own (a wave per (tracer, ypos) column holds a SIMD: 1 024 of them), years
        # side by side only take SIMDs from each other -- a resident one-launch year holds its SIMDs for its whole
        # length while another module's launches queue for the rest (round 3: 0.94 s per iteration of the three-module
        # mix at 416 x 416 against 0.70 s for its three years one after the other) -- so they run back to back.
        # NK2D_SERIAL_MODULES=1 / 0 forces one or the other.
        serial_env = os.environ.get("NK2D_SERIAL_MODULES")
        columns = sum(tms.eng.tc * tms.eng.ny for tms in self.tracer_modules)
        serial = (serial_env not in (None, "", "0")) if serial_env is not None else columns > self.CONCURRENT_MAX_COLUMNS
        if len(self.tracer_modules) > 1 and not serial:
            with ThreadPoolExecutor(max_workers=len(self.tracer_modules)) as pool:
                years = list(pool.map(forward_year, self.tracer_modules))
        else:
            years = [forward_year(tms) for tms in self.tracer_modules]
        for tms, (fx, st, hist) in zip(self.tracer_modules, years):
            if hist_fname is not None:
                tracers = {name: dict(meta.get("attrs", {}))
                           for name, meta in tms._tracer_module_def["tracers"].items()}
                if tms.eng.module_kind == 1:
                    # tracer-like history variable of the module (phosphorus.py:174-195) and the
                    # end-of-year po4 its preconditioner linearises about
                    po4_units = tracers["po4"].get("units", "1")
                    tracers["po4_uptake"] = {"long_name": "uptake of po4", "units": f"{po4_units} / s"}
                    uptake = tms.eng.po4_uptake(hist[:, 0])
                    hist = np.concatenate((hist, uptake[:, np.newaxis]), axis=1)
                    type(self)._hist_end.setdefault(os.path.abspath(hist_fname), {})[tms.name] = hist[-1, 0].copy()
                elif tms.eng.state_dependent_precond:
                    # forced module with a sink threshold: the tracer at the end of each third of the year
                    type(self)._hist_end.setdefault(os.path.abspath(hist_fname), {})[tms.name] = \
                        hist[self._third_end_indices(t_eval), 0].copy()
                hists.append((tracers, hist))
            mods.append(tms._like(fx))
            stats.append(st)
            if frozen is None:
                scheds[tms.name] = tms.eng.last_schedule()
        if hist_fname is not None and self.write_files:
            # the mixing coefficient samples come from the engine (not thread safe): here; the file itself -- reductions,
            # byte order, 423 MB at 416 x 416 -- on the writer's thread, with the stamp of now
            eng0 = self.tracer_modules[0].eng
            vmix_samples = np.stack([eng0.vmix_coeff(t) for t in t_eval])
            self.hist_writer.submit(hist_fname, self._grid, t_eval, hists, vmix_samples, hist_mod.hist_stamp(),
                                    background=self.async_hist)
        type(self).last_stats = stats
        res_ms = self._new(mods)
        res_ms._sched = scheds if frozen is None else None
        # zero_extra_tracers: no shadow tracers in the py_driver_2d modules handled here;
        # apply_region_mask is fused into the kernel that forms y(T) - x
        caller = f"{_class_name(self)}.comp_fcn_postprocess called from {_class_name(self)}.comp_fcn"
        res_ms.dump(res_fname, caller)       # (forgets any schedule known under this name)
        if res_fname is not None and res_ms._sched:
            by_name = type(self)._sched_by_name
            by_name[os.path.abspath(res_fname)] = res_ms._sched
            while len(by_name) > self.RESIDENT_MAX:
                by_name.pop(next(iter(by_name)))
            if self.write_files:
                crcs = {_CRC_PREFIX + tms.name: np.int64(_crc(tms.get_tracer_vals_all())) for tms in res_ms.tracer_modules}
                side_file, side_vals = sched_path(res_fname), dict(res_ms._sched, **crcs)

                def write_side_file():
                    os.makedirs(os.path.dirname(side_file), exist_ok=True)
                    np.savez(side_file[:-4], **side_vals)

                trail.submit(write_side_file)
        if solver_state is not None:
            solver_state.log_step(fcn_complete_step)
            modelinfo = self.model_config_obj.modelinfo
            if _strtobool(modelinfo.get("reinvoke", "False")):
                self.flush_files()        # the process that resumes reads what this one wrote
                cmd = [modelinfo["invoker_script_fname"], "--resume"]
                logger.info('cmd="%s"', " ".join(cmd))
                subprocess.Popen(cmd)
                raise SystemExit
        return res_ms

    def apply_region_mask(self):
        for tms in self.tracer_modules:
            tms.eng.apply_region_mask(tms.vec)
        return self

    # ---- preconditioner ------------------------------------------------------------------------------------
    def hist_vars_for_precond_list(self):
        """history variables the preconditioners in use need (model_state_base.py:379-389)"""
        defs = self.model_config_obj.precond_matrix_defs
        names = []
        for tms in self.tracer_modules:
            for metadata in tms._tracer_module_def["tracers"].values():
                if "precond_matrix" in metadata and metadata["precond_matrix"] not in names:
                    names.append(metadata["precond_matrix"])
        res = []
        for matrix_name in names + ["base"]:
            for varname in defs[matrix_name]["hist_to_precond_varnames"]:
                if varname not in res:
                    res.append(varname)
        return res

    def gen_precond_jacobian(self, hist_fname, precond_fname, solver_state):
        """file(s) the preconditioner reads: the listed history variables, optionally reduced
        over time (`:mean`, `:log_mean`) (model_state_base.py:404-481).  Without a history file
        (Krylov solver used on its own) only the `time` axis is written."""
        step = f"ModelStateBase.gen_precond_jacobian {precond_fname}"
        if solver_state is not None and solver_state.step_logged(step, per_iteration=False):
            return
        if hist_fname is not None:
            ends = self._hist_end.get(os.path.abspath(hist_fname))
            if ends:
                type(self)._precond_state[os.path.abspath(precond_fname)] = dict(ends)
        if self.write_files:
            time_attrs = {"long_name": "time", "units": "seconds since 0001-01-01", "calendar": "noleap"}
            dims, variables = {}, {}
            history = ncio.history_stamp(f"{_class_name(self)}.gen_precond_jacobian")
            rec = self.hist_writer.record(hist_fname)
            wanted = self.hist_vars_for_precond_list() if hist_fname is not None else []
            if rec is not None and wanted == ["time"]:
                # all the preconditioner file takes from this history is its time axis: from the record in memory, with
                # the attributes the history file gives it (the file itself may still be on its way to the disk)
                dims["time"] = len(rec["time"])
                variables["time"] = (("time",), ">f8", time_attrs, rec["time"])
                history = "\n".join([history, rec.get("stamp", "")]) if rec.get("stamp") else history
            elif hist_fname is not None and (self.hist_writer.wait(hist_fname) or os.path.exists(hist_fname)):
                wanted = self.hist_vars_for_precond_list()
                data, attrs = ncio.read_file(hist_fname, [name.partition(":")[0] for name in wanted])
                if "history" in attrs:
                    history = "\n".join([history, attrs["history"]])
                var_dims = ncio.read_var_dims(hist_fname, list(data))
                for spec in wanted:
                    name, _, time_op = spec.partition(":")
                    vals, vdims = data[name], list(var_dims[name])
                    vattrs = ncio.read_var_attrs(hist_fname, name)
                    if time_op == "mean":
                        vals, vdims, name = vals.mean(axis=0), vdims[1:], f"{name}_mean"
                    elif time_op == "log_mean":
                        vals, vdims, name = np.exp(np.log(vals).mean(axis=0)), vdims[1:], f"{name}_log_mean"
                    if "time" not in vdims:
                        vattrs.pop("cell_methods", None)
                    for dimname, dimlen in zip(vdims, vals.shape):
                        dims.setdefault(dimname, dimlen)
                    variables[name] = (tuple(vdims), ">f8", vattrs, vals)
            else:
                if hist_fname is not None and wanted != ["time"]:
                    # (round-3 ADVICE) the history is written behind the step log: a process killed in that window resumes
                    # with later steps logged and no file.  The time axis alone can be restated; history VARIABLES cannot
                    raise RuntimeError(f"gen_precond_jacobian: history file {hist_fname} is missing although its forward year "
                                       "is logged as complete (the run ended while the file was being written?): remove "
                                       f"'{step}' and the comp_fcn step before it from the solver state, or rerun the "
                                       "Newton iteration (--rewind), so that the year is integrated again")
                time = np.linspace(self.time_range[0], self.time_range[1], 61)
                dims["time"] = len(time)
                variables["time"] = (("time",), ">f8", time_attrs, time)
            ncio.write_vars_file(precond_fname, dims, variables, history)
        if solver_state is not None:
            solver_state.log_step(step, per_iteration=False)

    def apply_precond_jacobian(self, precond_fname, res_fname, solver_state):
        """M^-1 self, module by module (py_driver_2d/model_state.py:235-270)"""
        logger = logging.getLogger(__name__)
        fcn_complete_step = f"apply_precond_jacobian complete for {res_fname}"
        if solver_state is not None and solver_state.step_logged(fcn_complete_step):
            logger.debug('"%s" logged, returning result', fcn_complete_step)
            return type(self)(res_fname)
        for tms in self.tracer_modules:
            if tms.eng.module_kind == 1:
                self._ensure_state_precond(tms, precond_fname)
            elif tms.eng.state_dependent_precond:
                self._ensure_forced_precond(tms, precond_fname)
        res_ms = self._new([tms._like(tms.eng.precond_apply(tms.vec)) for tms in self.tracer_modules])
        # the file first, then the step that names it.  (The reference logs the step and THEN dumps, model_state.py:266-270: a run
        # killed in between has the step on record and no file, and its resumed run fails on the missing file.  The same files
        # and the same step log come out; tests/test_gpu_trail.py kills a solve at this point among others.)
        res_ms.dump(res_fname, f"{_class_name(self)}.apply_precond_jacobian")
        if solver_state is not None:
            solver_state.log_step(fcn_complete_step)
        return res_ms

    def _third_end_indices(self, time_vals):
        """samples of the precond file closest to the end of each third of the year (forced.py:222-233)"""
        t0, t1 = self.time_range
        return [int(np.argmin(abs(t0 + (k + 1.0) * (t1 - t0) / 3 - np.asarray(time_vals)))) for k in range(3)]

    def _ensure_forced_precond(self, tms, precond_fname):
        """preconditioner of a forced module whose Jacobian depends on the tracer (file source with a
        sink threshold): factorised once per precond file from the tracer at three times"""
        key = os.path.abspath(precond_fname)
        if getattr(tms.eng, "_state_precond_key", None) == key:
            return
        fields = self._precond_state.get(key, {}).get(tms.name)
        if fields is None:
            # resumed run: read the samples back from the precond file
            name = tms.tracer_names[0]
            data, _ = ncio.read_file(precond_fname, ["time", name])
            fields = data[name][self._third_end_indices(data["time"])]
        states = [tms.eng.upload(np.asarray(f, dtype=np.float64)[np.newaxis]) for f in fields]
        tms.eng.precond_setup_states(states)
        tms.eng._state_precond_key = key

    def _ensure_state_precond(self, tms, precond_fname):
        """factorise the state dependent (phosphorus) preconditioner of `precond_fname` once; the
        reference redoes its eigs + two sparse LU for every application (phosphorus.py:197-274)"""
        key = os.path.abspath(precond_fname)
        if getattr(tms.eng, "_state_precond_key", None) == key:
            return
        field = self._precond_state.get(key, {}).get(tms.name)
        if field is None:
            # resumed run: po4 at the sample closest to the end of the year (phosphorus.py:222-224)
            data, _ = ncio.read_file(precond_fname, ["time", "po4"])
            field = data["po4"][np.argmin(abs(self.time_range[1] - data["time"]))]
        pc = tms.eng.precond_setup_state(field, self.time_range)
        tms.eng._state_precond_key = key
        logger = logging.getLogger(__name__)
        for ind, val in enumerate(pc.e_vals[:5]):
            logger.info("small e_val[%d] = %e + %e j", ind, val.real, val.imag)
        if self.write_files:
            # null vector scaled to unit mean, as the reference leaves it next to the precond file
            e_vect = tms.eng.download(pc.e_hat)
            tracer_vals = {name: e_vect[ind] for ind, name in enumerate(tms.tracer_names)}
            ncio.write_state_file(os.path.join(os.path.dirname(key), "precond_null_space.nc"),
                                  [self.depth, self.ypos], tracer_vals,
                                  ncio.history_stamp(f"{_class_name(self)}.apply_precond_jacobian"))

    # ---- finite-difference Jacobian-vector product ---------------------------------------------------------
    def comp_jacobian_fcn_state_prod(self, fcn, direction, res_fname, solver_state):
        """(F(self + sigma d) - F(self)) / sigma with sigma = 1e-4 |self| per
        (module, region); assumes |d| = 1 (model_state_base.py:492-527)"""
        logger = logging.getLogger(__name__)
        fcn_complete_step = f"comp_jacobian_fcn_state_prod complete for {res_fname}"
        if solver_state.step_logged(fcn_complete_step):
            logger.debug('"%s" logged, returning result', fcn_complete_step)
            return type(self)(res_fname)
        sigma = 1.0e-4 * self.norm()
        sigma = np.where(sigma == 0.0, 1.0, sigma)
        perturb_ms = self + sigma * direction
        perturb_fcn_fname = os.path.join(
            solver_state.get_workdir(), f"perturb_fcn_{os.path.basename(res_fname)}")
        # Internal numerical differentiation: the perturbed year repeats the accepted steps of the year that produced
        # `fcn` (carried by it when it was computed in this process), so that the quotient below differentiates ONE
        # discrete map.  Two free-running years take different controller decisions here and there, and the
        # difference of their discretisation errors over sigma is 5 ... 90 % of the product (DESIGN.md section 3.5,
        # tools/probe_jvp_noise.py).  NK2D_JVP_FROZEN=0, or an `fcn` read back from a file, gives free-running years.
        frozen = getattr(fcn, "_sched", None) if os.environ.get("NK2D_JVP_FROZEN", "1") != "0" else None
        mode = ("frozen controller: the perturbed year repeats the accepted steps of the year behind F(x)" if frozen
                else "two free-running years (the reference's product)")
        logger.info("comp_jacobian_fcn_state_prod mode: %s", mode)
        type(self).last_jvp_mode = "frozen" if frozen else "free_running"
        perturb_fcn = perturb_ms.comp_fcn(perturb_fcn_fname, solver_state, frozen=frozen or None)
        caller = f"{_class_name(self)}.comp_jacobian_fcn_state_prod"
        res = ((perturb_fcn - fcn) / sigma).dump(res_fname, caller)
        solver_state.log_step(fcn_complete_step)
        return res


    # ---- what the Newton solver needs beyond the Krylov path ------------------------------------------
    def shadow_tracers_on(self):
        return False  # no py_driver_2d tracer module declares shadow tracers

    def copy_shadow_tracers_to_real_tracers(self):
        return self

    def copy_real_tracers_to_shadow_tracers(self):
        return self

    def apply_limiter(self, base):
        """scale self (an increment) so that base + scalef * self respects the tracer bounds,
        per region; returns scalef [ntm, nreg] (tracer_module_state_base.py:112-151 with
        utils.py:562-600).  Host side: once per Newton iteration, not on the hot path."""
        nreg = self.model_config_obj.region_cnt
        scalef = np.ones((len(self.tracer_modules), nreg))
        for ind, tms in enumerate(self.tracer_modules):
            module_def = tms._tracer_module_def
            bounded = "bounds" in module_def or any("bounds" in m for m in module_def["tracers"].values())
            if not bounded:
                continue
            inc = tms.get_tracer_vals_all()
            ref = base.tracer_modules[ind].get_tracer_vals_all()
            for tr, (tname, metadata) in enumerate(module_def["tracers"].items()):
                lob, upb = None, None
                for src in (module_def, metadata):
                    if "bounds" in src:
                        lob = src["bounds"].get("lob", lob)
                        upb = src["bounds"].get("upb", upb)
                mask = self.model_config_obj.grid_vars[metadata["region_mask_varname"]]["region_mask"]
                for bound, upper in ((lob, False), (upb, True)):
                    np.minimum(scalef[ind], scalef_for_bound(nreg, mask, ref[tr], inc[tr], bound, upper), out=scalef[ind])
            if (scalef[ind] < 1.0).any():
                tms.log_vals("applying scalef", scalef[ind])
                tms *= scalef[ind]
        return scalef

    # model-specific statistics read from a history file (model_state_base.py:138-177,
    # py_driver_2d/tracer_module_state.py:280-341)
    def def_stats_vars(self, stats_file, hist_fname, solver_state):
        step = "ModelStateBase.def_stats_vars"
        if solver_state is not None and solver_state.step_logged(step, per_iteration=False):
            return
        dims, vars_metadata = {}, {}
        for axis in (self.depth, self.ypos):
            dims.update(axis.dump_dimensions())
            vars_metadata.update(axis.dump_vars_metadata())
        names = [tname for tms in self.tracer_modules for tname in tms.tracer_names]
        mem = self.hist_writer.tracer_samples(hist_fname, names)
        if mem is None:
            self.hist_writer.wait(hist_fname)
        for tms in self.tracer_modules:
            for tname in tms.tracer_names:
                attrs = dict(mem[tname][1]) if mem is not None else ncio.read_var_attrs(hist_fname, tname)
                attrs.pop("cell_methods", None)
                vars_metadata[tname] = {
                    "dimensions": ("iteration", self.depth.axisname, self.ypos.axisname), "attrs": attrs}
                vars_metadata[f"{tname}_mean_{self.ypos.axisname}"] = {
                    "dimensions": ("iteration", self.depth.axisname), "attrs": attrs}
        stats_file.def_dimensions(dims)
        stats_file.def_vars(vars_metadata)
        if solver_state is not None:
            solver_state.log_step(step, per_iteration=False)

    def put_stats_vars_iteration_invariant(self, stats_file, hist_fname, solver_state):
        step = "ModelStateBase.put_stats_vars_iteration_invariant"
        if solver_state is not None and solver_state.step_logged(step, per_iteration=False):
            return
        vals = {}
        for axis in (self.depth, self.ypos):
            vals.update(axis.dump_vals_dict())
        stats_file.put_vars_iteration_invariant(vals)
        if solver_state is not None:
            solver_state.log_step(step, per_iteration=False)

    def put_stats_vars(self, stats_file, hist_fname, solver_state):
        step = "ModelStateBase.put_stats_vars"
        if solver_state is not None and solver_state.step_logged(step):
            return
        names = [tname for tms in self.tracer_modules for tname in tms.tracer_names]
        mem = self.hist_writer.tracer_samples(hist_fname, names)
        if mem is not None:
            data = {tname: vals for tname, (vals, _) in mem.items()}     # the samples the file is being written from
        else:
            self.hist_writer.wait(hist_fname)
            data, _ = ncio.read_file(hist_fname, names)
        weights = hist_mod.time_mean_weights(next(iter(data.values())).shape[0])
        ypos_weights = self.ypos.delta / self.ypos.delta.sum()
        vals = {}
        for tname in names:
            vals[tname] = np.einsum("i,i...", weights, data[tname])
            vals[f"{tname}_mean_{self.ypos.axisname}"] = np.einsum("j,...j", ypos_weights, vals[tname])
        stats_file.put_vars(solver_state.get_iteration(), vals)
        solver_state.log_step(step)


def lin_comb(res_type, coeff, fname_fcn, quantity):
    """file-name flavour of `ModelState.lin_comb_of` (reference signature,
    model_state_base.py:619-624)"""
    return res_type.lin_comb_of(
        coeff, [res_type(fname_fcn(quantity, j_val)) for j_val in range(coeff.shape[-2])])
