"""what the Newton and Krylov solvers share: workdir, file naming, tolerances, stats.

Follows `nk_ooc/solver_base.py`: files are `<workdir>/<quantity>_<NN>.nc` (`:50-54`),
tolerances come from `<solver>_rel_tol` / `<solver>_min_iter` of [solverinfo]
(`:56-69`), per-tracer-module statistics go to `<Solver>_stats.nc` (`:71-193`).
"""

import os

from .solver_state import SolverState
from .stats_file import StatsFile


class SolverBase:
    def __init__(self, solver_name, solverinfo, region_cnt, resume, rewind):
        self._solver_name = solver_name
        # configparser sections are case-insensitive (keys such as "Krylov_rel_tol" hit
        # "krylov_rel_tol" in the reference); keep that behaviour for plain dicts too
        self._solverinfo = {str(key).lower(): val for key, val in solverinfo.items()}
        workdir = self._get_workdir()
        os.makedirs(workdir, exist_ok=True)
        self._solver_state = SolverState(solver_name, workdir, resume, rewind)
        self._stats_file = StatsFile(solver_name, workdir, region_cnt, self._solver_state)
        self._stats_vars = {}

    def get_iteration(self):
        return self._solver_state.get_iteration()

    def _get_workdir(self):
        key = f"{self._solver_name}_workdir".lower()
        if key not in self._solverinfo:
            key = "workdir"
        return self._solverinfo[key]

    def _fname(self, quantity, iteration=None):
        if iteration is None:
            iteration = self.get_iteration()
        return os.path.join(self._get_workdir(), f"{quantity}_{iteration:02}.nc")

    def _get_rel_tol(self):
        return float(self._solverinfo[f"{self._solver_name}_rel_tol".lower()])

    def _get_min_iter(self):
        key = f"{self._solver_name}_min_iter".lower()
        return int(self._solverinfo[key]) if key in self._solverinfo else 0

    # ---- solver statistics (<Solver>_stats.nc) ------------------------------------------
    # categories as in the reference: "per_tracer_module" (one variable per module),
    # "model_state" (mean and norm of a model state, per module), "tracer_module_independent"
    def _def_solver_stats_vars(self, stats_vars_dict, tracer_modules):
        vars_def = {}
        for key, metadata in stats_vars_dict.items():
            dims = metadata["dimensions"]
            if "iteration" in dims and dims[0] != "iteration":
                raise ValueError("iteration must be first dimension, if present")
            category = metadata["category"]
            entry = {"category": category, "dimensions": dims}

            def module_var(tms, method=None):
                repl = {"tracer_module_name": tms.name, "tracer_module_units": tms.units,
                        "method": method}
                attrs = {k: v.format(**repl) for k, v in metadata["attrs"].items()}
                if attrs.get("units") == "None":
                    attrs["units"] = None
                return {"dimensions": dims, "attrs": attrs,
                        "datatype": metadata.get("datatype", "f8")}

            if category == "per_tracer_module":
                entry["names"] = []
                for tms in tracer_modules:
                    vars_def[f"{key}_{tms.name}"] = module_var(tms)
                    entry["names"].append(f"{key}_{tms.name}")
            elif category == "model_state":
                entry["names"] = {"mean": [], "norm": []}
                for method in ("mean", "norm"):
                    for tms in tracer_modules:
                        vars_def[f"{key}_{method}_{tms.name}"] = module_var(tms, method)
                        entry["names"][method].append(f"{key}_{method}_{tms.name}")
            elif category == "tracer_module_independent":
                vars_def[key] = {"dimensions": dims, "attrs": dict(metadata["attrs"]),
                                 "datatype": metadata.get("datatype", "f8")}
            else:
                raise ValueError(f"unknown category {category}")
            self._stats_vars[key] = entry
        step = f"define {self._solver_name} solver stats file vars"
        if not self._solver_state.step_logged(step, per_iteration=False):
            self._stats_file.def_vars(vars_def)
        self._solver_state.log_step(step, per_iteration=False)

    def _stats_vals(self, key, vals):
        entry = self._stats_vars[key]
        if entry["category"] == "per_tracer_module":
            return {name: vals[ind] for ind, name in enumerate(entry["names"])}
        if entry["category"] == "model_state":
            out = {}
            for method in ("mean", "norm"):
                reduced = vals.mean() if method == "mean" else vals.norm()
                for ind, name in enumerate(entry["names"][method]):
                    out[name] = reduced[ind]
            return out
        return {key: vals}

    def _put_solver_stats_vars_iteration_independent(self, **kwargs):
        vals_dict = {}
        for key, vals in kwargs.items():
            if "iteration" in self._stats_vars[key]["dimensions"]:
                raise ValueError("use _put_solver_stats_vars for vars with the iteration dimension")
            step = f"write {key} vals to stats file"
            if self._solver_state.step_logged(step, per_iteration=False):
                continue
            vals_dict.update(self._stats_vals(key, vals))
            self._solver_state.log_step(step, per_iteration=False)
        self._stats_file.put_vars_iteration_invariant(vals_dict)

    def _put_solver_stats_vars(self, **kwargs):
        vals_dict = {}
        for key, vals in kwargs.items():
            if "iteration" not in self._stats_vars[key]["dimensions"]:
                raise ValueError("use _put_solver_stats_vars_iteration_independent")
            step = f"write {key} vals to stats file"
            if self._solver_state.step_logged(step):
                continue
            vals_dict.update(self._stats_vals(key, vals))
            self._solver_state.log_step(step)
        self._stats_file.put_vars(self.get_iteration(), vals_dict)
