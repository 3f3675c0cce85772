"""Model configuration: cfg files, tracer-module definitions, region mask and weights.

Reads the SAME inputs as the reference: the configparser cfg files
(`input/py_driver_2d/newton_krylov.cfg`, `model_params.cfg`, overrides; reference
reader `nk_ooc/share.py:97-123`) and the tracer-module YAML
(`input/py_driver_2d/tracer_module_defs.yaml`; reference `nk_ooc/model_config.py:17-125`),
plus `grid_vars.nc` (`region_mask`, `grid_weight`; `nk_ooc/model_config.py:249-289`).
"""

import configparser
import copy
import os

import numpy as np
import yaml

from . import ncio

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_cfg_files(cfg_fnames, repo_root=None, overrides=None, write_cfg_out=False):
    """merge comma-separated cfg files in order, with the reference's interpolation
    variables (HOME, USER, repo_root).  overrides: {section: {key: value}}."""
    defaults = {key: os.environ.get(key, "") for key in ["HOME", "USER"]}
    defaults["repo_root"] = repo_root if repo_root is not None else _PKG_ROOT
    config = configparser.ConfigParser(defaults, allow_no_value=True)
    files_read = config.read(cfg_fnames.split(","))
    if len(files_read) == 0:
        raise RuntimeError(f"cfg_fnames not read: {cfg_fnames}")
    nva = config["DEFAULT"].get("no_value_allowed")
    allowed = nva.split(",") if nva is not None else []
    allowed.append("no_value_allowed")
    for section in config.sections():
        for name in config[section]:
            if config[section][name] is None and name not in allowed:
                raise ValueError(f"{name} not allowed to be empty in cfg files {cfg_fnames}")
    for section, items in (overrides or {}).items():
        for key, val in items.items():
            config[section][key] = val
    if write_cfg_out:
        out_fname = config["solverinfo"].get("cfg_out_fname")
        if out_fname is not None:
            os.makedirs(os.path.dirname(out_fname), exist_ok=True)
            with open(out_fname, "w") as fptr:
                config.write(fptr)
    return config


def _fmt(var, fmt):
    """apply str.format recursively (the `{suff}` expansion of parameterised modules)"""
    if isinstance(var, str):
        return var.format(**fmt)
    if isinstance(var, (list, tuple, set)):
        return type(var)(_fmt(item, fmt) for item in var)
    if isinstance(var, dict):
        return {_fmt(key, fmt): _fmt(val, fmt) for key, val in var.items()}
    return var


def _merge_base_matrix_def(base_def, matrix_def):
    for key, base_val in base_def.items():
        if key not in matrix_def:
            matrix_def[key] = copy.deepcopy(base_val)
        elif isinstance(base_val, list):
            have = [opt.split()[0] for opt in matrix_def[key]]
            for opt in base_val:
                if opt.split()[0] not in have:
                    matrix_def[key].append(opt)
        elif isinstance(base_val, dict):
            for sub in base_val:
                matrix_def[key].setdefault(sub, base_val[sub])
        else:
            raise TypeError(f"base defn type {type(base_val)} not supported")


def propagate_base_matrix_defs_to_all(matrix_defs):
    """copy the entries of the `base` preconditioner-matrix definition into every other
    definition (`nk_ooc/model_config.py:propagate_base_matrix_defs_to_all`): missing keys are
    added, list options are appended unless an option with the same leading word is already
    there, nothing a matrix defines itself is overridden"""
    if "base" not in matrix_defs:
        return
    for name, matrix_def in matrix_defs.items():
        if name != "base":
            _merge_base_matrix_def(matrix_defs["base"], matrix_def)


def gen_grid_vars(grid_vars_fname, region_mask_varname):
    """region_mask, grid_weight (both zeroed where either is), region_cnt"""
    data, _ = ncio.read_file(grid_vars_fname)
    attrs = ncio.read_var_attrs(grid_vars_fname, region_mask_varname)
    words = attrs["cell_measures"].split(":")
    if len(words) != 2:
        raise RuntimeError(f"unexpected number of words in {region_mask_varname}:cell_measures")
    weight_name = words[-1].split()[0]
    mask = np.array(data[region_mask_varname], dtype=np.int32)
    weight = np.array(data[weight_name], dtype=np.float64)
    mask[:] = np.where(weight == 0.0, 0, mask)
    weight[:] = np.where(mask == 0, 0.0, weight)
    return {"region_mask": mask, "grid_weight": weight, "region_cnt": int(mask.max())}


class ModelConfig:
    """tracer-module definitions + grid variables for one run"""

    def __init__(self, modelinfo):
        # own copy: the tracer module names are expanded below, and the caller's dictionary may configure
        # another ModelConfig later (set-up, then the driver)
        modelinfo = dict(modelinfo)
        self.modelinfo = modelinfo
        with open(modelinfo["tracer_module_defs_fname"], mode="r") as fptr:
            contents = yaml.safe_load(fptr)
        self.tracer_module_defs = contents["tracer_module_defs"]
        self.precond_matrix_defs = contents["precond_matrix_defs"]
        self._check_names(modelinfo["tracer_module_names"])
        propagate_base_matrix_defs_to_all(self.precond_matrix_defs)
        modelinfo["tracer_module_names"] = self._expand_all(modelinfo["tracer_module_names"])

        mask_names = set()
        for module_name in modelinfo["tracer_module_names"].split(","):
            module_def = self.tracer_module_defs[module_name]
            for tracer_name, metadata in module_def["tracers"].items():
                if "region_mask_varname" not in metadata:
                    if "region_mask_varname" not in module_def:
                        raise RuntimeError(
                            f"region_mask_varname not known for {tracer_name} in {module_name}")
                    metadata["region_mask_varname"] = module_def["region_mask_varname"]
                mask_names.add(metadata["region_mask_varname"])
        self.grid_vars = {name: gen_grid_vars(modelinfo["grid_vars_fname"], name)
                          for name in mask_names}
        counts = {gv["region_cnt"] for gv in self.grid_vars.values()}
        if len(counts) != 1:
            raise RuntimeError("not all region_masks have the same region_cnt")
        self.region_cnt = counts.pop()

    def _check_names(self, names):
        fmt = {"suff": "suff"}
        for name in names.split(","):
            has_suff = ":" in name
            root = name.partition(":")[0]
            if root not in self.tracer_module_defs:
                raise ValueError(f"unknown tracer module name {root}")
            if has_suff == (root.format(**fmt) == root):
                verb = "doesn't expect" if has_suff else "expects"
                raise ValueError(f"{root} {verb} suff")

    def _expand_all(self, names):
        expanded = []
        for name in names.split(","):
            if ":" not in name:
                expanded.append(name)
                continue
            root, _, suffs = name.partition(":")
            for suff in suffs.split(":"):
                fmt = {"suff": suff}
                new_name = root.format(**fmt)
                root_def = self.tracer_module_defs[root]
                self.tracer_module_defs[new_name] = _fmt(root_def, fmt)
                for metadata in root_def["tracers"].values():
                    if "precond_matrix" in metadata:
                        mname = metadata["precond_matrix"]
                        mname_new = mname.format(**fmt)
                        if mname_new != mname:
                            self.precond_matrix_defs[mname_new] = _fmt(
                                self.precond_matrix_defs[mname], fmt)
                expanded.append(new_name)
        return ",".join(expanded)
