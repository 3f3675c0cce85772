"""Solver checkpoint: the `<name>_state.json` step log of the out-of-core solver.

Format-compatible with the reference (`nk_ooc/solver_state.py:13-157`): a JSON
object `{"iteration": int, "step_log": [str, ...], <saved values>}` written with
indent 2; ndarrays are stored as `{"__ndarray__": nested list}`; per-iteration
step strings are prefixed with the two-digit iteration (`"03:<step>"`).  A
checkpoint written by the reference resumes here and vice versa.
"""

import functools
import json
import logging
import os

import numpy as np


class _NdarrayEncoder(json.JSONEncoder):
    def default(self, o):
        if isinstance(o, np.ndarray):
            return {"__ndarray__": o.tolist()}
        return super().default(o)


def _ndarray_hook(dct):
    if "__ndarray__" in dct:
        return np.asarray(dct["__ndarray__"])
    return dct


class SolverState:
    """iteration counter + step log + saved values of one iterative solver"""

    def __init__(self, name, workdir, resume=False, rewind=False):
        logger = logging.getLogger(__name__)
        os.makedirs(workdir, exist_ok=True)
        self._name = name
        self._workdir = workdir
        self._state_fname = os.path.join(workdir, f"{name}_state.json")
        self._rewound_step_string = None
        if resume:
            self._read_saved_state()
            if rewind:
                self._rewound_step_string = self._saved_state["step_log"].pop()
                logger.info('rewinding step "%s" for "%s"', self._rewound_step_string, name)
        else:
            if rewind:
                raise RuntimeError(f"rewind cannot be True if resume is False, name={name}")
            self._saved_state = {"iteration": 0, "step_log": []}
            self.log_step("__init__", per_iteration=False)
            logger.info('"%s" iteration now %d', name, self._saved_state["iteration"])

    def get_workdir(self):
        return self._workdir

    def get_iteration(self):
        return self._saved_state["iteration"]

    def inc_iteration(self):
        self._saved_state["iteration"] += 1
        self.log_step("inc_iteration")
        logging.getLogger(__name__).info(
            '"%s" iteration now %d', self._name, self._saved_state["iteration"])
        return self._saved_state["iteration"]

    def _step_string(self, stepval, per_iteration):
        return f"{self.get_iteration():02}:{stepval}" if per_iteration else stepval

    def step_logged(self, stepval, per_iteration=True):
        return self._step_string(stepval, per_iteration) in self._saved_state["step_log"]

    def log_step(self, stepval, per_iteration=True):
        if not self.step_logged(stepval, per_iteration):
            self._saved_state["step_log"].append(self._step_string(stepval, per_iteration))
            self._write_saved_state()

    def step_was_rewound(self, stepval, per_iteration=True):
        if self._rewound_step_string is None:
            return False
        return self._step_string(stepval, per_iteration) == self._rewound_step_string

    def set_value_saved_state(self, key, value):
        """store a value and confirm it survives the JSON round trip exactly"""
        self._saved_state[key] = value
        self._write_saved_state()
        self._read_saved_state()
        reread = self._saved_state[key]
        same = np.array_equal(reread, value) if isinstance(value, np.ndarray) else reread == value
        if not same:
            raise RuntimeError("saved_state value not recovered on reread")

    def get_value_saved_state(self, key):
        return self._saved_state[key]

    def _write_saved_state(self):
        with open(self._state_fname, mode="w") as fptr:
            json.dump(self._saved_state, fptr, indent=2, cls=_NdarrayEncoder)

    def _read_saved_state(self):
        with open(self._state_fname, mode="r") as fptr:
            self._saved_state = json.load(fptr, object_hook=_ndarray_hook)


def action_step_log_wrap(step, per_iteration=True, post_exit=False):
    """decorator: run the wrapped action once per step-log entry.  `solver_state`
    must be passed by keyword; `step` is formatted with the call's keyword arguments."""

    def outer(func):
        @functools.wraps(func)
        def inner(*args, **kwargs):
            solver_state = kwargs["solver_state"]
            label = step.format(**kwargs)
            if solver_state is not None and solver_state.step_logged(label, per_iteration):
                return
            func(*args, **kwargs)
            if solver_state is not None:
                solver_state.log_step(label, per_iteration)
            if post_exit:
                raise SystemExit

        return inner

    return outer
