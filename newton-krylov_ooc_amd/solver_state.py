"""Solver checkpoint: the `<name>_state.json` step log of the out-of-core solver.

Format-compatible with the reference (`nk_ooc/solver_state.py:13-157`): a JSON
object `{"iteration": int, "step_log": [str, ...], <saved values>}` written with
indent 2; ndarrays are stored as `{"__ndarray__": nested list}`; per-iteration
step strings are prefixed with the two-digit iteration (`"03:<step>"`).  A
checkpoint written by the reference resumes here and vice versa.

The file is rewritten after every change (a few hundred bytes) -- the run can be killed
between any two solver actions and resumed from the last completed one.  The write is a job of
the checkpoint trail (`trail.py`): at once by default, on the trail's writer thread behind the
files the logged steps stand for where the driver switched that on.
"""

import functools
import json
import logging
import os
import pickle

import numpy as np

from . import trail

LOG = logging.getLogger(__name__)
NDARRAY_TAG = "__ndarray__"


def _to_json(obj):
    """json.dump `default=` hook: the only non-JSON type stored is the ndarray"""
    if isinstance(obj, np.ndarray):
        return {NDARRAY_TAG: obj.tolist()}
    raise TypeError(f"{type(obj).__name__} is not JSON serializable")


def _from_json(mapping):
    """json.load `object_hook=`: tagged nested lists come back as ndarrays"""
    return np.asarray(mapping[NDARRAY_TAG]) if NDARRAY_TAG in mapping else mapping


def _equal(left, right):
    if isinstance(left, np.ndarray) or isinstance(right, np.ndarray):
        return np.array_equal(left, right)
    return left == right


class SolverState:
    """iteration counter + step log + saved values of one iterative solver"""

    def __init__(self, name, workdir, resume=False, rewind=False):
        if rewind and not resume:
            raise RuntimeError(f"rewind cannot be True if resume is False, name={name}")
        os.makedirs(workdir, exist_ok=True)
        self._name, self._workdir = name, workdir
        self._path = os.path.join(workdir, f"{name}_state.json")
        self._undone = None          # step string taken back by a rewind
        if resume:
            self._data = self._load()
            if rewind:
                self._undone = self._data["step_log"].pop()
                LOG.info('rewinding step "%s" for "%s"', self._undone, name)
        else:
            self._data = {"iteration": 0, "step_log": []}
            self.log_step("__init__", per_iteration=False)
            self._announce()

    # ---- file ------------------------------------------------------------------------------
    def _load(self):
        trail.flush()       # (the last store of this process may still be on its way to the disk)
        with open(self._path, mode="r") as fptr:
            return json.load(fptr, object_hook=_from_json)

    def _store(self):
        """the text json.dump(self._data, fptr, indent=2, default=_to_json) writes, byte for byte -- put together from the
        encoded text of every top-level value, re-encoded only when the value changed: the file is rewritten after every
        change and an indented dump runs in json's pure-Python encoder, over a Hessenberg that grows with every Krylov
        iteration (6 dumps per iteration: 2 ms of a 15 ms Krylov iteration at 26 x 26)"""
        cache = self.__dict__.setdefault("_text_cache", {})
        parts = []
        for key, value in self._data.items():
            if isinstance(value, np.ndarray):
                snap = (value.shape, value.dtype.str, value.tobytes())
            else:
                snap = pickle.dumps(value, protocol=pickle.HIGHEST_PROTOCOL)
            hit = cache.get(key)
            if hit is None or hit[0] != snap:
                text = json.dumps(value, indent=2, default=_to_json).replace("\n", "\n  ")
                hit = cache[key] = (snap, text)
            parts.append(f"  {json.dumps(key)}: {hit[1]}")
        for key in [k for k in cache if k not in self._data]:
            del cache[key]
        text = "{\n" + ",\n".join(parts) + "\n}" if parts else "{}"
        path = self._path

        def write():
            # under a temporary name, then renamed: whoever opens the file -- a resumed run after this one was killed -- finds
            # a complete step log, the one before or this one, never a truncated one (the rewrite-in-place of the reference,
            # solver_state.py:60-72, leaves an empty file to a run killed inside it)
            with open(path + ".partial", mode="w") as fptr:
                fptr.write(text)
            os.replace(path + ".partial", path)

        # behind the files the logged steps stand for (trail.py: one writer thread, program order)
        trail.submit(write)
        return text

    def _announce(self):
        LOG.info('"%s" iteration now %d', self._name, self._data["iteration"])

    # ---- iteration counter -------------------------------------------------------------------
    def get_workdir(self):
        return self._workdir

    def get_iteration(self):
        return self._data["iteration"]

    def inc_iteration(self):
        self._data["iteration"] += 1
        self.log_step("inc_iteration")
        self._announce()
        return self._data["iteration"]

    # ---- step log -------------------------------------------------------------------------------
    def _entry(self, stepval, per_iteration):
        """step strings of one iteration carry the iteration number in front"""
        if per_iteration:
            return f"{self._data['iteration']:02}:{stepval}"
        return stepval

    def step_logged(self, stepval, per_iteration=True):
        return self._entry(stepval, per_iteration) in self._data["step_log"]

    def log_step(self, stepval, per_iteration=True):
        entry = self._entry(stepval, per_iteration)
        if entry in self._data["step_log"]:
            return
        self._data["step_log"].append(entry)
        self._store()

    def step_was_rewound(self, stepval, per_iteration=True):
        return self._undone is not None and self._undone == self._entry(stepval, per_iteration)

    # ---- saved values ---------------------------------------------------------------------------
    def get_value_saved_state(self, key):
        return self._data[key]

    def set_value_saved_state(self, key, value):
        """store a value; the state is re-read from the file at once so that what the solver goes on
        with is exactly what a resumed run would see (and the round trip is verified)"""
        self._data[key] = value
        text = self._store()
        # (parsed from the text the file is written with: what _load() returns once the writer thread has got there)
        self._data = json.loads(text, object_hook=_from_json)
        if not _equal(self._data[key], value):
            raise RuntimeError("saved_state value not recovered on reread")


def action_step_log_wrap(step, per_iteration=True, post_exit=False):
    """decorator: run the wrapped action once per step-log entry.  `solver_state`
    must be passed by keyword; `step` is formatted with the call's keyword arguments."""

    def decorate(action):
        @functools.wraps(action)
        def guarded(*args, **kwargs):
            ledger = kwargs["solver_state"]
            label = step.format(**kwargs)
            if ledger is None:
                action(*args, **kwargs)
            elif not ledger.step_logged(label, per_iteration):
                action(*args, **kwargs)
                ledger.log_step(label, per_iteration)
            else:
                return
            if post_exit:
                raise SystemExit

        return guarded

    return decorate
