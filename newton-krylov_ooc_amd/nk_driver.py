"""Driver of the Newton-Krylov solve (reference `nk_ooc/nk_driver.py:38-67`).

    python -m nk_ooc_amd.nk_driver --workdir DIR [--cfg_fnames a.cfg,b.cfg] [--resume] [--rewind]
                                   [--depth_nlevs N --ypos_nlevs N] [--tracer_module_names ...]

Reads the same cfg files as the reference, configures the model, and iterates
`NewtonSolver.step()` until `converged().all()`.  `reinvoke` is forced off (the reference's
`--persist`): the process keeps the GPU contexts alive instead of exiting after every forward
year; `--resume` continues from the JSON checkpoints exactly as the reference does.
"""

import argparse
import logging
import os
import sys

from .model_config import ModelConfig, read_cfg_files
from . import trail
from .model_state import ModelState
from .newton_solver import NewtonSolver
from .setup_solver import default_cfg_fnames


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Newton-Krylov solver, MI355X path")
    parser.add_argument("--cfg_fnames", default=default_cfg_fnames())
    parser.add_argument("--workdir", default=None)
    parser.add_argument("--tracer_module_names", default=None)
    parser.add_argument("--depth_nlevs", default=None)
    parser.add_argument("--ypos_nlevs", default=None)
    parser.add_argument("--newton_max_iter", default=None)
    parser.add_argument("--newton_rel_tol", default=None)
    parser.add_argument("--resume", action="store_true")
    parser.add_argument("--rewind", action="store_true")
    return parser.parse_args(argv)


def config_from_args(args):
    overrides = {"DEFAULT": {}, "modelinfo": {"reinvoke": "False"}, "solverinfo": {}}
    if args.workdir is not None:
        overrides["DEFAULT"]["workdir"] = args.workdir
    for key in ("tracer_module_names", "depth_nlevs", "ypos_nlevs"):
        if getattr(args, key) is not None:
            overrides["modelinfo"][key] = getattr(args, key)
    for key in ("newton_max_iter", "newton_rel_tol"):
        if getattr(args, key) is not None:
            overrides["solverinfo"][key] = getattr(args, key)
    return read_cfg_files(args.cfg_fnames, overrides=overrides, write_cfg_out=True)


def run(config, resume=False, rewind=False, model_state_class=ModelState):
    """Newton iterations to convergence; returns the NewtonSolver"""
    logger = logging.getLogger(__name__)
    if os.path.exists("KILL"):
        logger.warning("KILL file detected, exiting")
        raise SystemExit
    model_state_class.reset_class()
    model_state_class.model_config_obj = ModelConfig(config["modelinfo"])
    # history files on a background thread while the driver runs (every access to them goes through the state class);
    # NK2D_ASYNC_HIST=0 keeps them synchronous
    was_async = getattr(model_state_class, "async_hist", None)
    if was_async is not None:
        model_state_class.async_hist = os.environ.get("NK2D_ASYNC_HIST", "1") != "0"
    # ... and the checkpoint trail (vector files, step logs, statistics) in program order on one writer thread (trail.py);
    # NK2D_ASYNC_TRAIL=0 keeps every write inside the call that asks for it
    trail_was = trail.set_enabled(os.environ.get("NK2D_ASYNC_TRAIL", "1") != "0")
    try:
        solver = NewtonSolver(model_state_class, solverinfo=config["solverinfo"], resume=resume, rewind=rewind)
        return _iterate(solver, model_state_class, logger)
    finally:
        try:
            trail.set_enabled(trail_was)        # (flushes)
        except Exception:                       # noqa: BLE001 -- reported, never in the way of the flush below
            logger.exception("checkpoint trail: a queued write failed")
        # also on an exception (or the SystemExit of a reinvoked run): no history file is left half written
        flush = getattr(model_state_class, "flush_files", None)
        if flush is not None:
            flush()
        if was_async is not None:
            model_state_class.async_hist = was_async


def _iterate(solver, model_state_class, logger):
    while True:
        if solver.converged().all():
            logger.info("Newton convergence criterion satisfied")
            solver.log()
            break
        solver.step()
    return solver


def main(argv=None):
    args = parse_args(argv)
    config = config_from_args(args)
    solverinfo = config["solverinfo"]
    os.makedirs(solverinfo["workdir"], exist_ok=True)
    logging.basicConfig(
        level=getattr(logging, solverinfo.get("logging_level", "INFO")),
        format="%(asctime)s:%(process)s:%(filename)s:%(funcName)s:%(message)s",
        handlers=[logging.FileHandler(solverinfo["logging_fname"], mode="a"), logging.StreamHandler(sys.stdout)],
    )
    run(config, resume=args.resume, rewind=args.rewind)


if __name__ == "__main__":
    main()
