"""per-call wall time and stats of the frozen years of a Newton-Krylov run (tools/probe_spinup.py's case)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd import engine as _engine  # noqa: E402

_orig = _engine.ModuleEngine.comp_fcn_frozen


def traced(self, x, sched, out=None):
    t0 = time.perf_counter()
    res = _orig(self, x, sched, out)
    self.sync()
    st = res[1]
    print(f"comp_fcn_frozen {1e3 * (time.perf_counter() - t0):8.1f} ms: launches {st['nlaunch']}, year {1e3 * st['seconds']:.1f} ms, "
          f"barrier timeouts {st['nbarrier_timeouts']}, resumed {st['nresumed']}, err checked {st['nerr_checked']}, "
          f"one-launch years {self.counter('frozen_persistent_years')}, cache builds {self.counter('frozen_cache_builds')}", flush=True)
    return res


_engine.ModuleEngine.comp_fcn_frozen = traced
opts = [kv.split("=") for kv in os.environ.get("PROBE_OPTS", "").split(",") if kv]      # e.g. PROBE_OPTS=frozen_cache_after=0
if opts:
    _init = _engine.ModuleEngine.__init__

    def _patched(self, *args, **kwargs):
        _init(self, *args, **kwargs)
        for key, val in opts:
            self.set_option(key, float(val))

    _engine.ModuleEngine.__init__ = _patched
sys.argv = [sys.argv[0]] + sys.argv[1:]
import runpy  # noqa: E402

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_spinup.py"), run_name="__main__")
