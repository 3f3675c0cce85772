"""does the process end cleanly?  modes: open (engine left open at exit), closed, hook (two hooked shards left open),
torch_open (torch imported first, engine left open)"""
import faulthandler
import sys

faulthandler.enable()
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
mode = sys.argv[1]
if mode.startswith("torch"):
    import torch  # noqa: F401
import numpy as np  # noqa: E402

from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

grid = Grid2d.default(26, 26)
eng = iage_engine(grid)
col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
y0 = np.stack([np.broadcast_to(col[:, None], (26, 26))] * 2).copy()
fx, _, _ = eng.comp_fcn(eng.upload(y0))
if mode == "hook":
    from nk_ooc_amd.dist import iage_shard_engine

    comm = type("C", (), {"allreduce": staticmethod(lambda a: np.array(a)), "allreduce_scalar": staticmethod(float)})
    sh = iage_shard_engine(grid, 0, comm)
    sh.comp_fcn(sh.upload(y0[:1]))
if mode in ("closed",):
    eng.close()
print(mode, "done", flush=True)
