"""round-3 verdict, weak 7: after bench.py's shard_e3 leg (phosphorus through dist.column_sharded_gmres) the launch-bound years of the
same process ran 2.7 times slower.  One process, one small iage engine whose frozen year is forced launch by launch
(2 341 launches of a few microseconds: a launch-gap meter), timed before and after each ingredient of that leg."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nk_ooc_amd.engine import iage_engine, phosphorus_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n_ph = int(sys.argv[1]) if len(sys.argv) > 1 else 416
n = 26
meter = iage_engine(Grid2d.default(n, n))
meter.set_option("stream_years", 0)
meter.set_option("frozen_persistent", 0)
col = np.interp(meter.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
x = meter.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
fx, st, sched = meter.comp_fcn(x, record=True)


def gap(tag):
    best = min(meter.comp_fcn_frozen(x, sched)[1]["seconds"] for _ in range(5))
    rep = meter.profile_replay(0, 400)
    _, stf = meter.comp_fcn_frozen(x, sched)
    print(f"{tag:70s} frozen year by {stf['nlaunch']} launches: {1e3 * best:7.2f} ms = {1e6 * best / stf['nlaunch']:5.2f} us per launch; "
          f"400 launches back to back: {rep['avg_us']:5.2f} us each", flush=True)
    return best


base = gap("fresh process")
# what bench.py has done before its auxiliary legs: the 416 x 416 iage workload with its 119 GB schedule cache, closed again
big = iage_engine(Grid2d.default(416, 416))
colb = np.interp(big.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
xb = big.upload(np.stack([np.broadcast_to(colb[:, None], (416, 416))] * 2).copy())
big.set_option("frozen_alloc_async", 0)
fb, _, schedb = big.comp_fcn(xb, record=True)
for _ in range(3):
    big.comp_fcn_frozen(xb, schedb)
print(f"    (iage 416^2: {big.counter('frozen_persistent_years')} one-launch years on a cache of {big.counter('frozen_cache_bytes') / 1e9:.0f} GB)")
gap("with a 416^2 iage engine and its schedule cache alive")
big.close()
gap("after closing it (119 GB given back)")
import torch  # noqa: E402

torch.cuda.set_device(0)
t = torch.zeros(16, device="cuda")
torch.cuda.synchronize()
gap("after torch.cuda initialisation and one tensor")
from nk_ooc_amd import dist as nkdist  # noqa: E402

comm = nkdist.ColumnComm(0, 1, torch.device("cuda", 0))
gap("after dist.ColumnComm (pinned staging buffer, device tensor)")
grid = Grid2d.default(n_ph, n_ph)
eng = phosphorus_engine(grid)
eng.set_region(np.ones((n_ph, n_ph), dtype=np.int32), np.outer(grid.depth.delta, grid.ypos.delta))
prof = [np.interp(grid.depth.mid, d, v) for d, v in (([1.3e2, 2.6e2], [5.5e-3, 4.1]), ([9.5e1, 1.4e2], [7.1e-2, 1.5e-4]),
                                                     ([1.7e2, 2.5e2], [1.8e-2, 7.9e-4]))]
x0 = np.stack([np.broadcast_to(p[:, None], (n_ph, n_ph)) for p in prof]).copy()
xp = eng.upload(x0)
gap(f"after creating a phosphorus engine at {n_ph}^2")
fp, stp, _ = eng.comp_fcn(xp)
schedp = eng.last_schedule()
gap("after its free-running year")
eng.precond_setup_state((x0 + eng.download(fp))[0])
gap("after its preconditioner (three factorisations, shift-invert Arnoldi)")
nkdist.column_sharded_gmres(eng, comm, xp, fp, 0.0, 0, 2, sched=schedp)
eng.sync()
gap("after two iterations of column_sharded_gmres")
eng.close()
last = gap("after closing that engine")
print(f"slowdown of the launch-bound year over the whole sequence: {last / base:.2f} x", flush=True)
