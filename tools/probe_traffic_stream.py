"""probe (torch-free) for rocprofv3 --pmc passes over the command-stream kernel: free-running forward years of iage n x n as command
streams (k_stream: one resident kernel per year executes the host controller's launches as commands); prints the algorithmic bytes of the
Newton commands of a year for the comparison with FETCH_SIZE / WRITE_SIZE"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
eng = iage_engine(Grid2d.default(n, n))
col = np.interp(eng.grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
x = eng.upload(np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy())
years = []
for _ in range(3):
    eng.profile_reset(0)
    fx, st, sched = eng.comp_fcn(x, record=True)
    tot = eng.profile_totals()
    years.append({"seconds": st["seconds"], "newton_commands": tot["launches"], "algorithmic_bytes_of_newton_commands": tot["bytes"],
                  "kernel_starts": st["nlaunch"], "nsteps": st["nsteps"], "nnewton": st["nnewton"]})
print(json.dumps({"grid": [n, n], "years_as_command_streams": eng.counter("stream_years_run"), "stream_timeouts": eng.counter("stream_timeouts"),
                  "years": years}))
eng.close()
