"""probe: set-up year of the ci_py_driver_2d_iage_column_regions case in the controller modes; largest
deviation of the history file from the reference's committed one in units of its CI tolerance
(atol 1e-6 + rtol 1e-3 |ref|)"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ooc_amd import ncio  # noqa: E402
from nk_ooc_amd.model_state import ModelState  # noqa: E402
from nk_ooc_amd.setup_solver import make_config, setup  # noqa: E402

BASE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_baselines",
                    "ci_py_driver_2d_iage_column_regions")
want, _ = ncio.read_file(os.path.join(BASE, "hist_0000.nc"))
for fresh, stage, cap in ((0, -1, 0), (1, -1, 0), (1, 1, 0), (1, 2, 0), (1, 1, 1), (1, 1, 2), (1, 1, 3), (0, -1, 1)):
    os.environ["NK2D_JAC_FRESH"], os.environ["NK2D_JAC_STAGE"], os.environ["NK2D_GROWTH_CAP"] = str(fresh), str(stage), str(cap)
    work = tempfile.mkdtemp()
    cfg = make_config(work, 20, 3, extra_modelinfo={"max_abs_vvel": "0.0", "horiz_mix_coeff": "0.0"})
    ModelState.reset_class()
    ModelState.write_files = True
    setup(cfg, fp_cnt=1)
    got, _ = ncio.read_file(os.path.join(work, "gen_init_iterate", "hist_0000.nc"))
    worst = {name: float(np.max(np.abs(got[name] - ref) / (1.0e-6 + 1.0e-3 * np.abs(ref)))) for name, ref in want.items()
             if got[name].dtype.kind == "f"}
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    st = ModelState.last_stats[0]
    print(f"jac_fresh={fresh} jac_stage={stage} growth_cap={cap}: Newton {st['nnewton']}, rejected {st['nrejected']}, steps {ModelState.last_stats[0]['nsteps']}, worst deviation / tolerance:",
          ", ".join(f"{k} {v:.2f}" for k, v in top), flush=True)
