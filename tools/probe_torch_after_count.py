"""which order of (engine, torch.cuda.device_count, torch CUDA init) loses torch its GPU?"""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402

from nk_ooc_amd.engine import iage_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402

mode = sys.argv[1]
if mode == "count_first":
    print("count", torch.cuda.device_count())
eng = iage_engine(Grid2d.default(26, 26))
if mode == "count_after_engine":
    print("count", torch.cuda.device_count())
if mode == "fork":
    import subprocess
    subprocess.run([sys.executable, "-c", "print('child')"])
try:
    x = torch.zeros(4, device="cuda")
    print(mode, "torch ok", x.sum().item())
except Exception as err:  # noqa: BLE001
    print(mode, "torch FAILED:", err)
