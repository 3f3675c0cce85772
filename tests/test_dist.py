"""N > 1 path on the CPU: world_size-2 gloo run of the module-per-rank Krylov loop
(nk_ooc_amd.dist): modules are partitioned over ranks and only the stopping test is a
collective (reference: `converged(...).all()` over all modules, krylov_solver.py:159)."""
import json
import os
import subprocess
import sys

import pytest

from nk_ooc_amd.dist import partition_modules

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_modules():
    names = ["iage", "phosphorus", "forced_a", "forced_b", "forced_c"]
    parts = partition_modules(names, 2)
    assert parts == [["iage", "forced_a", "forced_c"], ["phosphorus", "forced_b"]]
    assert sorted(sum(partition_modules(names, 4), [])) == sorted(names)
    assert partition_modules(names, 8)[5:] == [[], [], []]


def test_world_size_2_gloo(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = [json.load(open(tmp_path / f"result{r}.json")) for r in range(2)]
    assert out[0]["modules"] == ["iage"] and out[1]["modules"] == ["forced_a"]
    # rank 0 alone would have stopped after 1 iteration; the global AND keeps it going
    assert out[0]["iters"] == 3 and out[1]["iters"] == 3
    for o in out:
        assert o["resid"] < 0.2 * o["fcn_norm"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver calls it (no launcher, no WORLD_SIZE): the parent starts
    the ranks itself and relays rank 0's JSON line (here the CPU-only launch check over gloo)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"launch_check": 2, "rank_sum": 3}


def test_sharded_module_reductions_gloo(tmp_path):
    """SURVEY.md section 8(e) level 2 on the CPU: a module's tracers on two ranks, inner products and
    CGS-2 all-reduced over gloo (NumPy stand-in engines), against the unsharded numbers"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29534",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "shard"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = [json.load(open(tmp_path / f"shard{r}.json")) for r in range(2)]
    assert out[0] == out[1]                       # every rank holds the same reduced numbers
    assert out[0]["dot_err"] < 1e-14 and out[0]["h_err"] < 1e-12 and out[0]["ortho"] < 1e-14
    assert out[0]["allreduces"] == 1 + 2 + 1     # dot, CGS-2 (two passes), final check
    assert out[0]["gmres_resid_drop"] < 0.5 and out[0]["gmres_allreduces_per_iter"] <= 6.0


@pytest.mark.parametrize("world", [2, 4])
def test_column_sharded_gmres_gloo(tmp_path, world):
    """SURVEY.md section 8(e) level 3 on the CPU (BASELINE configs[4]): the Krylov basis columns dealt round-robin to 2 and 4
    ranks -- broadcast of the product, gathered projections, all-reduced partial sums -- against the same loop on one rank"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29535 + world),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "columns"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = [json.load(open(tmp_path / f"columns{r}.json")) for r in range(world)]
    for rank, o in enumerate(out):
        assert o["iters"] == 6 and o["alone_collectives"] == 0
        # basis columns 0..5 (the last product's direction is never stored) round-robin
        assert o["columns_here"] == [i for i in range(6) if i % world == rank]
        assert o["h_err"] < 1e-12 and o["resid_err"] < 1e-12 and o["inc_err"] < 1e-11
        assert o["resid_drop"] < 0.2
        # per Krylov iteration: 2 small all-reduces (CGS-2 projections) and 1 broadcast + 3 all-reduces of whole vectors
        assert o["small_allreduces_per_iter"] == 2.0 and o["vector_collectives_per_iter"] == 4.0
    assert all(o["h_err"] == out[0]["h_err"] for o in out)
