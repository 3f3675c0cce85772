"""N > 1 path with the real engines: two ranks (tracer module per rank, as BASELINE config 4
distributes them) against the single-process solves of the same modules, and the multi-rank leg of
bench.py.  The box has one GPU: the ranks share it and the stopping-test collectives run over gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(args, port, timeout=400, expect_ok=True):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", NK2D_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    if expect_ok:
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    return res


def test_module_per_rank_matches_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_gpu_worker
    from nk_ooc_amd.krylov_solver import KrylovSolver

    _launch([os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(tmp_path)], 29541)
    for rank, module in enumerate(dist_gpu_worker.MODULES):
        got = json.load(open(tmp_path / f"result{rank}.json"))
        want = dist_gpu_worker.solve_one(str(tmp_path / f"single_{module}"), module, KrylovSolver)
        assert got["module"] == module and got["iters"] == want["iters"] == 3
        # same kernels, same decisions: the ranks reproduce the single-process numbers
        assert np.allclose(got["beta"], want["beta"], rtol=1e-12, atol=0.0)
        assert np.allclose(got["h_mat"], want["h_mat"], rtol=1e-9, atol=1e-12)
        assert np.allclose(got["inc_norm"], want["inc_norm"], rtol=1e-9)


def _check_two_rank_line(stdout, backend, check_shard=True):
    line = [ln for ln in stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0.0
    assert out["config"]["parallelism"].endswith("x2") and out["config"]["collective_backend"] == backend
    for key in ("roofline", "metric", "unit", "ms_per_step", "dtype", "end_to_end"):
        assert key in out
    assert "cpu_baseline" not in out and "ladder" not in out   # rank 0 of single-GPU runs only
    # both products as top-level figures (round-2 ADVICE): frozen years (the default, = value) and the reference's two
    # free-running years, measured in the same run
    assert out["value_frozen"] == out["value"] and 0.0 < out["value_reference_semantic"] < out["value"] * 1.5
    assert "frozen" in out["metric"] and "roofline_year" in out
    # BASELINE.json configs[3]: the three-module mix on one GPU and dealt round-robin to the two ranks
    mix = out["config4_mix"]
    assert mix["modules"] == ["iage", "phosphorus", "forced_dye"] and mix["one_gpu"]["ms_per_krylov_iteration"] > 0.0
    assert mix["distributed"]["ranks_used"] == 2
    assert mix["distributed"]["layout"] == {"rank0": ["iage", "forced_dye"], "rank1": ["phosphorus"]}
    assert out["strong_scaling"]["n_gpus_used"] == 2 and out["strong_scaling"]["speedup"] > 0.0
    for counters in mix["one_gpu"]["counters"].values():
        assert counters["frozen_years_rejected"] == 0
    # BASELINE.json configs[4]: phosphorus with its Krylov basis columns on the two ranks
    e3 = out["shard_e3"]
    assert e3["sharded"]["ranks"] == 2 and e3["sharded"]["ms_per_jvp"] > 0.0 and e3["one_gpu"]["ms_per_jvp"] > 0.0
    # per Krylov iteration one broadcast and three all-reduces of whole vectors (+ one for the iterate at the end), two small ones
    assert 4.0 <= e3["sharded"]["vector_collectives_per_jvp"] <= 5.0 and e3["sharded"]["small_allreduces_per_jvp"] == 2.0
    assert e3["frozen_years_rejected"] == 0
    # SURVEY.md section 8(e) level 2, measured in the same run: ONE module, tracer per rank
    if check_shard:
        shard = out["shard_e2"]
        assert shard["backend"] == backend and shard["allreduce_latency_us"] > 0.0
        if "skipped" not in shard:
            assert shard["ms_per_jvp"] > 0.0
            # every Radau norm of the coupled year that gives F(x) is a collective; the perturbed year of the Krylov
            # iteration repeats its steps and exchanges nothing -- what is left are the iteration's own reductions
            assert shard["allreduces_of_the_coupled_year"] > 100
            assert shard["allreduces_per_jvp"] < 20
    else:
        assert "shard_e2" not in out
    return out


def test_bench_two_ranks():
    # the sharded-module leg on a tiny grid: two processes share ONE GPU here and the card switches contexts at
    # every all-reduce (13 ms per Newton iteration -- an artefact of this rehearsal, not of the layout)
    res = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "26",
                   "--cpu-baseline-seconds", "0", "--shard-grid", "12", "--shard-budget", "1000"], 29542)
    out = _check_two_rank_line(res.stdout, "gloo")
    assert "skipped" not in out["shard_e2"] and out["shard_e2"]["grid"] == [12, 12]


def test_bench_two_ranks_keeps_its_line_when_the_auxiliary_legs_run_out_of_time():
    """--aux-budget: the legs behind the headline get a wall-clock budget on all ranks; when it runs out (here at once) rank 0
    prints the line without them -- a leg stuck in a collective cannot cost the record -- and the ranks leave with a NON-ZERO
    status (round-3 ADVICE): the launcher reports a run that did not finish, the hang is not recorded as a success"""
    res = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "26",
                   "--cpu-baseline-seconds", "0", "--shard-grid", "12", "--aux-budget", "0.05"], 29543, expect_ok=False)
    assert res.returncode != 0
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0.0 and out["value_reference_semantic"] > 0.0
    assert out["aux_legs_timed_out"]["budget_s"] == 0.05 and "shard_e3" not in out


def test_bench_starts_its_own_ranks_on_the_gpu_box():
    """the driver's invocation: `python bench.py --gpus 2`, no launcher, no WORLD_SIZE"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMP_NUM_THREADS="1", NK2D_BENCH_BACKEND="gloo")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup",
                          "0", "--grid", "26", "--cpu-baseline-seconds", "0", "--no-shard"],
                         env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    _check_two_rank_line(res.stdout, "gloo", check_shard=False)


def test_bench_two_ranks_rccl():
    """the same over RCCL (backend nccl), one rank per GPU: needs two GPUs, skipped on the one-GPU box"""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL refuses two ranks on one GPU; this leg runs on multi-GPU nodes")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NK2D_BENCH_BACKEND")}
    env.update(OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup",
                          "0", "--grid", "52", "--cpu-baseline-seconds", "0"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    _check_two_rank_line(res.stdout, "nccl")


def test_vector_collectives_work_in_place_on_engine_memory():
    """the zero-copy path of dist.ColumnComm (BASELINE configs[4] over RCCL): a torch view of an engine vector's HBM
    (`ModuleEngine.vec_tensor`), collectives of a one-rank nccl (= RCCL) group on it, results seen by the engine"""
    import torch
    import torch.distributed as tdist

    from nk_ooc_amd import dist as nkdist
    from nk_ooc_amd.engine import iage_engine
    from nk_ooc_amd.grid import Grid2d

    eng = iage_engine(Grid2d.default(70, 12))
    rng = np.random.default_rng(4)
    host = rng.standard_normal(eng.shape)
    vec = eng.upload(host)
    view = eng.vec_tensor(vec)
    assert view.is_cuda and view.dtype == torch.float64 and view.numel() == 2 * 12 * 2 * 64     # packed columns, zero padded
    eng.sync()
    assert abs(float(view.sum().item()) - host.sum()) < 1e-9 * np.abs(host).sum()
    view.mul_(2.0)                                                  # in place, on torch's stream
    torch.cuda.synchronize()
    assert np.array_equal(eng.download(vec), 2.0 * host)
    started = False
    if not tdist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        tdist.init_process_group("nccl", rank=0, world_size=1)
        started = True
    try:
        tdist.all_reduce(view)
        tdist.broadcast(view, src=0)
        torch.cuda.synchronize()
        assert np.array_equal(eng.download(vec), 2.0 * host)
        # the communicator itself, device flavour: staging through the pinned buffer, vectors in place
        comm = nkdist.ColumnComm(0, 1, torch.device("cuda", 0))
        assert np.array_equal(comm.allreduce(np.arange(6.0).reshape(3, 2)), np.arange(6.0).reshape(3, 2))
        small = nkdist.ShardComm(torch.device("cuda", 0))
        assert small.allreduce_scalar(3.5) == 3.5
        # every other collective bench.py and dist.py issue on the RCCL branch, on this one-rank communicator: the
        # barrier, the MAX of the elapsed times (float64), the AND of the convergence flags (MIN over int32) on the
        # world group and on a sub-group, the sum of a small float64 array through the device staging path
        tdist.barrier()
        el = torch.tensor([1.25], dtype=torch.float64, device="cuda:0")
        tdist.all_reduce(el, op=tdist.ReduceOp.MAX)
        assert float(el.item()) == 1.25
        sub = tdist.new_group([0])
        for group in (None, sub):
            flag = nkdist._AllFlag(np.array([True, True]), torch.zeros(1, dtype=torch.int32, device="cuda:0"), group)
            # (a one-rank group answers locally; the collective itself, as _AllFlag issues it for more ranks:)
            assert flag.all() is True
            buf = torch.ones(1, dtype=torch.int32, device="cuda:0")
            tdist.all_reduce(buf, op=tdist.ReduceOp.MIN, group=group)
            assert int(buf.item()) == 1
        arr = small._buf[:3]
        arr.copy_(torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64))
        tdist.all_reduce(arr, op=tdist.ReduceOp.SUM, group=sub)
        assert arr.cpu().tolist() == [1.0, 2.0, 3.0]
    finally:
        if started:
            tdist.destroy_process_group()
    eng.close()
