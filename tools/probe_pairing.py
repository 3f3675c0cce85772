"""How many collectives does a coupled year of the tracer-sharded iage module take with the vector norm hook, by length of
the vector, against the norms its controller reads?  Two shards on two host threads of one process (the all-reduce is a
barrier), as tests/test_gpu_capi_krylov.py::test_norm_hook_couples_two_single_tracer_engines.
    python tools/probe_pairing.py [n] [lin_tol ...]"""
import collections
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from nk_ooc_amd.dist import iage_shard_engine  # noqa: E402
from nk_ooc_amd.grid import Grid2d  # noqa: E402


class BarrierComm:
    def __init__(self):
        self.barrier = threading.Barrier(2)
        self.slots = [0.0, 0.0]
        self.by_len = collections.Counter()

    def bind(self, rank):
        def allreduce(arr):
            self.slots[rank] = np.array(arr, dtype=np.float64)
            self.barrier.wait()
            total = self.slots[0] + self.slots[1]
            self.barrier.wait()
            if rank == 0:
                self.by_len[total.size] += 1
            return total

        return type("C", (), {"allreduce": staticmethod(allreduce),
                              "allreduce_scalar": staticmethod(lambda v: float(allreduce(np.array([v]))[0]))})


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    tols = [float(a) for a in sys.argv[2:]] or [1.0e-3, 3.0e-2]
    grid = Grid2d.default(n, n)
    col = np.interp(grid.depth.mid, [55.0, 200.0], [0.0, 2.0])
    y0 = np.stack([np.broadcast_to(col[:, None], (n, n))] * 2).copy()
    for tol in tols:
        for fresh in (0, 1):
            comm = BarrierComm()
            shards = [iage_shard_engine(grid, r, comm.bind(r), lin_tol=tol) for r in range(2)]
            for eng in shards:
                eng.set_option("jac_fresh", fresh)
            out = [None, None]

            def run(rank):
                eng = shards[rank]
                t0 = time.perf_counter()
                _, stats, _ = eng.comp_fcn(eng.upload(y0[rank:rank + 1]), record=True)
                out[rank] = (stats, time.perf_counter() - t0)

            threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=600)
            st = out[0][0]
            reads = st["nnewton"] + st["nsteps"] + st["nrejected"]
            calls = sum(comm.by_len.values())
            print(f"{n}x{n} lin_tol {tol:g} jac_fresh {fresh}: steps {st['nsteps']} rejected {st['nrejected']} newton {st['nnewton']} "
                  f"norms read {reads}, collectives {calls} ({calls / reads:.2f} per norm) by length {dict(comm.by_len)}, "
                  f"year {out[0][1]:.2f} s", flush=True)
            for eng in shards:
                eng.close()


if __name__ == "__main__":
    main()
