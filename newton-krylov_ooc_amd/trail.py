"""The checkpoint trail on ONE writer thread, in program order.

The reference is out-of-core: every solver action ends in a file (`model_state_base.py:93-111` NetCDF vector files,
`solver_state.py:60-72` the JSON step log rewritten after every change, `stats_file.py` appended per iteration), and the
next action re-reads them.  Here the vectors stay in HBM and the files are a trail for `--resume` and for whoever reads the
work directory afterwards -- nothing inside a solve waits for them.  With `enabled` set, every write of the trail is a job
on a FIFO queue served by one thread: the files appear on disk in exactly the order the synchronous code would have written
them, a few milliseconds later, while the GPU runs the next forward year (the host thread sits in a ctypes call with the
GIL released).  A run that is killed finds a PREFIX of the synchronous trail on disk -- never a step logged before the
file it stands for -- so `--resume` / `--rewind` see what they would have seen a moment earlier.

`flush()` returns when everything submitted so far is on disk (and re-raises what a job raised); everything that READS the
trail by name calls it first.  Disabled (the default of the library classes; `nk_driver.run` and `bench.py` switch it on,
`NK2D_ASYNC_TRAIL=0` keeps it off there), `submit` runs the job at once: the reference's contract, a file is on disk when
the call that writes it returns.
"""

import atexit
import os
import queue
import threading

_MAX_QUEUED = 48     # jobs in flight (a vector file holds its host copy until written: bounds the host memory)


class Trail:
    def __init__(self):
        self.enabled = os.environ.get("NK2D_ASYNC_TRAIL", "0") == "1"
        self._jobs = None
        self._thread = None
        self._error = None
        self.jobs_run = 0        # by the writer thread (tests, bench)

    # ---- the writer thread ----------------------------------------------------------------------
    def _serve(self):
        while True:
            job = self._jobs.get()
            try:
                if job is None:
                    return
                if self._error is None:      # after a failure nothing further is written: the trail stays a prefix
                    job()
                    self.jobs_run += 1
            except BaseException as exc:     # noqa: BLE001 -- handed to the submitting thread
                self._error = exc
            finally:
                self._jobs.task_done()

    def _start(self):
        self._jobs = queue.Queue(maxsize=_MAX_QUEUED)
        self._thread = threading.Thread(target=self._serve, name="nk2d-trail", daemon=True)
        self._thread.start()

    def _raise_pending(self):
        if self._error is not None:
            exc, self._error = self._error, None
            raise exc

    # ---- interface ------------------------------------------------------------------------------
    def submit(self, job):
        """run `job()` behind everything submitted before it (at once where the trail is synchronous)"""
        if not self.enabled:
            self.flush()
            job()
            return
        self._raise_pending()
        if self._thread is None or not self._thread.is_alive():
            self._start()
        self._jobs.put(job)

    def flush(self):
        """everything submitted so far is on disk; an exception a job raised is raised here"""
        if threading.current_thread() is self._thread:
            return
        if self._jobs is not None:
            self._jobs.join()
        self._raise_pending()

    def drain(self):
        """wait for the queue to empty WITHOUT taking delivery of a job's exception (it stays for the next submit / flush):
        for whoever must not pull the rug from under a queued job -- an engine about to destroy its context"""
        # (never from the writer thread itself -- an engine whose last reference goes away while a job runs is finalised there)
        if self._jobs is not None and threading.current_thread() is not self._thread:
            self._jobs.join()

    def pending(self):
        return 0 if self._jobs is None else self._jobs.unfinished_tasks


TRAIL = Trail()
submit = TRAIL.submit
flush = TRAIL.flush
drain = TRAIL.drain


def set_enabled(flag):
    """switch the background writer on or off (off: after a flush); returns the previous setting"""
    was = TRAIL.enabled
    if was and not flag:
        TRAIL.flush()
    TRAIL.enabled = bool(flag)
    return was


@atexit.register
def _flush_at_exit():
    try:
        TRAIL.flush()
    except BaseException:       # noqa: BLE001 -- the interpreter is going down; the error was the job's to report
        pass
