"""Host-side utilities the hot path touches: communicator, data log, trace points."""
import itertools
import os
import sys
import time


def _job_suffix():
    """``d<JOBID>`` under a batch scheduler (PBS, SLURM), else the wall-clock minute."""
    pbs, slurm = os.environ.get('PBS_JOBID'), os.environ.get('SLURM_JOBID')
    if pbs:
        return "d" + pbs.split('.')[0]
    if slurm:
        return "d" + slurm
    return time.strftime("%Y-%m-%d+%H:%M")


def create_output_path(basename=None, comm=None):
    """A fresh directory ``output/<basename>.<suffix>[+N]/`` for a run's results (what the reference's
    ``create_output_path`` provides, prosper/utils/__init__.py:17-68): ``basename`` defaults to the running script, the
    suffix names the batch job or the time, and ``+N`` is appended until the name is unused, so nothing is ever
    overwritten.  Rank 0 creates it; every rank gets the same path, with a trailing slash."""
    from .parallel import COMM_WORLD
    comm = COMM_WORLD if comm is None else comm
    path = None
    if comm.rank == 0:
        stem = "output/%s.%s" % (sys.argv[0] if basename is None else basename, _job_suffix())
        for n in itertools.count():
            path = stem if n == 0 else "%s+%d" % (stem, n)
            try:
                os.makedirs(path)
                break
            except FileExistsError:
                continue
    return comm.bcast(path) + "/"
