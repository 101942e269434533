"""Host-side utilities the hot path touches: communicator, data log, trace points."""
import errno
import os
import sys
import time as tm


def create_output_path(basename=None, comm=None):
    """Create ``output/<BASENAME>.<suffix>`` without ever reusing a directory (prosper/utils/__init__.py:17-68): the
    suffix is ``d<JOBID>`` under PBS / SLURM, else date and time; an existing directory gets ``+N`` appended.  Rank 0
    creates it, every rank gets the path (with a trailing slash, as upstream)."""
    from .parallel import COMM_WORLD
    comm = COMM_WORLD if comm is None else comm
    dirname = None
    if comm.rank == 0:
        if basename is None:
            basename = sys.argv[0]
        if 'PBS_JOBID' in os.environ:
            suffix = "d" + os.environ['PBS_JOBID'].split('.')[0]
        elif 'SLURM_JOBID' in os.environ:
            suffix = "d" + os.environ['SLURM_JOBID']
        else:
            suffix = tm.strftime("%Y-%m-%d+%H:%M")
        counter = 0
        dirname = "output/%s.%s" % (basename, suffix)
        while True:
            try:
                os.makedirs(dirname)
            except OSError as e:
                if e.errno != errno.EEXIST:
                    raise
                counter += 1
                dirname = "output/%s.%s+%d" % (basename, suffix, counter)
            else:
                break
    dirname = comm.bcast(dirname)
    return dirname + "/"
