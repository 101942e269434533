"""Single-rank stand-in for mpi4py, used ONLY by make_golden.py to import the reference
in the build container.  Never imported by the product."""
from . import MPI
