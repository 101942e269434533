"""Communicator + collective helpers for the truncated-EM hot path.

The reference talks to ``mpi4py.MPI.COMM_WORLD`` directly (captured as a default
argument at import time: prosper/em/__init__.py:34, prosper/em/camodels/__init__.py:60,
prosper/utils/parallel.py:27,44,91).  Here one process drives one MI355X and the
transport is ``torch.distributed`` -- backend ``nccl`` (= RCCL over xGMI) for device
buffers, ``gloo`` for host objects and for the CPU tests.  ``Comm`` keeps the mpi4py
call surface the models use (``rank, size, allreduce, Allreduce, Allgather, bcast,
Bcast, Barrier``) so model code keeps its ``comm.`` call sites; it adds
``allreduce_device`` for the one fused sufficient-statistics all-reduce per EM step.

Helper semantics follow prosper/utils/parallel.py:
  stride_data :44-84, allsort :87-110, allargsort :113-135, allmean :138-156, allsum :159-170.
"""
import sys

import numpy as np

try:  # torch is plumbing only (process groups, device buffers)
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover - torch is always present in the target image
    torch = None
    dist = None


# mpi4py datatype stand-ins: only ever used as opaque tags in [array, tag] pairs
DOUBLE, FLOAT, SHORT, INT, LONG = "DOUBLE", "FLOAT", "SHORT", "INT", "LONG"
UNSIGNED_SHORT, UNSIGNED_INT, UNSIGNED_LONG = "UNSIGNED_SHORT", "UNSIGNED_INT", "UNSIGNED_LONG"
SUM = "SUM"

typemap = {
    np.dtype('float64'): DOUBLE,
    np.dtype('float32'): FLOAT,
    np.dtype('int16'): SHORT,
    np.dtype('int32'): INT,
    np.dtype('int64'): LONG,
    np.dtype('uint16'): UNSIGNED_SHORT,
    np.dtype('uint32'): UNSIGNED_INT,
    np.dtype('uint64'): UNSIGNED_LONG,
}


def _buf(x):
    """mpi4py buffer specs are either an array or an ``[array, datatype]`` pair."""
    if isinstance(x, (list, tuple)):
        return x[0]
    return x


class Comm(object):
    """mpi4py-shaped communicator over ``torch.distributed`` (or a single process).

    ``group=None`` and an uninitialised process group means a one-rank world, which
    is what every serial script and the single-GPU bench use.
    """

    def __init__(self, group=None, force_collectives=None):
        self._group = group
        self._timed = None      # [(start event, end event)] of the device all-reduces when timing is on (bench.py)
        # a one-rank world normally short-cuts every collective; PM_FORCE_COLLECTIVES=1 (or force_collectives=True)
        # sends them through the process group anyway -- how the one-GPU box exercises the RCCL code path
        import os
        self._force = (os.environ.get("PM_FORCE_COLLECTIVES", "0") == "1") if force_collectives is None else bool(force_collectives)

    def _solo(self):
        """No collective needed: a one-rank world (unless collectives are forced through the process group)."""
        return self.size == 1 and not (self._force and self._live())

    # -- topology ---------------------------------------------------------
    def _live(self):
        return dist is not None and dist.is_available() and dist.is_initialized()

    @property
    def rank(self):
        return dist.get_rank(self._group) if self._live() else 0

    @property
    def size(self):
        return dist.get_world_size(self._group) if self._live() else 1

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def _host_backend_ok(self):
        # gloo handles CPU tensors; with an nccl-only group host objects travel as
        # device tensors (see _host_allreduce).
        return dist.get_backend(self._group) != "nccl"

    # -- object / scalar API (lower-case mpi4py) ----------------------------
    def allreduce(self, value, op=SUM):
        """Sum a python scalar or ndarray over ranks (mpi4py's pickled allreduce:
        bsc_et.py:225,258,266,387,417)."""
        if self._solo():
            return value
        arr = np.asarray(value)
        is_int = arr.dtype.kind in "iub"
        work = np.ascontiguousarray(arr, dtype=np.int64 if is_int else np.float64)
        out = self._host_allreduce(work)
        if np.isscalar(value) or arr.ndim == 0:
            return int(out) if is_int else float(out)
        return out.astype(arr.dtype, copy=False)

    def _host_allreduce(self, work):
        t = torch.from_numpy(work.copy().reshape(-1))
        if self._host_backend_ok():
            dist.all_reduce(t, group=self._group)
            return t.numpy().reshape(work.shape)
        d = t.cuda()
        dist.all_reduce(d, group=self._group)
        return d.cpu().numpy().reshape(work.shape)

    def allgather(self, value):
        if self._solo():
            return [value]
        out = [None] * self.size
        dist.all_gather_object(out, value, group=self._group)
        return out

    def bcast(self, value, root=0):
        if self._solo():
            return value
        box = [value]
        dist.broadcast_object_list(box, src=root, group=self._group)
        return box[0]

    # -- buffer API (upper-case mpi4py) --------------------------------------
    def Allreduce(self, sendbuf, recvbuf, op=SUM):
        send, recv = _buf(sendbuf), _buf(recvbuf)
        if self._solo():
            recv[...] = send
            return
        recv[...] = self._host_allreduce(np.ascontiguousarray(send)).reshape(recv.shape)

    def Allgather(self, sendbuf, recvbuf):
        """Equal-count gather along the leading layout of ``recv`` (parallel.py:107)."""
        send, recv = _buf(sendbuf), _buf(recvbuf)
        if self._solo():
            recv[...] = np.asarray(send).reshape(recv.shape)
            return
        parts = self.allgather(np.ascontiguousarray(send))
        recv.reshape(-1)[...] = np.concatenate([p.reshape(-1) for p in parts])

    def Bcast(self, buf, root=0):
        arr = _buf(buf)
        if self._solo():
            return
        arr[...] = self.bcast(arr if self.rank == root else None, root=root)

    def Barrier(self):
        if not self._solo():
            dist.barrier(group=self._group)

    # -- device path: the one fused statistics exchange per EM step -------------
    def allreduce_device(self, tensor):
        """In-place sum-all-reduce of a device (or CPU) torch tensor.  On GPUs this
        is a single ncclAllReduce over xGMI; replaces the eight MPI calls of
        bsc_et.py:225-417."""
        if not self._solo() and tensor.is_cuda and dist.get_backend(self._group) == "gloo":
            # gloo moves host memory: stage the device buffer through the host (two ranks sharing one GPU in the
            # tests -- RCCL refuses two ranks per device; production groups are nccl)
            import time
            t0 = time.perf_counter()
            host = tensor.detach().cpu()
            dist.all_reduce(host, group=self._group)
            tensor.copy_(host)
            if self._timed is not None:
                self._timed.append((time.perf_counter() - t0) * 1e3)     # (host wall time incl. the staging copies)
            return tensor
        if not self._solo():
            if self._timed is not None and tensor.is_cuda:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                dist.all_reduce(tensor, group=self._group)
                b.record()
                self._timed.append((a, b))
            else:
                dist.all_reduce(tensor, group=self._group)
        return tensor

    def time_collectives(self, on=True):
        """Record HIP events around every device all-reduce from now on (``collective_times`` reads them)."""
        self._timed = [] if on else None

    def collective_times(self):
        """Milliseconds of every timed device all-reduce so far (synchronises); the list is cleared."""
        if not self._timed:
            return []
        torch.cuda.synchronize()
        out = [t if isinstance(t, float) else t[0].elapsed_time(t[1]) for t in self._timed]
        self._timed = []
        return out


COMM_WORLD = Comm()


def Wtime():
    import time
    return time.time()


# ---------------------------------------------------------------------------
def pprint(obj="", comm=COMM_WORLD, end='\n'):
    """Rank-0-only print (parallel.py:27-41)."""
    if comm.rank != 0:
        return
    if isinstance(obj, str):
        sys.stdout.write(obj + end)
    else:
        sys.stdout.write(repr(obj))
        sys.stdout.write(end)
        sys.stdout.flush()


def stride_data(N, balanced=False, comm=COMM_WORLD):
    """Contiguous block distribution of N items (parallel.py:44-84): the first
    ``N % size`` ranks own one extra item unless ``balanced``."""
    my_N = N // comm.size
    residue = N % comm.size
    if balanced:
        return my_N * comm.rank, my_N * (comm.rank + 1)
    if comm.rank < residue:
        size = my_N + 1
        first = size * comm.rank
    else:
        size = my_N
        first = size * comm.rank + residue
    return first, first + size


def _gather_ragged(my_sorted, axis, comm):
    parts = comm.allgather(np.ascontiguousarray(my_sorted))
    return np.concatenate(parts, axis=axis)


def allsort(my_array, axis=-1, kind='quicksort', order=None, comm=COMM_WORLD):
    """Collective numpy.sort (parallel.py:87-110).  The reference Allgathers equal
    counts; this version also accepts ragged shards."""
    if my_array.dtype not in typemap:
        raise TypeError("Dont know how to handle arrays of type %s" % my_array.dtype)
    my_sorted = np.sort(my_array, axis, kind, order)
    all_array = _gather_ragged(my_sorted, axis, comm)
    return np.sort(all_array, axis, 'mergesort', order)


def allargsort(my_array, axis=-1, kind='quicksort', order=None, comm=COMM_WORLD):
    """Collective numpy.argsort (parallel.py:113-135): argsort of the gathered
    per-rank argsort arrays, as the reference does."""
    if my_array.dtype not in typemap:
        raise TypeError("Dont know how to handle arrays of type %s" % my_array.dtype)
    my_sorted = np.argsort(my_array, axis, kind, order)
    all_array = _gather_ragged(my_sorted, axis, comm)
    return np.argsort(all_array, axis, kind, order)


def allmean(my_a, axis=None, dtype=None, out=None, comm=COMM_WORLD):
    """Collective numpy.mean (parallel.py:138-156)."""
    shape = my_a.shape
    if axis is None:
        N = comm.allreduce(my_a.size)
    else:
        N = comm.allreduce(shape[axis])
    my_sum = np.sum(my_a, axis, dtype)
    return comm.allreduce(my_sum) / N


def allsum(my_a, axis=None, dtype=None, out=None, comm=COMM_WORLD):
    """Collective numpy.sum (parallel.py:159-170)."""
    my_sum = np.sum(my_a, axis, dtype)
    return comm.allreduce(my_sum)
