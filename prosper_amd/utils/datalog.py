"""Minimal ``dlog`` sink with the reference's call surface.

The hot path logs ``N``, ``L`` (free energy), ``N_use`` (bsc_et.py:261,267,436) and
every parameter / annealing value per step (camodels/__init__.py:190-191).  This keeps
prosper/utils/datalog.py's policy model: rank-0 only (:181,193), ordered
``(tblname, handler)`` policy with the ``'*'`` wildcard (:153-163), ``append`` /
``append_all`` / ``set_handler`` / ``remove_handler`` / ``ignored``.  ``StoreToH5`` writes
the reference's ``result.h5`` layout through ``utils/autotable.py`` (HDF5 C library via
ctypes: PyTables is not in the image); ``StoreToNpz`` is the dependency-free alternative.
"""
import os
from time import strftime

import numpy as np

from .parallel import pprint, COMM_WORLD


class DataHandler(object):
    """Base class for handlers (datalog.py:20-43)."""

    def register(self, tblname):
        pass

    def append(self, tblname, value):
        raise NotImplementedError

    def append_all(self, valdict):
        for key, val in valdict.items():
            self.append(key, val)

    def remove(self, tblname):
        pass

    def close(self):
        pass


class TextPrinter(DataHandler):
    def append(self, tblname, value):
        pprint("  %8s = %s " % (tblname, value))

    def append_all(self, valdict):
        for (name, val) in valdict.items():
            pprint("  %8s = %s \n" % (name, val), end="")


class StoreToTxt(DataHandler):
    def __init__(self, destination):
        self.txt_file = open(destination, 'w')

    def append(self, tblname, value):
        self.txt_file.write("%s = %s\n" % (tblname, value))

    def close(self):
        self.txt_file.close()


class StoreInMemory(DataHandler):
    """Keeps every appended value in ``self.tables[name]`` (one row per EM step, the
    layout AutoTable gives result.h5)."""

    def __init__(self):
        self.tables = {}

    def append(self, tblname, value):
        self.tables.setdefault(tblname, []).append(np.array(value, copy=True))


class StoreToNpz(StoreInMemory):
    """In-memory rows written as one ``.npz`` on close()."""

    def __init__(self, destination):
        StoreInMemory.__init__(self)
        self.destination = destination

    def close(self):
        np.savez(self.destination, **{k: np.stack(v) for k, v in self.tables.items()})


def resume_params(destination, names=None):
    """Last logged row of every table of a result file as a parameter dict -- the ``lparams`` to hand to ``EM`` to
    continue a run (``h5.root.W[steps-1]``, gsc_et.py:113-160).  ``destination``: a ``result.h5`` written by
    ``StoreToH5`` / ``AutoTable`` (or by the reference through PyTables), or a ``StoreToNpz`` file.  ``names`` restricts
    the keys (e.g. ``('W', 'pi', 'sigma')``)."""
    path = str(destination)
    if path.endswith((".h5", ".hdf5")) or (not path.endswith(".npz") and not os.path.exists(path + ".npz")):
        from . import autotable
        return autotable.read_last(path, names)
    with np.load(path if path.endswith(".npz") else path + ".npz") as f:
        keys = [k for k in f.files if names is None or k in names]
        out = {}
        for k in keys:
            last = f[k][-1]
            out[k] = last.item() if last.ndim == 0 else np.array(last)
    return out


class StoreToH5(DataHandler):
    """``result.h5`` writer of the reference (datalog.py:53-93): every logged name becomes an extendable array with
    one row per EM step, through ``AutoTable`` -- here on the HDF5 C library via ctypes (utils/autotable.py; PyTables
    and h5py are not in the image), same file layout.  ``destination``: a file name, an ``AutoTable``, or None (the
    first handler's table, else a file named after the running script), as upstream.  Rank 0 writes."""
    default_autotbl = None

    def __init__(self, destination=None):
        from .autotable import AutoTable
        self.destination = destination
        self.autotbl = None
        if COMM_WORLD.rank != 0:
            return
        if isinstance(destination, AutoTable):
            self.autotbl = destination
        elif isinstance(destination, str):
            self.autotbl = AutoTable(destination)
        elif destination is None:
            self.autotbl = StoreToH5.default_autotbl if StoreToH5.default_autotbl is not None else AutoTable()
        else:
            raise TypeError("Expects an AutoTable instance or a string as argument")
        if StoreToH5.default_autotbl is None:
            StoreToH5.default_autotbl = self.autotbl

    def __repr__(self):
        return "StoreToH5 into file %s" % self.destination

    def append(self, tblname, value):
        if self.autotbl is not None:
            self.autotbl.append(tblname, np.asarray(value) if not isinstance(value, str) else value)

    def append_all(self, valdict):
        for key, val in valdict.items():
            self.append(key, val)

    def close(self):
        if self.autotbl is not None:
            self.autotbl.close()
            if StoreToH5.default_autotbl is self.autotbl:
                StoreToH5.default_autotbl = None


class DataLog(object):
    def __init__(self, comm=COMM_WORLD):
        self.comm = comm
        self.policy = []
        self._lookup_cache = {}

    def _lookup(self, tblname):
        if tblname in self._lookup_cache:
            return self._lookup_cache[tblname]
        handlers = [h for (t, h) in self.policy if t == tblname or t == "*"]
        self._lookup_cache[tblname] = handlers
        return handlers

    def progress(self, message, completed=None):
        if self.comm.rank != 0:
            return
        if completed is None:
            print("[%s] %s" % (strftime("%H:%M:%S"), message))
        else:
            totlen = 65 - len(message)
            barlen = int(totlen * completed)
            print("[%s] %s [%s%s]" % (strftime("%H:%M:%S"), message, "*" * barlen,
                                      "-" * (totlen - barlen)))

    def append(self, tblname, value):
        if self.comm.rank != 0:
            return
        for h in self._lookup(tblname):
            h.append(tblname, value)

    def append_all(self, valdict):
        if self.comm.rank != 0:
            return
        all_handlers = []
        for tblname in valdict:
            for h in self._lookup(tblname):
                if h not in all_handlers:
                    all_handlers.append(h)
        for handler in all_handlers:
            handler.append_all({t: v for t, v in valdict.items() if handler in self._lookup(t)})

    def ignored(self, tblname):
        return self._lookup(tblname) == []

    def set_handler(self, tblname, handler_class, *args, **kargs):
        if self.comm.rank != 0:
            return
        if not issubclass(handler_class, DataHandler):
            raise TypeError("handler_class must be a subclass of DataHandler ")
        handler = handler_class(*args, **kargs)
        handler.register(tblname)
        if isinstance(tblname, str):
            self.policy.append((tblname, handler))
        elif hasattr(tblname, '__iter__'):
            for t in tblname:
                self.policy.append((t, handler))
        else:
            raise TypeError('Table-name must be a string (or a list of strings)')
        # the reference never invalidates its lookup cache here (datalog.py:234-254), so
        # handlers set after the first append are ignored upstream; we invalidate.
        self._lookup_cache = {}
        return handler

    def remove_handler(self, handler):
        if self.comm.rank != 0:
            return
        if not isinstance(handler, DataHandler):
            raise ValueError("Please provide valid DataHandler object.")
        self.policy = [(t, h) for (t, h) in self.policy if h is not handler]
        handler.close()
        self._lookup_cache = {}

    def close(self):
        if self.comm.rank != 0:
            return
        seen = []
        for _, h in self.policy:
            if h not in seen:
                seen.append(h)
                h.close()


dlog = DataLog()
