"""PyTables stand-in: the reference imports it at module level (gsc_et.py:28,
autotable.py:32) but the hot path never opens a file."""


def open_file(*a, **k):
    raise RuntimeError("tables stub: HDF5 storage is not available here")


openFile = open_file
