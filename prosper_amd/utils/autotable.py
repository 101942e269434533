"""AutoTable: the reference's HDF5 result store (prosper/utils/autotable.py:35-278) without PyTables.

The reference writes ``result.h5`` through PyTables: one extendable array (EArray) per logged name, one row
per ``append`` -- ``h5.create_earray(h5.root, name, atom, (0,) + value.shape, filters=Filters(complevel=1,
complib='zlib', shuffle=True))`` (autotable.py:262-269).  Neither PyTables nor h5py is in the target image, but the
HDF5 C library is (``libhdf5.so``); this module drives it through ``ctypes`` and writes the same file layout:

    /<name>   dataset, shape (rows, *value.shape), maxshape (unlimited, *value.shape), chunked, shuffle + deflate(1),
              attributes CLASS='EARRAY', VERSION='1.1', TITLE='', EXTDIM=0   (what PyTables stamps on an EArray)
    /         attributes CLASS='GROUP', VERSION='1.0', TITLE='', PYTABLES_FORMAT_VERSION='2.1'

so PyTables (``h5.root.W[-1]``), h5py (``f['W'][-1]``), ``h5dump`` and Matlab's ``hdf5read`` open it like a file the
reference wrote.  ``read_last`` / ``read_table`` / ``table_names`` read such files back (also ones PyTables wrote:
the library undoes shuffle + zlib itself) -- the basis of ``resume_params`` and ``GSC.resume_init``.

Same call surface as upstream for numeric data: ``append``, ``append_all``, ``assign``, ``close``, context manager,
``compression_level``.  Strings (upstream: VLArray of VLStringAtom, autotable.py:270-276) are stored as
variable-length UTF-8 string datasets of the same extendable shape -- readable by h5py / ``h5dump`` / this module, but
NOT a PyTables VLArray (that is an H5T_VLEN of uint8 with a PSEUDOATOM attribute), so they carry no PyTables CLASS
stamp: PyTables lists them as unknown leaves instead of misreading them.  Numeric tables are what upstream logs.
"""
import ctypes
import ctypes.util
import glob
import os
import sys

import numpy as np

_UNLIMITED = ctypes.c_uint64(-1).value
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5S_SELECT_SET = 0
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING, _H5T_ENUM = 0, 1, 3, 8


class HDF5Unavailable(ImportError):
    pass


class _Lib(object):
    """libhdf5 through ctypes: prototypes for the two dozen calls used here."""

    _instance = None

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    @staticmethod
    def _candidates():
        env = os.environ.get("PM_HDF5_LIB")
        if env:
            yield env
        found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
        if found:
            yield found
        for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*",
                    "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*",
                    os.path.join(sys.prefix, "lib", "libhdf5.so*")):
            for p in sorted(glob.glob(pat)):
                yield p

    def __init__(self):
        lib, tried = None, []
        for cand in self._candidates():
            try:
                lib = ctypes.CDLL(cand)
                break
            except OSError as e:
                tried.append("%s (%s)" % (cand, e))
        if lib is None:
            raise HDF5Unavailable("no HDF5 C library found (set PM_HDF5_LIB=/path/to/libhdf5.so); tried: %s. "
                                  "StoreToNpz / resume_params keep the same one-array-per-name, one-row-per-step "
                                  "layout without it" % (", ".join(tried) or "nothing"))
        self.lib = lib
        c_uint = ctypes.c_uint
        maj, mnr, rel = c_uint(), c_uint(), c_uint()
        lib.H5open()
        lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
        self.version = (maj.value, mnr.value, rel.value)
        hid = ctypes.c_int64 if self.version >= (1, 10, 0) else ctypes.c_int
        self.hid = hid
        hs, hsp = ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)
        vp, cp, ci = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int
        protos = {
            "H5Fcreate": (hid, [cp, c_uint, hid, hid]), "H5Fopen": (hid, [cp, c_uint, hid]), "H5Fclose": (ci, [hid]),
            "H5Fflush": (ci, [hid, ci]),
            "H5Screate_simple": (hid, [ci, hsp, hsp]), "H5Screate": (hid, [ci]), "H5Sclose": (ci, [hid]),
            "H5Sget_simple_extent_ndims": (ci, [hid]), "H5Sget_simple_extent_dims": (ci, [hid, hsp, hsp]),
            "H5Sselect_hyperslab": (ci, [hid, ci, hsp, hsp, hsp, hsp]),
            "H5Pcreate": (hid, [hid]), "H5Pclose": (ci, [hid]), "H5Pset_chunk": (ci, [hid, ci, hsp]),
            "H5Pset_shuffle": (ci, [hid]), "H5Pset_deflate": (ci, [hid, c_uint]),
            "H5Dcreate2": (hid, [hid, cp, hid, hid, hid, hid, hid]), "H5Dopen2": (hid, [hid, cp, hid]),
            "H5Dclose": (ci, [hid]), "H5Dset_extent": (ci, [hid, hsp]), "H5Dget_space": (hid, [hid]),
            "H5Dget_type": (hid, [hid]), "H5Dwrite": (ci, [hid, hid, hid, hid, hid, vp]),
            "H5Dread": (ci, [hid, hid, hid, hid, hid, vp]),
            "H5Tcopy": (hid, [hid]), "H5Tset_size": (ci, [hid, ctypes.c_size_t]), "H5Tclose": (ci, [hid]),
            "H5Tget_class": (ci, [hid]), "H5Tget_size": (ctypes.c_size_t, [hid]), "H5Tget_sign": (ci, [hid]),
            "H5Tis_variable_str": (ci, [hid]), "H5Tset_cset": (ci, [hid, ci]),
            "H5Acreate2": (hid, [hid, cp, hid, hid, hid, hid]), "H5Awrite": (ci, [hid, hid, vp]),
            "H5Aclose": (ci, [hid]),
            "H5Gopen2": (hid, [hid, cp, hid]), "H5Gclose": (ci, [hid]), "H5Gget_info": (ci, [hid, vp]),
            "H5Lget_name_by_idx": (ctypes.c_ssize_t, [hid, cp, ci, ci, hs, cp, ctypes.c_size_t, hid]),
            "H5Lexists": (ci, [hid, cp, hid]), "H5Ldelete": (ci, [hid, cp, hid]),
            "H5Eset_auto2": (ci, [hid, vp, vp]),
        }
        for name, (res, args) in protos.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        # variable-length read buffers are handed back through H5Treclaim (HDF5 >= 1.12) or the deprecated
        # H5Dvlen_reclaim (absent from builds without deprecated symbols): resolved lazily, whichever exists
        self.vlen_reclaim = None
        for name in ("H5Treclaim", "H5Dvlen_reclaim"):
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = ci, [hid, hid, hid, vp]
                self.vlen_reclaim = fn
                break
        lib.H5Eset_auto2(0, None, None)             # errors are reported through return codes (-> exceptions below)

    def glob_id(self, symbol, *fallbacks):
        for name in (symbol,) + fallbacks:
            try:
                return self.hid.in_dll(self.lib, name).value
            except ValueError:
                continue
        raise IOError("HDF5: none of %s is exported by this libhdf5" % ", ".join((symbol,) + fallbacks))

    def native(self, dtype):
        names = {"float64": "H5T_NATIVE_DOUBLE_g", "float32": "H5T_NATIVE_FLOAT_g", "int64": "H5T_NATIVE_INT64_g",
                 "int32": "H5T_NATIVE_INT32_g", "int16": "H5T_NATIVE_INT16_g", "int8": "H5T_NATIVE_INT8_g",
                 "uint64": "H5T_NATIVE_UINT64_g", "uint32": "H5T_NATIVE_UINT32_g", "uint16": "H5T_NATIVE_UINT16_g",
                 "uint8": "H5T_NATIVE_UINT8_g", "bool": "H5T_NATIVE_UINT8_g"}
        key = np.dtype(dtype).name
        if key not in names:
            raise TypeError("unknown dtype '%s'" % key)
        return self.glob_id(names[key])


def _check(rc, what):
    if rc < 0:
        raise IOError("HDF5: %s failed" % what)
    return rc


def _dims(values):
    return (ctypes.c_uint64 * len(values))(*values)


class AutoTable(object):
    """Store data into HDF5 files (autotable.py:35-231): ``append(name, value)`` adds one row to the table ``name``."""

    def __init__(self, fname=None, compression_level=1):
        self._h5 = _Lib.get()
        self.warnings = True
        if fname is None:
            fname = self._guess_fname()
        self.fname = os.path.expanduser(fname)
        self.compression_level = compression_level
        self.tables = {}            # name -> [dataset id, row shape, numpy dtype or str, rows]
        self.types = {}
        self._file = _check(self._h5.lib.H5Fcreate(self.fname.encode(), _H5F_ACC_TRUNC, 0, 0), "H5Fcreate " + self.fname)
        root = _check(self._h5.lib.H5Gopen2(self._file, b"/", 0), "H5Gopen2 /")
        for k, v in (("CLASS", "GROUP"), ("VERSION", "1.0"), ("TITLE", ""), ("PYTABLES_FORMAT_VERSION", "2.1")):
            self._str_attr(root, k, v)
        self._h5.lib.H5Gclose(root)

    # -- context manager, close ----------------------------------------------------------------------
    def __enter__(self):
        return self

    def __exit__(self, *exc_info):
        self.close()

    def close(self):
        if self._file is None:
            return
        for rec in self.tables.values():
            self._h5.lib.H5Dclose(rec[0])
        self.tables = {}
        self._h5.lib.H5Fclose(self._file)
        self._file = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _guess_fname():
        base, _ = os.path.splitext(sys.argv[0] or "autotable")
        return base + ".h5"

    # -- attributes ----------------------------------------------------------------------------------
    def _str_attr(self, obj, name, value):
        lib = self._h5.lib
        data = value.encode() + b"\0"
        t = lib.H5Tcopy(self._h5.glob_id("H5T_C_S1_g"))
        lib.H5Tset_size(t, len(data))
        s = lib.H5Screate(0)                                      # H5S_SCALAR
        a = _check(lib.H5Acreate2(obj, name.encode(), t, s, 0, 0), "H5Acreate2 " + name)
        lib.H5Awrite(a, t, ctypes.c_char_p(data))
        lib.H5Aclose(a)
        lib.H5Sclose(s)
        lib.H5Tclose(t)

    def _int_attr(self, obj, name, value):
        lib = self._h5.lib
        t = self._h5.native(np.int32)
        s = lib.H5Screate(0)
        a = _check(lib.H5Acreate2(obj, name.encode(), t, s, 0, 0), "H5Acreate2 " + name)
        v = ctypes.c_int32(value)
        lib.H5Awrite(a, t, ctypes.byref(v))
        lib.H5Aclose(a)
        lib.H5Sclose(s)

    # -- tables --------------------------------------------------------------------------------------
    def _create_table(self, name, example):
        """A new extendable array whose row shape and datatype are those of ``example`` (autotable.py:234-278)."""
        lib = self._h5.lib
        if isinstance(example, str):
            shape, dt = (), str
            ftype = lib.H5Tcopy(self._h5.glob_id("H5T_C_S1_g"))
            lib.H5Tset_size(ftype, ctypes.c_size_t(-1).value)     # H5T_VARIABLE
            lib.H5Tset_cset(ftype, 1)                             # H5T_CSET_UTF8
            row_bytes, own_type = 16, True
        else:
            shape, dt = tuple(example.shape), example.dtype
            try:
                ftype = self._h5.native(dt)
            except TypeError:
                raise TypeError("Could not create table %s because of unknown dtype '%s'" % (name, dt))
            row_bytes, own_type = max(1, int(np.prod(shape, dtype=np.int64)) * dt.itemsize), False
        rank = 1 + len(shape)
        rows_per_chunk = max(1, min(1024, (64 * 1024) // row_bytes))
        space = _check(lib.H5Screate_simple(rank, _dims((0,) + shape), _dims((_UNLIMITED,) + shape)), "H5Screate_simple")
        plist = _check(lib.H5Pcreate(self._h5.glob_id("H5P_CLS_DATASET_CREATE_ID_g", "H5P_CLS_DATASET_CREATE_g")), "H5Pcreate")
        lib.H5Pset_chunk(plist, rank, _dims((rows_per_chunk,) + shape))
        if self.compression_level and dt is not str:
            lib.H5Pset_shuffle(plist)
            lib.H5Pset_deflate(plist, int(self.compression_level))
        ds = _check(lib.H5Dcreate2(self._file, name.encode(), ftype, space, 0, plist, 0), "H5Dcreate2 " + name)
        lib.H5Pclose(plist)
        lib.H5Sclose(space)
        if dt is not str:
            for k, v in (("CLASS", "EARRAY"), ("VERSION", "1.1"), ("TITLE", "")):
                self._str_attr(ds, k, v)
            self._int_attr(ds, "EXTDIM", 0)
        else:
            # strings are a plain extendable dataset of variable-length UTF-8 strings (h5dump / h5py read it as such).
            # PyTables' own VLArray(VLStringAtom) is an H5T_VLEN of uint8 + PSEUDOATOM, which this is NOT: no PyTables
            # CLASS stamp, so PyTables sees an unknown node instead of misreading it.
            self._str_attr(ds, "TITLE", "")
        self.tables[name] = [ds, shape, dt, 0, ftype if own_type else None]
        self.types[name] = str if dt is str else np.ndarray

    def _delete_table(self, name):
        rec = self.tables.pop(name)
        self._h5.lib.H5Dclose(rec[0])
        self._h5.lib.H5Ldelete(self._file, name.encode(), 0)
        del self.types[name]

    def append(self, name, value):
        """Append ``value`` (int, float, ndarray or str) as one row of the table ``name``; the first append creates
        the table with that row shape and datatype (autotable.py:87-127)."""
        if isinstance(value, np.ma.core.MaskedArray):
            value = value.data
        if not isinstance(value, str):
            if np.isscalar(value):
                value = np.asarray(value)
            if not isinstance(value, np.ndarray):
                raise TypeError("Don't know how to handle values of type '%s'" % type(value))
        if name not in self.tables:
            self._create_table(name, value)
        ds, shape, dt, rows, _ = self.tables[name]
        lib = self._h5.lib
        if dt is str:
            if not isinstance(value, str):
                raise TypeError('Wrong datatype "%s" for "%s" field' % (type(value), name))
            raw = ctypes.c_char_p(value.encode("utf-8"))
            buf, mtype = ctypes.byref(raw), self.tables[name][4]
        else:
            if isinstance(value, str) or tuple(value.shape) != shape:
                raise TypeError('Wrong datatype "%s" for "%s" field' % (getattr(value, "dtype", type(value)), name))
            try:
                arr = np.ascontiguousarray(value.astype(dt, casting="same_kind", copy=False))
            except TypeError:
                raise TypeError('Wrong datatype "%s" for "%s" field' % (value.dtype, name))
            if dt == np.bool_:
                arr = arr.view(np.uint8)
            buf, mtype = arr.ctypes.data_as(ctypes.c_void_p), self._h5.native(dt)
        rank = 1 + len(shape)
        _check(lib.H5Dset_extent(ds, _dims((rows + 1,) + shape)), "H5Dset_extent " + name)
        fspace = lib.H5Dget_space(ds)
        lib.H5Sselect_hyperslab(fspace, _H5S_SELECT_SET, _dims((rows,) + (0,) * len(shape)), None,
                                _dims((1,) + shape), None)
        mspace = lib.H5Screate_simple(rank, _dims((1,) + shape), None)
        rc = lib.H5Dwrite(ds, mtype, mspace, fspace, 0, buf)
        lib.H5Sclose(mspace)
        lib.H5Sclose(fspace)
        _check(rc, "H5Dwrite " + name)
        self.tables[name][3] = rows + 1
        lib.H5Fflush(self._file, 0)                    # upstream flushes the table after every row

    def append_all(self, valdict):
        for name, value in valdict.items():
            self.append(name, value)

    def assign(self, name, value):
        """Replace the table ``name`` by the rows of ``value`` (autotable.py:129-171)."""
        if isinstance(value, str):
            self.append(name, value)
            return
        if np.isscalar(value):
            value = np.asarray(value).reshape((1,))
        if not isinstance(value, np.ndarray):
            raise TypeError("Don't know how to handle values of type '%s'" % type(value))
        if name in self.tables:
            if self.warnings:
                print("Warning! The previous data with key %s is being overwritten" % name)
            self._delete_table(name)
        for ii in range(value.shape[0]):
            self.append(name, value[ii])


# ---- reading -----------------------------------------------------------------------------------------
class _G_info(ctypes.Structure):
    _fields_ = [("storage_type", ctypes.c_int), ("nlinks", ctypes.c_uint64), ("max_corder", ctypes.c_int64),
                ("mounted", ctypes.c_int)]


def _open(fname):
    h5 = _Lib.get()
    f = h5.lib.H5Fopen(os.path.expanduser(str(fname)).encode(), _H5F_ACC_RDONLY, 0)
    if f < 0:
        raise IOError("HDF5: cannot open %s" % fname)
    return h5, f


def table_names(fname):
    """Names of the datasets in the root group of ``fname``."""
    h5, f = _open(fname)
    try:
        g = h5.lib.H5Gopen2(f, b"/", 0)
        info = _G_info()
        _check(h5.lib.H5Gget_info(g, ctypes.byref(info)), "H5Gget_info")
        names = []
        buf = ctypes.create_string_buffer(1024)
        for i in range(info.nlinks):
            n = h5.lib.H5Lget_name_by_idx(g, b".", 0, 0, i, buf, 1024, 0)      # H5_INDEX_NAME, H5_ITER_INC
            if n > 0:
                names.append(buf.value.decode())
        h5.lib.H5Gclose(g)
        return names
    finally:
        h5.lib.H5Fclose(f)


def _np_dtype(h5, ftype):
    cls, size = h5.lib.H5Tget_class(ftype), h5.lib.H5Tget_size(ftype)
    if cls == _H5T_FLOAT:
        return np.dtype("f%d" % size)
    if cls in (_H5T_INTEGER, _H5T_ENUM):
        signed = cls == _H5T_INTEGER and h5.lib.H5Tget_sign(ftype) != 0
        return np.dtype("%s%d" % ("i" if signed else "u", size))
    if cls == _H5T_STRING and h5.lib.H5Tis_variable_str(ftype) > 0:
        return str
    raise TypeError("unsupported HDF5 datatype class %d" % cls)


def read_table(fname, name, rows=None):
    """The table ``name`` of ``fname`` as an array of shape (rows, ...); ``rows=(lo, hi)`` reads a row range (negative
    values count from the end, as in a slice)."""
    h5, f = _open(fname)
    lib = h5.lib
    try:
        ds = lib.H5Dopen2(f, name.encode(), 0)
        if ds < 0:
            raise KeyError("no table '%s' in %s" % (name, fname))
        fspace = lib.H5Dget_space(ds)
        rank = lib.H5Sget_simple_extent_ndims(fspace)
        dims = (ctypes.c_uint64 * max(rank, 1))()
        lib.H5Sget_simple_extent_dims(fspace, dims, None)
        shape = tuple(int(d) for d in dims[:rank])
        ftype = lib.H5Dget_type(ds)
        dt = _np_dtype(h5, ftype)
        total = shape[0] if rank else 1
        lo, hi, _ = slice(*(rows if rows is not None else (None, None))).indices(total)
        n = max(0, hi - lo)
        if rank == 0:
            out_shape, mspace = (), lib.H5Screate(0)
        else:
            out_shape = (n,) + shape[1:]
            lib.H5Sselect_hyperslab(fspace, _H5S_SELECT_SET, _dims((lo,) + (0,) * (rank - 1)), None,
                                    _dims(tuple(max(1, v) for v in out_shape)), None)
            mspace = lib.H5Screate_simple(rank, _dims(tuple(max(1, v) for v in out_shape)), None)
        if dt is str:
            count = int(np.prod(out_shape, dtype=np.int64)) if out_shape else 1
            ptrs = (ctypes.c_char_p * max(count, 1))()
            if count:
                _check(lib.H5Dread(ds, ftype, mspace, fspace, 0, ptrs), "H5Dread " + name)
            out = np.array([(p or b"").decode("utf-8") for p in ptrs[:count]], dtype=object).reshape(out_shape)
            if count and h5.vlen_reclaim is not None:
                h5.vlen_reclaim(ftype, mspace, 0, ptrs)
        else:
            out = np.empty(out_shape, dtype=dt)
            if out.size:
                _check(lib.H5Dread(ds, h5.native(dt), mspace, fspace, 0, out.ctypes.data_as(ctypes.c_void_p)),
                       "H5Dread " + name)
        lib.H5Sclose(mspace)
        lib.H5Tclose(ftype)
        lib.H5Sclose(fspace)
        lib.H5Dclose(ds)
        return out
    finally:
        lib.H5Fclose(f)


def read_last(fname, names=None):
    """Last row of every (or every named) table: the parameter dict to resume a run from (``h5.root.W[steps-1]``,
    gsc_et.py:113-160)."""
    out = {}
    for name in (table_names(fname) if names is None else names):
        try:
            last = read_table(fname, name, rows=(-1, None))
        except (KeyError, TypeError):
            if names is None:
                continue
            raise
        if last.shape[0] == 0:
            continue
        row = last[0]
        out[name] = row.item() if np.ndim(row) == 0 else np.array(row)
    return out
