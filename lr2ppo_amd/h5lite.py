"""The slice of h5py the reference's data readers use, on the HDF5 C library itself (ctypes, no h5py).

The reference opens `LRMovieNet/clean_feat.h5` and the LETOR `train.h5` / `test.h5` with h5py and touches exactly this surface
(finetune/ppo.py:92,118-125; pointwise.py:85,137-143; reward_pair_dataloader.py:95,181-187; ppo_trad.py:70-74;
pointwise_trad.py:94-98,107; datasets_trad/convert_to_h5py.py:41-43):

    h5py.File(path, 'r' | 'w')        f.keys()   len(f)   name in f   f[name] -> Group | Dataset   f.close() / with
    group[name]                        dataset[:]   dataset[()]   dataset[index]   dataset.shape / .dtype / len()
    f.create_dataset(name, data=array)  f.create_group(name)                    (write side: the conversion script's calls)

This image has no h5py, but it does have libhdf5 (HDF5 1.10, /opt/conda/lib); the readers fall back on this module when
`import h5py` fails (`open_file`), so real HDF5 files -- made by h5py anywhere else, or by `create_dataset` here -- are read
through the same library h5py wraps.  Keys come back in increasing name order (H5_INDEX_NAME / H5_ITER_INC), which is h5py's
iteration order for files written without link-creation-order tracking (its default).  Numeric datasets only (integer and IEEE
float of 1-8 bytes, any byte order, any storage layout or filter the library build supports): that is everything the reference
stores.  Host-side data plumbing; nothing here is on the GPU path.
"""
import ctypes
import ctypes.util
import glob
import os
import threading

import numpy as np

_hid = ctypes.c_int64
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5I_GROUP, _H5I_DATASET = 2, 5
_H5T_INTEGER, _H5T_FLOAT = 0, 1
_NATIVE = {"f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g", "i1": "H5T_NATIVE_INT8_g", "i2": "H5T_NATIVE_INT16_g",
           "i4": "H5T_NATIVE_INT32_g", "i8": "H5T_NATIVE_INT64_g", "u1": "H5T_NATIVE_UINT8_g", "u2": "H5T_NATIVE_UINT16_g",
           "u4": "H5T_NATIVE_UINT32_g", "u8": "H5T_NATIVE_UINT64_g"}
_ITER_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int64, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)   # H5L_iterate_t
_lib = None
_lock = threading.RLock()          # libhdf5 is not built thread-safe by default: one call at a time, like h5py's global lock


def _candidates():
    env = os.environ.get("LR2_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for pat in ("/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/lib64/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*", "/opt/conda/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat), key=len):
            yield p


def _load():
    global _lib
    if _lib is not None:
        return _lib
    errors = []
    for path in _candidates():
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:
            errors.append(f"{path}: {e}")
            continue
        if not hasattr(lib, "H5Dread"):
            continue
        sig = {
            "H5open": (ctypes.c_int, []), "H5get_libversion": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint)] * 3),
            "H5Eset_auto2": (ctypes.c_int, [_hid, ctypes.c_void_p, ctypes.c_void_p]),
            "H5Fopen": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]), "H5Fcreate": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
            "H5Fclose": (ctypes.c_int, [_hid]), "H5Fflush": (ctypes.c_int, [_hid, ctypes.c_int]),
            "H5Oopen": (_hid, [_hid, ctypes.c_char_p, _hid]), "H5Oclose": (ctypes.c_int, [_hid]), "H5Iget_type": (ctypes.c_int, [_hid]),
            "H5Lexists": (ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
            "H5Gget_info": (ctypes.c_int, [_hid, ctypes.c_void_p]), "H5Gcreate2": (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid]),
            "H5Lget_name_by_idx": (ctypes.c_ssize_t, [_hid, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p,
                                                      ctypes.c_size_t, _hid]),
            "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]),
            "H5Dcreate2": (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
            "H5Dread": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            "H5Dwrite": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            "H5Sget_simple_extent_ndims": (ctypes.c_int, [_hid]),
            "H5Sget_simple_extent_dims": (ctypes.c_int, [_hid, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
            "H5Screate_simple": (_hid, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
            "H5Screate": (_hid, [ctypes.c_int]), "H5Sclose": (ctypes.c_int, [_hid]),
            "H5Tget_class": (ctypes.c_int, [_hid]), "H5Tget_size": (ctypes.c_size_t, [_hid]), "H5Tget_sign": (ctypes.c_int, [_hid]),
            "H5Tget_native_type": (_hid, [_hid, ctypes.c_int]), "H5Tclose": (ctypes.c_int, [_hid]),
            "H5Pcreate": (_hid, [_hid]), "H5Pset_fclose_degree": (ctypes.c_int, [_hid, ctypes.c_int]), "H5Pclose": (ctypes.c_int, [_hid]),
            "H5Pset_chunk": (ctypes.c_int, [_hid, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
            "H5Pset_deflate": (ctypes.c_int, [_hid, ctypes.c_uint]), "H5Zfilter_avail": (ctypes.c_int, [ctypes.c_int]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.H5open() < 0:
            errors.append(f"{path}: H5open failed")
            continue
        lib.H5Eset_auto2(0, None, None)          # no error stack on stderr: failures become Python exceptions below
        lib._path = path
        _lib = lib
        return lib
    raise ImportError("lr2ppo_amd.h5lite: no HDF5 C library found (set LR2_HDF5_LIB to a libhdf5.so, or install h5py)"
                      + ("".join("\n  " + e for e in errors)))


def library():
    """-> (path, (major, minor, release)) of the HDF5 library in use."""
    lib = _load()
    v = [ctypes.c_uint() for _ in range(3)]
    lib.H5get_libversion(*[ctypes.byref(x) for x in v])
    return lib._path, tuple(int(x.value) for x in v)


def _native_id(dtype):
    key = np.dtype(dtype).str.lstrip("<>=|")
    if key not in _NATIVE:
        raise TypeError(f"h5lite stores numeric arrays only (got dtype {np.dtype(dtype)})")
    return _hid.in_dll(_load(), _NATIVE[key]).value


class Dataset:
    """An open dataset.  `ds[:]`, `ds[()]` and `ds[...]` read it whole; any other index is applied to the whole array."""

    def __init__(self, did, name, file=None):
        self._id, self.name, self.file = did, name, file
        lib = _load()
        with _lock:
            space = lib.H5Dget_space(did)
            nd = lib.H5Sget_simple_extent_ndims(space)
            dims = (ctypes.c_uint64 * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(space, dims, None)
            lib.H5Sclose(space)
            ftype = lib.H5Dget_type(did)
            cls, size, sign = lib.H5Tget_class(ftype), lib.H5Tget_size(ftype), lib.H5Tget_sign(ftype)
            lib.H5Tclose(ftype)
        if nd < 0:
            raise OSError(f"{name}: not a simple dataspace")
        self.shape = tuple(int(dims[i]) for i in range(nd))
        if cls == _H5T_FLOAT and size in (4, 8):
            self.dtype = np.dtype(f"f{size}")
        elif cls == _H5T_INTEGER and size in (1, 2, 4, 8):
            self.dtype = np.dtype(("i" if sign else "u") + str(size))
        else:
            raise TypeError(f"{name}: h5lite reads integer / IEEE float datasets only (HDF5 class {cls}, {size} bytes)")

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of a scalar dataset")
        return self.shape[0]

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def _read(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if out.size:
            with _lock:
                rc = _load().H5Dread(self._id, _native_id(self.dtype), 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p))
            if rc < 0:
                raise OSError(f"{self.name}: H5Dread failed")
        return out

    def __getitem__(self, index):
        whole = self._read()
        if index is Ellipsis or (isinstance(index, tuple) and len(index) == 0):
            return whole if self.shape else whole[()]
        return whole[index]

    def __array__(self, dtype=None, copy=None):
        a = self._read()
        return a if dtype is None else a.astype(dtype)

    def __del__(self):
        try:
            if self._id > 0 and _lib is not None:
                with _lock:
                    _lib.H5Oclose(self._id)
        except Exception:
            pass
        self._id = 0


class Group:
    def __init__(self, gid, name, file=None):
        self._id, self.name, self.file = gid, name, file       # members hold their File: it stays open while they are in use

    # -- reading ---------------------------------------------------------------------------
    def __len__(self):
        info = ctypes.create_string_buffer(64)                    # H5G_info_t: { int storage_type; hsize_t nlinks; ... }
        with _lock:
            if _load().H5Gget_info(self._id, info) < 0:
                raise OSError(f"{self.name}: H5Gget_info failed")
        return int(ctypes.c_uint64.from_buffer(info, 8).value)

    def keys(self):
        """Member names in increasing name order.  One H5Literate pass (the by-index lookup walks the group's B-tree from the start
        for every index: quadratic on the thousands of items of a real clean_feat.h5); by-index only where that symbol is missing."""
        lib, out = _load(), []
        iterate = getattr(lib, "H5Literate", None) or getattr(lib, "H5Literate1", None)
        with _lock:
            if iterate is not None:
                @_ITER_CB
                def visit(_group, name, _info, _data):
                    out.append(name.decode("utf-8"))
                    return 0
                iterate.restype = ctypes.c_int
                iterate.argtypes = [_hid, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), _ITER_CB, ctypes.c_void_p]
                if iterate(self._id, 0, 0, None, visit, None) < 0:                         # H5_INDEX_NAME, H5_ITER_INC
                    raise OSError(f"{self.name}: H5Literate failed")
                return out
            for i in range(len(self)):
                n = lib.H5Lget_name_by_idx(self._id, b".", 0, 0, i, None, 0, 0)
                if n < 0:
                    raise OSError(f"{self.name}: H5Lget_name_by_idx failed")
                buf = ctypes.create_string_buffer(n + 1)
                lib.H5Lget_name_by_idx(self._id, b".", 0, 0, i, buf, n + 1, 0)
                out.append(buf.value.decode("utf-8"))
        return out

    def __iter__(self):
        return iter(self.keys())

    def __contains__(self, name):
        if not isinstance(name, str) or not name:
            return False
        lib, at = _load(), ""
        with _lock:
            for part in name.strip("/").split("/"):               # H5Lexists wants every intermediate link to exist
                at = f"{at}/{part}" if at else part
                if lib.H5Lexists(self._id, at.encode(), 0) <= 0:
                    return False
        return True

    def __getitem__(self, name):
        if not isinstance(name, str):
            raise TypeError("group members are addressed by name")
        lib = _load()
        with _lock:
            oid = lib.H5Oopen(self._id, name.encode(), 0)
            if oid < 0:
                raise KeyError(f"Unable to open object (object '{name}' doesn't exist)")
            kind = lib.H5Iget_type(oid)
        full = f"{self.name.rstrip('/')}/{name}"
        if kind == _H5I_GROUP:
            return Group(oid, full, self.file)
        if kind == _H5I_DATASET:
            return Dataset(oid, full, self.file)
        with _lock:
            lib.H5Oclose(oid)
        raise TypeError(f"{full}: neither a group nor a dataset")

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    # -- writing (what datasets_trad/convert_to_h5py.py:41-43 and a feature-extraction script call) ----------------------
    def create_group(self, name):
        with _lock:
            gid = _load().H5Gcreate2(self._id, name.encode(), 0, 0, 0)
        if gid < 0:
            raise ValueError(f"Unable to create group '{name}' (exists already, or the file is read-only)")
        return Group(gid, f"{self.name.rstrip('/')}/{name}", self.file)

    def create_dataset(self, name, data=None, shape=None, dtype=None, chunks=None, compression=None, compression_opts=4):
        """h5py's create_dataset for numeric arrays.  chunks: a chunk shape (or True: one chunk per leading index);
        compression="gzip" (needs chunks; h5py picks them itself, here True is implied) with level `compression_opts`."""
        lib = _load()
        arr = np.array(np.zeros(shape, dtype or "f4") if data is None else data, dtype=dtype, order="C")   # 0-d stays 0-d
        tid = _native_id(arr.dtype)
        dims = (ctypes.c_uint64 * max(arr.ndim, 1))(*arr.shape)
        if compression not in (None, "gzip"):
            raise ValueError("h5lite writes compression=None or 'gzip'")
        if compression and not chunks:
            chunks = True
        dcpl = 0
        with _lock:
            if chunks and arr.ndim and arr.size:
                cshape = ((1,) + arr.shape[1:]) if chunks is True else tuple(int(c) for c in chunks)
                if len(cshape) != arr.ndim or min(cshape) < 1:
                    raise ValueError("chunks must give one positive extent per dimension")
                dcpl = lib.H5Pcreate(_hid.in_dll(lib, "H5P_CLS_DATASET_CREATE_ID_g").value)
                lib.H5Pset_chunk(dcpl, arr.ndim, (ctypes.c_uint64 * arr.ndim)(*cshape))
                if compression:
                    if lib.H5Zfilter_avail(1) <= 0:                      # H5Z_FILTER_DEFLATE
                        lib.H5Pclose(dcpl)
                        raise ValueError("this HDF5 library was built without the gzip filter")
                    lib.H5Pset_deflate(dcpl, int(compression_opts))
            space = lib.H5Screate_simple(arr.ndim, dims, None) if arr.ndim else lib.H5Screate(0)      # 0: H5S_SCALAR
            did = lib.H5Dcreate2(self._id, name.encode(), tid, space, 0, dcpl, 0)
            lib.H5Sclose(space)
            if dcpl:
                lib.H5Pclose(dcpl)
            if did < 0:
                raise ValueError(f"Unable to create dataset '{name}' (exists already, or the file is read-only)")
            if arr.size and lib.H5Dwrite(did, tid, 0, 0, 0, arr.ctypes.data_as(ctypes.c_void_p)) < 0:
                lib.H5Oclose(did)
                raise OSError(f"{name}: H5Dwrite failed")
        return Dataset(did, f"{self.name.rstrip('/')}/{name}", self.file)

    def __del__(self):
        try:
            if self._id > 0 and _lib is not None and not isinstance(self, File):
                with _lock:
                    _lib.H5Oclose(self._id)
        except Exception:
            pass
        if not isinstance(self, File):
            self._id = 0


class File(Group):
    """h5py.File(path, mode): 'r' (default) opens read-only, 'w' creates / truncates."""

    def __init__(self, path, mode="r"):
        lib = _load()
        p = os.fspath(path).encode()
        with _lock:
            # close degree STRONG: close() ends every member still open, like h5py's File.close(); with the default (weak)
            # degree a lingering member would keep the file -- and its write intent -- alive behind a later read-only open
            fapl = lib.H5Pcreate(_hid.in_dll(lib, "H5P_CLS_FILE_ACCESS_ID_g").value)
            lib.H5Pset_fclose_degree(fapl, 3)
            try:
                if mode == "r":
                    if not os.path.exists(path):
                        raise FileNotFoundError(f"Unable to open file (unable to open file: name = '{path}', No such file or directory)")
                    fid = lib.H5Fopen(p, _H5F_ACC_RDONLY, fapl)
                elif mode == "w":
                    fid = lib.H5Fcreate(p, _H5F_ACC_TRUNC, 0, fapl)
                else:
                    raise ValueError("h5lite.File modes: 'r', 'w'")
            finally:
                lib.H5Pclose(fapl)
        if fid < 0:
            raise OSError(f"Unable to open file '{path}' (not an HDF5 file, or not accessible)")
        super().__init__(fid, "/", self)
        self.filename, self.mode = os.fspath(path), mode

    def flush(self):
        with _lock:
            _load().H5Fflush(self._id, 1)

    def close(self):
        if self._id > 0:
            with _lock:
                _load().H5Fclose(self._id)           # STRONG close degree: members still open are closed with it
            self._id = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_file(path, mode="r"):
    """What the readers call where the reference calls `h5py.File(path, mode)`: h5py when it is installed, this module otherwise."""
    try:
        import h5py
    except ImportError:
        return File(path, mode)
    return h5py.File(path, mode)
