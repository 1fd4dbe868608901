"""Minimal HDF5 reader/writer over the system libhdf5 (ctypes) — just what the replay buffer needs.

The reference appends finished games to `Self_Play_Data.h5` through h5py (Self_Play.py:178-208, file created in
<Game>/main.py:83-86 with libver="latest"): a `game_stats` u32[6] dataset plus `boards_k` / `policies_k` / `values_k`
datasets created with `maxshape=(None, ...)`.  h5py is not installed in this image but libhdf5 1.10 is, so this module
binds the dozen C entry points involved.  Storage layout details (chunk shape) are invisible to readers such as
Dataloader.py:83-110; names, dtypes and shapes are what matter and are kept.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_LIB = None
hid_t = C.c_int64
hsize_t = C.c_uint64
H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_int)]


def available():
    try:
        _lib()
        return True
    except OSError:
        return False


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    cands = [os.environ.get("GAZ_LIBHDF5"), ctypes.util.find_library("hdf5"), "/opt/conda/lib/libhdf5.so", "libhdf5.so", "libhdf5_serial.so"]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            L = C.CDLL(c)
            break
        except OSError as e:
            err = e
    else:
        raise OSError(f"libhdf5 not found ({err})")
    L.H5open()
    for f, res, args in [
        ("H5Pcreate", hid_t, [hid_t]), ("H5Pclose", C.c_int, [hid_t]), ("H5Pset_libver_bounds", C.c_int, [hid_t, C.c_int, C.c_int]),
        ("H5Pset_chunk", C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]), ("H5Fclose", C.c_int, [hid_t]),
        ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), ("H5Sclose", C.c_int, [hid_t]),
        ("H5Sget_simple_extent_ndims", C.c_int, [hid_t]), ("H5Sget_simple_extent_dims", C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]),
        ("H5Dclose", C.c_int, [hid_t]), ("H5Dget_space", hid_t, [hid_t]), ("H5Dget_type", hid_t, [hid_t]),
        ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        ("H5Tget_class", C.c_int, [hid_t]), ("H5Tget_size", C.c_size_t, [hid_t]), ("H5Tget_sign", C.c_int, [hid_t]), ("H5Tclose", C.c_int, [hid_t]),
        ("H5Gget_info", C.c_int, [hid_t, C.POINTER(_GInfo)]), ("H5Gopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Gclose", C.c_int, [hid_t]),
        ("H5Pset_create_intermediate_group", C.c_int, [hid_t, C.c_uint]), ("H5Eset_auto2", C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
        ("H5Lexists", C.c_int, [hid_t, C.c_char_p, hid_t]), ("H5Fflush", C.c_int, [hid_t, C.c_int]),
        ("H5Lget_name_by_idx", C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        ("H5Pset", C.c_int, [hid_t, C.c_char_p, C.c_void_p]), ("H5Ldelete", C.c_int, [hid_t, C.c_char_p, hid_t]),
        ("H5Eget_auto2", C.c_int, [hid_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ]:
        fn = getattr(L, f); fn.restype = res; fn.argtypes = args
    _LIB = L
    return L


def _g(name):
    return hid_t.in_dll(_lib(), name).value


class _quiet:
    """libhdf5's automatic error printing switched off for the calls inside the block only (a failure there is reported through an OSError
    or is an expected probe); the process-wide handler is put back afterwards, so other users of libhdf5 keep their diagnostics."""

    def __enter__(self):
        L = _lib()
        self.func, self.data = C.c_void_p(), C.c_void_p()
        L.H5Eget_auto2(0, C.byref(self.func), C.byref(self.data))
        L.H5Eset_auto2(0, None, None)

    def __exit__(self, *a):
        _lib().H5Eset_auto2(0, self.func, self.data)
        return False


def superblock_status_flags(path):
    """The "file consistency flags" byte of an HDF5 version 2 / 3 superblock at offset 0 (bit 0: open for write, bit 2: SWMR write) — set
    while a writer has the file open and left set by a writer that was killed; None when the file has no such superblock."""
    try:
        with open(path, "rb") as f:
            head = f.read(12)
    except OSError:
        return None
    if len(head) < 12 or head[:8] != b"\x89HDF\r\n\x1a\n" or head[8] < 2:
        return None
    return head[11]


_FILE_TYPES = {np.dtype(np.int8): "H5T_STD_I8LE_g", np.dtype(np.uint8): "H5T_STD_U8LE_g", np.dtype(np.int32): "H5T_STD_I32LE_g",
               np.dtype(np.uint32): "H5T_STD_U32LE_g", np.dtype(np.int64): "H5T_STD_I64LE_g", np.dtype(np.float32): "H5T_IEEE_F32LE_g",
               np.dtype(np.float64): "H5T_IEEE_F64LE_g"}
_MEM_TYPES = {np.dtype(np.int8): "H5T_NATIVE_INT8_g", np.dtype(np.uint8): "H5T_NATIVE_UINT8_g", np.dtype(np.int32): "H5T_NATIVE_INT32_g",
              np.dtype(np.uint32): "H5T_NATIVE_UINT32_g", np.dtype(np.int64): "H5T_NATIVE_INT64_g", np.dtype(np.float32): "H5T_NATIVE_FLOAT_g",
              np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g"}


class H5File:
    """`with H5File(path, "w" | "r" | "r+") as f:`  f.keys(), f.read(name), f.create_dataset(name, data, maxshape), f.write(name, data)"""

    def __init__(self, path, mode="r", clear_status_flags=False):
        L = _lib()
        fapl = L.H5Pcreate(_g("H5P_CLS_FILE_ACCESS_ID_g"))
        L.H5Pset_libver_bounds(fapl, 2, 2)                         # H5F_LIBVER_LATEST, like h5py's libver="latest" (main.py:83)
        if clear_status_flags:
            # what `h5clear -s` does: a writer that was killed with the file open leaves the superblock's "open for write" flag
            # set and libhdf5 then refuses every open; this file-access property makes H5Fopen reset the flag first
            one = C.c_uint(1)
            if L.H5Pset(fapl, b"clear_status_flags", C.byref(one)) < 0:
                L.H5Pclose(fapl)
                raise OSError("this libhdf5 has no clear_status_flags file-access property")
        with _quiet():                                             # failures are reported through the OSError below, not on stderr
            if mode == "w":
                self.fid = L.H5Fcreate(path.encode(), 2, 0, fapl)  # H5F_ACC_TRUNC
            else:
                self.fid = L.H5Fopen(path.encode(), 1 if mode == "r+" else 0, fapl)
        L.H5Pclose(fapl)
        if self.fid < 0:
            raise OSError(f"cannot open {path} (mode {mode})")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False

    def close(self):
        if self.fid >= 0:
            _lib().H5Fclose(self.fid)
            self.fid = -1

    def n_links(self):
        """number of links in the root group (O(1); listing the names of a large group is not)"""
        info = _GInfo()
        _lib().H5Gget_info(self.fid, C.byref(info))
        return int(info.nlinks)

    def flush(self):
        _lib().H5Fflush(self.fid, 1)                               # H5F_SCOPE_GLOBAL

    def exists(self, name):
        return _lib().H5Lexists(self.fid, name.encode(), 0) > 0

    def delete(self, name):
        if _lib().H5Ldelete(self.fid, name.encode(), 0) < 0:
            raise KeyError(name)

    def keys(self):
        L = _lib()
        info = _GInfo()
        L.H5Gget_info(self.fid, C.byref(info))
        out = []
        for i in range(info.nlinks):
            n = L.H5Lget_name_by_idx(self.fid, b".", 0, 0, i, None, 0, 0)
            buf = C.create_string_buffer(n + 1)
            L.H5Lget_name_by_idx(self.fid, b".", 0, 0, i, buf, n + 1, 0)
            out.append(buf.value.decode())
        return out

    def walk(self, group=""):
        """paths of every dataset below `group`, in link-name order (groups are told from datasets by trying to open them)"""
        L = _lib()
        out = []

        def rec(gid, prefix):
            info = _GInfo()
            L.H5Gget_info(gid, C.byref(info))
            for i in range(info.nlinks):
                n = L.H5Lget_name_by_idx(gid, b".", 0, 0, i, None, 0, 0)
                buf = C.create_string_buffer(n + 1)
                L.H5Lget_name_by_idx(gid, b".", 0, 0, i, buf, n + 1, 0)
                name = buf.value.decode()
                sub = L.H5Gopen2(gid, name.encode(), 0)
                if sub >= 0:
                    rec(sub, prefix + name + "/")
                    L.H5Gclose(sub)
                else:
                    out.append(prefix + name)

        with _quiet():                                             # probing a dataset with H5Gopen2 is expected to fail quietly
            if group:
                gid = L.H5Gopen2(self.fid, group.encode(), 0)
                if gid < 0:
                    raise KeyError(group)
                rec(gid, group.rstrip("/") + "/")
                L.H5Gclose(gid)
            else:
                rec(self.fid, "")
        return out

    def create_dataset(self, name, data, maxshape=None, dtype=None):
        L = _lib()
        arr = np.ascontiguousarray(data, dtype=dtype)
        rank = arr.ndim
        dims = (hsize_t * rank)(*arr.shape)
        if maxshape is None:
            maxshape = arr.shape
        maxd = (hsize_t * rank)(*[H5S_UNLIMITED if m is None else m for m in maxshape])
        space = L.H5Screate_simple(rank, dims, maxd)
        dcpl = L.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
        if any(m is None for m in maxshape):                       # unlimited dimensions need a chunked layout
            chunk = (hsize_t * rank)(*[max(1, s) for s in arr.shape])
            L.H5Pset_chunk(dcpl, rank, chunk)
        lcpl = L.H5Pcreate(_g("H5P_CLS_LINK_CREATE_ID_g"))
        L.H5Pset_create_intermediate_group(lcpl, 1)                # "layers/conv2d/vars/0" creates the groups on the way
        ds = L.H5Dcreate2(self.fid, name.encode(), _g(_FILE_TYPES[arr.dtype]), space, lcpl, dcpl, 0)
        L.H5Pclose(lcpl)
        if ds < 0:
            raise OSError(f"cannot create dataset {name}")
        if arr.size:
            L.H5Dwrite(ds, _g(_MEM_TYPES[arr.dtype]), 0, 0, 0, arr.ctypes.data)
        L.H5Dclose(ds); L.H5Pclose(dcpl); L.H5Sclose(space)

    def write(self, name, data):
        """overwrite an existing dataset of the same shape (file["game_stats"][i] = ..., Self_Play.py:182-188)"""
        L = _lib()
        ds = L.H5Dopen2(self.fid, name.encode(), 0)
        if ds < 0:
            raise KeyError(name)
        arr = np.ascontiguousarray(data)
        L.H5Dwrite(ds, _g(_MEM_TYPES[arr.dtype]), 0, 0, 0, arr.ctypes.data)
        L.H5Dclose(ds)

    def read(self, name):
        L = _lib()
        ds = L.H5Dopen2(self.fid, name.encode(), 0)
        if ds < 0:
            raise KeyError(name)
        space = L.H5Dget_space(ds)
        rank = L.H5Sget_simple_extent_ndims(space)
        dims = (hsize_t * max(rank, 1))()
        L.H5Sget_simple_extent_dims(space, dims, None)
        t = L.H5Dget_type(ds)
        cls, size, sign = L.H5Tget_class(t), L.H5Tget_size(t), L.H5Tget_sign(t)
        if cls == 0:                                               # H5T_INTEGER
            dt = np.dtype(("i" if sign == 1 else "u") + str(size))
        elif cls == 1:                                             # H5T_FLOAT
            dt = np.dtype("f" + str(size))
        else:
            raise TypeError(f"dataset {name}: unsupported HDF5 type class {cls}")
        out = np.empty(tuple(dims[i] for i in range(rank)), dt)
        if out.size:
            L.H5Dread(ds, _g(_MEM_TYPES[dt]), 0, 0, 0, out.ctypes.data)
        L.H5Tclose(t); L.H5Sclose(space); L.H5Dclose(ds)
        return out
