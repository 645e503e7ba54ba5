"""MAT-file level 7.3 (HDF5) -- the format the reference's `save(..., '-v7.3')` writes (SPH_Poiseuille.m:609, 614) --
read and written through the system's libhdf5 with ctypes (no h5py in the image; HDF5 1.10 ships in /opt/conda/lib).

What a v7.3 file is (checked against a file written by MATLAB itself, scipy's test datum testhdf5_7.4_GLNX86.mat, see
tests/test_restart_files.py): a 512-byte user block -- 116 bytes of text "MATLAB 7.3 MAT-file, Platform: ..., Created on:
... HDF5 schema 1.00 .", 8 zero bytes, the version 0x0200 and the endian mark "IM" -- followed by an ordinary HDF5 file in
which every variable is an object of the root group carrying the attribute MATLAB_class:
  double [m x n]   dataset of IEEE doubles with the dimensions REVERSED (n, m): column-major data as it lies in MATLAB
  char   [1 x n]   dataset of uint16 code units (n, 1), MATLAB_class "char", MATLAB_int_decode = 2
  logical          dataset of uint8, MATLAB_class "logical", MATLAB_int_decode = 1
  empty arrays     dataset of uint64 holding the dimensions, attribute MATLAB_empty = 1
  struct (scalar)  group, MATLAB_class "struct", one member per field, field order in the attribute MATLAB_fields
Only these are handled -- they are all the reference's restart / post-process files contain (SPH_Poiseuille.m:434-445,
617-639).  Struct arrays and cells (datasets of object references into /#refs#) raise Mat73Error.

    save(path, {"state": {...}, "config_signature": "..."})      load(path) -> {"state": {...}, ...}
numpy arrays map to double matrices (1-D arrays to columns), Python floats / ints to 1 x 1 doubles, bool to logical, str to
char rows, dict to scalar structs.  `load` returns 2-D float64 arrays in MATLAB's shapes, str for char rows, dict for structs.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
import time

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5I_GROUP, H5I_DATASET = 2, 5
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_REFERENCE = 0, 1, 3, 7
H5S_ALL, H5P_DEFAULT = 0, 0
H5_INDEX_NAME, H5_ITER_INC = 0, 0


class Mat73Error(RuntimeError):
    pass


class Mat73Unavailable(Mat73Error):
    """no libhdf5 on this machine"""


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_int)]


class _Hvl(C.Structure):
    _fields_ = [("len", C.c_size_t), ("p", C.c_void_p)]


_L = None


def _candidates():
    env = os.environ.get("SPHX_HDF5_LIB")
    if env:
        return [env]                              # an explicit choice is not second-guessed
    out = []
    found = ctypes.util.find_library("hdf5")
    if found:
        out.append(found)
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*", "/usr/lib64/libhdf5.so*"):
        out += sorted(p for p in glob.glob(pat) if "_hl" not in p and "_cpp" not in p and "fortran" not in p)
    return out


def lib():
    """libhdf5 (>= 1.10: 64-bit hid_t), loaded once."""
    global _L
    if _L is not None:
        return _L
    errors = []
    for cand in _candidates():
        try:
            L = C.CDLL(cand)
        except OSError as e:
            errors.append(f"{cand}: {e}")
            continue
        if L.H5open() < 0:
            errors.append(f"{cand}: H5open failed")
            continue
        maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
        L.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
        if (maj.value, mnr.value) < (1, 10):
            errors.append(f"{cand}: HDF5 {maj.value}.{mnr.value} (need >= 1.10)")
            continue
        _declare(L)
        L.H5Eset_auto2(0, None, None)             # errors are reported through return codes, not printed by the library
        _L = L
        return L
    raise Mat73Unavailable("MAT v7.3 files need libhdf5 (>= 1.10), none could be loaded"
                           + (": " + "; ".join(errors) if errors else "") + " -- set SPHX_HDF5_LIB, or use format='5'")


def available() -> bool:
    try:
        lib()
        return True
    except Mat73Unavailable:
        return False


def _declare(L):
    sig = {
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fclose": (C.c_int, [hid_t]), "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (C.c_int, [hid_t]),
        "H5Pset_userblock": (C.c_int, [hid_t, hsize_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gclose": (C.c_int, [hid_t]),
        "H5Gget_info": (C.c_int, [hid_t, C.POINTER(_GInfo)]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (C.c_int, [hid_t]), "H5Iget_type": (C.c_int, [hid_t]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
        "H5Sclose": (C.c_int, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]), "H5Dclose": (C.c_int, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]), "H5Tclose": (C.c_int, [hid_t]), "H5Tvlen_create": (hid_t, [hid_t]),
        "H5Tis_variable_str": (C.c_int, [hid_t]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]),
        "H5Aread": (C.c_int, [hid_t, hid_t, C.c_void_p]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Aexists": (C.c_int, [hid_t, C.c_char_p]), "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]),
        "H5Aclose": (C.c_int, [hid_t]), "H5Dvlen_reclaim": (C.c_int, [hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args


def _gid(name: str) -> int:
    """A predefined identifier: the C macros (H5T_NATIVE_DOUBLE, H5P_FILE_CREATE ...) read global variables of the library."""
    return hid_t.in_dll(lib(), name).value


def _ok(rc, what):
    if rc < 0:
        raise Mat73Error(f"HDF5: {what} failed")
    return rc


# ------------------------------------------------------------------------------------------------ writing
def _set_str_attr(obj, name: str, value: str):
    L = lib()
    raw = value.encode("ascii")
    t = _ok(L.H5Tcopy(_gid("H5T_C_S1_g")), "H5Tcopy")
    L.H5Tset_size(t, max(len(raw), 1))
    sp = _ok(L.H5Screate(0), "H5Screate")                     # H5S_SCALAR
    a = _ok(L.H5Acreate2(obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2 " + name)
    buf = C.create_string_buffer(raw, max(len(raw), 1))
    _ok(L.H5Awrite(a, t, buf), "H5Awrite " + name)
    L.H5Aclose(a); L.H5Sclose(sp); L.H5Tclose(t)


def _set_int_attr(obj, name: str, value: int, file_type="H5T_STD_I32LE_g"):
    L = lib()
    sp = _ok(L.H5Screate(0), "H5Screate")
    a = _ok(L.H5Acreate2(obj, name.encode(), _gid(file_type), sp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2 " + name)
    v = C.c_int32(value)
    _ok(L.H5Awrite(a, _gid("H5T_NATIVE_INT32_g"), C.byref(v)), "H5Awrite " + name)
    L.H5Aclose(a); L.H5Sclose(sp)


def _set_fields_attr(obj, names):
    """MATLAB_fields: one variable-length sequence of 1-byte characters per field (keeps the field order)."""
    L = lib()
    base = _ok(L.H5Tcopy(_gid("H5T_C_S1_g")), "H5Tcopy")
    vl = _ok(L.H5Tvlen_create(base), "H5Tvlen_create")
    dims = (hsize_t * 1)(len(names))
    sp = _ok(L.H5Screate_simple(1, dims, None), "H5Screate_simple")
    a = _ok(L.H5Acreate2(obj, b"MATLAB_fields", vl, sp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2 MATLAB_fields")
    keep = [C.create_string_buffer(n.encode("ascii"), len(n)) for n in names]
    arr = (_Hvl * len(names))(*[_Hvl(len(n), C.cast(b, C.c_void_p)) for n, b in zip(names, keep)])
    _ok(L.H5Awrite(a, vl, arr), "H5Awrite MATLAB_fields")
    L.H5Aclose(a); L.H5Sclose(sp); L.H5Tclose(vl); L.H5Tclose(base)


def _write_dataset(loc, name: str, data: np.ndarray, file_type: str, mem_type: str, matlab_class: str, int_decode=0, empty_shape=None):
    L = lib()
    data = np.ascontiguousarray(data)
    dims = (hsize_t * data.ndim)(*data.shape)
    sp = _ok(L.H5Screate_simple(data.ndim, dims, None), "H5Screate_simple")
    d = _ok(L.H5Dcreate2(loc, name.encode(), _gid(file_type), sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Dcreate2 " + name)
    _ok(L.H5Dwrite(d, _gid(mem_type), H5S_ALL, H5S_ALL, H5P_DEFAULT, data.ctypes.data_as(C.c_void_p)), "H5Dwrite " + name)
    _set_str_attr(d, "MATLAB_class", matlab_class)
    if int_decode:
        _set_int_attr(d, "MATLAB_int_decode", int_decode)
    if empty_shape is not None:
        _set_int_attr(d, "MATLAB_empty", 1, "H5T_STD_U8LE_g")
    L.H5Dclose(d); L.H5Sclose(sp)


def _write_value(loc, name: str, value):
    L = lib()
    if isinstance(value, dict):
        g = _ok(L.H5Gcreate2(loc, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2 " + name)
        _set_str_attr(g, "MATLAB_class", "struct")
        names = [str(k) for k in value]
        if names:
            _set_fields_attr(g, names)
        for k in names:
            _write_value(g, k, value[k])
        L.H5Gclose(g)
        return
    if isinstance(value, str):
        units = np.frombuffer(value.encode("utf-16-le"), dtype="<u2")
        if units.size == 0:
            _write_dataset(loc, name, np.array([0, 0], dtype=np.uint64), "H5T_STD_U64LE_g", "H5T_NATIVE_UINT64_g", "char", 0, (0, 0))
        else:   # a 1 x n row: HDF5 dims (n, 1)
            _write_dataset(loc, name, units.astype(np.uint16).reshape(-1, 1), "H5T_STD_U16LE_g", "H5T_NATIVE_UINT16_g", "char", 2)
        return
    if isinstance(value, (bool, np.bool_)):
        _write_dataset(loc, name, np.array([[1 if value else 0]], dtype=np.uint8), "H5T_STD_U8LE_g", "H5T_NATIVE_UINT8_g", "logical", 1)
        return
    arr = np.asarray(value)
    if arr.dtype == np.bool_:
        raise Mat73Error(f"{name}: logical arrays are not needed by the reference's files and not written")
    if arr.dtype.kind not in "fiu":
        raise Mat73Error(f"{name}: cannot store {type(value).__name__} / dtype {arr.dtype} in a MAT v7.3 file")
    arr = np.asarray(arr, dtype=np.float64)
    if arr.ndim == 0:
        arr = arr.reshape(1, 1)
    elif arr.ndim == 1:
        arr = arr.reshape(-1, 1)           # oned_as="column", as restart.py always asked of savemat
    elif arr.ndim != 2:
        raise Mat73Error(f"{name}: only matrices are stored, got {arr.ndim} dimensions")
    if arr.size == 0:
        _write_dataset(loc, name, np.array(arr.shape, dtype=np.uint64), "H5T_STD_U64LE_g", "H5T_NATIVE_UINT64_g", "double", 0, arr.shape)
        return
    # MATLAB [m x n], column-major  ==  C-order (n, m): the transpose, contiguous
    _write_dataset(loc, name, np.ascontiguousarray(arr.T), "H5T_IEEE_F64LE_g", "H5T_NATIVE_DOUBLE_g", "double")


def header(platform="GLNXA64", when=None) -> bytes:
    text = "MATLAB 7.3 MAT-file, Platform: %s, Created on: %s HDF5 schema 1.00 ." % (platform, time.strftime("%a %b %e %H:%M:%S %Y", when or time.localtime()))
    head = text.encode("ascii").ljust(116, b" ") + b"\x00" * 8 + b"\x00\x02" + b"IM"
    return head.ljust(512, b"\x00")


def save(path: str, variables: dict) -> None:
    L = lib()
    fcpl = _ok(L.H5Pcreate(_gid("H5P_CLS_FILE_CREATE_ID_g")), "H5Pcreate")
    _ok(L.H5Pset_userblock(fcpl, 512), "H5Pset_userblock")
    f = L.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, fcpl, H5P_DEFAULT)
    L.H5Pclose(fcpl)
    _ok(f, f"H5Fcreate {path}")
    try:
        for name, value in variables.items():
            _write_value(f, str(name), value)
    finally:
        L.H5Fclose(f)
    with open(path, "r+b") as fh:           # the user block is the application's: HDF5 leaves it alone
        fh.write(header())


# ------------------------------------------------------------------------------------------------ reading
def _attr_str(obj, name: str):
    L = lib()
    if L.H5Aexists(obj, name.encode()) <= 0:
        return None
    a = _ok(L.H5Aopen(obj, name.encode(), H5P_DEFAULT), "H5Aopen " + name)
    t = L.H5Aget_type(a)
    try:
        if L.H5Tget_class(t) != H5T_STRING:
            return None
        if L.H5Tis_variable_str(t) > 0:
            p = C.c_char_p()
            _ok(L.H5Aread(a, t, C.byref(p)), "H5Aread " + name)
            return (p.value or b"").decode("ascii", "replace")
        n = L.H5Tget_size(t)
        buf = C.create_string_buffer(n + 1)
        _ok(L.H5Aread(a, t, buf), "H5Aread " + name)
        return buf.raw[:n].split(b"\x00")[0].decode("ascii", "replace")
    finally:
        L.H5Tclose(t); L.H5Aclose(a)


def _attr_fields(obj):
    """field order of a struct (MATLAB_fields), None when the attribute is absent"""
    L = lib()
    if L.H5Aexists(obj, b"MATLAB_fields") <= 0:
        return None
    a = _ok(L.H5Aopen(obj, b"MATLAB_fields", H5P_DEFAULT), "H5Aopen MATLAB_fields")
    sp, t = L.H5Aget_space(a), L.H5Aget_type(a)
    try:
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        n = int(np.prod(list(dims)[:nd])) if nd else 1
        arr = (_Hvl * n)()
        base = _ok(L.H5Tcopy(_gid("H5T_C_S1_g")), "H5Tcopy")
        vl = _ok(L.H5Tvlen_create(base), "H5Tvlen_create")
        _ok(L.H5Aread(a, vl, arr), "H5Aread MATLAB_fields")
        out = [C.string_at(e.p, e.len).decode("ascii", "replace") for e in arr]
        L.H5Dvlen_reclaim(vl, sp, H5P_DEFAULT, arr)
        L.H5Tclose(vl); L.H5Tclose(base)
        return out
    finally:
        L.H5Tclose(t); L.H5Sclose(sp); L.H5Aclose(a)


def _members(group):
    L = lib()
    gi = _GInfo()
    _ok(L.H5Gget_info(group, C.byref(gi)), "H5Gget_info")
    names = []
    for k in range(gi.nlinks):
        n = _ok(L.H5Lget_name_by_idx(group, b".", H5_INDEX_NAME, H5_ITER_INC, k, None, 0, H5P_DEFAULT), "H5Lget_name_by_idx")
        buf = C.create_string_buffer(n + 1)
        L.H5Lget_name_by_idx(group, b".", H5_INDEX_NAME, H5_ITER_INC, k, buf, n + 1, H5P_DEFAULT)
        names.append(buf.value.decode())
    return names


def _read_dataset(d, where: str):
    L = lib()
    cls = _attr_str(d, "MATLAB_class")
    sp, t = L.H5Dget_space(d), L.H5Dget_type(d)
    try:
        if L.H5Tget_class(t) == H5T_REFERENCE or cls in ("cell", "function_handle"):
            raise Mat73Error(f"{where}: class {cls or 'reference'} (cell / struct array) is not supported")
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        shape = tuple(int(x) for x in list(dims)[:nd])
        empty = L.H5Aexists(d, b"MATLAB_empty") > 0
        if empty:
            dd = np.zeros(shape or (1,), dtype=np.uint64)
            _ok(L.H5Dread(d, _gid("H5T_NATIVE_UINT64_g"), H5S_ALL, H5S_ALL, H5P_DEFAULT, dd.ctypes.data_as(C.c_void_p)), "H5Dread " + where)
            mshape = tuple(int(x) for x in dd.ravel())
            return "" if cls == "char" else np.zeros(mshape, dtype=np.float64)
        if cls == "char":
            buf = np.zeros(shape, dtype=np.uint16)
            _ok(L.H5Dread(d, _gid("H5T_NATIVE_UINT16_g"), H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.ctypes.data_as(C.c_void_p)), "H5Dread " + where)
            rows = buf.T if buf.ndim == 2 else buf.reshape(1, -1)       # back to MATLAB's [m x n]
            text = ["".join(chr(c) for c in r) for r in rows]
            return text[0] if len(text) == 1 else text
        if L.H5Tget_class(t) not in (H5T_INTEGER, H5T_FLOAT):
            raise Mat73Error(f"{where}: unsupported HDF5 type class {L.H5Tget_class(t)} (MATLAB class {cls})")
        buf = np.zeros(shape, dtype=np.float64)       # the library converts any numeric file type (and undoes chunking / deflate)
        _ok(L.H5Dread(d, _gid("H5T_NATIVE_DOUBLE_g"), H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.ctypes.data_as(C.c_void_p)), "H5Dread " + where)
        out = np.asfortranarray(buf.T) if buf.ndim >= 2 else buf.reshape(-1, 1)
        return out.astype(bool) if cls == "logical" else out
    finally:
        L.H5Tclose(t); L.H5Sclose(sp)


def _read_object(loc, name: str, where: str):
    L = lib()
    o = _ok(L.H5Oopen(loc, name.encode(), H5P_DEFAULT), "H5Oopen " + where)
    try:
        kind = L.H5Iget_type(o)
        if kind == H5I_DATASET:
            return _read_dataset(o, where)
        if kind == H5I_GROUP:
            cls = _attr_str(o, "MATLAB_class")
            if cls not in (None, "struct"):
                raise Mat73Error(f"{where}: group of MATLAB class {cls} is not supported")
            present = _members(o)
            order = [n for n in (_attr_fields(o) or []) if n in present]
            order += [n for n in present if n not in order]
            return {n: _read_object(o, n, where + "." + n) for n in order}
        raise Mat73Error(f"{where}: unexpected HDF5 object type {kind}")
    finally:
        L.H5Oclose(o)


def is_mat73(path: str) -> bool:
    with open(path, "rb") as f:
        head = f.read(520)
    return head[:8] == b"\x89HDF\r\n\x1a\n" or head[512:520] == b"\x89HDF\r\n\x1a\n"


def load(path: str, names=None) -> dict:
    """All variables of a v7.3 file (or just `names`, like MATLAB's load(path, 'a', 'b')); '#refs#' and '#subsystem#' are skipped."""
    L = lib()
    f = L.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    _ok(f, f"H5Fopen {path}")
    try:
        root = _ok(L.H5Oopen(f, b"/", H5P_DEFAULT), "H5Oopen /")
        try:
            out = {}
            for n in _members(root):
                if n.startswith("#") or (names is not None and n not in names):
                    continue
                out[n] = _read_object(root, n, n)
            return out
        finally:
            L.H5Oclose(root)
    finally:
        L.H5Fclose(f)
