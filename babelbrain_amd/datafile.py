"""`SaveToH5py` / `ReadFromH5py` for the files either side of the acoustic step (`*DataForSim.h5`).

The reference writes and reads its result dictionaries with `BabelViscoFDTD.H5pySimple`
(BASE:1584-1586 writes `DataForSim`; `_BabelBaseTx.py:134-135`, `CalculateTemperatureEffects.py:684-743`
read it back). That package is absent from the reference tree; what pins the on-disk layout is a file
the reference ships that was written by it, `TranscranialModeling/MapPichardo.h5` (read at BASE:61):
every ndarray is one HDF5 dataset of the array's own dtype and shape carrying a scalar attribute
`type` = "ndarray" stored as a variable-length UTF-8 string. `tests/golden/harness_golden.json`
(`h5pysimple_layout`) holds that structure and `tests/test_datafile.py` holds this writer to it.
The other value kinds a `DataForSim` dictionary carries (python/numpy scalars, bool, str, None, nested
dict / list / tuple) are not represented in that file; they follow the same scheme (dataset or group +
`type` attribute: "scalar", "str", "None", "dict", "list", "tuple", items of sequences named `item_<n>`)
and are UNPINNED against the real package.

Backend: `h5py` when it is importable (the reference's own environment), otherwise libhdf5 through
ctypes (this image has /opt/conda/lib/libhdf5.so but no h5py for /usr/bin/python3). Both produce the
same objects; complex arrays are the compound {r, i} and booleans the FALSE/TRUE enum h5py uses.
"""
import ctypes
import ctypes.util
import glob
import os
import sys

import numpy as np

__all__ = ['SaveToH5py', 'ReadFromH5py', 'describe', 'backend']

_hid = ctypes.c_int64
_hsize = ctypes.c_uint64
_H5T_VARIABLE = ctypes.c_size_t(-1).value
_CLS_INT, _CLS_FLOAT, _CLS_STRING, _CLS_COMPOUND, _CLS_ENUM = 0, 1, 3, 6, 8
_H5I_GROUP, _H5I_DATASET = 2, 5
_lib = None


def _find_lib():
    cands = []
    if os.environ.get('BABELFDTD_HDF5_LIB'):
        cands.append(os.environ['BABELFDTD_HDF5_LIB'])
    n = ctypes.util.find_library('hdf5')
    if n:
        cands.append(n)
    for root in (sys.prefix, '/opt/conda', '/usr', '/usr/local'):
        for pat in ('lib/libhdf5.so*', 'lib/*/libhdf5.so*', 'lib/*/libhdf5_serial.so*'):
            cands += sorted(glob.glob(os.path.join(root, pat)))
    for c in cands:
        try:
            return ctypes.CDLL(c)
        except OSError:
            continue
    raise ImportError('neither h5py nor a loadable libhdf5 was found (set BABELFDTD_HDF5_LIB)')


def _h5():
    """libhdf5 with the prototypes this module uses (1.10+ API names only)."""
    global _lib
    if _lib is not None:
        return _lib
    L = _find_lib()

    def proto(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)
    vp, cp, ci, cu, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t
    proto('H5open', ci)
    proto('H5Fcreate', _hid, cp, cu, _hid, _hid)
    proto('H5Fopen', _hid, cp, cu, _hid)
    proto('H5Fclose', ci, _hid)
    proto('H5Gcreate2', _hid, _hid, cp, _hid, _hid, _hid)
    proto('H5Gclose', ci, _hid)
    proto('H5Gget_info', ci, _hid, vp)
    proto('H5Lget_name_by_idx', ctypes.c_ssize_t, _hid, cp, ci, ci, _hsize, vp, sz, _hid)
    proto('H5Oopen', _hid, _hid, cp, _hid)
    proto('H5Oclose', ci, _hid)
    proto('H5Iget_type', ci, _hid)
    proto('H5Screate', _hid, ci)
    proto('H5Screate_simple', _hid, ci, vp, vp)
    proto('H5Sclose', ci, _hid)
    proto('H5Sget_simple_extent_ndims', ci, _hid)
    proto('H5Sget_simple_extent_dims', ci, _hid, vp, vp)
    proto('H5Sget_simple_extent_type', ci, _hid)
    proto('H5Dcreate2', _hid, _hid, cp, _hid, _hid, _hid, _hid, _hid)
    proto('H5Dopen2', _hid, _hid, cp, _hid)
    proto('H5Dwrite', ci, _hid, _hid, _hid, _hid, _hid, vp)
    proto('H5Dread', ci, _hid, _hid, _hid, _hid, _hid, vp)
    proto('H5Dget_space', _hid, _hid)
    proto('H5Dget_type', _hid, _hid)
    proto('H5Dclose', ci, _hid)
    proto('H5Acreate2', _hid, _hid, cp, _hid, _hid, _hid, _hid)
    proto('H5Aopen', _hid, _hid, cp, _hid)
    proto('H5Aexists', ci, _hid, cp)
    proto('H5Awrite', ci, _hid, _hid, vp)
    proto('H5Aread', ci, _hid, _hid, vp)
    proto('H5Aget_type', _hid, _hid)
    proto('H5Aclose', ci, _hid)
    proto('H5Tcopy', _hid, _hid)
    proto('H5Tclose', ci, _hid)
    proto('H5Tset_size', ci, _hid, sz)
    proto('H5Tset_cset', ci, _hid, ci)
    proto('H5Tget_cset', ci, _hid)
    proto('H5Tget_class', ci, _hid)
    proto('H5Tget_size', sz, _hid)
    proto('H5Tget_sign', ci, _hid)
    proto('H5Tis_variable_str', ci, _hid)
    proto('H5Tcreate', _hid, ci, sz)
    proto('H5Tinsert', ci, _hid, cp, sz, _hid)
    proto('H5Tget_nmembers', ci, _hid)
    proto('H5Tget_member_type', _hid, _hid, cu)
    proto('H5Tenum_create', _hid, _hid)
    proto('H5Tenum_insert', ci, _hid, cp, vp)
    proto('H5free_memory', ci, vp)
    proto('H5Eset_auto2', ci, _hid, vp, vp)
    proto('H5Zregister', ci, vp)
    proto('H5Zfilter_avail', ci, ci)
    if L.H5open() < 0:
        raise ImportError('libhdf5 failed to initialise')
    L.H5Eset_auto2(0, None, None)          # errors are reported through return codes below
    _lib = L
    _register_blosc_decoder(L)
    return L


# The reference's writer compresses arrays with the Blosc filter (HDF5 registered filter 32001, seen in
# MapPichardo.h5: chunked, cd_values {2 2 itemsize chunkbytes 9 1 1}). h5py finds it through hdf5plugin; for the
# ctypes backend a decode-only filter is registered here on top of libblosc when that library is present.
_BLOSC_ID = 32001
_keep = []


def _register_blosc_decoder(L):
    if L.H5Zfilter_avail(_BLOSC_ID) > 0:
        return
    cands = [os.environ.get('BABELFDTD_BLOSC_LIB'), ctypes.util.find_library('blosc')]
    for root in (sys.prefix, '/opt/conda', '/usr', '/usr/local'):
        cands += sorted(glob.glob(os.path.join(root, 'lib/libblosc.so*'))) + sorted(glob.glob(os.path.join(root, 'lib/*/libblosc.so*')))
    blosc = None
    for c in cands:
        if c:
            try:
                blosc = ctypes.CDLL(c)
                break
            except OSError:
                continue
    if blosc is None:
        return
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    blosc.blosc_cbuffer_sizes.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_size_t)] * 3
    blosc.blosc_cbuffer_sizes.restype = None
    blosc.blosc_decompress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    blosc.blosc_decompress.restype = ctypes.c_int
    FILTER = ctypes.CFUNCTYPE(ctypes.c_size_t, ctypes.c_uint, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint), ctypes.c_size_t,
                              ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p))

    def decode(flags, ncd, cd, nbytes, buf_size, buf):
        if not (flags & 0x0100):                     # H5Z_FLAG_REVERSE: only reading is supported
            return 0
        src = buf[0]
        nb, cb, bs = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        blosc.blosc_cbuffer_sizes(src, ctypes.byref(nb), ctypes.byref(cb), ctypes.byref(bs))
        if nb.value == 0 or cb.value > nbytes:
            return 0
        out = libc.malloc(nb.value)
        if not out:
            return 0
        n = blosc.blosc_decompress(src, out, nb.value)
        if n <= 0:
            libc.free(out)
            return 0
        libc.free(src)
        buf[0] = out
        buf_size[0] = nb.value
        return n

    class H5ZClass(ctypes.Structure):
        _fields_ = [('version', ctypes.c_int), ('id', ctypes.c_int), ('encoder_present', ctypes.c_uint),
                    ('decoder_present', ctypes.c_uint), ('name', ctypes.c_char_p), ('can_apply', ctypes.c_void_p),
                    ('set_local', ctypes.c_void_p), ('filter', FILTER)]
    cb = FILTER(decode)
    cls = H5ZClass(1, _BLOSC_ID, 0, 1, b'blosc (decode only, babelbrain_amd)', None, None, cb)
    _keep.extend([cb, cls, blosc, libc])
    L.H5Zregister(ctypes.byref(cls))


def backend():
    try:
        import h5py  # noqa: F401
        return 'h5py'
    except ImportError:
        _h5()
        return 'libhdf5-ctypes'


def _g(name):
    return _hid.in_dll(_h5(), name).value


_STD = {'f4': 'H5T_IEEE_F32LE_g', 'f8': 'H5T_IEEE_F64LE_g', 'i1': 'H5T_STD_I8LE_g', 'u1': 'H5T_STD_U8LE_g',
        'i2': 'H5T_STD_I16LE_g', 'u2': 'H5T_STD_U16LE_g', 'i4': 'H5T_STD_I32LE_g', 'u4': 'H5T_STD_U32LE_g',
        'i8': 'H5T_STD_I64LE_g', 'u8': 'H5T_STD_U64LE_g'}


class _Err(IOError):
    pass


def _ck(v, what):
    if v < 0:
        raise _Err('HDF5 call failed: ' + what)
    return v


def _type_for(dt):
    """numpy dtype -> (hid of a new HDF5 type to be closed by the caller, numpy dtype actually written)."""
    L = _h5()
    dt = np.dtype(dt)
    if dt.kind == 'c':
        part = np.dtype('f4' if dt.itemsize == 8 else 'f8')
        t = _ck(L.H5Tcreate(_CLS_COMPOUND, dt.itemsize), 'H5Tcreate')
        L.H5Tinsert(t, b'r', 0, _g(_STD[part.str[1:]]))
        L.H5Tinsert(t, b'i', part.itemsize, _g(_STD[part.str[1:]]))
        return t, dt.newbyteorder('<')
    if dt.kind == 'b':
        t = _ck(L.H5Tenum_create(_g('H5T_STD_I8LE_g')), 'H5Tenum_create')
        for nm, v in ((b'FALSE', 0), (b'TRUE', 1)):
            L.H5Tenum_insert(t, nm, ctypes.byref(ctypes.c_int8(v)))
        return t, np.dtype('i1')
    key = dt.newbyteorder('<').str[1:]
    if key not in _STD:
        raise TypeError('unsupported array dtype for HDF5: %s' % dt)
    return _ck(L.H5Tcopy(_g(_STD[key])), 'H5Tcopy'), np.dtype('<' + key)


def _vlen_str_type():
    L = _h5()
    t = _ck(L.H5Tcopy(_g('H5T_C_S1_g')), 'H5Tcopy')
    L.H5Tset_size(t, _H5T_VARIABLE)
    L.H5Tset_cset(t, 1)                     # H5T_CSET_UTF8
    return t


def _set_type_attr(obj, value):
    L = _h5()
    t = _vlen_str_type()
    s = _ck(L.H5Screate(0), 'H5Screate')    # H5S_SCALAR
    a = _ck(L.H5Acreate2(obj, b'type', t, s, 0, 0), 'H5Acreate2')
    buf = (ctypes.c_char_p * 1)(value.encode())
    _ck(L.H5Awrite(a, t, buf), 'H5Awrite')
    L.H5Aclose(a); L.H5Sclose(s); L.H5Tclose(t)


def _write_array(loc, name, arr, kind):
    L = _h5()
    arr = np.asarray(arr)
    t, wdt = _type_for(arr.dtype)
    data = np.ascontiguousarray(arr.astype(wdt, copy=False))
    if arr.ndim == 0:
        s = _ck(L.H5Screate(0), 'H5Screate')
    else:
        dims = (_hsize * arr.ndim)(*arr.shape)
        s = _ck(L.H5Screate_simple(arr.ndim, dims, None), 'H5Screate_simple')
    d = _ck(L.H5Dcreate2(loc, name.encode(), t, s, 0, 0, 0), 'H5Dcreate2 ' + name)
    if data.size:
        _ck(L.H5Dwrite(d, t, 0, 0, 0, data.ctypes.data_as(ctypes.c_void_p)), 'H5Dwrite ' + name)
    _set_type_attr(d, kind)
    L.H5Dclose(d); L.H5Sclose(s); L.H5Tclose(t)


def _write_str(loc, name, text, kind='str'):
    L = _h5()
    t = _vlen_str_type()
    s = _ck(L.H5Screate(0), 'H5Screate')
    d = _ck(L.H5Dcreate2(loc, name.encode(), t, s, 0, 0, 0), 'H5Dcreate2 ' + name)
    buf = (ctypes.c_char_p * 1)(text.encode())
    _ck(L.H5Dwrite(d, t, 0, 0, 0, buf), 'H5Dwrite ' + name)
    _set_type_attr(d, kind)
    L.H5Dclose(d); L.H5Sclose(s); L.H5Tclose(t)


def _kind_of(v):
    if isinstance(v, np.ndarray):
        return 'ndarray'
    if isinstance(v, dict):
        return 'dict'
    if isinstance(v, list):
        return 'list'
    if isinstance(v, tuple):
        return 'tuple'
    if isinstance(v, str):
        return 'str'
    if v is None:
        return 'None'
    if isinstance(v, (bool, int, float, complex, np.generic)):
        return 'scalar'
    raise TypeError('cannot store a %s in an HDF5 dictionary file' % type(v).__name__)


def _items_of(v, kind):
    if kind == 'dict':
        for k in v:
            if not isinstance(k, str) or '/' in k or not k:
                raise TypeError('dictionary keys must be non-empty strings without "/": %r' % (k,))
        return list(v.items())
    return [('item_%d' % n, x) for n, x in enumerate(v)]


# ---------------------------------------------------------------------------------------- ctypes backend
def _save_ct(loc, key, v):
    L = _h5()
    kind = _kind_of(v)
    if kind in ('dict', 'list', 'tuple'):
        g = _ck(L.H5Gcreate2(loc, key.encode(), 0, 0, 0), 'H5Gcreate2 ' + key)
        _set_type_attr(g, kind)
        for k, x in _items_of(v, kind):
            _save_ct(g, k, x)
        L.H5Gclose(g)
    elif kind == 'str':
        _write_str(loc, key, v)
    elif kind == 'None':
        _write_str(loc, key, 'None', 'None')
    else:
        _write_array(loc, key, np.asarray(v), kind)


def _read_type_attr(obj):
    L = _h5()
    if L.H5Aexists(obj, b'type') <= 0:
        return None
    a = _ck(L.H5Aopen(obj, b'type', 0), 'H5Aopen')
    ft = L.H5Aget_type(a)
    out = None
    if L.H5Tget_class(ft) == _CLS_STRING:
        if L.H5Tis_variable_str(ft) > 0:
            t = _vlen_str_type()
            p = ctypes.c_void_p()
            if L.H5Aread(a, t, ctypes.byref(p)) >= 0 and p.value:
                out = ctypes.string_at(p.value).decode()
                L.H5free_memory(p)
            L.H5Tclose(t)
        else:
            n = L.H5Tget_size(ft)
            buf = ctypes.create_string_buffer(n + 1)
            if L.H5Aread(a, ft, buf) >= 0:
                out = buf.value.decode()
    L.H5Tclose(ft); L.H5Aclose(a)
    return out


def _np_dtype_of(ft):
    """File datatype -> (numpy dtype to read into, memory type hid to close, post-processing tag)."""
    L = _h5()
    cls = L.H5Tget_class(ft)
    size = L.H5Tget_size(ft)
    if cls == _CLS_FLOAT and size in (4, 8):
        dt = np.dtype('<f%d' % size)
    elif cls == _CLS_INT and size in (1, 2, 4, 8):
        dt = np.dtype('<%s%d' % ('i' if L.H5Tget_sign(ft) == 1 else 'u', size))
    elif cls == _CLS_ENUM and size == 1:
        t, _ = _type_for(np.bool_)
        return np.dtype('i1'), t, 'bool'
    elif cls == _CLS_COMPOUND and L.H5Tget_nmembers(ft) == 2:
        m = L.H5Tget_member_type(ft, 0)
        part = L.H5Tget_size(m)
        L.H5Tclose(m)
        dt = np.dtype('<c%d' % (2 * part))
    elif cls == _CLS_STRING:
        return None, None, 'str'
    else:
        raise TypeError('unsupported HDF5 datatype class %d size %d' % (cls, size))
    t, _ = _type_for(dt)
    return dt, t, None


def _shape_of(space):
    L = _h5()
    nd = L.H5Sget_simple_extent_ndims(space)
    if nd <= 0:
        return ()
    dims = (_hsize * nd)()
    L.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(x) for x in dims)


def _read_dataset(d):
    L = _h5()
    ft = L.H5Dget_type(d)
    sp = L.H5Dget_space(d)
    shape = _shape_of(sp)
    dt, mt, tag = _np_dtype_of(ft)
    try:
        if tag == 'str':
            if L.H5Tis_variable_str(ft) > 0:
                t = _vlen_str_type()
                p = ctypes.c_void_p()
                _ck(L.H5Dread(d, t, 0, 0, 0, ctypes.byref(p)), 'H5Dread')
                s = ctypes.string_at(p.value).decode() if p.value else ''
                if p.value:
                    L.H5free_memory(p)
                L.H5Tclose(t)
                return s, 'str', shape
            buf = ctypes.create_string_buffer(L.H5Tget_size(ft) + 1)
            _ck(L.H5Dread(d, ft, 0, 0, 0, buf), 'H5Dread')
            return buf.value.decode(), 'str', shape
        out = np.empty(shape, dt)
        if out.size:
            _ck(L.H5Dread(d, mt, 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p)),
                'H5Dread (a Blosc-compressed dataset needs libblosc, see BABELFDTD_BLOSC_LIB)')
        if tag == 'bool':
            out = out.astype(np.bool_)
        return out, str(out.dtype), shape
    finally:
        if mt:
            L.H5Tclose(mt)
        L.H5Tclose(ft); L.H5Sclose(sp)


def _children(g):
    L = _h5()
    info = (ctypes.c_uint64 * 4)()           # H5G_info_t {int storage_type; hsize_t nlinks; int64 max_corder; hbool_t mounted}
    _ck(L.H5Gget_info(g, info), 'H5Gget_info')
    names = []
    for idx in range(int(info[1])):
        n = L.H5Lget_name_by_idx(g, b'.', 0, 0, idx, None, 0, 0)
        buf = ctypes.create_string_buffer(n + 1)
        L.H5Lget_name_by_idx(g, b'.', 0, 0, idx, buf, n + 1, 0)
        names.append(buf.value.decode())
    return names


def _restore(kind, value):
    if kind == 'scalar':
        return value.reshape(()).item() if isinstance(value, np.ndarray) else value
    if kind == 'None':
        return None
    return value


def _load_ct(g, describe_only=False):
    L = _h5()
    out = {}
    for name in _children(g):
        o = _ck(L.H5Oopen(g, name.encode(), 0), 'H5Oopen ' + name)
        it = L.H5Iget_type(o)
        kind = _read_type_attr(o)
        if it == _H5I_GROUP:
            sub = _load_ct(o, describe_only)
            if describe_only:
                out[name] = {'object': 'group', 'type_attr': kind, 'members': sub}
            elif kind in ('list', 'tuple'):
                seq = [sub['item_%d' % n] for n in range(len(sub))]
                out[name] = seq if kind == 'list' else tuple(seq)
            else:
                out[name] = sub
        elif it == _H5I_DATASET:
            val, dts, shape = _read_dataset(o)
            if describe_only:
                out[name] = {'object': 'dataset', 'dtype': dts, 'shape': list(shape), 'type_attr': kind,
                             'type_attr_storage': _type_attr_storage(o)}
            else:
                out[name] = _restore(kind, val)
        L.H5Oclose(o)
    return out


def _type_attr_storage(obj):
    L = _h5()
    if L.H5Aexists(obj, b'type') <= 0:
        return None
    a = L.H5Aopen(obj, b'type', 0)
    ft = L.H5Aget_type(a)
    r = {'class': 'string' if L.H5Tget_class(ft) == _CLS_STRING else 'other', 'variable_length': L.H5Tis_variable_str(ft) > 0,
         'utf8': L.H5Tget_cset(ft) == 1}
    L.H5Tclose(ft); L.H5Aclose(a)
    return r


# ---------------------------------------------------------------------------------------- h5py backend
def _save_h5py(loc, key, v):
    kind = _kind_of(v)
    if kind in ('dict', 'list', 'tuple'):
        g = loc.create_group(key)
        g.attrs['type'] = kind
        for k, x in _items_of(v, kind):
            _save_h5py(g, k, x)
    elif kind == 'None':
        loc.create_dataset(key, data='None').attrs['type'] = 'None'
    else:
        loc.create_dataset(key, data=v).attrs['type'] = kind


def _load_h5py(g):
    import h5py
    out = {}
    for name, o in g.items():
        kind = o.attrs.get('type')
        kind = kind.decode() if isinstance(kind, bytes) else kind
        if isinstance(o, h5py.Group):
            sub = _load_h5py(o)
            if kind in ('list', 'tuple'):
                seq = [sub['item_%d' % n] for n in range(len(sub))]
                out[name] = seq if kind == 'list' else tuple(seq)
            else:
                out[name] = sub
        else:
            v = o[()]
            if isinstance(v, bytes):
                v = v.decode()
            out[name] = _restore(kind, v)
    return out


# ---------------------------------------------------------------------------------------- public
def SaveToH5py(MyDict, f_name, use_h5py=None):
    """Writes a (nested) dictionary to `f_name`, truncating it (BASE:1584-1586 usage)."""
    if not isinstance(MyDict, dict):
        raise TypeError('SaveToH5py needs a dictionary')
    if use_h5py is None:
        use_h5py = backend() == 'h5py'
    if use_h5py:
        import h5py
        with h5py.File(f_name, 'w') as f:
            for k, v in _items_of(MyDict, 'dict'):
                _save_h5py(f, k, v)
        return
    L = _h5()
    f = L.H5Fcreate(os.fsencode(f_name), 2, 0, 0)       # H5F_ACC_TRUNC
    if f < 0:
        raise IOError('cannot create %s' % f_name)
    try:
        for k, v in _items_of(MyDict, 'dict'):
            _save_ct(f, k, v)
    finally:
        L.H5Fclose(f)


def _open_ro(f_name):
    L = _h5()
    if not os.path.isfile(f_name):
        raise FileNotFoundError(f_name)
    f = L.H5Fopen(os.fsencode(f_name), 0, 0)            # H5F_ACC_RDONLY
    if f < 0:
        raise IOError('%s is not a readable HDF5 file' % f_name)
    return f


def ReadFromH5py(f_name, use_h5py=None):
    """Reads a file written by `SaveToH5py` (this one or the reference's) back into a dictionary."""
    if use_h5py is None:
        use_h5py = backend() == 'h5py'
    if use_h5py:
        import h5py
        with h5py.File(f_name, 'r') as f:
            return _load_h5py(f)
    f = _open_ro(f_name)
    try:
        return _load_ct(f)
    finally:
        _h5().H5Fclose(f)


def describe(f_name):
    """Structure of a file (object kinds, dtypes, shapes, how the `type` attribute is stored), libhdf5 backend."""
    f = _open_ro(f_name)
    try:
        return _load_ct(f, describe_only=True)
    finally:
        _h5().H5Fclose(f)
