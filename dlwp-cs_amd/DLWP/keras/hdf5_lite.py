"""
Minimal pure-Python reader for the HDF5 subset that Keras / h5py write for `model.save_weights(path, save_format='h5')` and
`model.save(path)` (reference DLWP/util.py:139-142, DLWP/custom.py:184-191): h5py's default `libver='earliest'` files --
version 0/1 superblock, old-style groups (symbol table: v1 B-tree + local heap + SNOD nodes), version 1 object headers
(with continuation blocks), contiguous / compact datasets of fixed-size numeric types, and attributes holding fixed-length
strings, variable-length strings (global heap) or numeric arrays.  New-style compact groups (link messages) and version 2
object headers are read as well; chunked / filtered datasets and dense link / attribute storage are not (clear error).

This stack carries neither h5py nor TensorFlow; the reader exists so that weights and models written by the reference
(TensorFlow 2.1 + h5py) load into the engine without a conversion step, the writer at the end of the file so that what the engine
saves under the reference's file names (`save_weights(save_format='h5')`, `<name>.keras`) is an HDF5 file Keras / h5py can open.
No dependencies beyond numpy.
Layout knowledge: HDF5 File Format Specification version 2.0/3.0 (public document); validated against files written by the
real HDF5 library (tests/golden/gen_golden_h5.py, run with h5py 3.3 / libhdf5 1.10.6 in the build container).
"""
import struct

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5Error(ValueError):
    pass


def is_hdf5(path):
    try:
        with open(path, 'rb') as f:
            for off in (0, 512, 1024, 2048):
                f.seek(off)
                if f.read(8) == SIGNATURE:
                    return True
    except OSError:
        pass
    return False


class _Buf(object):
    def __init__(self, data):
        self.d = data

    def u(self, off, n):
        return int.from_bytes(self.d[off:off + n], 'little')

    def bytes(self, off, n):
        b = self.d[off:off + n]
        if len(b) != n:
            raise HDF5Error('truncated file (wanted %d bytes at %d)' % (n, off))
        return b


class Dataset(object):
    def __init__(self, f, name, dtype, shape, layout):
        self._f, self.name, self.dtype, self.shape, self._layout = f, name, dtype, tuple(shape), layout
        self.attrs = {}

    def __getitem__(self, key):
        return self.read()[key]

    def read(self):
        kind, a, b = self._layout
        n = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        if kind == 'compact':
            raw = a
        elif kind == 'contiguous':
            if a == UNDEF or n == 0:
                raw = b'\x00' * n
            else:
                raw = self._f._b.bytes(self._f._base + a, n)
        else:
            raise HDF5Error('dataset %s: %s storage is not supported by this reader (Keras writes contiguous datasets)'
                            % (self.name, kind))
        return np.frombuffer(raw[:n], dtype=self.dtype).reshape(self.shape).copy()


class Group(object):
    def __init__(self, f, name):
        self._f, self.name = f, name
        self._links = {}        # name -> object header address
        self.attrs = {}
        self._cache = {}

    def keys(self):
        return list(self._links)

    def __contains__(self, k):
        try:
            self[k]
            return True
        except KeyError:
            return False

    def __iter__(self):
        return iter(self._links)

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            if part not in node._cache:
                node._cache[part] = node._f._object(node._links[part], (node.name.rstrip('/') + '/' + part))
            node = node._cache[part]
        return node

    def visit_datasets(self, prefix=''):
        for k in self._links:
            obj = self[k]
            if isinstance(obj, Group):
                for item in obj.visit_datasets(prefix + k + '/'):
                    yield item
            else:
                yield prefix + k, obj


class File(Group):
    def __init__(self, path):
        with open(path, 'rb') as fh:
            data = fh.read()
        self._b = _Buf(data)
        Group.__init__(self, self, '/')
        self._parse_superblock()

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    # -- superblock ------------------------------------------------------------------------------------------------ #
    def _parse_superblock(self):
        b = self._b
        start = None
        for off in (0, 512, 1024, 2048, 4096):
            if b.d[off:off + 8] == SIGNATURE:
                start = off
                break
        if start is None:
            raise HDF5Error('not an HDF5 file')
        ver = b.u(start + 8, 1)
        if ver in (0, 1):
            self._O, self._L = b.u(start + 13, 1), b.u(start + 14, 1)
            p = start + 24 + (4 if ver == 1 else 0)
            self._base = b.u(p, self._O)
            p += 4 * self._O
            # root group symbol table entry
            root_hdr = b.u(p + self._O, self._O)
        elif ver in (2, 3):
            self._O, self._L = b.u(start + 9, 1), b.u(start + 10, 1)
            p = start + 12
            self._base = b.u(p, self._O)
            root_hdr = b.u(p + 3 * self._O, self._O)
        else:
            raise HDF5Error('superblock version %d is not supported' % ver)
        if self._O != 8 or self._L != 8:
            raise HDF5Error('only 8-byte offsets / lengths are supported')
        root = self._object(root_hdr, '/')
        self._links, self.attrs = root._links, root.attrs

    # -- object headers -------------------------------------------------------------------------------------------- #
    def _messages(self, addr):
        """yield (type, flags, body bytes) of every header message of the object at `addr` (v1 and v2 headers)."""
        b = self._b
        a = self._base + addr
        if b.d[a:a + 4] == b'OHDR':
            flags = b.u(a + 5, 1)
            p = a + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szf = 1 << (flags & 3)
            chunk0 = b.u(p, szf)
            p += szf
            blocks = [(p, chunk0)]
            track = bool(flags & 0x04)
            i = 0
            while i < len(blocks):
                q, size = blocks[i]
                end = q + size
                while q + 4 <= end:
                    mtype, msize, mflags = b.u(q, 1), b.u(q + 1, 2), b.u(q + 3, 1)
                    q += 4 + (2 if track else 0)
                    body = b.bytes(q, msize)
                    if mtype == 0x10:
                        caddr, clen = struct.unpack_from('<QQ', body)
                        blocks.append((self._base + caddr + 4, clen - 8))       # skip 'OCHK', drop the checksum
                    elif mtype != 0:
                        yield mtype, mflags, body
                    q += msize
                i += 1
            return
        ver = b.u(a, 1)
        if ver != 1:
            raise HDF5Error('object header version %d at %d is not supported' % (ver, addr))
        nmsg = b.u(a + 2, 2)
        hsize = b.u(a + 8, 4)
        blocks = [(a + 16, hsize)]
        seen = 0
        i = 0
        while i < len(blocks) and seen < nmsg:
            q, size = blocks[i]
            end = q + size
            while q + 8 <= end and seen < nmsg:
                mtype, msize, mflags = b.u(q, 2), b.u(q + 2, 2), b.u(q + 4, 1)
                body = b.bytes(q + 8, msize)
                seen += 1
                if mtype == 0x10:
                    caddr, clen = struct.unpack_from('<QQ', body)
                    blocks.append((self._base + caddr, clen))
                elif mtype != 0:
                    yield mtype, mflags, body
                q += 8 + msize
            i += 1

    def _object(self, addr, name):
        msgs = list(self._messages(addr))
        types = [m[0] for m in msgs]
        attrs = {}
        for mtype, mflags, body in msgs:
            if mtype == 0x0C:
                k, v = self._attribute(body)
                attrs[k] = v
        if 0x08 in types:                        # data layout -> dataset
            dtype = shape = layout = None
            for mtype, mflags, body in msgs:
                if mtype == 0x03:
                    dtype = self._datatype(body)[0]
                elif mtype == 0x01:
                    shape = self._dataspace(body)
                elif mtype == 0x08:
                    layout = self._layout(body)
            if isinstance(dtype, tuple) or dtype is None:
                raise HDF5Error('dataset %s: unsupported datatype' % name)
            ds = Dataset(self, name, dtype, shape, layout)
            ds.attrs = attrs
            return ds
        g = Group(self, name)
        g.attrs = attrs
        for mtype, mflags, body in msgs:
            if mtype == 0x11:                    # symbol table: old-style group
                btree, heap = struct.unpack_from('<QQ', body)
                g._links.update(self._symbol_table(btree, heap))
            elif mtype == 0x06:                  # link message: new-style compact group
                k, target = self._link(body)
                if target is not None:
                    g._links[k] = target
            elif mtype == 0x02:
                fheap = struct.unpack_from('<Q', body, 2 + (8 if body[1] & 1 else 0))[0]
                if fheap != UNDEF:
                    raise HDF5Error('group %s uses dense link storage, which this reader does not support' % name)
        return g

    # -- groups ---------------------------------------------------------------------------------------------------- #
    def _heap_name(self, heap_data_addr, off):
        d = self._b.d
        a = self._base + heap_data_addr + off
        e = d.index(b'\x00', a)
        return d[a:e].decode('utf8')

    def _symbol_table(self, btree_addr, heap_addr):
        b = self._b
        h = self._base + heap_addr
        if b.d[h:h + 4] != b'HEAP':
            raise HDF5Error('bad local heap signature')
        data_addr = b.u(h + 24, 8)
        out = {}

        def walk(addr):
            a = self._base + addr
            if b.d[a:a + 4] != b'TREE':
                raise HDF5Error('bad B-tree signature')
            level, used = b.u(a + 5, 1), b.u(a + 6, 2)
            p = a + 24
            for i in range(used):
                child = b.u(p + 8 + i * 16, 8)          # key_i (8) child_i (8) ...
                if level > 0:
                    walk(child)
                else:
                    s = self._base + child
                    if b.d[s:s + 4] != b'SNOD':
                        raise HDF5Error('bad symbol node signature')
                    n = b.u(s + 6, 2)
                    for j in range(n):
                        e = s + 8 + j * 40
                        out[self._heap_name(data_addr, b.u(e, 8))] = b.u(e + 8, 8)
        walk(btree_addr)
        return out

    def _link(self, body):
        flags = body[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nsz = 1 << (flags & 3)
        nlen = int.from_bytes(body[p:p + nsz], 'little')
        p += nsz
        name = body[p:p + nlen].decode('utf8')
        p += nlen
        if ltype != 0:
            return name, None                    # soft / external links: ignored
        return name, int.from_bytes(body[p:p + 8], 'little')

    # -- messages -------------------------------------------------------------------------------------------------- #
    @staticmethod
    def _dataspace(body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if body[3] == 2:                     # null dataspace
                return (0,)
            p = 4
        else:
            raise HDF5Error('dataspace version %d' % ver)
        return tuple(int.from_bytes(body[p + 8 * i:p + 8 * i + 8], 'little') for i in range(rank))

    def _datatype(self, body):
        """-> (numpy dtype | ('vlen_str',) | ('str', n), bytes consumed)"""
        cls, ver = body[0] & 0x0F, body[0] >> 4
        bits0 = body[1]
        size = int.from_bytes(body[4:8], 'little')
        order = '>' if (bits0 & 1) else '<'
        if cls == 0:
            signed = bool(bits0 & 0x08)
            return np.dtype('%s%s%d' % (order, 'i' if signed else 'u', size)), 8 + 4
        if cls == 1:
            return np.dtype('%sf%d' % (order, size)), 8 + 12
        if cls == 3:
            return ('str', size), 8
        if cls == 9:
            if bits0 & 0x0F == 1:
                return ('vlen_str',), 8 + self._datatype(body[8:])[1]
            raise HDF5Error('variable-length sequences are not supported')
        if cls == 8:                             # enum (h5py stores bool as an int8 enum): read the base integer type
            base, used = self._datatype(body[8:])
            return base, 8 + used
        raise HDF5Error('datatype class %d is not supported' % cls)

    def _layout(self, body):
        ver = body[0]
        if ver == 3:
            cls = body[1]
            if cls == 0:
                n = int.from_bytes(body[2:4], 'little')
                return ('compact', bytes(body[4:4 + n]), None)
            if cls == 1:
                addr, size = struct.unpack_from('<QQ', body, 2)
                return ('contiguous', addr, size)
            return ('chunked', None, None)
        if ver in (1, 2):
            rank, cls = body[1], body[2]
            if cls == 1:
                return ('contiguous', struct.unpack_from('<Q', body, 8)[0], None)
            return ('chunked' if cls == 2 else 'compact-v1', None, None)
        if ver == 4:
            cls = body[1]
            if cls == 1:
                addr, size = struct.unpack_from('<QQ', body, 2)
                return ('contiguous', addr, size)
            if cls == 0:
                n = int.from_bytes(body[2:4], 'little')
                return ('compact', bytes(body[4:4 + n]), None)
            return ('chunked', None, None)
        raise HDF5Error('data layout version %d' % ver)

    def _global_heap_object(self, coll_addr, index):
        b = self._b
        a = self._base + coll_addr
        if b.d[a:a + 4] != b'GCOL':
            raise HDF5Error('bad global heap signature')
        size = b.u(a + 8, 8)
        p, end = a + 16, a + size
        while p + 16 <= end:
            idx, osize = b.u(p, 2), b.u(p + 8, 8)
            if idx == 0:
                break
            if idx == index:
                return b.bytes(p + 16, osize)
            p += 16 + ((osize + 7) // 8) * 8
        raise HDF5Error('global heap object %d not found' % index)

    def _attribute(self, body):
        ver = body[0]
        nsz, tsz, ssz = struct.unpack_from('<HHH', body, 2)
        if ver == 1:
            p = 8
            pad = lambda n: ((n + 7) // 8) * 8       # noqa: E731
        elif ver == 2:
            p = 8
            pad = lambda n: n                         # noqa: E731
        elif ver == 3:
            p = 9
            pad = lambda n: n                         # noqa: E731
        else:
            raise HDF5Error('attribute message version %d' % ver)
        name = body[p:p + nsz].split(b'\x00')[0].decode('utf8')
        p += pad(nsz)
        dtype, _ = self._datatype(body[p:p + tsz])
        p += pad(tsz)
        shape = self._dataspace(body[p:p + ssz]) if ssz else ()
        p += pad(ssz)
        data = body[p:]
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if isinstance(dtype, tuple) and dtype[0] == 'str':
            L = dtype[1]
            vals = [data[i * L:(i + 1) * L].split(b'\x00')[0] for i in range(n)]
            arr = np.array(vals, dtype='S%d' % max(L, 1)).reshape(shape) if shape else vals[0]
            return name, arr
        if isinstance(dtype, tuple) and dtype[0] == 'vlen_str':
            vals = []
            for i in range(n):
                ln, coll, idx = struct.unpack_from('<IQI', data, i * 16)
                vals.append(self._global_heap_object(coll, idx)[:ln] if ln else b'')
            if shape:
                return name, np.array(vals, dtype=object).reshape(shape)
            return name, vals[0]
        arr = np.frombuffer(data[:n * dtype.itemsize], dtype=dtype).copy()
        return name, (arr.reshape(shape) if shape else arr[0])


# ---------------------------------------------------------------------------------------------------------------------- #
# Keras layouts on top of the reader
# ---------------------------------------------------------------------------------------------------------------------- #

def _as_str(v):
    if isinstance(v, (bytes, np.bytes_)):
        return bytes(v).decode('utf8')
    return str(v)


def _attr_list(group, name):
    """Keras' save_attributes_to_hdf5_group: `name`, or chunks `name0`, `name1`, ... when the attribute exceeded 64 KB."""
    if name in group.attrs:
        return [_as_str(v) for v in np.atleast_1d(group.attrs[name])]
    out, i = [], 0
    while '%s%d' % (name, i) in group.attrs:
        out += [_as_str(v) for v in np.atleast_1d(group.attrs['%s%d' % (name, i)])]
        i += 1
    return out


def read_keras_weights(path):
    """
    -> (ordered list of (layer_name, [(weight_name, ndarray), ...]), model_config json str | None).
    Accepts both layouts Keras writes: a weights file (root attrs `layer_names`) and a full model file (group `model_weights`
    + root attr `model_config`).
    """
    f = File(path)
    root = f['model_weights'] if 'model_weights' in f._links else f
    cfg = f.attrs.get('model_config')
    layers = []
    for lname in _attr_list(root, 'layer_names'):
        g = root[lname]
        ws = []
        for wname in _attr_list(g, 'weight_names'):
            ws.append((wname, np.asarray(g[wname].read())))
        layers.append((lname, ws))
    return layers, (None if cfg is None else _as_str(cfg))


# ---------------------------------------------------------------------------------------------------------------------- #
# Writer: the same subset, as h5py's default `libver='earliest'` lays it out (version 0 superblock, version 1 object headers,
# old-style groups: one v1 B-tree node + one symbol node + a local heap per group, contiguous datasets, attributes as header
# messages).  Enough for what Keras writes in `save_weights(save_format='h5')` / `model.save()` -- and read back by the real
# HDF5 library: tests/test_hdf5_writer.py opens the files with h5py 3.3 / libhdf5 1.10.6 where the build container has them.
# ---------------------------------------------------------------------------------------------------------------------- #

_ATTR_LIMIT = 64512         # keras' HDF5_OBJECT_HEADER_LIMIT: larger attributes are split into name0, name1, ...


def _pad8(b):
    return b + b'\x00' * (-len(b) % 8)


class _WGroup(object):
    def __init__(self):
        self.attrs = []             # [(name, value)]: bytes | str | list of bytes/str | numpy array / scalar
        self.children = {}          # name -> _WGroup | numpy array

    def group(self, name):
        g = self.children.get(name)
        if g is None:
            g = self.children[name] = _WGroup()
        return g


def _dataspace_msg(shape):
    shape = tuple(int(v) for v in shape)
    return struct.pack('<BBB5x', 1, len(shape), 0) + b''.join(struct.pack('<Q', v) for v in shape)


def _datatype_msg(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == 'S':
        # fixed-length string, null-padded, ASCII (what h5py makes of numpy 'S' arrays)
        return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, max(dtype.itemsize, 1))
    if dtype.kind == 'f' and dtype.itemsize in (4, 8):
        if dtype.itemsize == 4:
            return struct.pack('<BBBBI', 0x11, 0x20, 31, 0, 4) + struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)
        return struct.pack('<BBBBI', 0x11, 0x20, 63, 0, 8) + struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)
    if dtype.kind in 'iu' and dtype.itemsize in (1, 2, 4, 8):
        return struct.pack('<BBBBI', 0x10, 0x08 if dtype.kind == 'i' else 0x00, 0, 0, dtype.itemsize) + \
            struct.pack('<HH', 0, 8 * dtype.itemsize)
    raise HDF5Error('writer: dtype %s is not supported' % dtype)


def _attr_value(v):
    """-> numpy array (fixed-length bytes, or numeric) in the form h5py would store it (FIXED-length strings, as h5py 2.10 did
    for the reference's TensorFlow 2.1 environment)"""
    if isinstance(v, str):
        v = v.encode('utf8')
    if isinstance(v, (bytes, np.bytes_)):
        return np.array(bytes(v), dtype='S%d' % max(len(v), 1))
    if isinstance(v, (list, tuple)):
        if len(v) == 0:
            return np.zeros((0,), dtype=np.float64)             # h5py stores an empty list as a float64 array of shape (0,)
        if all(isinstance(x, (str, bytes, np.bytes_)) for x in v):
            bs = [x.encode('utf8') if isinstance(x, str) else bytes(x) for x in v]
            return np.array(bs, dtype='S%d' % max(max(len(x) for x in bs), 1))
    return np.asarray(v)


def _attr_msg(name, value):
    arr = _attr_value(value)
    nm = name.encode('utf8') + b'\x00'
    dt, ds = _datatype_msg(arr.dtype), _dataspace_msg(arr.shape)
    data = arr.tobytes()
    if arr.dtype.kind == 'S' and arr.dtype.itemsize == 0:
        data = b'\x00' * max(arr.size, 1)
    return struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data


def _split_attr(name, value):
    """keras' save_attributes_to_hdf5_group: a list attribute beyond the object-header limit becomes name0, name1, ..."""
    arr = _attr_value(value)
    if arr.nbytes <= _ATTR_LIMIT or arr.ndim == 0:
        if arr.nbytes > 0xFF00:
            raise HDF5Error('writer: attribute %r of %d bytes does not fit an object-header message' % (name, arr.nbytes))
        return [(name, value)]
    n = 1
    chunks = np.array_split(arr, n)
    while any(c.nbytes > _ATTR_LIMIT for c in chunks):
        n += 1
        chunks = np.array_split(arr, n)
    return [('%s%d' % (name, i), list(c)) for i, c in enumerate(chunks)]


class _Writer(object):
    def __init__(self, leaf_k, internal_k=16):
        self.buf = bytearray(96)                # superblock, patched at the end
        self.leaf_k, self.internal_k = leaf_k, internal_k

    def alloc(self, data):
        self.buf += b'\x00' * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    @staticmethod
    def header(messages):
        """version 1 object header: [(type, body)]"""
        body = b''
        for mtype, mbody in messages:
            mbody = _pad8(mbody)
            body += struct.pack('<HHB3x', mtype, len(mbody), 0) + mbody
        return struct.pack('<BBHII4x', 1, 0, len(messages), 1, len(body)) + body

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.dtype.byteorder == '>':
            arr = arr.astype(arr.dtype.newbyteorder('<'))
        raw = arr.tobytes()
        data_addr = self.alloc(raw) if raw else UNDEF
        layout = struct.pack('<BBQQ', 3, 1, data_addr, len(raw))                    # version 3, contiguous
        fill = struct.pack('<BBBB', 2, 2, 0, 0)                                     # version 2, late allocation, undefined
        return self.alloc(self.header([(0x01, _dataspace_msg(arr.shape)), (0x03, _datatype_msg(arr.dtype)), (0x05, fill),
                                       (0x08, layout)]))

    def group(self, g):
        """-> (object header address, B-tree address, local heap address)"""
        links = []
        for name in sorted(g.children, key=lambda s: s.encode('utf8')):             # symbol nodes are sorted by name (strcmp)
            child = g.children[name]
            links.append((name, self.group(child)[0] if isinstance(child, _WGroup) else self.dataset(child)))
        # local heap: the link names, 8-byte aligned, the empty string at offset 0; the tail is one free block
        heap, offs = bytearray(8), []
        for name, _ in links:
            offs.append(len(heap))
            heap += _pad8(name.encode('utf8') + b'\x00')
        free_off = len(heap)
        heap += struct.pack('<QQ', 1, 32) + b'\x00' * 16                            # free block: next = 1 (none), size 32
        heap_hdr_addr = self.alloc(b'')
        heap_addr = self.alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), free_off, heap_hdr_addr + 32) + bytes(heap))
        # Symbol nodes of <= 2 * leaf_k links each under a v1 B-tree (node type 0) whose nodes hold <= 2 * internal_k children:
        # key i of a node is the heap offset of the LAST name below child i - 1 (key 0 of the leftmost node: the empty string at
        # offset 0), so a group of any size costs its own links only (the first version raised the file-wide leaf K to the
        # largest group and padded every group's node to it: O(groups x max_links) bytes).
        cap = 2 * self.leaf_k
        snod_size = 8 + cap * 40
        level = []                                                                  # [(address, heap offset of the last name)]
        for lo in range(0, len(links), cap):
            part = list(zip(links[lo:lo + cap], offs[lo:lo + cap]))
            snod = b'SNOD' + struct.pack('<BBH', 1, 0, len(part))
            for (name, addr), off in part:
                snod += struct.pack('<QQII16x', off, addr, 0, 0)
            snod += b'\x00' * (snod_size - len(snod))
            level.append((self.alloc(snod), part[-1][1]))
        fan = 2 * self.internal_k
        node_size = 24 + (2 * fan + 1) * 8
        depth = 0
        while True:
            groups_ = [level[lo:lo + fan] for lo in range(0, len(level), fan)] or [[]]
            base = self.alloc(b'\x00' * (node_size * len(groups_)))                 # the nodes of a level are contiguous
            nxt = []
            for i, kids in enumerate(groups_):
                left = base + (i - 1) * node_size if i > 0 else UNDEF
                right = base + (i + 1) * node_size if i + 1 < len(groups_) else UNDEF
                node = b'TREE' + struct.pack('<BBHQQ', 0, depth, len(kids), left, right)
                first_key = groups_[i - 1][-1][1] if i > 0 else 0
                node += struct.pack('<Q', first_key)
                for addr, last in kids:
                    node += struct.pack('<QQ', addr, last)
                node += b'\x00' * (node_size - len(node))
                self.buf[base + i * node_size:base + (i + 1) * node_size] = node
                nxt.append((base + i * node_size, kids[-1][1] if kids else 0))
            level = nxt
            depth += 1
            if len(level) == 1:
                break
        tree_addr = level[0][0]
        msgs = [(0x11, struct.pack('<QQ', tree_addr, heap_addr))]
        for name, value in g.attrs:
            for n2, v2 in _split_attr(name, value):
                msgs.append((0x0C, _attr_msg(n2, v2)))
        return self.alloc(self.header(msgs)), tree_addr, heap_addr

    def finish(self, root):
        hdr, tree, heap = self.group(root)
        self.buf += b'\x00' * (-len(self.buf) % 8)
        sb = SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, self.leaf_k, self.internal_k, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack('<QQII', 0, hdr, 1, 0) + struct.pack('<QQ', tree, heap)   # root entry, cached symbol-table addresses
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_tree(path, root):
    """root: _WGroup.  Atomic (temporary file + rename)."""
    import os
    data = _Writer(4).finish(root)          # libhdf5's defaults: group leaf K = 4, internal K = 16
    tmp = '%s.tmp%d' % (path, os.getpid())
    with open(tmp, 'wb') as f:
        f.write(data)
    os.replace(tmp, path)


def write_keras_file(path, layers, model_config=None, training_config=None, keras_version='2.2.4-tf', backend='tensorflow',
                     optimizer_weights=None, extra_attrs=None):
    """
    Write what Keras writes (keras.engine.saving.save_weights_to_hdf5_group / save_model_to_hdf5, TF 2.1): `layers` is the
    ordered list [(layer_name, [(weight_name, ndarray), ...])] of ALL layers (weightless ones with an empty list).
    model_config None: a weights file (`model.save_weights(path, save_format='h5')`, reference DLWP/custom.py:186) -- root
    attributes layer_names / backend / keras_version, one group per layer with attribute weight_names and one dataset per weight
    at <layer>/<weight name> (a weight name like 'conv/kernel:0' makes a sub-group, as in h5py).  With model_config (JSON str):
    a model file (`model.save(path)`, DLWP/util.py:139) -- the same tree under /model_weights, model_config and training_config
    as root attributes; optimizer_weights [(name, ndarray)] (keras: the optimizer's `weights` in order -- iterations, then the
    slots) go to /optimizer_weights with attribute weight_names; extra_attrs [(name, value)] are added to the root.
    """
    root = _WGroup()
    wroot = root
    if model_config is not None:
        root.attrs += [('keras_version', keras_version), ('backend', backend), ('model_config', model_config)]
        if training_config is not None:
            root.attrs.append(('training_config', training_config))
        wroot = root.group('model_weights')
        if optimizer_weights:
            og = root.group('optimizer_weights')
            og.attrs.append(('weight_names', [n for n, _ in optimizer_weights]))
            for wname, arr in optimizer_weights:
                parts = wname.split('/')
                h = og
                for p in parts[:-1]:
                    h = h.group(p)
                h.children[parts[-1]] = np.asarray(arr)
    root.attrs += list(extra_attrs or [])
    wroot.attrs += [('layer_names', [n for n, _ in layers]), ('backend', backend), ('keras_version', keras_version)]
    for lname, ws in layers:
        g = wroot.group(lname)
        g.attrs.append(('weight_names', [n for n, _ in ws]))
        for wname, arr in ws:
            parts = wname.split('/')
            h = g
            for p in parts[:-1]:
                h = h.group(p)
            h.children[parts[-1]] = np.asarray(arr)
    write_tree(path, root)
