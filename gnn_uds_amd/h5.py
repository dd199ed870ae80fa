"""Minimal HDF5 reader for Keras weight files (`model.h5`, reference emulator.py:814-822,833-838).

h5py is not part of this image, and all `Emulator.load` needs from a `model.save_weights('model.h5')` file is the set
of float arrays keyed by their group path (`dense_4/dense_4/kernel:0`, ...).  This module reads exactly that subset of
the HDF5 file format, written against the published format specification (HDF5 File Format Specification 2.0 / 3.0):

  * superblock version 0 or 1 (what libhdf5 writes with the default `libver='earliest'`, which Keras / h5py use),
  * old-style groups: symbol-table message -> version-1 B-tree of symbol-table nodes + local heap for the names,
  * version-1 object headers incl. continuation blocks,
  * datasets with contiguous or compact layout (layout message version 3), fixed-point and IEEE float datatypes,
    dataspace message versions 1 and 2.

Anything else (superblock 2 / 3, `OHDR` version-2 object headers, link messages of new-style groups, chunked or filtered
datasets, variable-length / compound types) raises `NotImplementedError` naming the feature, so a file written with
`libver='latest'` or with compression fails loudly instead of being misread.  Attributes (`layer_names`,
`weight_names`, `keras_version`) are not needed: the importer matches arrays by group path.

Status: the reference ships no `.h5` file and nothing here can write one with libhdf5, so the reader is tested on files
built byte by byte from the same specification by `tests/h5_writer.py` (structure-faithful: superblock 0, symbol-table
groups, nested groups, continuation blocks, compact and contiguous data) -- not on a file produced by h5py itself.
"""
import struct

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF

MSG_DATASPACE, MSG_LINK_INFO, MSG_DATATYPE, MSG_LINK, MSG_LAYOUT, MSG_FILTER = 0x01, 0x02, 0x03, 0x06, 0x08, 0x0B
MSG_CONTINUATION, MSG_SYMBOL_TABLE = 0x10, 0x11


class H5FormatError(ValueError):
    pass


class _Reader:
    def __init__(self, buf):
        self.b = buf
        base = 0
        while buf[base:base + 8] != SIGNATURE:          # the superblock may sit at 0, 512, 1024, ... (user block)
            base = 512 if base == 0 else base * 2
            if base + 8 > len(buf):
                raise H5FormatError('not an HDF5 file (no superblock signature)')
        ver = buf[base + 8]
        if ver not in (0, 1):
            raise NotImplementedError('HDF5 superblock version %d (file written with libver="latest"?): only versions 0 / 1, '
                                      'the libhdf5 default that Keras uses, are read' % ver)
        self.so, self.sl = buf[base + 13], buf[base + 14]          # size of offsets / lengths
        if (self.so, self.sl) != (8, 8):
            raise NotImplementedError('HDF5 file with %d-byte offsets / %d-byte lengths (8 / 8 expected)' % (self.so, self.sl))
        p = base + 24 + (4 if ver == 1 else 0)
        self.base_addr = self.u64(p)
        p += 4 * 8                                       # base, free-space info, end of file, driver info
        # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch pad
        self.root = self.u64(p + 8)

    # --- primitives
    def u8(self, p):
        return self.b[p]

    def u16(self, p):
        return struct.unpack_from('<H', self.b, p)[0]

    def u32(self, p):
        return struct.unpack_from('<I', self.b, p)[0]

    def u64(self, p):
        return struct.unpack_from('<Q', self.b, p)[0]

    def at(self, addr):
        return self.base_addr + addr

    # --- object headers
    def messages(self, addr):
        """[(type, flags, offset of the message data, size)] of a version-1 object header, continuation blocks followed."""
        p = self.at(addr)
        if self.b[p:p + 4] == b'OHDR':
            raise NotImplementedError('version-2 object header (file written with libver="latest")')
        if self.u8(p) != 1:
            raise H5FormatError('object header version %d at %#x' % (self.u8(p), p))
        n_msg, size = self.u16(p + 2), self.u32(p + 8)
        blocks = [(p + 16, size)]                        # 12 bytes of prefix + 4 of padding to the 8-byte boundary
        out = []
        while blocks and len(out) < n_msg:
            q, left = blocks.pop(0)
            while left >= 8 and len(out) < n_msg:
                typ, sz, flags = self.u16(q), self.u16(q + 2), self.u8(q + 4)
                data = q + 8
                if typ == MSG_CONTINUATION:
                    blocks.append((self.at(self.u64(data)), self.u64(data + 8)))
                out.append((typ, flags, data, sz))
                q += 8 + sz
                left -= 8 + sz
        return out

    # --- groups
    def heap_name(self, heap_addr, off):
        p = self.at(heap_addr)
        if self.b[p:p + 4] != b'HEAP':
            raise H5FormatError('local heap signature missing at %#x' % p)
        data = self.at(self.u64(p + 24))
        end = self.b.index(b'\x00', data + off)
        return bytes(self.b[data + off:end]).decode('utf-8')

    def group_entries(self, btree_addr, heap_addr):
        """[(name, object header address)] of an old-style group."""
        out = []
        p = self.at(btree_addr)
        if self.b[p:p + 4] != b'TREE':
            raise H5FormatError('B-tree signature missing at %#x' % p)
        if self.u8(p + 4) != 0:
            raise H5FormatError('group B-tree node of type %d' % self.u8(p + 4))
        level, used = self.u8(p + 5), self.u16(p + 6)
        q = p + 8 + 16                                   # left / right sibling addresses
        for i in range(used):
            child = self.u64(q + 8)                      # key i (8 bytes), child i
            q += 16
            if level > 0:
                out += self.group_entries(child, heap_addr)
                continue
            s = self.at(child)
            if self.b[s:s + 4] != b'SNOD':
                raise H5FormatError('symbol-table node signature missing at %#x' % s)
            n = self.u16(s + 6)
            for k in range(n):
                e = s + 8 + 40 * k
                out.append((self.heap_name(heap_addr, self.u64(e)), self.u64(e + 8)))
        return out

    # --- datasets
    def datatype(self, p):
        cls, ver = self.u8(p) & 0x0F, self.u8(p) >> 4
        bits0, size = self.u8(p + 1), self.u32(p + 4)
        if ver not in (1, 2, 3):
            raise H5FormatError('datatype message version %d' % ver)
        order = '>' if bits0 & 1 else '<'
        if cls == 1 and size in (2, 4, 8):
            return np.dtype(order + 'f%d' % size)
        if cls == 0 and size in (1, 2, 4, 8):
            return np.dtype(order + ('i' if bits0 & 8 else 'u') + '%d' % size)
        raise NotImplementedError('HDF5 datatype class %d of %d bytes (only fixed-point and IEEE float arrays are read)' % (cls, size))

    def dataspace(self, p):
        ver, rank = self.u8(p), self.u8(p + 1)
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if self.u8(p + 3) == 2:                      # null dataspace
                return (0,)
            q = p + 4
        else:
            raise H5FormatError('dataspace message version %d' % ver)
        return tuple(self.u64(q + 8 * i) for i in range(rank))

    def dataset(self, msgs, path):
        dtype = shape = None
        data = None
        for typ, flags, p, sz in msgs:
            if typ == MSG_FILTER:
                raise NotImplementedError('%s: filtered (compressed) dataset' % path)
            if typ == MSG_DATATYPE:
                dtype = self.datatype(p)
            elif typ == MSG_DATASPACE:
                shape = self.dataspace(p)
            elif typ == MSG_LAYOUT:
                ver, cls = self.u8(p), self.u8(p + 1)
                if ver != 3:
                    raise NotImplementedError('%s: data layout message version %d' % (path, ver))
                if cls == 1:
                    data = ('contiguous', self.u64(p + 2), self.u64(p + 10))
                elif cls == 0:
                    data = ('compact', p + 4, self.u16(p + 2))
                else:
                    raise NotImplementedError('%s: chunked dataset (Keras writes contiguous ones)' % path)
        if dtype is None or shape is None or data is None:
            raise H5FormatError('%s: dataset without datatype / dataspace / layout message' % path)
        count = int(np.prod(shape)) if len(shape) else 1
        if data[0] == 'contiguous':
            if data[1] == UNDEF or count == 0:
                return np.zeros(shape, dtype=dtype.newbyteorder('='))
            start = self.at(data[1])
        else:
            start = data[1]
        arr = np.frombuffer(self.b, dtype=dtype, count=count, offset=start).reshape(shape)
        return arr.astype(dtype.newbyteorder('='))

    def walk(self, addr, prefix, out, seen):
        if addr in seen:                                 # hard links / cycles: visit once
            return
        seen.add(addr)
        msgs = self.messages(addr)
        types = {m[0] for m in msgs}
        if MSG_SYMBOL_TABLE in types:
            for typ, _, p, _ in msgs:
                if typ == MSG_SYMBOL_TABLE:
                    for name, child in self.group_entries(self.u64(p), self.u64(p + 8)):
                        self.walk(child, prefix + [name], out, seen)
        elif MSG_LINK_INFO in types or MSG_LINK in types:
            raise NotImplementedError('new-style group (link messages) at /%s: file written with libver="latest"' % '/'.join(prefix))
        elif MSG_LAYOUT in types:
            out['/'.join(prefix)] = self.dataset(msgs, '/'.join(prefix))
        # anything else (committed datatypes, empty objects) carries no weights


def read_datasets(path):
    """{group path: ndarray} of every dataset in the file, e.g. 'dense_4/dense_4/kernel:0' for a Keras `save_weights` file
    (a full `model.save` file keeps the same tree under 'model_weights/')."""
    with open(path, 'rb') as fh:
        buf = fh.read()
    r = _Reader(buf)
    out = {}
    r.walk(r.root, [], out, set())
    return out


def read_keras_weights(path):
    """`read_datasets` with the 'model_weights/' prefix of full-model files removed: what `Emulator.load_keras_weights` takes."""
    out = {}
    for k, v in read_datasets(path).items():
        out[k[len('model_weights/'):] if k.startswith('model_weights/') else k] = v
    return out
