"""Byte-level writer of the HDF5 subset `gnn_uds_amd/h5.py` reads (TEST INFRASTRUCTURE): superblock version 0, old-style
groups (symbol-table message -> version-1 B-tree -> symbol-table node + local heap), version-1 object headers with an
optional continuation block, contiguous or compact datasets of IEEE floats / integers.  Written from the HDF5 File Format
Specification -- the structure libhdf5 produces with the default `libver='earliest'`, which is what Keras' `save_weights`
goes through -- because nothing in this image can write a genuine file (no h5py, no libhdf5 bindings)."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Writer:
    def __init__(self):
        self.buf = bytearray(b'\x00' * 96)              # superblock 0 with 8-byte offsets: 56 + 40 bytes

    def alloc(self, data, align=8):
        while len(self.buf) % align:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    @staticmethod
    def _msg(typ, data, flags=0):
        data = bytes(data)
        data += b'\x00' * (-len(data) % 8)
        return struct.pack('<HHB3x', typ, len(data), flags) + data

    def _object_header(self, msgs, split=False):
        """Version-1 object header; split=True moves all but the first message into a continuation block."""
        if split and len(msgs) > 1:
            tail = b''.join(msgs[1:])
            tail_addr = self.alloc(tail)
            first = msgs[0] + self._msg(0x10, struct.pack('<QQ', tail_addr, len(tail)))
            n = len(msgs) + 1
        else:
            first, n = b''.join(msgs), len(msgs)
        return self.alloc(struct.pack('<BxHII4x', 1, n, 1, len(first)) + first)

    def dataset(self, arr, compact=False, split=False, dataspace_version=1):
        arr = np.asarray(arr, order="C")          # (ascontiguousarray would turn a scalar into shape (1,))
        dt = arr.dtype
        if dt.kind == 'f':
            # class 1 (float), version 1; bit field: little endian, IEEE; properties: bit offset, precision, exponent / mantissa layout
            props = {4: struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127), 8: struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)}[dt.itemsize]
            dtype_msg = struct.pack('<BBBBI', 0x11, 0x20, 0x1F if dt.itemsize == 4 else 0x3F, 0, dt.itemsize) + props
        else:
            dtype_msg = struct.pack('<BBBBI', 0x10, 0x08 if dt.kind == 'i' else 0, 0, 0, dt.itemsize) + struct.pack('<HH', 0, 8 * dt.itemsize)
        if dataspace_version == 1:
            space = struct.pack('<BBB5x', 1, arr.ndim, 0) + b''.join(struct.pack('<Q', d) for d in arr.shape)
        else:
            space = struct.pack('<BBBB', 2, arr.ndim, 0, 1) + b''.join(struct.pack('<Q', d) for d in arr.shape)
        raw = arr.astype(dt.newbyteorder('<')).tobytes()
        if compact:
            layout = struct.pack('<BBH', 3, 0, len(raw)) + raw
        else:
            addr = self.alloc(raw) if raw else UNDEF
            layout = struct.pack('<BBQQ', 3, 1, addr, len(raw))
        return self._object_header([self._msg(0x01, space), self._msg(0x03, dtype_msg, 1), self._msg(0x08, layout)], split)

    def group(self, entries, nodes=1):
        """entries: {name: object header address}; `nodes` symbol-table nodes under one B-tree node (names sorted, as the format
        requires).  Returns the object header address of the group."""
        names = sorted(entries)
        heap_data = bytearray(b'\x00' * 8)              # offset 0: the empty string (name of the root link)
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += n.encode() + b'\x00'
            heap_data += b'\x00' * (-len(heap_data) % 8)
        data_addr = self.alloc(bytes(heap_data))
        heap = self.alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap_data), UNDEF, data_addr))
        chunks = [names[i::nodes] for i in range(nodes)] if nodes > 1 else [names]
        chunks = [sorted(c) for c in chunks if c]
        # contiguous runs keep the sort order across nodes
        per = -(-len(names) // max(1, len(chunks))) if names else 0
        chunks = [names[i:i + per] for i in range(0, len(names), per)] if names else [[]]
        snods, keys = [], [0]
        for c in chunks:
            body = b''.join(struct.pack('<QQII16x', offs[n], entries[n], 0, 0) for n in c)
            snods.append(self.alloc(b'SNOD' + struct.pack('<BxH', 1, len(c)) + body))
            keys.append(offs[c[-1]] if c else 0)
        tree = b'TREE' + struct.pack('<BBHQQ', 0, 0, len(snods), UNDEF, UNDEF)
        for k, child in zip(keys, snods):
            tree += struct.pack('<QQ', k, child)
        tree += struct.pack('<Q', keys[-1])
        btree = self.alloc(tree)
        return self._object_header([self._msg(0x11, struct.pack('<QQ', btree, heap))])

    def finish(self, root):
        sb = b'\x89HDF\r\n\x1a\n' + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack('<QQII16x', 0, root, 0, 0)
        assert len(sb) == 96, len(sb)
        self.buf[:96] = sb
        return bytes(self.buf)


def write_tree(path, tree, **opts):
    """tree: nested dict, leaves are arrays.  opts: compact / split (per-dataset callables name -> bool), nodes per group."""
    w = Writer()

    def build(node, prefix):
        entries = {}
        for name, v in node.items():
            full = prefix + [name]
            if isinstance(v, dict):
                entries[name] = build(v, full)
            else:
                key = '/'.join(full)
                entries[name] = w.dataset(v, compact=opts.get('compact', lambda k: False)(key), split=opts.get('split', lambda k: False)(key),
                                          dataspace_version=opts.get('dataspace_version', lambda k: 1)(key))
        return w.group(entries, nodes=opts.get('nodes', 1))
    data = w.finish(build(tree, []))
    with open(path, 'wb') as fh:
        fh.write(data)
    return data


def keras_tree(weights):
    """{'layer/weight:0': array} -> the nested dict Keras' save_weights lays out: /layer/layer/weight:0 (the weight name itself
    starts with the layer name, and '/' in it nests groups)."""
    tree = {}
    for key, arr in weights.items():
        layer = key.split('/')[0]
        node = tree.setdefault(layer, {})
        parts = key.split('/')
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = np.asarray(arr)
    return tree
