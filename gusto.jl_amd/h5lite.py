"""Minimal HDF5 writer for the trajectory hand-off of examples/freeflyerSE2.ipynb cell 6 (`h5open(...)`: group `traj`
with x_traj / u_traj / t_traj, groups ind_x / ind_u of scalar integers).

This image has no HDF5 library (no libhdf5, h5py, HDF5.jl), so the container is written directly to the "HDF5 File
Format Specification" (version 1.1 of the format: superblock version 0, version-1 object headers, version-1 group
B-trees + local heaps + symbol table nodes, contiguous little-endian data sets) -- the subset every HDF5 library since
1.0 reads.  Supported leaves: float64, int64 and int32 arrays of any rank, and scalars of those types (rank-0
dataspace, what HDF5.jl writes for `g["x"] = 0`).  Groups may nest; a group holds at most 2 * LEAF_K entries.

Array convention: a numpy array of shape (d0, ..., dk) in C order is stored with HDF5 dimensions (d0, ..., dk); HDF5.jl
(column-major) then sees the reversed shape, so the C-ABI layout X[k][i] (shape [N, n]) is the notebook's X[n, N].
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 16, 16          # symbol table nodes hold 2 * LEAF_K entries, B-tree nodes 2 * INTERNAL_K children
SIGNATURE = b"\x89HDF\r\n\x1a\n"


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _message(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(messages):
    body = b"".join(messages)
    # version 1, reserved, number of messages, reference count, size of the message block; 4 bytes align it to 8
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


def _datatype(dt):
    if dt == np.float64:
        # class 1 (floating point), version 1; little endian, mantissa normalisation 2 (implied msb), sign bit 63
        return struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    size = {np.dtype(np.int64): 8, np.dtype(np.int32): 4}[np.dtype(dt)]
    # class 0 (fixed point), version 1; little endian, two's complement signed
    return struct.pack("<BBBBI", 0x10, 0x08, 0, 0, size) + struct.pack("<HH", 0, 8 * size)


def _dataspace(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


class _Writer:
    def __init__(self):
        self.buf = bytearray()

    def tell(self):
        return len(self.buf)

    def put(self, b):
        """append at the next 8-byte boundary, return the address"""
        self.buf += b"\0" * (-len(self.buf) % 8)
        at = len(self.buf)
        self.buf += b
        return at

    def dataset(self, value):
        a = np.asarray(value)
        if a.dtype == np.bool_ or (a.dtype.kind in "iu" and a.dtype != np.int32):
            a = a.astype(np.int64)
        elif a.dtype.kind == "f":
            a = a.astype(np.float64)
        elif a.dtype != np.int32:
            raise TypeError(f"h5lite: unsupported leaf type {a.dtype}")
        raw = np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes()
        data_at = self.put(raw) if raw else UNDEF
        msgs = [_message(0x0001, _dataspace(a.shape)),
                _message(0x0003, _datatype(a.dtype), flags=1),                       # (constant message)
                _message(0x0005, struct.pack("<BBBB", 2, 2, 0, 0)),                  # fill value v2: late allocation, none defined
                _message(0x0008, struct.pack("<BBQQ", 3, 1, data_at, len(raw)))]     # layout v3, contiguous
        return self.put(_object_header(msgs))

    def group(self, tree):
        """writes the members, then heap, symbol table node, B-tree node and the group's object header; returns
        (object header address, B-tree address, heap address)"""
        names = sorted(tree, key=lambda s: s.encode())
        if len(names) > 2 * LEAF_K:
            raise ValueError(f"h5lite: a group holds at most {2 * LEAF_K} entries")
        entries = []
        for name in names:
            v = tree[name]
            if isinstance(v, dict):
                oh, bt, hp = self.group(v)
                entries.append((name, oh, 1, struct.pack("<QQ", bt, hp)))
            else:
                entries.append((name, self.dataset(v), 0, b"\0" * 16))
        # local heap: offset 0 is the empty string (key 0 of the B-tree), names null-terminated on 8-byte boundaries
        seg, off = bytearray(b"\0" * 8), {}
        for name in names:
            off[name] = len(seg)
            seg += _pad8(name.encode() + b"\0")
        free = len(seg)
        seg += struct.pack("<QQ", 1, 16)                    # one free block closing the segment: (next = none, size 16)
        seg_at = self.put(bytes(seg))
        heap_at = self.put(b"HEAP" + struct.pack("<B3xQQQ", 0, len(seg), free, seg_at))
        snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(entries))
        for name, oh, cache, scratch in entries:
            snod += struct.pack("<QQII", off[name], oh, cache, 0) + scratch
        snod += b"\0" * (40 * (2 * LEAF_K - len(entries)))
        snod_at = self.put(snod)
        tree_node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if entries else 0, UNDEF, UNDEF)
        tree_node += struct.pack("<QQQ", 0, snod_at, off[names[-1]] if names else 0)
        tree_node += b"\0" * (8 * (2 * INTERNAL_K + 1) + 8 * 2 * INTERNAL_K - 24)
        bt_at = self.put(tree_node)
        oh_at = self.put(_object_header([_message(0x0011, struct.pack("<QQ", bt_at, heap_at))]))
        return oh_at, bt_at, heap_at


def write_h5(path, tree):
    """`tree`: nested dicts (groups) of arrays / scalars (data sets).  Writes `path`, returns the number of bytes."""
    w = _Writer()
    w.buf += b"\0" * 96                                      # superblock, filled in last
    oh, bt, hp = w.group(tree)
    w.buf += b"\0" * (-len(w.buf) % 8)
    eof = len(w.buf)
    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)          # base, free-space info, end of file, driver info
    sb += struct.pack("<QQII", 0, oh, 1, 0) + struct.pack("<QQ", bt, hp)   # root group symbol table entry
    assert len(sb) == 96
    w.buf[0:96] = sb
    with open(path, "wb") as f:
        f.write(bytes(w.buf))
    return eof
