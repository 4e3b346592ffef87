"""Independent reader of the HDF5 subset gusto.jl_amd/h5lite.py writes, following the HDF5 File Format Specification
by the object graph (superblock -> root symbol table entry -> object header -> symbol table message -> B-tree ->
symbol table node -> local heap names -> data set messages).  Test infrastructure: it shares no code with the writer
and checks the structural invariants an HDF5 library checks on open (signatures, node sizes implied by the K values of
the superblock, sorted names, heap bounds, free list, end-of-file address)."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _u(fmt, b, at):
    return struct.unpack_from("<" + fmt, b, at)


def read_h5(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89HDF\r\n\x1a\n"
    sbv, fsv, rgv, _, shv, so, sl, _, leaf_k, int_k, flags = _u("BBBBBBBBHHI", b, 8)
    assert (sbv, fsv, rgv, shv, so, sl, flags) == (0, 0, 0, 0, 8, 8, 0)
    base, freesp, eof, drv = _u("QQQQ", b, 24)
    assert base == 0 and freesp == UNDEF and drv == UNDEF and eof == len(b)
    name_off, oh, cache, _r = _u("QQII", b, 56)
    bt, hp = _u("QQ", b, 80)
    assert cache == 1 and name_off == 0
    ctx = dict(b=b, leaf_k=leaf_k, int_k=int_k)
    return _group(ctx, oh, (bt, hp))


def _messages(b, at):
    ver, _r, n, refc, size = _u("BBHII", b, at)
    assert ver == 1 and refc >= 1 and at % 8 == 0
    p, end, out = at + 16, at + 16 + size, []
    for _ in range(n):
        mtype, msize, mflags = _u("HHB", b, p)
        assert msize % 8 == 0
        out.append((mtype, b[p + 8:p + 8 + msize]))
        p += 8 + msize
    assert p == end
    return out


def _group(ctx, oh, cached=None):
    b = ctx["b"]
    msgs = dict(_messages(b, oh))
    assert 0x0011 in msgs
    bt, hp = _u("QQ", msgs[0x0011], 0)
    if cached:
        assert cached == (bt, hp)
    # local heap
    assert b[hp:hp + 4] == b"HEAP" and b[hp + 4] == 0
    seg_size, free, seg_at = _u("QQQ", b, hp + 8)
    seg = b[seg_at:seg_at + seg_size]
    assert len(seg) == seg_size and seg_size % 8 == 0
    while free != 1:                       # free list: (next, size) blocks inside the segment; 1 terminates it
        assert free % 8 == 0 and free + 16 <= seg_size
        nxt, fsz = _u("QQ", seg, free)
        assert fsz >= 16 and free + fsz <= seg_size
        free = nxt
    name = lambda off: seg[off:seg.index(b"\0", off)].decode()
    # B-tree: one leaf node of type 0 (group), node size fixed by the internal K of the superblock
    assert b[bt:bt + 4] == b"TREE"
    ntype, level, used, left, right = _u("BBHQQ", b, bt + 4)
    assert (ntype, level, left, right) == (0, 0, UNDEF, UNDEF) and used <= 1
    assert bt + 24 + 8 * (2 * ctx["int_k"] + 1) + 8 * 2 * ctx["int_k"] <= len(b)
    out = {}
    if used == 0:
        return out
    key0, child, key1 = _u("QQQ", b, bt + 24)
    assert name(key0) == ""
    assert b[child:child + 4] == b"SNOD" and b[child + 4] == 1
    nsym, = _u("H", b, child + 6)
    assert nsym <= 2 * ctx["leaf_k"] and child + 8 + 40 * 2 * ctx["leaf_k"] <= len(b)
    names = []
    for i in range(nsym):
        e = child + 8 + 40 * i
        noff, ohx, cache, _r = _u("QQII", b, e)
        nm = name(noff)
        names.append(nm)
        if cache == 1:
            out[nm] = _group(ctx, ohx, _u("QQ", b, e + 24))
        else:
            assert cache == 0
            out[nm] = _dataset(b, ohx)
    assert names == sorted(names, key=lambda s: s.encode()) and name(key1) == names[-1]
    return out


def _dataset(b, oh):
    msgs = dict(_messages(b, oh))
    sp, dt, lay = msgs[0x0001], msgs[0x0003], msgs[0x0008]
    ver, rank, fl = _u("BBB", sp, 0)
    assert ver == 1 and fl == 0
    shape = _u("Q" * rank, sp, 8) if rank else ()
    cv, b0, b1, b2, size = _u("BBBBI", dt, 0)
    assert cv >> 4 == 1
    if cv & 15 == 1:
        assert (b0, b1, size) == (0x20, 63, 8) and _u("HHBBBBI", dt, 8) == (0, 64, 52, 11, 0, 52, 1023)
        np_dt = "<f8"
    else:
        assert cv & 15 == 0 and b0 == 0x08 and _u("HH", dt, 8) == (0, 8 * size)
        np_dt = {8: "<i8", 4: "<i4"}[size]
    lv, lc, addr, nbytes = _u("BBQQ", lay, 0)
    assert (lv, lc) == (3, 1) and nbytes == size * int(np.prod(shape, dtype=np.int64))
    if 0x0005 in msgs:
        assert msgs[0x0005][0] == 2 and msgs[0x0005][3] == 0
    a = np.frombuffer(b, dtype=np_dt, count=nbytes // size, offset=addr).reshape(shape) if nbytes else np.zeros(shape, np_dt)
    return a[()] if rank == 0 else a.copy()
