"""A small pure-Python reader/writer for the subset of HDF5 that Keras 2.2.x checkpoints use (there is no h5py in this
image).  The reference saves models with ``ModelCheckpoint`` / ``model.save`` (experiments/train_siamese.py:74-80) and
loads them with ``keras.models.load_model``; the file layout is the one libhdf5 writes with its default ("earliest")
format bounds:

  superblock v0/v1 -> root symbol-table entry -> groups as (B-tree v1 + local heap + symbol-table nodes) -> object
  headers v1 (with continuation blocks) -> datasets with contiguous / compact (or unfiltered chunked) layout, fixed-point,
  floating-point and fixed-length string datatypes; attributes v1-v3 (fixed-length and variable-length strings through the
  global heap, numeric and string arrays).

Reading follows "HDF5 File Format Specification Version 2.0" (sections III.A-E, IV.A); the writer emits the same subset
(superblock v0, one symbol-table group per Keras group, contiguous little-endian datasets, fixed-length string attributes).
The API mirrors the slice of h5py that Keras' saving code touches: ``File(path)[name]``, ``.keys()``, ``.attrs``,
``dataset[()]`` / ``np.asarray(dataset)``, ``.shape``, ``.dtype``.
"""
from __future__ import annotations

import struct
from collections import OrderedDict
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(ValueError):
    pass


# =========================================================================================================
# reading
# =========================================================================================================
class _Reader:
    def __init__(self, buf: bytes):
        self.buf = buf
        start = 0
        while buf[start:start + 8] != SIGNATURE:  # the superblock may sit at 0, 512, 1024, ... (spec III.A)
            start = 512 if start == 0 else start * 2
            if start >= len(buf):
                raise Hdf5FormatError("not an HDF5 file (no superblock signature)")
        ver = buf[start + 8]
        if ver > 1:
            raise Hdf5FormatError("superblock version %d is not supported (written with libver='latest'?)" % ver)
        self.O, self.L = buf[start + 13], buf[start + 14]
        if self.O != 8 or self.L != 8:
            raise Hdf5FormatError("only 8-byte offsets/lengths are supported")
        pos = start + 24 + (4 if ver == 1 else 0)
        self.base = self.u64(pos)
        pos += 4 * 8  # base, free-space info, end of file, driver info
        self.root_entry = self.symbol_entry(pos)
        self._gheaps: Dict[int, Dict[int, bytes]] = {}

    # ---- primitives ----
    def u8(self, p):
        return self.buf[p]

    def u16(self, p):
        return struct.unpack_from("<H", self.buf, p)[0]

    def u32(self, p):
        return struct.unpack_from("<I", self.buf, p)[0]

    def u64(self, p):
        return struct.unpack_from("<Q", self.buf, p)[0]

    def symbol_entry(self, p):
        """(link name offset, object header address, cache type, scratch) -- spec III.C."""
        return self.u64(p), self.u64(p + 8), self.u32(p + 16), self.buf[p + 24:p + 40]

    # ---- object headers (v1) ----
    def messages(self, addr) -> List[Tuple[int, int, bytes]]:
        """[(type, flags, body)] of the version-1 object header at addr, continuation blocks followed."""
        a = self.base + addr
        if self.buf[a:a + 4] == b"OHDR":
            raise Hdf5FormatError("version-2 object headers are not supported (file written with libver='latest')")
        if self.u8(a) != 1:
            raise Hdf5FormatError("object header version %d" % self.u8(a))
        nmsg, size = self.u16(a + 2), self.u32(a + 8)
        blocks = [(a + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, remaining = blocks.pop(0)
            end = p + remaining
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self.u16(p), self.u16(p + 2), self.u8(p + 4)
                body = self.buf[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:  # continuation
                    blocks.append((self.base + struct.unpack_from("<Q", body, 0)[0], struct.unpack_from("<Q", body, 8)[0]))
                out.append((mtype, flags, body))
        return out

    # ---- groups ----
    def group_links(self, btree_addr, heap_addr) -> "OrderedDict[str, int]":
        h = self.base + heap_addr
        if self.buf[h:h + 4] != b"HEAP":
            raise Hdf5FormatError("bad local heap signature")
        data = self.base + self.u64(h + 24)
        links: "OrderedDict[str, int]" = OrderedDict()

        def name_at(off):
            e = self.buf.index(b"\x00", data + off)
            return self.buf[data + off:e].decode("utf8")

        def walk(addr):
            a = self.base + addr
            if self.buf[a:a + 4] == b"TREE":
                if self.u8(a + 4) != 0:
                    raise Hdf5FormatError("group B-tree of node type %d" % self.u8(a + 4))
                used = self.u16(a + 6)
                p = a + 8 + 16  # siblings
                for i in range(used):
                    child = self.u64(p + 8 + i * 16)  # key_i (8) child_i (8) ...
                    walk(child)
            elif self.buf[a:a + 4] == b"SNOD":
                n = self.u16(a + 6)
                for i in range(n):
                    off, ohdr, _, _ = self.symbol_entry(a + 8 + i * 40)
                    links[name_at(off)] = ohdr
            else:
                raise Hdf5FormatError("bad group node signature %r" % self.buf[a:a + 4])
        walk(btree_addr)
        return links

    # ---- datatypes / dataspaces ----
    def datatype(self, b: bytes, p: int = 0):
        """-> (numpy dtype | ('vlen_str',) | ('vlen', base), size in bytes)."""
        cls, ver = b[p] & 0x0F, b[p] >> 4
        bits = b[p + 1] | (b[p + 2] << 8) | (b[p + 3] << 16)
        size = struct.unpack_from("<I", b, p + 4)[0]
        order = ">" if (bits & 1) else "<"
        if cls == 0:
            return np.dtype("%s%s%d" % (order, "i" if bits & 8 else "u", size)), size
        if cls == 1:
            return np.dtype("%sf%d" % (order, size)), size
        if cls == 3:
            return np.dtype("S%d" % size), size
        if cls == 9:
            if (bits & 0x0F) == 1:
                return ("vlen_str",), size
            base, _ = self.datatype(b, p + 8)
            return ("vlen", base), size
        if cls == 6 and ver in (1, 2, 3):  # compound: not needed for Keras files
            raise Hdf5FormatError("compound datatypes are not supported")
        raise Hdf5FormatError("datatype class %d is not supported" % cls)

    def dataspace(self, b: bytes) -> Tuple[int, ...]:
        ver, rank = b[0], b[1]
        if ver == 1:
            p = 8
        elif ver == 2:
            if b[3] == 2:  # null dataspace
                return (0,)
            p = 4
        else:
            raise Hdf5FormatError("dataspace version %d" % ver)
        return tuple(struct.unpack_from("<Q", b, p + 8 * i)[0] for i in range(rank))

    def gheap_object(self, addr, index) -> bytes:
        if addr not in self._gheaps:
            a = self.base + addr
            if self.buf[a:a + 4] != b"GCOL":
                raise Hdf5FormatError("bad global heap signature")
            size = self.u64(a + 8)
            objs, p = {}, a + 16
            while p + 16 <= a + size:
                idx, osize = self.u16(p), self.u64(p + 8)
                if idx == 0:
                    break
                objs[idx] = self.buf[p + 16:p + 16 + osize]
                p += 16 + ((osize + 7) // 8) * 8
            self._gheaps[addr] = objs
        return self._gheaps[addr][index]

    def decode(self, dt, shape, raw: bytes):
        n = int(np.prod(shape)) if len(shape) else 1
        if isinstance(dt, tuple):
            vals = []
            for i in range(n):
                ln, addr, idx = struct.unpack_from("<IQI", raw, 16 * i)
                payload = self.gheap_object(addr, idx) if ln else b""
                if dt[0] == "vlen_str":
                    vals.append(payload[:ln].decode("utf8"))
                else:
                    vals.append(np.frombuffer(payload, dtype=dt[1], count=ln).copy())
            if not shape:
                return vals[0]
            arr = np.empty(n, dtype=object)
            arr[:] = vals
            return arr.reshape(shape)
        arr = np.frombuffer(raw, dtype=dt, count=n).copy().reshape(shape)
        if dt.kind in "iuf":
            arr = arr.astype(dt.newbyteorder("="))
        return arr if shape else arr[()]

    def attribute(self, body: bytes):
        ver = body[0]
        nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
        if ver == 1:
            pad = lambda n: (n + 7) // 8 * 8
            p = 8
        elif ver in (2, 3):
            pad = lambda n: n
            p = 8 if ver == 2 else 9
        else:
            raise Hdf5FormatError("attribute message version %d" % ver)
        name = body[p:p + nsz].split(b"\x00")[0].decode("utf8")
        p += pad(nsz)
        dt, esize = self.datatype(body, p)
        p += pad(tsz)
        shape = self.dataspace(body[p:p + ssz])
        p += pad(ssz)
        n = int(np.prod(shape)) if len(shape) else 1
        return name, self.decode(dt, shape, body[p:p + n * esize])


class _Object:
    def __init__(self, rd: _Reader, addr: int, name: str):
        self._rd, self._addr, self.name = rd, addr, name
        self._msgs = rd.messages(addr)
        self._attrs: Optional[Dict[str, object]] = None

    @property
    def attrs(self) -> Dict[str, object]:
        if self._attrs is None:
            self._attrs = OrderedDict()
            for t, _, body in self._msgs:
                if t == 0x000C:
                    k, v = self._rd.attribute(body)
                    self._attrs[k] = v
                elif t == 0x0015:
                    raise Hdf5FormatError("dense attribute storage is not supported")
        return self._attrs


class Dataset(_Object):
    def __init__(self, rd, addr, name):
        super().__init__(rd, addr, name)
        self.shape: Tuple[int, ...] = ()
        self._dt, self._esize, self._layout = None, 0, None
        for t, _, body in self._msgs:
            if t == 0x0001:
                self.shape = rd.dataspace(body)
            elif t == 0x0003:
                self._dt, self._esize = rd.datatype(body)
            elif t == 0x0008:
                self._layout = body
            elif t == 0x000B:
                raise Hdf5FormatError("filtered (compressed) datasets are not supported: %s" % name)
        if self._dt is None or self._layout is None:
            raise Hdf5FormatError("dataset %s lacks a datatype or layout message" % name)

    @property
    def dtype(self):
        return self._dt if not isinstance(self._dt, tuple) else np.dtype(object)

    def _raw(self) -> bytes:
        rd, b = self._rd, self._layout
        n = (int(np.prod(self.shape)) if self.shape else 1) * self._esize
        if b[0] != 3:
            raise Hdf5FormatError("data layout message version %d" % b[0])
        cls = b[1]
        if cls == 0:  # compact
            size = struct.unpack_from("<H", b, 2)[0]
            return b[4:4 + size]
        if cls == 1:  # contiguous
            addr = struct.unpack_from("<Q", b, 2)[0]
            if addr == UNDEF:
                return b"\x00" * n
            return rd.buf[rd.base + addr:rd.base + addr + n]
        if cls == 2:  # chunked, unfiltered: gather the chunks of the v1 B-tree
            rank = b[2]
            btree = struct.unpack_from("<Q", b, 3)[0]
            cdims = struct.unpack_from("<%dI" % rank, b, 11)[:-1]
            out = np.zeros(self.shape, dtype=np.dtype("V%d" % self._esize))
            if btree == UNDEF:
                return out.tobytes()

            def walk(addr):
                a = rd.base + addr
                if rd.buf[a:a + 4] != b"TREE" or rd.u8(a + 4) != 1:
                    raise Hdf5FormatError("bad chunk B-tree node")
                level, used = rd.u8(a + 5), rd.u16(a + 6)
                ksz = 8 + 8 * rank
                p = a + 24
                for i in range(used):
                    csize = rd.u32(p)
                    offs = struct.unpack_from("<%dQ" % rank, rd.buf, p + 8)[:-1]
                    child = rd.u64(p + ksz)
                    if level:
                        walk(child)
                    else:
                        chunk = np.frombuffer(rd.buf, dtype=out.dtype, count=csize // self._esize, offset=rd.base + child)
                        chunk = chunk.reshape(cdims)
                        sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, self.shape))
                        out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
                    p += ksz + 8
            walk(btree)
            return out.tobytes()
        raise Hdf5FormatError("data layout class %d" % cls)

    def __array__(self, dtype=None, copy=None):
        a = self._rd.decode(self._dt, self.shape, self._raw())
        a = np.asarray(a)
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, key):
        a = np.asarray(self)
        return a[key] if key != () else (a if a.shape else a[()])

    @property
    def value(self):
        return self[()]

    def __len__(self):
        return self.shape[0]


class Group(_Object):
    def __init__(self, rd, addr, name, scratch: Optional[Tuple[int, int]] = None):
        super().__init__(rd, addr, name)
        st = scratch
        for t, _, body in self._msgs:
            if t == 0x0011:
                st = struct.unpack_from("<QQ", body, 0)
            elif t in (0x0002, 0x0006):
                raise Hdf5FormatError("new-style (link message) groups are not supported: %s" % name)
        if st is None:
            raise Hdf5FormatError("%s is not a group" % name)
        self._links = rd.group_links(st[0], st[1])

    def keys(self):
        return list(self._links.keys())

    def __iter__(self) -> Iterator[str]:
        return iter(self._links)

    def __len__(self):
        return len(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def _child(self, name):
        if name not in self._links:
            raise KeyError("%s/%s" % (self.name.rstrip("/"), name))
        addr = self._links[name]
        full = "%s/%s" % (self.name.rstrip("/"), name)
        types = {t for t, _, _ in self._rd.messages(addr)}
        return Group(self._rd, addr, full) if 0x0011 in types else Dataset(self._rd, addr, full)

    def __getitem__(self, path: str):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            node = node._child(part)
        return node

    def items(self):
        return [(k, self._child(k)) for k in self._links]

    def visit_datasets(self, prefix=""):
        for k, v in self.items():
            if isinstance(v, Group):
                yield from v.visit_datasets(prefix + k + "/")
            else:
                yield prefix + k, v


class File(Group):
    """Read-only view of an HDF5 file: ``File(path)["model_weights/dense_1/dense_1/kernel:0"][()]``."""

    def __init__(self, path: str, mode: str = "r"):
        if mode != "r":
            raise ValueError("hdf5_lite.File is read-only; use hdf5_lite.write_file to create a file")
        with open(path, "rb") as f:
            rd = _Reader(f.read())
        _, ohdr, cache, scratch = rd.root_entry
        st = struct.unpack_from("<QQ", scratch, 0) if cache == 1 else None
        super().__init__(rd, ohdr, "/", st)
        self.filename = path

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# =========================================================================================================
# writing
# =========================================================================================================
def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * (-len(b) % 8)


def _dtype_msg(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, max(dt.itemsize, 1))  # class 3 v1, null-terminated, ASCII
    if dt.kind == "f":
        if dt.itemsize == 4:
            return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        if dt.itemsize == 8:
            return struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    if dt.kind in "iu":
        return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    raise TypeError("cannot store dtype %s" % dt)


def _space_msg(shape) -> bytes:
    shape = tuple(int(s) for s in shape)
    return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(struct.pack("<Q", s) for s in shape)


def _as_storable(value) -> np.ndarray:
    if isinstance(value, str):
        value = value.encode("utf8")
    if isinstance(value, bytes):
        return np.array(value, dtype="S%d" % max(len(value), 1))
    a = np.asarray(value)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf8")
    if a.dtype.kind == "O":
        a = np.array([x.encode("utf8") if isinstance(x, str) else x for x in a.ravel()]).reshape(a.shape)
    if a.dtype.kind == "f" and a.dtype.itemsize not in (4, 8):
        a = a.astype(np.float32)
    if a.dtype.kind == "b":
        a = a.astype(np.uint8)
    a = a.astype(a.dtype.newbyteorder("<")) if a.dtype.kind in "iuf" else a
    return a.copy() if a.ndim == 0 else np.ascontiguousarray(a)  # (ascontiguousarray would turn a scalar into shape (1,))


def _attr_msg(name: str, value) -> bytes:
    a = _as_storable(value)
    nm = name.encode("utf8") + b"\x00"
    dtm, spm = _dtype_msg(a.dtype), _space_msg(a.shape)
    body = struct.pack("<BBHHH", 1, 0, len(nm), len(dtm), len(spm)) + _pad8(nm) + _pad8(dtm) + _pad8(spm) + a.tobytes()
    if len(body) > 0xFFF0:
        raise ValueError("attribute %s is too large for an object-header message (%d bytes)" % (name, len(body)))
    return body


class _Writer:
    """Append-only image builder; every structure is written once its children's addresses are known."""

    def __init__(self):
        self.buf = bytearray(b"\x00" * 96)  # superblock v0 (56 bytes + root symbol-table entry 40 bytes), patched last

    def alloc(self, data: bytes) -> int:
        self.buf += b"\x00" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def object_header(self, msgs: List[Tuple[int, bytes]]) -> int:
        body = b"".join(struct.pack("<HHB3x", t, len(_pad8(m)), 0) + _pad8(m) for t, m in msgs)
        return self.alloc(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    def dataset(self, arr, attrs) -> int:
        a = _as_storable(arr)
        data_addr = self.alloc(a.tobytes()) if a.nbytes else UNDEF
        msgs = [(0x0001, _space_msg(a.shape)), (0x0003, _dtype_msg(a.dtype)),
                (0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),  # fill value v2: allocate late, write never, undefined
                (0x0008, struct.pack("<BBQQ", 3, 1, data_addr, a.nbytes))]
        msgs += [(0x000C, _attr_msg(k, v)) for k, v in attrs.items()]
        return self.object_header(msgs)

    def group(self, children: "OrderedDict[str, int]", attrs) -> Tuple[int, int, int]:
        """-> (object header address, B-tree address, local heap address)."""
        names = sorted(children)  # symbol-table entries are ordered by name (spec III.B: the B-tree keys are name offsets)
        heap = bytearray(b"\x00" * 8)  # offset 0: the empty string (key of the left-most B-tree edge)
        offs = {}
        for nm in names:
            offs[nm] = len(heap)
            heap += _pad8(nm.encode("utf8") + b"\x00")
        free = len(heap)
        heap += struct.pack("<QQ", 1, 16 + (-(len(heap) + 16) % 8))  # one free block: next = 1 (none), its size
        heap += b"\x00" * (-len(heap) % 8)
        heap_data = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free, heap_data))
        # symbol-table nodes of at most 2K = 8 entries (group leaf node K = 4 in the superblock), one B-tree level
        snods, keys = [], [0]
        for i in range(0, max(len(names), 1), 8):
            part = names[i:i + 8]
            ent = b"".join(struct.pack("<QQII16x", offs[nm], children[nm], 0, 0) for nm in part)
            ent += b"\x00" * (40 * (8 - len(part)))
            snods.append(self.alloc(b"SNOD" + struct.pack("<BBH", 1, 0, len(part)) + ent))
            keys.append(offs[part[-1]] if part else 0)
        if len(snods) > 32:
            raise ValueError("more than 256 links in one group are not supported by this writer")
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF)
        for i, s in enumerate(snods):
            node += struct.pack("<QQ", keys[i], s)
        node += struct.pack("<Q", keys[-1])
        node += b"\x00" * ((2 * 16 + 1) * 16 + 24 - len(node))  # full-size node (internal node K = 16)
        btree = self.alloc(node)
        msgs = [(0x0011, struct.pack("<QQ", btree, heap_addr))] + [(0x000C, _attr_msg(k, v)) for k, v in attrs.items()]
        return self.object_header(msgs), btree, heap_addr

    def finish(self, root: Tuple[int, int, int]) -> bytes:
        ohdr, btree, heap = root
        eof = len(self.buf) + (-len(self.buf) % 8)
        self.buf += b"\x00" * (eof - len(self.buf))
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, ohdr, 1, 0) + struct.pack("<QQ", btree, heap)
        self.buf[0:len(sb)] = sb
        return bytes(self.buf)


class NodeSpec:
    """In-memory tree handed to ``write_file``: groups hold children (NodeSpec) and attrs; leaves hold an array."""

    def __init__(self, data=None):
        self.data = data
        self.children: "OrderedDict[str, NodeSpec]" = OrderedDict()
        self.attrs: "OrderedDict[str, object]" = OrderedDict()

    def require_group(self, path: str) -> "NodeSpec":
        node = self
        for part in [p for p in path.split("/") if p]:
            node = node.children.setdefault(part, NodeSpec())
        return node

    def create_dataset(self, path: str, data) -> "NodeSpec":
        parts = [p for p in path.split("/") if p]
        g = self.require_group("/".join(parts[:-1]))
        g.children[parts[-1]] = NodeSpec(np.asarray(data))
        return g.children[parts[-1]]


def write_file(path: str, root: NodeSpec):
    w = _Writer()

    def emit(node: NodeSpec):
        if node.data is not None:
            return w.dataset(node.data, node.attrs), None
        kids = OrderedDict((k, emit(v)[0]) for k, v in node.children.items())
        g = w.group(kids, node.attrs)
        return g[0], g
    _, g = emit(root)
    with open(path, "wb") as f:
        f.write(w.finish(g))
