"""Nested-dict <-> HDF5 without silx / h5py: the ``mcmc.h5`` writer and the ``observables.h5`` reader.

The reference stores its results with ``silx.io.dictdump.dicttoh5`` and reads them back with ``h5todict``
(ref: data_IO.py:217-257; written at mcmc.py:111-125, consumed by plot_mcmc.py:44-58 and by every reader of
``observables.h5``).  Neither silx nor h5py is installed on the GPU image, so this module

* writes a nested dict of numpy arrays / scalars / strings / ``None`` as a classic HDF5 file -- superblock
  version 0, version-1 object headers, symbol-table groups, contiguous datasets: the layout h5py itself produces
  with its default ``libver='earliest'``, readable by any HDF5 library -- with silx's conventions: a dataset per
  array at the place of its key, a group per nested dict, an EMPTY group for ``None`` or an empty dict;
* reads such files back, including the ones h5py / silx wrote (old-style groups whose symbol tables span several
  B-tree leaves, contiguous or compact datasets of IEEE floats, two's-complement integers and fixed-length
  strings).  Chunked / filtered datasets and new-style (``libver='latest'``) groups are not supported and raise.

``h5py`` is used instead whenever it is importable.  ``install_silx_shim()`` registers ``silx.io.dictdump`` with
these two functions when silx is absent, so that the reference's untouched ``data_IO`` imports and works.
File format: "HDF5 File Format Specification Version 1.1" (superblock 0, sections III.A-G, IV.A).
"""
from __future__ import annotations

import os
import struct
import sys
import types

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF
_INTERNAL_K = 16


def _pad8(n):
    return (n + 7) // 8 * 8


# ----------------------------------------------------------------------------------------------------------------
# writer
# ----------------------------------------------------------------------------------------------------------------
def _as_array(value):
    if isinstance(value, str):
        value = value.encode("utf-8")
    if isinstance(value, (bytes, np.bytes_)):
        return np.array(value, dtype=f"S{max(len(value), 1)}")
    arr = np.asarray(value)
    if arr.dtype.kind == "U":
        arr = np.char.encode(arr, "utf-8")
    if arr.dtype.kind == "b":
        arr = arr.astype(np.int8)
    if arr.dtype.kind == "O":
        raise TypeError("object arrays cannot be written to HDF5 by this writer")
    if arr.dtype.kind not in "fiuS":
        raise TypeError(f"unsupported dtype {arr.dtype}")
    arr = arr.astype(arr.dtype.newbyteorder("<"), copy=False)
    return arr if arr.flags.c_contiguous else arr.copy(order="C")       # (ascontiguousarray would turn 0-d into 1-d)


def _datatype_message(dt, utf8=False):
    if dt.kind == "f":
        size = dt.itemsize
        ebits, mbits, bias = {2: (5, 10, 15), 4: (8, 23, 127), 8: (11, 52, 1023)}[size]
        return (bytes([0x11, 0x20, size * 8 - 1, 0]) + struct.pack("<I", size) +
                struct.pack("<HHBBBBI", 0, size * 8, mbits, ebits, 0, mbits, bias))
    if dt.kind in "iu":
        return (bytes([0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0]) + struct.pack("<I", dt.itemsize) +
                struct.pack("<HH", 0, dt.itemsize * 8))
    if dt.kind == "S":
        # null padded; character set ASCII, or UTF-8 (bits 4-7 = 1) when the data hold non-ASCII bytes
        return bytes([0x13, 0x11 if utf8 else 0x01, 0, 0]) + struct.pack("<I", dt.itemsize)
    raise TypeError(dt)


def _message(mtype, data, flags=0):
    body = data + b"\0" * (_pad8(len(data)) - len(data))
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header(messages):
    body = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body


class _Writer:
    def __init__(self):
        self.chunks = []           # (address, bytes)
        self.pos = 96              # after the superblock
        self.max_entries = 1

    def alloc(self, nbytes):
        addr = self.pos
        self.pos = _pad8(self.pos + nbytes)
        return addr

    def put(self, addr, data):
        self.chunks.append((addr, data))

    def dataset(self, arr):
        rank = arr.ndim
        space = struct.pack("<BBB5x", 1, rank, 0) + b"".join(struct.pack("<Q", n) for n in arr.shape)
        # the array's own memory is written to the file later: no copy of a (possibly 500 MB) chain
        raw = memoryview(arr if arr.flags.c_contiguous else np.ascontiguousarray(arr)).cast("B") if arr.size else b""
        utf8 = arr.dtype.kind == "S" and arr.size > 0 and bool((np.frombuffer(arr.tobytes(), dtype=np.uint8) >= 0x80).any())
        msgs = [_message(0x0001, space), _message(0x0003, _datatype_message(arr.dtype, utf8), flags=1),
                _message(0x0005, bytes([2, 2, 2, 1]) + struct.pack("<I", 0), flags=1)]
        hdr_len = 16 + sum(len(m) for m in msgs) + 8 + 24
        addr = self.alloc(hdr_len)
        data_addr = self.alloc(len(raw)) if raw else _UNDEF
        msgs.append(_message(0x0008, struct.pack("<BBQQ", 3, 1, data_addr, len(raw))))
        self.put(addr, _object_header(msgs))
        if raw:
            self.put(data_addr, raw)
        return addr

    def group(self, tree, leaf_k):
        """Writes the group of ``tree`` (a dict); returns (object header address, B-tree address, heap address)."""
        names = sorted(tree, key=lambda s: s.encode("utf-8"))
        children = []
        for name in names:
            value = tree[name]
            if value is None or (isinstance(value, dict) and not value):
                children.append((name,) + self.group({}, leaf_k))
            elif isinstance(value, dict):
                children.append((name,) + self.group(value, leaf_k))
            else:
                children.append((name, self.dataset(_as_array(value)), None, None))
        # local heap: "" at offset 0, the names, one free block at the end
        seg = bytearray(8)
        offsets = []
        for name in names:
            offsets.append(len(seg))
            enc = name.encode("utf-8") + b"\0"
            seg += enc + b"\0" * (_pad8(len(enc)) - len(enc))
        free_at = len(seg)
        seg += struct.pack("<QQ", 1, 16)
        heap_addr = self.alloc(32)
        seg_addr = self.alloc(len(seg))
        self.put(heap_addr, b"HEAP" + struct.pack("<B3xQQQ", 0, len(seg), free_at, seg_addr))
        self.put(seg_addr, bytes(seg))
        # one symbol node (the file's leaf K is chosen so that every group fits) under a one-entry B-tree
        btree_addr = self.alloc(24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8)
        node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, _UNDEF, _UNDEF))
        if names:
            snod_addr = self.alloc(8 + 2 * leaf_k * 40)
            snod = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(names)))
            for (name, ohdr, bt, hp), off in zip(children, offsets):
                if bt is None:
                    snod += struct.pack("<QQII16x", off, ohdr, 0, 0)
                else:
                    snod += struct.pack("<QQIIQQ", off, ohdr, 1, 0, bt, hp)
            snod += b"\0" * (8 + 2 * leaf_k * 40 - len(snod))
            self.put(snod_addr, bytes(snod))
            node += struct.pack("<QQQ", 0, snod_addr, offsets[-1])
        node += b"\0" * (24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8 - len(node))
        self.put(btree_addr, bytes(node))
        ohdr_addr = self.alloc(16 + 24)
        self.put(ohdr_addr, _object_header([_message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))]))
        return ohdr_addr, btree_addr, heap_addr


def _max_group_size(tree):
    n = len(tree)
    for v in tree.values():
        if isinstance(v, dict):
            n = max(n, _max_group_size(v))
    return n


def _check_keys(tree, where=""):
    for k, v in tree.items():
        if not isinstance(k, str) or not k or "/" in k:
            raise ValueError(f"HDF5 member names must be non-empty strings without '/': {where}/{k!r}")
        if isinstance(v, dict):
            _check_keys(v, f"{where}/{k}")


def _write_native(tree, path):
    _check_keys(tree)
    leaf_k = max(4, (_max_group_size(tree) + 1) // 2)
    if leaf_k > 0x7FFF:
        raise ValueError("too many members in one group for this writer")
    w = _Writer()
    root, btree, heap = w.group(tree, leaf_k)
    eof = w.pos
    sb = (_SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", leaf_k, _INTERNAL_K, 0) +
          struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF) + struct.pack("<QQIIQQ", 0, root, 1, 0, btree, heap))
    assert len(sb) == 96
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "wb") as f:
        f.write(sb)
        for addr, data in sorted(w.chunks, key=lambda c: c[0]):     # alignment gaps stay holes, i.e. zeros
            f.seek(addr)
            f.write(data)
        f.truncate(eof)
    os.replace(tmp, path)


# ----------------------------------------------------------------------------------------------------------------
# reader
# ----------------------------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != _SIG:
            raise ValueError("not an HDF5 file (no signature at offset 0; user blocks are not supported)")
        if buf[8] != 0:
            raise NotImplementedError(f"HDF5 superblock version {buf[8]} (libver='latest' files) is not supported "
                                      "by the built-in reader; install h5py")
        if buf[13] != 8 or buf[14] != 8:
            raise NotImplementedError("only 8-byte offsets and lengths are supported")
        self.base = struct.unpack_from("<Q", buf, 24)[0]
        self.root = struct.unpack_from("<Q", buf, 56 + 8)[0]

    def u(self, fmt, off):
        return struct.unpack_from(fmt, self.b, off)

    def messages(self, addr):
        """[(type, flags, bytes)] of a version-1 object header, following continuation blocks."""
        addr += self.base
        version, nmsg, _ref, size = self.u("<BxHII", addr)
        if version != 1:
            raise NotImplementedError(f"object header version {version} (new-style file) is not supported")
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self.u("<HHB", pos)
                data = bytes(self.b[pos + 8:pos + 8 + msize])
                pos += 8 + msize
                if mtype == 0x0010:
                    caddr, clen = struct.unpack("<QQ", data[:16])
                    blocks.append((caddr + self.base, clen))
                out.append((mtype, flags, data))
        return out

    def heap_name(self, heap_addr, off):
        seg = self.u("<Q", heap_addr + self.base + 24)[0] + self.base
        end = self.b.find(b"\0", seg + off)
        if end < 0:
            raise ValueError("corrupt local heap (unterminated name)")
        return bytes(self.b[seg + off:end]).decode("utf-8")

    def symbol_nodes(self, btree_addr):
        a = btree_addr + self.base
        if bytes(self.b[a:a + 4]) != b"TREE":
            raise ValueError("corrupt group B-tree")
        ntype, level, used = self.u("<BBH", a + 4)
        if ntype != 0:
            raise ValueError("not a group B-tree")
        for i in range(used):
            child = self.u("<Q", a + 24 + 8 + 16 * i)[0]
            if level > 0:
                yield from self.symbol_nodes(child)
            else:
                yield child

    def group(self, btree_addr, heap_addr):
        out = {}
        for snod in self.symbol_nodes(btree_addr):
            a = snod + self.base
            if bytes(self.b[a:a + 4]) != b"SNOD":
                raise ValueError("corrupt symbol table node")
            n = self.u("<H", a + 6)[0]
            for i in range(n):
                off, ohdr, _cache = self.u("<QQI", a + 8 + 40 * i)
                out[self.heap_name(heap_addr, off)] = self.obj(ohdr)
        return out

    def obj(self, ohdr_addr):
        msgs = self.messages(ohdr_addr)
        for mtype, _f, data in msgs:
            if mtype == 0x0011:
                bt, hp = struct.unpack("<QQ", data[:16])
                return self.group(bt, hp)
            if mtype in (0x0002, 0x0006):
                raise NotImplementedError("new-style groups (link messages) are not supported; install h5py")
        return self.dataset(msgs)

    @staticmethod
    def _dtype(data):
        cls, ver = data[0] & 0x0F, data[0] >> 4
        b0 = data[1]
        size = struct.unpack("<I", data[4:8])[0]
        order = ">" if (b0 & 1) else "<"
        if cls == 0:
            return np.dtype(f"{order}{'i' if b0 & 0x08 else 'u'}{size}")
        if cls == 1:
            if size not in (2, 4, 8):
                raise NotImplementedError(f"{size}-byte floats")
            return np.dtype(f"{order}f{size}")
        if cls == 3:
            return np.dtype(f"S{size}")
        raise NotImplementedError(f"HDF5 datatype class {cls} (version {ver}) is not supported by the built-in reader")

    def dataset(self, msgs):
        shape = dtype = None
        layout = None
        for mtype, _f, data in msgs:
            if mtype == 0x0001:
                ver, rank = data[0], data[1]
                if ver == 1:
                    shape = struct.unpack_from(f"<{rank}Q", data, 8) if rank else ()
                elif ver == 2:
                    shape = struct.unpack_from(f"<{rank}Q", data, 4) if rank else ()
                    if data[3] == 2:
                        shape = None                                  # null dataspace
                else:
                    raise NotImplementedError(f"dataspace message version {ver}")
            elif mtype == 0x0003:
                dtype = self._dtype(data)
            elif mtype == 0x0008:
                layout = data
            elif mtype == 0x000B:
                raise NotImplementedError("filtered (compressed) datasets are not supported; install h5py")
        if dtype is None or layout is None:
            raise ValueError("object is neither a group nor a dataset")
        if shape is None:
            return np.empty(0, dtype=dtype)
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        nbytes = count * dtype.itemsize
        if layout[0] == 3:
            cls = layout[1]
            if cls == 1:
                addr, _size = struct.unpack_from("<QQ", layout, 2)
                if addr == _UNDEF:
                    src, off = b"\0" * nbytes, 0
                else:
                    src, off = self.b, addr + self.base
            elif cls == 0:
                csize = struct.unpack_from("<H", layout, 2)[0]
                src, off = layout[4:4 + csize][:nbytes], 0
            else:
                raise NotImplementedError("chunked datasets are not supported by the built-in reader; install h5py")
        elif layout[0] in (1, 2):
            rank, cls = layout[1], layout[2]
            if cls != 1:
                raise NotImplementedError("only contiguous datasets are supported for layout versions 1-2")
            addr = struct.unpack_from("<Q", layout, 8)[0]
            src, off = self.b, addr + self.base
        else:
            raise NotImplementedError(f"data layout message version {layout[0]}")
        if off + nbytes > len(src):
            raise ValueError("dataset extends past the end of the file")
        # a view on the mapped file, then ONE copy into memory the caller owns (byte order made native on the way)
        arr = np.frombuffer(src, dtype=dtype, count=count, offset=off).reshape(shape)
        arr = arr.astype(dtype.newbyteorder("="), copy=True)
        if dtype.kind == "S" and arr.shape == ():
            return arr[()].decode("utf-8", "replace")
        return arr[()] if arr.shape == () else arr

    def read(self):
        return self.obj(self.root)


def _read_native(path):
    """The file is mapped, not read: a 574 MB mcmc.h5 then costs its datasets once, not three times."""
    import mmap
    with open(path, "rb") as f:
        if os.fstat(f.fileno()).st_size == 0:
            raise ValueError("not an HDF5 file (empty)")
        with mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as buf:
            return _Reader(buf).read()


# ----------------------------------------------------------------------------------------------------------------
# public
# ----------------------------------------------------------------------------------------------------------------
def _h5py():
    if os.environ.get("GPEMU_NO_H5PY"):
        return None
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def _h5py_write(h5py, tree, group):
    for key, value in tree.items():
        if value is None or (isinstance(value, dict) and not value):
            group.create_group(key)
        elif isinstance(value, dict):
            _h5py_write(h5py, value, group.create_group(key))
        else:
            group.create_dataset(key, data=_as_array(value))


def _h5py_read(h5py, group):
    out = {}
    for key, item in group.items():
        if isinstance(item, h5py.Group):
            out[key] = _h5py_read(h5py, item)
        else:
            v = item[()]
            out[key] = v.decode("utf-8", "replace") if isinstance(v, bytes) else v
    return out


def dicttoh5(treedict, h5file, h5path="/", mode="w", **_ignored):
    """``silx.io.dictdump.dicttoh5`` for the cases the reference needs (ref: data_IO.py:232): the whole file is
    (over)written from its root.  silx's other modes -- appending to a file, writing below a sub-path -- are refused
    rather than silently turned into an overwrite."""
    if h5path not in ("/", "", None):
        raise NotImplementedError(f"dicttoh5(h5path={h5path!r}): only the file's root is supported by gpemu.h5io")
    if mode not in ("w", "w-", "x"):
        raise NotImplementedError(f"dicttoh5(mode={mode!r}): only (over)writing a whole file is supported by gpemu.h5io")
    if mode in ("w-", "x") and os.path.exists(os.fspath(h5file)):
        raise FileExistsError(os.fspath(h5file))
    h5py = _h5py()
    if h5py is not None:
        with h5py.File(h5file, "w") as f:
            _h5py_write(h5py, treedict, f)
        return
    _write_native(treedict, os.fspath(h5file))


def h5todict(h5file, path="/", **_ignored):
    """``silx.io.dictdump.h5todict``: nested dict of arrays; an empty group comes back as ``{}``."""
    h5py = _h5py()
    if h5py is not None:
        with h5py.File(h5file, "r") as f:
            tree = _h5py_read(h5py, f)
    else:
        tree = _read_native(os.fspath(h5file))
    for part in [p for p in path.split("/") if p]:
        tree = tree[part]
    return tree


def write_dict_to_h5(results, output_dir, filename, verbose=True):
    """Same call as the reference's ``data_IO.write_dict_to_h5`` (ref: data_IO.py:217-236)."""
    os.makedirs(output_dir, exist_ok=True)
    dicttoh5(results, os.path.join(output_dir, filename))


def read_dict_from_h5(input_dir, filename, verbose=True):
    """Same call as the reference's ``data_IO.read_dict_from_h5`` (ref: data_IO.py:239-257)."""
    return h5todict(os.path.join(input_dir, filename))


def install_silx_shim():
    """Make ``from silx.io.dictdump import dicttoh5, h5todict`` (ref: data_IO.py:32) work without silx: the two
    names are served by this module.  Does nothing when silx is importable."""
    if "silx.io.dictdump" in sys.modules:
        return False
    try:
        import silx.io.dictdump  # noqa: F401
        return False
    except ImportError:
        pass
    silx = sys.modules.setdefault("silx", types.ModuleType("silx"))
    sio = sys.modules.setdefault("silx.io", types.ModuleType("silx.io"))
    dd = types.ModuleType("silx.io.dictdump")
    dd.dicttoh5, dd.h5todict = dicttoh5, h5todict
    dd.__doc__ = "gpemu.h5io stand-in for silx.io.dictdump (silx is not installed)"
    sys.modules["silx.io.dictdump"] = dd
    silx.io = sio
    sio.dictdump = dd
    return True
