"""Minimal fast5 (HDF5) reader -- what chiron/utils/extract_sig_ref.py:149-193 takes from h5py,
without h5py: superblock v0/v1, v1 object headers, symbol-table groups (v1 B-tree + SNOD + local
heap), contiguous / compact / chunked datasets (v1 chunk B-tree) with the deflate and shuffle
filters, v1-v3 attribute messages, fixed-point / float / fixed-string / vlen-string datatypes
(global heap).  That covers the MinKNOW-era single- and multi-read fast5 files of the reference's
example data.  Anything else raises Fast5FormatError (the caller logs and skips the read, like the
reference does for unreadable files, extract_sig_ref.py:97-117).
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Fast5FormatError(Exception):
    pass


def _pad8(n):
    return (n + 7) & ~7


class _H5File(object):
    def __init__(self, path):
        with open(path, "rb") as f:
            self.d = f.read()
        d = self.d
        if d[:8] != b"\x89HDF\r\n\x1a\n":
            raise Fast5FormatError("not an HDF5 file")
        ver = d[8]
        if ver not in (0, 1):
            raise Fast5FormatError("HDF5 superblock version %d is not supported" % ver)
        if d[13] != 8 or d[14] != 8:
            raise Fast5FormatError("only 8-byte offsets/lengths are supported")
        p = 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", d, p)[0]
        # root group symbol table entry follows the four addresses
        ent = p + 32
        self.root = self.base + struct.unpack_from("<Q", d, ent + 8)[0]

    # ---- object headers -------------------------------------------------------------------------
    def messages(self, addr):
        """-> list of (type, flags, bytes) for a version-1 object header."""
        d = self.d
        if d[addr:addr + 4] == b"OHDR":
            raise Fast5FormatError("version-2 object headers are not supported")
        if d[addr] != 1:
            raise Fast5FormatError("object header version %d" % d[addr])
        nmsg, = struct.unpack_from("<H", d, addr + 2)
        size, = struct.unpack_from("<I", d, addr + 8)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", d, pos)
                body = d[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x10:          # continuation
                    off, ln = struct.unpack_from("<QQ", body, 0)
                    blocks.append((self.base + off, ln))
                out.append((mtype, mflags, body))
        return out

    # ---- groups -------------------------------------------------------------------------------
    def _heap_name(self, heap_addr, off):
        d = self.d
        if d[heap_addr:heap_addr + 4] != b"HEAP":
            raise Fast5FormatError("bad local heap")
        seg, = struct.unpack_from("<Q", d, heap_addr + 24)
        s = self.base + seg + off
        e = d.index(b"\x00", s)
        return d[s:e].decode("utf-8", "replace")

    def _walk_group_btree(self, node, heap, out):
        d = self.d
        if d[node:node + 4] == b"SNOD":
            n, = struct.unpack_from("<H", d, node + 6)
            for i in range(n):
                e = node + 8 + 40 * i
                name_off, obj = struct.unpack_from("<QQ", d, e)
                out[self._heap_name(heap, name_off)] = self.base + obj
            return
        if d[node:node + 4] != b"TREE" or d[node + 4] != 0:
            raise Fast5FormatError("bad group B-tree node")
        used, = struct.unpack_from("<H", d, node + 6)
        p = node + 24
        for i in range(used):
            child, = struct.unpack_from("<Q", d, p + 8)        # key(8) child(8) key(8) ...
            self._visit()
            self._walk_group_btree(self.base + child, heap, out)
            p += 16

    def _visit(self):
        """A B-tree of a damaged file may point back at itself: no real tree has more nodes than the file has 24-byte
        pieces (the same budget as csrc/fast5.cpp)."""
        self._budget -= 1
        if self._budget < 0:
            raise Fast5FormatError("B-tree has more nodes than the file can hold (a loop)")

    def links(self, addr):
        """children of the group whose object header is at addr: {name: object header address}.
        Old-style groups keep a symbol table (B-tree + local heap); new-style groups with compact
        storage keep one Link message per child in the object header."""
        out = {}
        for mtype, _, body in self.messages(addr):
            if mtype == 0x11:
                btree, heap = struct.unpack_from("<QQ", body, 0)
                self._budget = len(self.d) // 24 + 16          # per walk
                self._walk_group_btree(self.base + btree, self.base + heap, out)
            elif mtype == 0x02:            # Link Info: dense storage lives in a fractal heap
                flags = body[1]
                p = 2 + (8 if flags & 1 else 0)
                fheap, = struct.unpack_from("<Q", body, p)
                if fheap != UNDEF:
                    raise Fast5FormatError("dense (fractal heap) link storage is not supported")
            elif mtype == 0x06:            # Link message
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
                nlen = int.from_bytes(body[p:p + nsz], "little")
                p += nsz
                name = body[p:p + nlen].decode("utf-8", "replace")
                p += nlen
                if ltype == 0:
                    out[name] = self.base + struct.unpack_from("<Q", body, p)[0]
        return out

    def resolve(self, path, start=None):
        addr = self.root if start is None else start
        for part in [p for p in path.split("/") if p]:
            ch = self.links(addr)
            if part not in ch:
                raise KeyError(path)
            addr = ch[part]
        return addr

    # ---- datatypes / dataspaces -----------------------------------------------------------------
    def _dtype(self, body):
        cls = body[0] & 0x0F
        bits0 = body[1]
        size, = struct.unpack_from("<I", body, 4)
        if size == 0 or size > (1 << 20) or (cls == 0 and size not in (1, 2, 4, 8)) or (cls == 1 and size not in (4, 8)):
            raise Fast5FormatError("datatype size %d" % size)
        if cls == 0:
            if bits0 & 1:
                raise Fast5FormatError("big-endian integers are not supported")
            return ("int", np.dtype("<%s%d" % ("i" if bits0 & 8 else "u", size)), size)
        if cls == 1:
            return ("float", np.dtype("<f%d" % size), size)
        if cls == 3:
            return ("str", None, size)
        if cls == 9:
            if (bits0 & 0x0F) != 1:
                raise Fast5FormatError("only variable-length strings are supported")
            return ("vstr", None, size)
        raise Fast5FormatError("datatype class %d is not supported" % cls)

    @staticmethod
    def _dims(body):
        ver, rank, flags = body[0], body[1], body[2]
        off = 8 if ver == 1 else 4
        return list(struct.unpack_from("<%dQ" % rank, body, off)) if rank else []

    def _vlen(self, raw):
        ln, coll, idx = struct.unpack_from("<IQI", raw, 0)
        d = self.d
        c = self.base + coll
        if d[c:c + 4] != b"GCOL":
            raise Fast5FormatError("bad global heap collection")
        csize, = struct.unpack_from("<Q", d, c + 8)
        p, end = c + 16, c + csize
        while p + 16 <= end:
            oi, _, _, osz = struct.unpack_from("<HHIQ", d, p)
            if oi == 0:
                break
            if osz > end - (p + 16):
                raise Fast5FormatError("global heap object runs past its collection")
            if oi == idx:
                return d[p + 16:p + 16 + ln]
            p += 16 + _pad8(osz)
        raise Fast5FormatError("global heap object %d not found" % idx)

    def _decode(self, kind, dt, size, dims, raw):
        n = int(np.prod(dims)) if dims else 1
        if kind in ("int", "float"):
            a = np.frombuffer(raw, dtype=dt, count=n)
            return a.reshape(dims) if dims else a[0]
        if kind == "str":
            vals = [raw[i * size:(i + 1) * size].split(b"\x00")[0] for i in range(n)]
        else:
            vals = [self._vlen(raw[i * 16:(i + 1) * 16]) for i in range(n)]
        return vals if dims else vals[0]

    # ---- attributes -----------------------------------------------------------------------------
    def attrs(self, addr):
        out = {}
        for mtype, _, body in self.messages(addr):
            if mtype != 0x0C:
                continue
            ver = body[0]
            if ver == 1:
                nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
                p = 8
                name = body[p:p + nsz].split(b"\x00")[0].decode()
                p += _pad8(nsz)
                tb = body[p:p + tsz]
                p += _pad8(tsz)
                sb = body[p:p + ssz]
                p += _pad8(ssz)
            elif ver in (2, 3):
                nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
                p = 8 + (1 if ver == 3 else 0)
                name = body[p:p + nsz].split(b"\x00")[0].decode()
                p += nsz
                tb = body[p:p + tsz]
                p += tsz
                sb = body[p:p + ssz]
                p += ssz
            else:
                raise Fast5FormatError("attribute message version %d" % ver)
            kind, dt, size = self._dtype(tb)
            out[name] = self._decode(kind, dt, size, self._dims(sb), body[p:])
        return out

    # ---- datasets -------------------------------------------------------------------------------
    def _chunks(self, node, ndim, out):
        d = self.d
        if d[node:node + 4] != b"TREE" or d[node + 4] != 1:
            raise Fast5FormatError("bad chunk B-tree node")
        level = d[node + 5]
        used, = struct.unpack_from("<H", d, node + 6)
        ksz = 8 + 8 * ndim
        p = node + 24
        for _ in range(used):
            csize, fmask = struct.unpack_from("<II", d, p)
            offs = struct.unpack_from("<%dQ" % ndim, d, p + 8)
            child, = struct.unpack_from("<Q", d, p + ksz)
            self._visit()
            if level == 0:
                out.append((offs, csize, fmask, self.base + child))
            else:
                self._chunks(self.base + child, ndim, out)
            p += ksz + 8

    def dataset(self, addr):
        layout = dt = dims = None
        filters = []
        for mtype, _, body in self.messages(addr):
            if mtype == 0x01:
                dims = self._dims(body)
            elif mtype == 0x03:
                dt = self._dtype(body)
            elif mtype == 0x08:
                layout = body
            elif mtype == 0x0B:
                ver, nf = body[0], body[1]
                p = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid, nlen, _, ncd = struct.unpack_from("<HHHH", body, p) if (ver == 1 or struct.unpack_from("<H", body, p)[0] >= 256) \
                        else (struct.unpack_from("<H", body, p)[0], 0) + struct.unpack_from("<HH", body, p + 2)
                    if ver == 1:
                        p += 8 + _pad8(nlen) + 4 * ncd + (4 if ncd % 2 else 0)
                    else:
                        p += (8 + nlen if fid >= 256 else 6) + 4 * ncd
                    filters.append(fid)
        if layout is None or dt is None or dims is None:
            raise Fast5FormatError("not a dataset")
        kind, npdt, esize = dt
        if layout[0] != 3:
            raise Fast5FormatError("data layout message version %d" % layout[0])
        cls = layout[1]
        count = 1
        for v in dims:
            count *= int(v)                  # Python integers: no wrap, but a damaged dimension must not reach bytearray()
        total = count * esize
        if count > (1 << 40) or total > (1 << 34):
            raise Fast5FormatError("dataset too large")
        if cls == 0:
            sz, = struct.unpack_from("<H", layout, 2)
            raw = layout[4:4 + sz]
        elif cls == 1:
            a, sz = struct.unpack_from("<QQ", layout, 2)
            raw = b"" if a == UNDEF else self.d[self.base + a:self.base + a + sz]
        elif cls == 2:
            ndim = layout[2]
            btree, = struct.unpack_from("<Q", layout, 3)
            cdims = struct.unpack_from("<%dI" % ndim, layout, 11)
            if len(dims) != 1:
                raise Fast5FormatError("only 1-D chunked datasets are supported")
            if any(f not in (1, 2) for f in filters):
                raise Fast5FormatError("unsupported filter %s" % filters)
            buf = bytearray(total)
            chunks = []
            self._budget = len(self.d) // 24 + 16
            if btree != UNDEF:
                self._chunks(self.base + btree, ndim, chunks)
            cbytes = cdims[0] * esize
            for offs, csize, fmask, caddr in chunks:
                blob = self.d[caddr:caddr + csize]
                for k, fid in reversed(list(enumerate(filters))):
                    if fmask & (1 << k):
                        continue
                    if fid == 1:
                        blob = zlib.decompress(blob)
                    elif fid == 2:                       # byte shuffle
                        n = len(blob) // esize
                        blob = np.frombuffer(blob, np.uint8).reshape(esize, n).T.tobytes()
                s = offs[0] * esize
                e = min(s + cbytes, total)
                buf[s:e] = blob[:e - s]
            raw = bytes(buf)
        else:
            raise Fast5FormatError("layout class %d" % cls)
        return self._decode(kind, npdt, esize, dims, raw[:total] if kind in ("int", "float") else raw)


def _text(v):
    if isinstance(v, (list, tuple)):
        v = v[0] if v else b""
    if isinstance(v, bytes):
        return v.decode("utf-8", "replace")
    return str(v)


def _read_record(h5, raw_group, analyses_root, suffix):
    rec = {"suffix": suffix, "signal": np.asarray(h5.dataset(h5.resolve("Signal", raw_group)))}
    rec["read_id"] = _text(h5.attrs(raw_group).get("read_id", b""))
    rec["fastq"] = ""
    for p in ("Analyses/Basecall_1D_000/BaseCalled_template/Fastq", "Analyses/Alignment_000/Aligned_template/Fasta"):
        try:
            rec["fastq"] = _text(h5.dataset(h5.resolve(p, analyses_root)))
            break
        except (KeyError, Fast5FormatError):
            continue
    return rec


def read_fast5(path):
    """-> list of records {suffix, signal (int16 ndarray), read_id, fastq, channel}.
    Single-read files (extract_file, extract_sig_ref.py:149-175): first group under /Raw/Reads.
    Multi-read files (extract_file_v2, :178-193): one record per top-level read group."""
    h5 = _H5File(path)
    top = h5.links(h5.root)
    out = []
    if "Raw" in top:
        reads = h5.links(h5.resolve("Raw/Reads"))
        if not reads:
            raise Fast5FormatError("no read under /Raw/Reads")
        first = sorted(reads)[0]
        rec = _read_record(h5, reads[first], h5.root, "")
        try:
            ch = h5.attrs(h5.resolve("UniqueGlobalKey/channel_id"))
            rec["channel"] = {k: (_text(v) if isinstance(v, (bytes, list)) else float(v)) for k, v in ch.items()}
        except (KeyError, Fast5FormatError):
            rec["channel"] = None
        out.append(rec)
    else:
        for name in sorted(top):
            sub = h5.links(top[name])
            if "Raw" not in sub:
                continue
            out.append(_read_record(h5, sub["Raw"], top[name], name))
    return out


def read_fast5_native(path, reverse=False):
    """The same records through the native reader of libchiron_amd.so (csrc/fast5.cpp: file image, B-tree walk and zlib
    inflate in C++, no GIL): -> list of {suffix, signal (float32 ndarray, reversed when `reverse`), read_id, fastq}.
    This is the reader of the hot path (`chiron call` on fast5 input); the Python classes above remain for
    what the hot path does not need (channel attributes for the pA conversion, arbitrary datasets) and as its cross-check."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    h = C.c_void_p()
    st = lib.chiron_fast5_open(path.encode() if isinstance(path, str) else path, C.byref(h))
    if st != _lib.OK:
        raise Fast5FormatError(lib.chiron_last_error().decode("utf-8", "replace"))
    try:
        out = []
        suffix, rid = C.create_string_buffer(256), C.create_string_buffer(256)
        for i in range(lib.chiron_fast5_read_count(h)):
            n, fq = C.c_int64(), C.c_int64()
            if lib.chiron_fast5_read_info(h, i, suffix, 256, rid, 256, C.byref(n), C.byref(fq)) != _lib.OK:
                raise Fast5FormatError(lib.chiron_last_error().decode("utf-8", "replace"))
            sig = np.empty(n.value, dtype=np.float32)
            if lib.chiron_fast5_signal(h, i, sig.ctypes.data_as(C.c_void_p), n.value, 1 if reverse else 0) != _lib.OK:
                raise Fast5FormatError(lib.chiron_last_error().decode("utf-8", "replace"))
            fastq = ""
            if fq.value > 0:
                buf = C.create_string_buffer(fq.value + 1)
                if lib.chiron_fast5_fastq(h, i, buf, fq.value + 1) == _lib.OK:
                    fastq = buf.value.decode("utf-8", "replace")
            out.append({"suffix": suffix.value.decode("utf-8", "replace"), "signal": sig,
                        "read_id": rid.value.decode("utf-8", "replace"), "fastq": fastq})
        return out
    finally:
        lib.chiron_fast5_close(h)


def write_signal_text(path, signal, delimiter="\n"):
    """extract_sig_ref.py:122-123 for integer DAC counts: delimiter.join(str(v) for v in signal), natively."""
    import ctypes as C
    from . import _lib
    sig = np.ascontiguousarray(signal, dtype=np.float32)
    _lib.check(_lib.load().chiron_write_signal_text(path.encode(), sig.ctypes.data_as(C.c_void_p), sig.shape[0], delimiter.encode()))


def read_raw_signal(path):
    recs = read_fast5(path)
    if not recs:
        raise Fast5FormatError("no raw signal in %s" % path)
    return recs[0]["signal"]
