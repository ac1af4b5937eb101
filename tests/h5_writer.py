"""Test-only writer of small HDF5 files (fast5 layouts) so that chiron_amd/fast5.py can be exercised on multi-read
files, which the reference tree ships no example of (extract_sig_ref.py:178-193 extract_file_v2).  Independent of the
reader: it emits what the HDF5 File Format Specification 1.x describes -- superblock v0, version-1 object headers,
new-style groups as Link messages, int16 datasets (contiguous, or chunked with the deflate filter behind a v1 chunk
B-tree), scalar fixed-string datasets, and version-1 attribute messages with fixed-length strings."""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b):
    return b + b"\x00" * ((-len(b)) % 8)


def _msg(mtype, body):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _header(msgs):
    data = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(data)) + data          # 16 bytes, then the messages


def _dt_int16():
    return struct.pack("<BBBBI", 0x10, 0x08, 0, 0, 2) + struct.pack("<HH", 0, 16)    # class 0 v1, signed LE, size 2


def _dt_str(n):
    return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, n)                                # class 3 v1, null terminated ASCII


def _space(dims):
    return struct.pack("<BBB5x", 1, len(dims), 0) + b"".join(struct.pack("<Q", d) for d in dims)


class H5Builder(object):
    def __init__(self):
        self.blob = bytearray(b"\x00" * 96)      # superblock (56 bytes + root symbol table entry 40 bytes)

    def _put(self, data):
        while len(self.blob) % 8:
            self.blob += b"\x00"
        addr = len(self.blob)
        self.blob += data
        return addr

    def attr(self, name, value):
        v = value.encode() + b"\x00"
        nm = name.encode() + b"\x00"
        dt, sp = _dt_str(len(v)), _space([])
        body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + v
        return _msg(0x0C, body)

    def dataset_int16(self, values, chunk=None, attrs=()):
        a = np.asarray(values, dtype="<i2")
        msgs = [_msg(0x01, _space([a.shape[0]])), _msg(0x03, _dt_int16())]
        if chunk is None:
            addr = self._put(a.tobytes())
            msgs.append(_msg(0x08, struct.pack("<BBQQ", 3, 1, addr, a.nbytes)))
        else:
            keys = []
            for off in range(0, a.shape[0], chunk):
                piece = a[off:off + chunk]
                raw = piece.tobytes() + b"\x00" * (2 * (chunk - piece.shape[0]))     # chunks are stored whole
                comp = zlib.compress(raw, 4)
                keys.append((len(comp), off, self._put(comp)))
            node = b"TREE" + struct.pack("<BBHQQ", 1, 0, len(keys), UNDEF, UNDEF)
            for size, off, addr in keys:
                node += struct.pack("<IIQQ", size, 0, off, 0) + struct.pack("<Q", addr)
            node += struct.pack("<IIQQ", 0, 0, a.shape[0], 0)                          # final key
            bt = self._put(node)
            msgs.append(_msg(0x0B, struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<I4x", 4)))
            msgs.append(_msg(0x08, struct.pack("<BBBQ", 3, 2, 2, bt) + struct.pack("<II", chunk, 2)))
        return self._put(_header(msgs + list(attrs)))

    def dataset_str(self, text):
        v = text.encode() + b"\x00"
        addr = self._put(v)
        msgs = [_msg(0x01, _space([])), _msg(0x03, _dt_str(len(v))), _msg(0x08, struct.pack("<BBQQ", 3, 1, addr, len(v)))]
        return self._put(_header(msgs))

    def group(self, children, attrs=()):
        """children: {name: object header address}"""
        msgs = []
        for name, addr in children.items():
            nm = name.encode()
            msgs.append(_msg(0x06, struct.pack("<BBB", 1, 0, len(nm)) + nm + struct.pack("<Q", addr)))
        return self._put(_header(msgs + list(attrs)))

    def finish(self, root_addr, path):
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.blob), UNDEF)
        sb += struct.pack("<QQI4x16x", 0, root_addr, 0)                               # root symbol table entry
        assert len(sb) == 96
        self.blob[:96] = sb
        with open(path, "wb") as f:
            f.write(bytes(self.blob))


def write_multi_read_fast5(path, reads, chunk=None):
    """reads: list of (read group name, read_id, int16 signal, fastq text or None)"""
    b = H5Builder()
    top = {}
    for name, read_id, signal, fastq in reads:
        sig = b.dataset_int16(signal, chunk=chunk)
        raw = b.group({"Signal": sig}, attrs=[b.attr("read_id", read_id)])
        kids = {"Raw": raw}
        if fastq is not None:
            fq = b.dataset_str(fastq)
            kids["Analyses"] = b.group({"Basecall_1D_000": b.group({"BaseCalled_template": b.group({"Fastq": fq})})})
        top[name] = b.group(kids)
    b.finish(b.group(top), path)
