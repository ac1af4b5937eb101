"""Test helper: write a TensorFlow tensor-bundle (V2) checkpoint (<prefix>.index + .data-00000-of-00001)
so the reader (chiron_amd/tf_bundle.py) can be round-tripped without TensorFlow."""
import struct

import numpy as np

MAGIC = 0xDB4775248B80FB57


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


_TABLE = None


def crc32c(data):
    global _TABLE
    if _TABLE is None:
        _TABLE = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            _TABLE.append(c)
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _block(entries, restart_interval=16):
    buf = bytearray()
    restarts = []
    prev = b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        buf += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def _entry_proto(dtype, shape, offset, size, crc):
    dims = b"".join(b"\x12" + _varint(len(d)) + d for d in [b"\x08" + _varint(s) for s in shape])
    out = b"\x08" + _varint(dtype) + b"\x12" + _varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _varint(offset)
    out += b"\x28" + _varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


def write_bundle(prefix, tensors, extra_int32=None):
    """tensors: {name: float32 ndarray}; extra_int32: {name: int} scalars (e.g. global_step)."""
    items = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in tensors.items()}
    for k, v in (extra_int32 or {}).items():
        items[k] = np.asarray(v, dtype=np.int32)
    entries = [(b"", b"\x08\x01\x12\x02\x08\x01")]      # BundleHeaderProto{num_shards=1, version{producer=1}}
    off = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(items):
            a = items[name]
            raw = a.tobytes()
            f.write(raw)
            dt = 1 if a.dtype == np.float32 else 3
            entries.append((name.encode(), _entry_proto(dt, a.shape, off, len(raw), _mask(crc32c(raw)))))
            off += len(raw)
    table = bytearray()
    handles = []
    for i in range(0, len(entries), 40):          # several data blocks, like a real index
        blk = _block(entries[i:i + 40])
        handles.append((entries[min(i + 39, len(entries) - 1)][0], len(table), len(blk)))
        table += blk + b"\x00" + struct.pack("<I", _mask(crc32c(blk + b"\x00")))
    meta = _block([])
    meta_h = (len(table), len(meta))
    table += meta + b"\x00" + struct.pack("<I", _mask(crc32c(meta + b"\x00")))
    idx = _block([(k, _varint(o) + _varint(s)) for k, o, s in handles], restart_interval=1)
    idx_h = (len(table), len(idx))
    table += idx + b"\x00" + struct.pack("<I", _mask(crc32c(idx + b"\x00")))
    foot = _varint(meta_h[0]) + _varint(meta_h[1]) + _varint(idx_h[0]) + _varint(idx_h[1])
    foot += b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC)
    table += foot
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(table))
