"""TensorFlow checkpoint "tensor bundle" (V2) reader -- replaces tf.train.Saver.restore /
tf.train.latest_checkpoint as used by the reference (chiron_eval.py:272-276).

`<prefix>.index` is a LevelDB-style immutable table: 48-byte footer
(metaindex handle, index handle, magic), blocks of prefix-compressed entries with a restart array;
the value of key "" is a BundleHeaderProto, every other key is a variable name whose value is a
BundleEntryProto {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: crc32c}.
Tensor bytes live at [offset, offset+size) of `<prefix>.data-00000-of-00001`, little endian, row-major.
"""
import os
import struct

import numpy as np

MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_NP = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}


def latest_checkpoint(model_dir):
    """tf.train.latest_checkpoint (chiron_eval.py:276): parse the text proto
    `checkpoint` file, return the prefix path or None."""
    path = os.path.join(model_dir, "checkpoint")
    if not os.path.exists(path):
        return None
    for line in open(path):
        line = line.strip()
        if line.startswith("model_checkpoint_path:"):
            name = line.split(":", 1)[1].strip().strip('"')
            return name if os.path.isabs(name) else os.path.join(model_dir, name)
    return None


def _varint(buf, p):
    v = 0
    shift = 0
    while True:
        b = buf[p]
        p += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, p
        shift += 7


def _block_entries(buf, off, size):
    """entries of one table block (handle = offset/size, trailer excluded)."""
    blk = buf[off:off + size]
    if buf[off + size] != 0:
        raise ValueError("compressed table blocks are not supported (type %d)" % buf[off + size])
    nrestart, = struct.unpack_from("<I", blk, size - 4)
    end = size - 4 - 4 * nrestart
    p = 0
    key = b""
    out = []
    while p < end:
        shared, p = _varint(blk, p)
        non_shared, p = _varint(blk, p)
        vlen, p = _varint(blk, p)
        key = key[:shared] + blk[p:p + non_shared]
        p += non_shared
        out.append((key, blk[p:p + vlen]))
        p += vlen
    return out


def _proto_fields(buf):
    """minimal protobuf wire decoder -> list of (field, wiretype, value)."""
    p = 0
    out = []
    while p < len(buf):
        tag, p = _varint(buf, p)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = _varint(buf, p)
        elif wt == 1:
            v = buf[p:p + 8]
            p += 8
        elif wt == 2:
            ln, p = _varint(buf, p)
            v = buf[p:p + ln]
            p += ln
        elif wt == 5:
            v = buf[p:p + 4]
            p += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((f, wt, v))
    return out


def _entry(value):
    e = {"dtype": 0, "shape": (), "shard": 0, "offset": 0, "size": 0, "crc32c": None}
    for f, wt, v in _proto_fields(value):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            dims = []
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:          # TensorShapeProto.dim
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1:
                            size = v3
                    dims.append(size)
            e["shape"] = tuple(dims)
        elif f == 3:
            e["shard"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
    return e


def read_index(index_path):
    """-> {variable name: {dtype, shape, shard, offset, size, crc32c}} ('' = bundle header, skipped)."""
    buf = open(index_path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError("%s is not a tensor-bundle index (bad table magic)" % index_path)
    foot = buf[-48:]
    p = 0
    _, p = _varint(foot, p)      # metaindex handle
    _, p = _varint(foot, p)
    ioff, p = _varint(foot, p)   # index handle
    isz, p = _varint(foot, p)
    out = {}
    for _, handle in _block_entries(buf, ioff, isz):
        boff, q = _varint(handle, 0)
        bsz, q = _varint(handle, q)
        for key, value in _block_entries(buf, boff, bsz):
            if key == b"":
                continue
            out[key.decode("utf-8")] = _entry(value)
    return out


def crc32c(data):
    """CRC-32C (Castagnoli) through the native library (chiron_crc32c, host code: no GPU involved)."""
    import ctypes
    from . import _lib
    data = bytes(data)
    out = ctypes.c_uint32()
    _lib.check(_lib.load().chiron_crc32c(data, len(data), ctypes.byref(out)))
    return out.value


def masked_crc32c(data):
    """TF's lib/hash/crc32c.h Mask(): rotate right by 15 and add a constant -- what BundleEntryProto.crc32c stores."""
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def read_tensors(prefix, entries=None, names=None, verify=True):
    """Load `names` (default: all float32 entries) from <prefix>.data-00000-of-00001.  Every tensor's bytes are checked
    against the masked CRC-32C its BundleEntryProto records (tensor_bundle.cc verifies the same on restore); a
    mismatch raises IOError naming the variable."""
    if entries is None:
        entries = read_index(prefix + ".index")
    data_path = prefix + ".data-00000-of-00001"
    out = {}
    with open(data_path, "rb") as f:
        for name in (names if names is not None else sorted(entries)):
            if name not in entries:
                raise KeyError("variable %r not in checkpoint %s" % (name, prefix))
            e = entries[name]
            if e["shard"] != 0:
                raise ValueError("multi-shard bundles are not supported")
            if e["dtype"] not in _NP:
                if names is None:
                    continue
                raise ValueError("variable %r has unsupported dtype %d" % (name, e["dtype"]))
            f.seek(e["offset"])
            raw = f.read(e["size"])
            if len(raw) != e["size"]:
                raise IOError("%s is truncated at variable %r" % (data_path, name))
            if verify and e["crc32c"] is not None and masked_crc32c(raw) != e["crc32c"]:
                raise IOError("%s: checksum mismatch in variable %r (stored crc32c %08x): the checkpoint data is corrupt"
                              % (data_path, name, e["crc32c"]))
            out[name] = np.frombuffer(raw, dtype=_NP[e["dtype"]]).reshape(e["shape"]).copy()
    return out
