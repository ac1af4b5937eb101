"""Serving surface of the hot path: the SavedModel PREDICT signature of the reference,
    inputs  {x: float32 [B, L], seq_len: int32 [B]}
    outputs {indices, values, dense_shape, logits, prob_logits, log_prob}
(chiron/export_test.py:24-41, :103-112), served from one engine over a local socket, plus a client that follows
chiron/chiron_client.py: fixed-size zero-padded batches per file (data_iterator :140-157), concurrent requests
with a throttle (_Result_Collection :59-109), per-file collection in batch order, sparse2dense (:111-131), the
`simple` overlap consensus with quality scores and the chiron_eval writers (do_inference :191-255).

The reference speaks gRPC to TensorFlow Serving; neither TF Serving's protos nor a network exist here, so the wire
is a length-prefixed binary frame over `multiprocessing.connection` on 127.0.0.1: a JSON header (method, scalar
fields, and name / dtype / shape of every tensor) followed by the tensors' raw little-endian bytes.  Nothing is
unpickled on either side (only send_bytes / recv_bytes are used), dtypes are restricted to the signature's, and the
connection handshake is HMAC-authenticated with a per-deployment key: the server generates one (or takes
CHIRON_SERVE_AUTHKEY / --authkey-file) -- there is no built-in default.  The request/response field names and
semantics are the signature's.  As in export_test.py:34 the server divides
seq_len by the model's ratio and rounds (round-half-even, tf.round) before decoding, and decodes with its own
beam width (export_test.py:36-39: ctc_beam_search_decoder, merge_repeated=False); beam_width 0 selects greedy.
"""
import json
import os
import struct
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from multiprocessing.connection import Client, Listener

import numpy as np

from . import assembly
from . import eval as ce
from . import signal_io
from .engine import seq_len_for_engine

SIGNATURE_INPUTS = ("x", "seq_len")
SIGNATURE_OUTPUTS = ("indices", "values", "dense_shape", "logits", "prob_logits", "log_prob")
WIRE_DTYPES = {"float32": np.dtype("<f4"), "int32": np.dtype("<i4"), "int64": np.dtype("<i8")}
MAX_FRAME = 1 << 31


def make_authkey():
    """A fresh per-deployment key for the connection handshake (hex so that it survives files and env vars)."""
    return os.urandom(24).hex().encode("ascii")


def authkey_from_env(path=None):
    """Key from a 0600 file (`path`) or CHIRON_SERVE_AUTHKEY; None when neither is set."""
    if path:
        with open(path, "rb") as f:
            return f.read().strip()
    v = os.environ.get("CHIRON_SERVE_AUTHKEY")
    return v.encode("ascii") if v else None


def pack_frame(fields, tensors=None):
    """-> bytes: u32 header length | JSON header | raw tensor bytes in header order"""
    tensors = tensors or {}
    arrays, blobs = [], []
    for name, a in tensors.items():
        if a is None:
            continue
        a = np.ascontiguousarray(a)
        if a.dtype.name not in WIRE_DTYPES:
            raise TypeError("tensor %r has dtype %s; the wire carries %s" % (name, a.dtype, sorted(WIRE_DTYPES)))
        a = a.astype(WIRE_DTYPES[a.dtype.name], copy=False)
        arrays.append({"name": name, "dtype": a.dtype.name, "shape": list(a.shape)})
        blobs.append(a.tobytes())
    header = json.dumps({"fields": fields, "tensors": arrays}).encode("utf-8")
    return b"".join([struct.pack("<I", len(header)), header] + blobs)


def unpack_frame(buf):
    """-> (fields, {name: ndarray}); rejects anything that is not exactly header + declared tensors"""
    buf = memoryview(buf)
    if len(buf) < 4:
        raise ValueError("short frame")
    hlen, = struct.unpack_from("<I", buf, 0)
    if 4 + hlen > len(buf):
        raise ValueError("frame header overruns the frame")
    head = json.loads(bytes(buf[4:4 + hlen]).decode("utf-8"))
    pos = 4 + hlen
    tensors = {}
    for d in head.get("tensors", []):
        dt = WIRE_DTYPES.get(d.get("dtype"))
        shape = d.get("shape")
        if dt is None or not isinstance(shape, list) or any((not isinstance(v, int)) or v < 0 for v in shape) or len(shape) > 4:
            raise ValueError("bad tensor descriptor %r" % (d,))
        n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        if pos + n > len(buf):
            raise ValueError("tensor %r overruns the frame" % d.get("name"))
        tensors[str(d["name"])] = np.frombuffer(buf[pos:pos + n], dtype=dt).reshape(shape)
        pos += n
    if pos != len(buf):
        raise ValueError("%d trailing bytes in frame" % (len(buf) - pos))
    fields = head.get("fields", {})
    if not isinstance(fields, dict):
        raise ValueError("frame fields must be an object")
    return fields, tensors


class PredictServer(object):
    """One engine, many connections.  Every connection thread takes an engine slot for the duration of a request
    (engines have `n_slots` independent streams), so `n_slots` requests are in flight on the GPU at once."""

    def __init__(self, engine, address=("127.0.0.1", 0), beam_width=0, authkey=None):
        self.engine = engine
        self.beam_width = int(beam_width)
        self.authkey = authkey if authkey else make_authkey()     # hand this to the clients of THIS server
        self._listener = Listener(address, authkey=self.authkey)
        self.address = self._listener.address
        self._slots = list(range(engine.n_slots))
        self._slot_cv = threading.Condition()
        self._threads = []
        self._stop = False
        self._accept_thread = threading.Thread(target=self._accept_loop, daemon=True)
        self._accept_thread.start()

    # -- engine access ---------------------------------------------------------------------------------------
    def _take_slot(self):
        with self._slot_cv:
            while not self._slots:
                self._slot_cv.wait()
            return self._slots.pop()

    def _give_slot(self, slot):
        with self._slot_cv:
            self._slots.append(slot)
            self._slot_cv.notify()

    def predict(self, x, seq_len, want_logits=True):
        """The signature itself (also usable in-process)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        seq_len = np.asarray(seq_len).reshape(-1)
        if x.ndim != 2 or x.shape[1] != self.engine.segment_len:
            raise ValueError("x must be [batch, %d], got %s" % (self.engine.segment_len, (x.shape,)))
        if seq_len.shape[0] != x.shape[0]:
            raise ValueError("seq_len has %d entries for %d rows" % (seq_len.shape[0], x.shape[0]))
        sl = seq_len_for_engine(seq_len, self.engine.ratio)          # export_test.py:34
        B, mb = x.shape[0], self.engine.max_batch
        parts = []
        slot = self._take_slot()
        try:
            for a in range(0, max(B, 1), mb):
                parts.append(self.engine.infer(x[a:a + mb], sl[a:a + mb], beam_width=self.beam_width, want_prob=True,
                                               want_logits=want_logits, slot=slot) if B else None)
        finally:
            self._give_slot(slot)
        if not B:
            z = np.zeros
            return {"indices": z((0, 2), np.int64), "values": z((0,), np.int64), "dense_shape": z((2,), np.int64),
                    "logits": z((0, self.engine.T, 5), np.float32), "prob_logits": z((0, 1), np.float32),
                    "log_prob": z((0, 1), np.float32)}
        idx, val, row0, width = [], [], 0, 0
        for r in parts:                                              # re-base the row index of each engine batch
            i = r.decoded.indices.copy()
            i[:, 0] += row0
            idx.append(i)
            val.append(r.decoded.values)
            row0 += int(r.decoded.dense_shape[0])
            width = max(width, int(r.decoded.dense_shape[1]))
        out = {"indices": np.concatenate(idx), "values": np.concatenate(val),
               "dense_shape": np.asarray([B, width], dtype=np.int64),
               "prob_logits": np.concatenate([r.prob_logits for r in parts]),
               "log_prob": np.concatenate([r.log_prob for r in parts])}
        out["logits"] = np.concatenate([r.logits for r in parts]) if want_logits else None
        return out

    # -- wire ------------------------------------------------------------------------------------------------
    def _accept_loop(self):
        while not self._stop:
            try:
                conn = self._listener.accept()
            except (OSError, EOFError):
                return
            except Exception:          # failed handshake (wrong authkey): keep serving
                continue
            t = threading.Thread(target=self._serve, args=(conn,), daemon=True)
            t.start()
            self._threads.append(t)

    def _serve(self, conn):
        with conn:
            while True:
                try:
                    raw = conn.recv_bytes(MAX_FRAME)
                except (EOFError, OSError):
                    return
                try:
                    req, tensors = unpack_frame(raw)
                    if req.get("method") == "signature":
                        rep = pack_frame({"inputs": list(SIGNATURE_INPUTS), "outputs": list(SIGNATURE_OUTPUTS),
                                          "segment_len": self.engine.segment_len, "T": self.engine.T, "ratio": self.engine.ratio,
                                          "max_batch": self.engine.max_batch, "beam_width": self.beam_width})
                    elif req.get("method") == "predict":
                        missing = [k for k in SIGNATURE_INPUTS if k not in tensors]
                        if missing:
                            raise KeyError("missing inputs %s" % missing)
                        rep = pack_frame({}, self.predict(tensors["x"], tensors["seq_len"], want_logits=bool(req.get("want_logits", True))))
                    else:
                        raise ValueError("unknown method %r" % (req.get("method"),))
                except Exception as exc:                             # the failure travels to the caller, like a gRPC status
                    rep = pack_frame({"error": "%s: %s" % (type(exc).__name__, exc)})
                try:
                    conn.send_bytes(rep)
                except (OSError, EOFError):
                    return

    def close(self):
        self._stop = True
        try:
            self._listener.close()
        except OSError:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class PredictError(RuntimeError):
    pass


class PredictClient(object):
    """`stub.Predict` / `stub.Predict.future` of chiron_client.py:208-227 over the local wire.  Each in-flight
    request uses its own connection (a pool of `concurrency` of them)."""

    def __init__(self, address, authkey, concurrency=4):
        if not authkey:
            raise ValueError("the server's authkey is required (PredictServer.authkey, --authkey-file or CHIRON_SERVE_AUTHKEY)")
        self.address = address
        self.authkey = authkey
        self._pool = ThreadPoolExecutor(max_workers=max(1, concurrency))
        self._local = threading.local()
        self._conns = []
        self._lock = threading.Lock()

    def _conn(self):
        c = getattr(self._local, "conn", None)
        if c is None:
            c = Client(self.address, authkey=self.authkey)
            self._local.conn = c
            with self._lock:
                self._conns.append(c)
        return c

    def _call(self, fields, tensors=None):
        c = self._conn()
        c.send_bytes(pack_frame(fields, tensors))
        rep, out = unpack_frame(c.recv_bytes(MAX_FRAME))
        if "error" in rep:
            raise PredictError(rep["error"])
        return rep, out

    def signature(self):
        return self._call({"method": "signature"})[0]

    def predict(self, x, seq_len, want_logits=True):
        x = np.ascontiguousarray(x, dtype=np.float32)
        seq_len = np.ascontiguousarray(np.asarray(seq_len).reshape(-1), dtype=np.int32)
        out = self._call({"method": "predict", "want_logits": bool(want_logits)}, {"x": x, "seq_len": seq_len})[1]
        out.setdefault("logits", None)
        return out

    def predict_future(self, x, seq_len, want_logits=False):
        return self._pool.submit(self.predict, x, seq_len, want_logits)

    def close(self):
        self._pool.shutdown(wait=True)
        with self._lock:
            for c in self._conns:
                try:
                    c.close()
                except OSError:
                    pass
            self._conns = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# -------------------------------------------------------------------------------------------------------------
# the reference client's job: a folder of .signal files -> result/ segments/ meta/ through the server
# -------------------------------------------------------------------------------------------------------------
def gen_file_list(input_path):
    """chiron_client.py:132-139 (os.walk for *.signal), sorted for determinism."""
    out = []
    for root, _, names in os.walk(input_path):
        out += [os.path.join(root, n) for n in names if n.endswith(".signal")]
    return sorted(out)


def data_iterator(file_list, batch_size, start=0, segment_len=400, jump_step=30):
    """chiron_client.py:140-157: per file, fixed-size batches, the last one zero padded (not wrapped)."""
    for f_p in file_list:
        ds = signal_io.read_data_for_eval(f_p, start, jump_step, segment_len)
        reads_n = ds.reads_n
        n_batches = len(range(0, reads_n, batch_size))
        for index, _ in enumerate(range(0, reads_n, batch_size)):
            bx, sl = ds.next_batch(batch_size, shuffle=False)[:2]
            bx = np.pad(bx, ((0, batch_size - len(bx)), (0, 0)), mode="constant")
            sl = np.pad(sl, (0, batch_size - len(sl)), mode="constant")
            yield bx, sl, index, f_p, n_batches, reads_n


class ClientFlags(object):
    """FLAGS of chiron_client.py:257-283 (+ DNA_CONF / RNA_CONF :49-57)."""

    def __init__(self, input, output, server, mode="dna", batch_size=100, concurrency=4, extension="fastq", concise=False,
                 segment_len=None, jump=None, start=0, authkey=None):
        if mode not in ("dna", "rna"):
            raise ValueError("Mode has to be either rna or dna.")
        self.input, self.output, self.server, self.mode = input, output, server, mode
        self.batch_size, self.concurrency, self.extension, self.concise = batch_size, concurrency, extension, concise
        self.segment_len = segment_len if segment_len else (400 if mode == "dna" else 2000)
        self.jump = jump if jump else (30 if mode == "dna" else 200)
        self.start = start
        self.authkey = authkey
        self.recursive = True
        self.beam = 0
        self.model = "served"


def do_inference(FLAGS, client=None):
    """chiron_client.py:191-255.  Returns {file stem: consensus string}."""
    own = client is None
    if own:
        host, port = FLAGS.server.rsplit(":", 1)
        client = PredictClient((host, int(port)), FLAGS.authkey or authkey_from_env(), concurrency=FLAGS.concurrency)
    files = gen_file_list(FLAGS.input)
    pending = {}                                                     # file -> {batch index: (reads, probs)}
    expect = {}
    lock = threading.Lock()
    throttle = threading.Semaphore(max(1, FLAGS.concurrency))        # _Result_Collection.throttle
    futures = []

    def done_cb(f_p, i, n_batches):
        def cb(fut):
            throttle.release()
            out = fut.result()
            # chiron_client.py:111-131: the rows that decoded to something, in order
            (reads,), (uniq,) = ce.sparse2dense(([ce.SparseTensor(out["indices"], out["values"], out["dense_shape"])], None))
            with lock:
                pending.setdefault(f_p, {})[i] = (reads, out["prob_logits"][uniq])
                expect[f_p] = n_batches
        return cb

    for bx, sl, i, f_p, n_batches, _ in data_iterator(files, FLAGS.batch_size, FLAGS.start, FLAGS.segment_len, FLAGS.jump):
        throttle.acquire()
        fut = client.predict_future(bx, sl, want_logits=False)
        fut.add_done_callback(done_cb(f_p, i, n_batches))
        futures.append(fut)
    for fut in futures:
        fut.result()                                                 # re-raises a server-side failure
    results = {}
    kernal = ce.get_assembler_kernal(FLAGS.jump, FLAGS.segment_len)
    for f_p in files:
        if f_p not in pending:
            continue
        reads, probs = [], []
        for i in range(expect[f_p]):                                 # batch order == window order within the file
            reads += pending[f_p][i][0]
            probs.append(pending[f_p][i][1])
        if not reads:
            continue
        probs = np.concatenate(probs)
        bpreads = ce.bases_of_reads(reads)
        consensus, qs_consensus = assembly.simple_assembly_qs(bpreads, probs, FLAGS.jump / FLAGS.segment_len, kernal=kernal)
        qs_string = ce.qs(consensus, qs_consensus)
        c_bpread = ce.index2base(np.argmax(consensus, axis=0))
        file_pre = os.path.basename(os.path.splitext(f_p)[0])
        ce.write_output(bpreads, c_bpread, [np.nan] * 4, file_pre, concise=FLAGS.concise, suffix=FLAGS.extension,
                        q_score=qs_string, global_setting=FLAGS)
        results[file_pre] = c_bpread
    if own:
        client.close()
    return results


def start_server(a):
    """engine + PredictServer for the parsed `server` arguments -> (server, engine)"""
    from .engine import Engine
    from .model import load_model
    seg = a.segment_len if a.segment_len else (400 if a.mode == "dna" else 2000)
    spec, weights, _ = load_model(a.model, allow_synthetic=a.synthetic_weights)
    key = None
    if a.authkey_file and os.path.exists(a.authkey_file):
        key = authkey_from_env(a.authkey_file)
    key = key or authkey_from_env() or make_authkey()
    if a.authkey_file and not os.path.exists(a.authkey_file):
        fd = os.open(a.authkey_file, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(key)
    # the same dtype / calibration switches as `chiron call`: a read decodes to the same string behind either entry point
    eng = Engine(spec, weights, max_batch=a.batch_size, segment_len=seg, n_slots=a.slots, max_beam=a.beam, dtype=getattr(a, "dtype", "fp32"),
                 calibrate=not getattr(a, "no_calibration", False))
    return PredictServer(eng, ("127.0.0.1", a.port), beam_width=a.beam, authkey=key), eng


def main(argv=None):
    """python -m chiron_amd.serve server -m <model dir> [--port P] | client -i <signals> -o <out> --server host:port"""
    import argparse
    ap = argparse.ArgumentParser(prog="chiron_amd.serve")
    sub = ap.add_subparsers(dest="cmd", required=True)
    sp = sub.add_parser("server")
    sp.add_argument("-m", "--model", required=True)
    sp.add_argument("--port", type=int, default=8500)
    sp.add_argument("--mode", default="dna")
    sp.add_argument("-b", "--batch_size", type=int, default=400)
    sp.add_argument("-l", "--segment_len", type=int, default=None)
    sp.add_argument("--beam", type=int, default=50)                 # export_test.py beam_width flag
    sp.add_argument("--slots", type=int, default=2)
    sp.add_argument("--synthetic-weights", action="store_true")
    sp.add_argument("--dtype", default="fp32", choices=["fp32", "fp16", "fp16-w2", "fp32-split"])
    sp.add_argument("--no-calibration", dest="no_calibration", action="store_true", help="--dtype fp16: as `chiron call --no-calibration`")
    sp.add_argument("--authkey-file", default=None, help="file holding the handshake key; created (0600) with a fresh key when absent")
    cp = sub.add_parser("client")
    cp.add_argument("--authkey-file", default=None, help="the key file the server wrote (or set CHIRON_SERVE_AUTHKEY)")
    cp.add_argument("-i", "--input", required=True)
    cp.add_argument("-o", "--output", required=True)
    cp.add_argument("--server", default="127.0.0.1:8500")
    cp.add_argument("--mode", default="dna")
    cp.add_argument("-b", "--batch_size", type=int, default=100)
    cp.add_argument("--concurrency", type=int, default=4)
    cp.add_argument("-e", "--extension", default="fastq")
    cp.add_argument("--concise", action="store_true")
    a = ap.parse_args(argv)
    if a.cmd == "client":
        res = do_inference(ClientFlags(a.input, a.output, a.server, a.mode, a.batch_size, a.concurrency, a.extension, a.concise,
                                       authkey=authkey_from_env(a.authkey_file)))
        print("%d reads written to %s" % (len(res), a.output))
        return 0
    srv, eng = start_server(a)
    print("serving %s on %s:%d (beam %d); Ctrl-C to stop" % (a.model, srv.address[0], srv.address[1], a.beam))
    try:
        while True:
            time.sleep(3600)
    except KeyboardInterrupt:
        srv.close()
        eng.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
