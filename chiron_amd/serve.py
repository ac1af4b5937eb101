"""Serving surface of the hot path: the SavedModel PREDICT signature of the reference,
    inputs  {x: float32 [B, L], seq_len: int32 [B]}
    outputs {indices, values, dense_shape, logits, prob_logits, log_prob}
(chiron/export_test.py:24-41, :103-112), served from one engine over a local socket, plus a client that follows
chiron/chiron_client.py: fixed-size zero-padded batches per file (data_iterator :140-157), concurrent requests
with a throttle (_Result_Collection :59-109), per-file collection in batch order, sparse2dense (:111-131), the
`simple` overlap consensus with quality scores and the chiron_eval writers (do_inference :191-255).

The reference speaks gRPC to TensorFlow Serving; neither TF Serving's protos nor a network exist here, so the wire
is `multiprocessing.connection` (length-prefixed pickles of numpy arrays, HMAC-authenticated) on 127.0.0.1 -- the
request/response field names and semantics are the signature's.  As in export_test.py:34 the server divides
seq_len by the model's ratio and rounds (round-half-even, tf.round) before decoding, and decodes with its own
beam width (export_test.py:36-39: ctc_beam_search_decoder, merge_repeated=False); beam_width 0 selects greedy.
"""
import os
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
DEFAULT_AUTHKEY = b"chiron-predict"


class PredictServer(object):
    """One engine, many connections.  Every connection thread takes an engine slot for the duration of a request
    (engines have `n_slots` independent streams), so `n_slots` requests are in flight on the GPU at once."""

    def __init__(self, engine, address=("127.0.0.1", 0), beam_width=0, authkey=DEFAULT_AUTHKEY):
        self.engine = engine
        self.beam_width = int(beam_width)
        self._listener = Listener(address, authkey=authkey)
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
                    req = conn.recv()
                except (EOFError, OSError):
                    return
                try:
                    if req.get("method") == "signature":
                        rep = {"inputs": SIGNATURE_INPUTS, "outputs": SIGNATURE_OUTPUTS, "segment_len": self.engine.segment_len,
                               "T": self.engine.T, "ratio": self.engine.ratio, "max_batch": self.engine.max_batch,
                               "beam_width": self.beam_width}
                    else:
                        missing = [k for k in SIGNATURE_INPUTS if k not in req.get("inputs", {})]
                        if missing:
                            raise KeyError("missing inputs %s" % missing)
                        rep = {"outputs": self.predict(req["inputs"]["x"], req["inputs"]["seq_len"],
                                                       want_logits=req.get("want_logits", True))}
                except Exception as exc:                             # the failure travels to the caller, like a gRPC status
                    rep = {"error": "%s: %s" % (type(exc).__name__, exc)}
                try:
                    conn.send(rep)
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

    def __init__(self, address, concurrency=4, authkey=DEFAULT_AUTHKEY):
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

    def _call(self, req):
        c = self._conn()
        c.send(req)
        rep = c.recv()
        if "error" in rep:
            raise PredictError(rep["error"])
        return rep

    def signature(self):
        return self._call({"method": "signature"})

    def predict(self, x, seq_len, want_logits=True):
        return self._call({"method": "predict", "inputs": {"x": x, "seq_len": seq_len}, "want_logits": want_logits})["outputs"]

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
def sparse2dense(indices, values):
    """chiron_client.py:111-131: rows of the SparseTensor in order, and the batch rows that decoded to something."""
    unique, counts = np.unique(indices[:, 0], return_counts=True)
    reads, pos = [], 0
    for c in counts:
        reads.append(values[pos:pos + c])
        pos += c
    return reads, unique


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
                 segment_len=None, jump=None, start=0):
        if mode not in ("dna", "rna"):
            raise ValueError("Mode has to be either rna or dna.")
        self.input, self.output, self.server, self.mode = input, output, server, mode
        self.batch_size, self.concurrency, self.extension, self.concise = batch_size, concurrency, extension, concise
        self.segment_len = segment_len if segment_len else (400 if mode == "dna" else 2000)
        self.jump = jump if jump else (30 if mode == "dna" else 200)
        self.start = start
        self.recursive = True
        self.beam = 0
        self.model = "served"


def do_inference(FLAGS, client=None):
    """chiron_client.py:191-255.  Returns {file stem: consensus string}."""
    own = client is None
    if own:
        host, port = FLAGS.server.rsplit(":", 1)
        client = PredictClient((host, int(port)), concurrency=FLAGS.concurrency)
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
            reads, uniq = sparse2dense(out["indices"], out["values"])
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
        bpreads = [ce.index2base(r) for r in reads]
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
    cp = sub.add_parser("client")
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
        res = do_inference(ClientFlags(a.input, a.output, a.server, a.mode, a.batch_size, a.concurrency, a.extension, a.concise))
        print("%d reads written to %s" % (len(res), a.output))
        return 0
    from .engine import Engine
    from .model import load_model
    seg = a.segment_len if a.segment_len else (400 if a.mode == "dna" else 2000)
    spec, weights = load_model(a.model, allow_synthetic=a.synthetic_weights)
    eng = Engine(spec, weights, max_batch=a.batch_size, segment_len=seg, n_slots=a.slots, max_beam=a.beam)
    srv = PredictServer(eng, ("127.0.0.1", a.port), beam_width=a.beam)
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
