"""Host-side signal reading and windowing (counterpart of the eval part of
chiron/chiron_input.py: read_signal :527-539, read_signal_fast5 :541-555,
read_data_for_eval :253-292, padding :681-692, DataSet.next_batch eval branch
:194-250).  numpy arrays instead of Python lists; same values."""
import os

import numpy as np

MEDIAN = 0          # chiron_input.py:30-31
MEAN = 1
SIG_NORM = None     # chiron_input.py:32-39: FLAGS.sig_norm is never set -> no normalisation at inference


def _mad(x):
    """statsmodels.robust.mad: median(|x - median(x)|) / 0.6744897501960817."""
    x = np.asarray(x, dtype=np.float64)
    return np.median(np.abs(x - np.median(x))) / 0.6744897501960817


def _normalize(signal, stat_source, normalize):
    if normalize == MEAN:
        return (signal - np.mean(stat_source)) / float(np.std(stat_source))
    if normalize == MEDIAN:
        return (signal - np.median(stat_source)) / float(_mad(stat_source))
    return signal


def _parse_signal_text(data):
    """bytes of a .signal file -> float32 array.  The native parser (chiron_parse_signal_text in libchiron_amd.so,
    20x faster than str.split + numpy and it releases the GIL, so reader threads scale) gives the same values as the
    numpy conversion below, which is used when the library has not been built."""
    try:
        from . import _lib
        lib = _lib.load()
    except (ImportError, OSError):
        return np.asarray(data.split(), dtype=np.float32)
    import ctypes as C
    out = np.empty(len(data) // 2 + 1, dtype=np.float32)
    n = C.c_size_t()
    st = lib.chiron_parse_signal_text(data, len(data), out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(n))
    if st != _lib.OK:
        raise ValueError(lib.chiron_last_error().decode("utf-8", "replace"))
    return out[:n.value].copy()


def read_signal(file_path, normalize=None):
    """chiron_input.py:527-539: whitespace/newline separated numbers -> float32."""
    with open(file_path, "rb") as f:
        signal = _parse_signal_text(f.read())
    if signal.shape[0] == 0:
        return signal
    return _normalize(signal, signal, normalize)


def read_signal_fast5(fast5_path, normalize=None):
    """chiron_input.py:541-555 (statistics over the UNIQUE values, as the reference does)."""
    from . import fast5
    try:
        recs = fast5.read_fast5_native(fast5_path)          # csrc/fast5.cpp (B-tree walk + inflate without the GIL)
        if not recs:
            raise fast5.Fast5FormatError("no raw signal in %s" % fast5_path)
        signal = recs[0]["signal"]
    except ImportError:                                     # library not built: the Python reader gives the same samples
        signal = np.asarray(fast5.read_raw_signal(fast5_path))
    if signal.shape[0] == 0:
        return signal.astype(np.float32)
    uniq = np.unique(signal)
    return _normalize(signal, uniq, normalize)


def window_signal(signal, start_index, step, seg_length):
    """chiron_input.py:276-286: windows signal[i:i+L] for i in range(0, n, step), true lengths,
    zero padded (padding(), :681-692).  -> (event float32 [n_win, L], event_length int32 [n_win])."""
    sig = np.asarray(signal, dtype=np.float32)[start_index:]
    n = sig.shape[0]
    n_win = -(-n // step) if n > 0 else 0
    ln = np.minimum(n - step * np.arange(n_win), seg_length).astype(np.int32)
    if n_win == 0:
        return np.zeros((0, seg_length), dtype=np.float32), ln
    # ONE zero-padded copy of the signal; the windows are overlapping VIEWS of it (row r starts r * step samples in): nothing is
    # gathered -- the engine copies row by row into its staging buffer (chiron_engine_submit_pieces with row stride = step), and
    # np.concatenate / slicing / comparison see an ordinary [n_win, seg_length] array.  Read-only, because rows share memory.
    buf = np.zeros((n_win - 1) * step + seg_length, dtype=np.float32)
    m = min(n, buf.shape[0])           # jump > seg_length: the samples between the last window's end and the signal's are in no window
    buf[:m] = sig[:m]
    ev = np.lib.stride_tricks.as_strided(buf, shape=(n_win, seg_length), strides=(step * 4, 4), writeable=False)
    if os.environ.get("CHIRON_WINDOW_COPY"):      # A/B switch: materialise the windows as rounds 1 .. 4 did (a [n_win, seg_length] copy per read)
        ev = np.ascontiguousarray(ev)
    return ev, ln


class DataSet(object):
    """Evaluation data set (chiron_input.py DataSet with for_eval=True)."""

    def __init__(self, event, event_length):
        self.event = event
        self.event_length = event_length
        self._index = 0
        self.epochs_completed = 0

    @property
    def reads_n(self):
        return len(self.event_length)

    def next_batch(self, batch_size, shuffle=False):
        """chiron_input.py:194-250 eval branch: sequential slices; the call that reaches the end
        returns the remainder and sets epochs_completed."""
        start = self._index
        if start + batch_size >= self.reads_n:
            self.epochs_completed += 1
            end = self.reads_n
            self._index = 0
        else:
            end = start + batch_size
            self._index = end
        return self.event[start:end], self.event_length[start:end].astype(np.int32), []


def read_data_for_eval(file_path, start_index=0, step=20, seg_length=200, reverse_fast5=False):
    """chiron_input.py:253-292."""
    if file_path.endswith(".signal"):
        f_signal = read_signal(file_path, normalize=SIG_NORM)
    elif file_path.endswith(".fast5"):
        f_signal = read_signal_fast5(file_path, normalize=SIG_NORM)
        if reverse_fast5:
            f_signal = f_signal[::-1]
    else:
        raise TypeError("Input file should be a signal file or fsat5 file, but a %s file is given." % (file_path))
    ev, ln = window_signal(f_signal, start_index, step, seg_length)
    return DataSet(ev, ln)
