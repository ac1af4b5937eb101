"""Wall / user / sys timer around the whole evaluation (counterpart of
chiron/utils/unix_time.py:11-27; feeds the meta/all.meta line, chiron_eval.py:535-544)."""
import resource
import time


def unix_time(function, *args, **kwargs):
    t0, r0 = time.time(), resource.getrusage(resource.RUSAGE_SELF)
    function(*args, **kwargs)
    r1, t1 = resource.getrusage(resource.RUSAGE_SELF), time.time()
    return {"real": t1 - t0, "sys": r1.ru_stime - r0.ru_stime, "user": r1.ru_utime - r0.ru_utime}
