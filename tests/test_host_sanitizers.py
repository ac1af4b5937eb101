"""The host-side C++ of the C ABI (chiron_assemble, chiron_parse_signal_text) under AddressSanitizer and
UndefinedBehaviorSanitizer (the reference has no sanitizer coverage; its Python cannot overrun a
buffer, this C++ can), plus a randomized differential test of the native consensus vote against the Python form of
easy_assembler.py:302-335 / :393-432."""
import os
import subprocess

import numpy as np
import pytest

from chiron_amd import assembly

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_fuzz_host_entry_points_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(HERE, "native", "fuzz_host.cpp"), os.path.join(ROOT, "chiron_amd", "csrc", "assemble.cpp"),
           "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, "3000"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "clean" in r.stdout


def _random_case(rng, related):
    n = int(rng.randint(1, 14))
    genome = "".join(rng.choice(list("ACGT"), size=600))
    segs, pos = [], 0
    for _ in range(n):
        ln = int(rng.randint(1, 70))
        if related:
            s = list(genome[pos:pos + ln]) or ["A"]
            for i in range(len(s)):
                if rng.rand() < 0.06:
                    s[i] = "ACGT"[rng.randint(4)]
            segs.append("".join(s))
            pos = max(0, min(len(genome) - 80, pos + int(rng.randint(0, ln + 4))))
        else:
            segs.append("".join(rng.choice(list("ACGT"), size=ln)))
    return segs


@pytest.mark.parametrize("kernal", ["glue", "stick"])
def test_native_vote_equals_python_form_on_random_segments(built, kernal):
    """bit-exact: the vote is integer counts (float64 holders) and per-base sums of per-segment qualities"""
    rng = np.random.RandomState(31 if kernal == "glue" else 32)
    for it in range(400):
        segs = _random_case(rng, related=bool(it & 1))
        qs = rng.uniform(0, 20, size=(len(segs), 1))
        want = assembly._python_assembly(segs, None, 0.975, 0.2, kernal)
        got = assembly.simple_assembly(segs, 0.975, kernal=kernal)
        assert got.shape == want.shape and np.array_equal(got, want), (it, segs)
        wc, wq = assembly._python_assembly(segs, qs, 0.975, 0.2, kernal)
        gc, gq = assembly.simple_assembly_qs(segs, qs, 0.975, kernal=kernal)
        assert np.array_equal(gc, wc), (it, segs)
        np.testing.assert_allclose(gq, wq, rtol=1e-13, atol=0)
