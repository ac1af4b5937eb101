"""The host-side C++ of the C ABI (chiron_assemble, chiron_parse_signal_text) under AddressSanitizer and
UndefinedBehaviorSanitizer (the reference has no sanitizer coverage; its Python cannot overrun a
buffer, this C++ can), plus randomized differential tests of the native displacement kernels and consensus vote against
oracle/assembly_oracle.py (whose `simple` kernel runs the real difflib.SequenceMatcher)."""
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


def test_fuzz_native_fast5_reader_under_asan_ubsan(tmp_path):
    """csrc/fast5.cpp under AddressSanitizer + UBSan on thousands of damaged copies of real fast5 files (the reference's DNA and RNA
    examples, a multi-read file with chunked + deflate signals and reference FASTQ, a contiguous one): random bytes, size / offset /
    count fields set to 0, all ones, 2^63, 2^32 +- 1 ..., damage behind the B-tree / heap signatures, truncations.  No crash, no
    out-of-bounds access, no signed overflow, no hang (time limit), nothing written past a capacity; every undamaged seed reads."""
    import sys
    sys.path.insert(0, HERE)
    import h5_writer
    rng = np.random.RandomState(5)
    reads = [("read_%d" % k, "id-%d" % k, rng.randint(-300, 1200, size=n).astype(np.int16), ("@x\nACGT\n+\n!!!!" if k % 2 else None))
             for k, n in enumerate((3000, 1, 2345))]
    multi, plain = str(tmp_path / "multi.fast5"), str(tmp_path / "plain.fast5")
    h5_writer.write_multi_read_fast5(multi, reads, chunk=700)
    h5_writer.write_multi_read_fast5(plain, reads[:1], chunk=None)
    rna = sorted(n for n in os.listdir(os.path.join(HERE, "golden", "example_rna")) if n.endswith(".fast5"))[0]
    seeds = [os.path.join(HERE, "golden", "example_dna", "read1.fast5"), os.path.join(HERE, "golden", "example_rna", rna), multi, plain]
    exe = str(tmp_path / "fuzz_fast5")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(HERE, "native", "fuzz_fast5.cpp"), os.path.join(ROOT, "chiron_amd", "csrc", "fast5.cpp"), "-o", exe, "-lz", "-ldl"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1",
               CHIRON_NO_LIBDEFLATE="1")           # zlib only: the sanitizers see every byte the inflate path touches
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, "4000", str(tmp_path / "scratch.fast5")] + seeds, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "clean" in r.stdout
    # the same damaged files through libdeflate where the box has it (no sanitizer inside the shared object, but its results are checked)
    env.pop("CHIRON_NO_LIBDEFLATE")
    r = subprocess.run([exe, "1500", str(tmp_path / "scratch.fast5")] + seeds, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]


@pytest.mark.parametrize("kernal", ["glue", "stick", "simple"])
def test_native_vote_equals_the_oracle_on_random_segments(built, kernal):
    """bit-exact: the vote is integer counts (float64 holders) and per-base sums of per-segment qualities"""
    from oracle import assembly_oracle as ao
    rng = np.random.RandomState({"glue": 31, "stick": 32, "simple": 33}[kernal])
    jr = 0.975 if kernal != "simple" else 0.075
    for it in range(400):
        segs = _random_case(rng, related=bool(it & 1))
        qs = rng.uniform(0, 20, size=(len(segs), 1))
        want, _ = ao.vote(segs, None, jr, 0.2, kernal)
        got = assembly.simple_assembly(segs, jr, kernal=kernal)
        assert got.shape == want.shape and np.array_equal(got, want), (it, segs)
        wc, wq = ao.vote(segs, qs, jr, 0.2, kernal)
        gc, gq = assembly.simple_assembly_qs(segs, qs, jr, kernal=kernal)
        assert np.array_equal(gc, wc), (it, segs)
        np.testing.assert_allclose(gq, wq, rtol=1e-13, atol=0)


def test_native_matching_blocks_equal_difflib(built):
    """chiron_overlap_displacement(simple) against difflib on the shapes that exercise its corners: short and long
    segments, unrelated / overlapping / identical / shifted pairs, and prev >= 200 bases, where difflib's autojunk drops
    every base from its index (each of A, C, G, T is "popular") and matches only grow from rectangle corners."""
    from oracle import assembly_oracle as ao
    rng = np.random.RandomState(41)
    pairs = [("ACGT", "ACGT"), ("A", "A"), ("A", "C"), ("ACGTACGT", "TTTT"), ("AAAA", "AAAAAAAA"), ("GATTACA", "TACAGATTACA")]
    for it in range(1500):
        genome = "".join(rng.choice(list("ACGT"), size=700))
        la, lb = (int(rng.randint(1, 60)), int(rng.randint(1, 60))) if it % 3 else (int(rng.randint(150, 320)), int(rng.randint(180, 330)))
        start = int(rng.randint(0, 300))
        prev = genome[start:start + lb]
        shift = int(rng.randint(-10, lb + 5))
        cur = list(genome[max(0, start + shift):max(0, start + shift) + la]) or ["A"]
        for i in range(len(cur)):
            if rng.rand() < 0.08:
                cur[i] = "ACGT"[rng.randint(4)]
        if rng.rand() < 0.1:
            del cur[len(cur) // 2:len(cur) // 2 + 2]
        pairs.append(("".join(cur) or "A", prev))
    for cur, prev in pairs:
        for jr in (0.075, 0.5, 0.975):
            want = ao.simple_displacement(cur, prev, 0.2, jr)
            got = assembly.simple_assembly_kernal(cur, prev, 0.2, jr)
            assert got[0] == want[0], (cur, prev, jr, got, want)
            assert abs(got[1] - want[1]) <= 1e-12 * max(1.0, abs(want[1]))


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_native_pipeline_threads_under_sanitizers(tmp_path, sanitizer):
    """csrc/pipeline.cpp (round 6: the host side of `chiron call` as C++ reader / packer / finisher threads) under ThreadSanitizer and
    under ASan + UBSan, with fast5.cpp and assemble.cpp, behind its null engine: 24 files (multi-read, chunked + deflate, references,
    one damaged), three runs each with 6 and 2 threads.  No data race, no leak, no overrun; every read is finished."""
    import sys
    sys.path.insert(0, HERE)
    import h5_writer
    rng = np.random.RandomState(17)
    inp = tmp_path / "in"
    inp.mkdir()
    files = []
    for k in range(23):
        reads = [("read_%d" % j if k % 3 == 0 else "", "id-%d-%d" % (k, j), rng.randint(200, 1000, size=int(rng.randint(300, 40000))).astype(np.int16),
                  "@x\nACGT\n+\n!!!!" if (k + j) % 4 == 0 else None) for j in range(3 if k % 3 == 0 else 1)]
        files.append(str(inp / ("f%02d.fast5" % k)))
        h5_writer.write_multi_read_fast5(files[-1], reads, chunk=None if k % 2 else 3000)
    files.append(str(inp / "damaged.fast5"))
    with open(files[-1], "wb") as f:
        f.write(b"\x89HDF\r\n\x1a\n" + b"\x01" * 500)
    exe = str(tmp_path / "tsan_pipeline")
    csrc = os.path.join(ROOT, "chiron_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", os.path.join(HERE, "native", "tsan_pipeline.cpp"),
           os.path.join(csrc, "pipeline.cpp"), os.path.join(csrc, "fast5.cpp"), os.path.join(csrc, "assemble.cpp"), "-o", exe, "-lz", "-ldl", "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    for threads in ("6", "2"):
        out = tmp_path / ("out" + threads)
        for sub in ("raw", "reference", "result", "segments", "meta"):
            (out / sub).mkdir(parents=True)
        r = subprocess.run([exe, str(out), threads] + files, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "clean" in r.stdout, r.stdout + r.stderr[-4000:]
        assert len(os.listdir(str(out / "result"))) == 8 * 3 + 15
