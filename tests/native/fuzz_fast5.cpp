// Seeded mutation fuzz of the native fast5 reader (chiron_amd/csrc/fast5.cpp behind chiron_fast5_*), built by
// tests/test_host_sanitizers.py under -fsanitize=address,undefined.  Seeds: real fast5 files (the reference's DNA / RNA examples,
// files written by tests/h5_writer.py).  Every iteration damages a copy -- random bytes, or an aligned 2 / 4 / 8-byte field set to a
// value a size / offset / count field must not take (0, all ones, 2^63, 2^32 +- 1, the file size) -- writes it out and walks the
// whole API over it.  The contract: a status code and a reason, never a crash, an out-of-bounds access, signed overflow, or a hang
// (the Python test runs this under a time limit); whatever a successful read reports must be consistent with the buffer it filled.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/chiron_amd.h"

namespace chiron {
chiron_status set_error(chiron_status st, const char* fmt, ...) {
  (void)fmt;
  return st;
}
}  // namespace chiron

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned long long rnd64() {
  rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
  return rng_state;
}
static unsigned rnd() { return (unsigned)(rnd64() >> 32); }

static std::vector<unsigned char> slurp(const char* path) {
  std::vector<unsigned char> d;
  FILE* f = fopen(path, "rb");
  if (!f) return d;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  d.resize(n > 0 ? (size_t)n : 0);
  if (!d.empty() && fread(d.data(), 1, d.size(), f) != d.size()) d.clear();
  fclose(f);
  return d;
}

static int walk(const char* path, long long* reads_ok) {
  chiron_fast5* f = nullptr;
  if (chiron_fast5_open(path, &f) != CHIRON_OK) return f == nullptr ? 0 : 1;   // a failed open leaves no handle
  const int n = chiron_fast5_read_count(f);
  if (n < 0) return 1;
  for (int i = 0; i < n && i < 8; ++i) {
    char suffix[64], rid[64];
    int64_t ns = -1, fq = -1;
    if (chiron_fast5_read_info(f, i, suffix, sizeof(suffix), rid, sizeof(rid), &ns, &fq) != CHIRON_OK) return 1;
    if (ns < 0 || fq < 0 || strlen(suffix) >= sizeof(suffix) || strlen(rid) >= sizeof(rid)) return 1;
    // capacity: what the file claims, bounded (a damaged count may claim 2^40 samples: the reader must then refuse, not write)
    const int64_t cap = ns < (1 << 22) ? ns : (1 << 22);
    std::vector<float> out((size_t)cap + 1, -12345.0f);
    const chiron_status st = chiron_fast5_signal(f, i, out.data(), cap, (int)(rnd() & 1));
    if (st == CHIRON_OK) {
      if (ns > cap) return 1;                       // more samples than capacity must be CHIRON_ERR_OVERFLOW
      if (out[(size_t)cap] != -12345.0f) return 1;  // nothing past the capacity
      ++*reads_ok;
    }
    if (fq > 0 && fq < (1 << 20)) {
      std::vector<char> txt((size_t)fq + 1);
      (void)chiron_fast5_fastq(f, i, txt.data(), fq + 1);
      std::vector<char> small(4);
      (void)chiron_fast5_fastq(f, i, small.data(), 4);   // too small: an error, no overrun (ASan watches)
    }
  }
  chiron_fast5_close(f);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: fuzz_fast5 <iterations> <scratch file> <seed file>...\n");
    return 2;
  }
  const int iters = atoi(argv[1]);
  const char* scratch = argv[2];
  std::vector<std::vector<unsigned char>> seeds;
  for (int a = 3; a < argc; ++a) {
    seeds.push_back(slurp(argv[a]));
    if (seeds.back().size() < 96) {
      fprintf(stderr, "fuzz_fast5: cannot read seed %s\n", argv[a]);
      return 2;
    }
    long long ok = 0;
    if (walk(argv[a], &ok) != 0 || ok == 0) {   // the undamaged seed must read
      fprintf(stderr, "fuzz_fast5: seed %s does not read cleanly\n", argv[a]);
      return 1;
    }
  }
  static const unsigned long long evil[] = {0ull, 1ull, 0xFFFFFFFFFFFFFFFFull, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFFull,
                                            0x100000000ull, 0x100000001ull, 0xFFFFFFFFFFFFFFF0ull, 0x10ull, 0xFFFFull, 0x4000000000ull};
  long long reads_ok = 0, opened = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<unsigned char> d = seeds[it % seeds.size()];
    const int kind = rnd() % 4;
    const int hits = 1 + rnd() % 3;
    for (int h = 0; h < hits; ++h) {
      if (kind == 0) {                                   // random bytes anywhere
        d[rnd() % d.size()] = (unsigned char)rnd();
      } else if (kind == 1) {                            // an aligned field takes an evil value
        const int w = 1 << (1 + rnd() % 3);              // 2, 4, 8 bytes
        const size_t pos = (rnd() % (d.size() / w)) * w;
        unsigned long long v = evil[rnd() % (sizeof(evil) / sizeof(evil[0]))];
        if (rnd() % 4 == 0) v = d.size() + (rnd() % 64) - 32;
        memcpy(&d[pos], &v, w);
      } else if (kind == 2) {                            // metadata lives in the first KBs and behind "TREE" / "SNOD" / "HEAP" / "GCOL" signatures
        const char* sig[] = {"TREE", "SNOD", "HEAP", "GCOL"};
        const char* sg = sig[rnd() % 4];
        size_t at = 0;
        for (size_t p = rnd() % d.size(), n = 0; n < d.size() - 4; ++n, p = (p + 1) % (d.size() - 4))
          if (memcmp(&d[p], sg, 4) == 0) { at = p; break; }
        const size_t pos = at + 4 + rnd() % 60;
        if (pos + 8 < d.size()) {
          const unsigned long long v = (rnd() & 1) ? evil[rnd() % 12] : rnd64();
          memcpy(&d[pos], &v, 1 << (rnd() % 4));
        }
      } else {                                           // truncation or a damaged header region
        if (rnd() & 1) d.resize(96 + rnd() % (d.size() - 96));
        else d[rnd() % 2048 % d.size()] ^= (unsigned char)(1u << (rnd() % 8));
      }
    }
    FILE* f = fopen(scratch, "wb");
    if (!f || fwrite(d.data(), 1, d.size(), f) != d.size()) {
      fprintf(stderr, "fuzz_fast5: cannot write %s\n", scratch);
      return 2;
    }
    fclose(f);
    const long long before = reads_ok;
    if (walk(scratch, &reads_ok) != 0) {
      fprintf(stderr, "fuzz_fast5: contract violated at iteration %d (kind %d)\n", it, kind);
      return 1;
    }
    opened += reads_ok > before;
  }
  printf("fuzz_fast5: %d damaged files, %lld still readable, clean\n", iters, opened);
  return 0;
}
