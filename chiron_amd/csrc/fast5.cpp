// fast5 (HDF5) reader of the hot path, without libhdf5 / h5py: what chiron/utils/extract_sig_ref.py:149-193
// (extract_file / extract_file_v2) takes from a fast5 file -- the raw int16 signal of every read, its read_id attribute
// and the reference FASTQ if the file carries one -- plus the writer of the reference's `.signal` text format
// (extract_sig_ref.py:122-123).  SURVEY.md 8(f)1: `chiron call` feeds the engine from the decoded samples directly; the
// text file is still written for output-tree fidelity, but nobody parses it back.
//
// Covered (the MinKNOW-era files of the reference's example data, and what tests/h5_writer.py emits): superblock v0 / v1,
// version-1 object headers with continuation blocks, old-style groups (symbol table: v1 B-tree + SNOD + local heap) and
// new-style groups with compact link storage (Link messages), contiguous / compact / chunked 1-D datasets (v1 chunk
// B-tree) with the deflate and shuffle filters, fixed-point and IEEE datatypes, fixed and variable-length strings (global
// heap), attribute messages v1-v3.  Anything else is reported as CHIRON_ERR_INVALID with a reason; the caller logs and
// skips the read, as the reference does for unreadable files (extract_sig_ref.py:97-117).  Host code: no GPU needed,
// thread-safe (every handle owns its bytes), releases nothing Python-side.
#include <zlib.h>
#include <dlfcn.h>
#include <cstdlib>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/chiron_amd.h"

namespace chiron {
chiron_status set_error(chiron_status st, const char* fmt, ...);
}

// Optional accelerator for the deflate filter: libdeflate (same streams, 2 .. 3 x the speed of zlib's inflate), looked up at run
// time because the image ships the shared object without its header; zlib stays the reference path (CHIRON_NO_LIBDEFLATE=1
// forces it, and any stream libdeflate does not accept is handed to zlib).  Behind the fp16 engine the fast5 side of
// `chiron call` is host-bound, and inflate is its largest part (18 ns per sample against 8 for the raw/*.signal text).
struct Deflater {
  void* (*alloc)() = nullptr;
  int (*zlib_decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
  void (*release)(void*) = nullptr;
  Deflater() {
    if (getenv("CHIRON_NO_LIBDEFLATE")) return;
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    alloc = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
    zlib_decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_zlib_decompress"));
    release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
    if (!alloc || !zlib_decompress || !release) alloc = nullptr;
  }
};
static const Deflater& deflater() {
  static const Deflater d;   // thread-safe initialisation
  return d;
}
// one decompressor per thread (libdeflate's are not shareable between concurrent calls)
static void* thread_decompressor() {
  struct Holder {
    void* p = nullptr;
    ~Holder() { if (p) deflater().release(p); }
  };
  static thread_local Holder h;
  if (!h.p && deflater().alloc) h.p = deflater().alloc();
  return h.p;
}
// true: `out` holds the inflated stream (exactly its size).  false: not available / not accepted -- the caller takes zlib.
static bool fast_inflate(const uint8_t* in, size_t n_in, std::vector<uint8_t>& out, size_t expect) {
  void* dc = thread_decompressor();
  if (!dc) return false;
  out.resize(std::max<size_t>(expect, 64));
  size_t actual = 0;
  if (deflater().zlib_decompress(dc, in, n_in, out.data(), out.size(), &actual) != 0) return false;   // LIBDEFLATE_SUCCESS == 0
  out.resize(actual);
  return true;
}

namespace {

constexpr uint64_t UNDEF = 0xFFFFFFFFFFFFFFFFull;

struct FormatError : std::runtime_error {
  explicit FormatError(const std::string& m) : std::runtime_error(m) {}
};

struct Msg {
  uint16_t type;
  uint8_t flags;
  uint64_t off;   // body offset in the file image
  uint32_t size;
};

struct DType {
  int kind = 0;   // 0 int, 1 float, 3 fixed string, 9 vlen string
  bool sign = false;
  uint32_t size = 0;
};

struct ReadRec {
  std::string suffix, read_id, fastq;
  uint64_t signal_addr = 0;     // object header of the Signal dataset
  int64_t n_samples = 0;
};

struct H5 {
  std::vector<uint8_t> d;
  uint64_t base = 0, root = 0;

  void need(uint64_t off, uint64_t n) const {
    if (off > d.size() || n > d.size() - off) throw FormatError("truncated file (offset past the end)");
  }
  uint8_t u8(uint64_t o) const { need(o, 1); return d[o]; }
  uint16_t u16(uint64_t o) const { need(o, 2); uint16_t v; memcpy(&v, &d[o], 2); return v; }
  uint32_t u32(uint64_t o) const { need(o, 4); uint32_t v; memcpy(&v, &d[o], 4); return v; }
  uint64_t u64(uint64_t o) const { need(o, 8); uint64_t v; memcpy(&v, &d[o], 8); return v; }
  bool sig(uint64_t o, const char* s) const { need(o, 4); return memcmp(&d[o], s, 4) == 0; }

  void open(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) throw FormatError(std::string("cannot open ") + path);
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    d.resize(n > 0 ? (size_t)n : 0);
    const size_t got = d.empty() ? 0 : fread(d.data(), 1, d.size(), f);
    fclose(f);
    if (got != d.size()) throw FormatError("short read");
    static const uint8_t magic[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    if (d.size() < 96 || memcmp(d.data(), magic, 8) != 0) throw FormatError("not an HDF5 file");
    const int ver = d[8];
    if (ver != 0 && ver != 1) throw FormatError("HDF5 superblock version " + std::to_string(ver) + " is not supported");
    if (d[13] != 8 || d[14] != 8) throw FormatError("only 8-byte offsets/lengths are supported");
    const uint64_t p = 24 + (ver == 1 ? 4 : 0);
    base = u64(p);
    root = base + u64(p + 32 + 8);   // root group symbol table entry follows the four addresses
  }

  // ---- version-1 object headers
  std::vector<Msg> messages(uint64_t addr) const {
    if (sig(addr, "OHDR")) throw FormatError("version-2 object headers are not supported");
    if (u8(addr) != 1) throw FormatError("object header version " + std::to_string(u8(addr)));
    const unsigned nmsg = u16(addr + 2);
    std::vector<std::pair<uint64_t, uint64_t>> blocks{{addr + 16, u32(addr + 8)}};
    std::vector<Msg> out;
    for (size_t b = 0; b < blocks.size() && out.size() < nmsg; ++b) {
      uint64_t pos = blocks[b].first;
      const uint64_t end = pos + blocks[b].second;
      need(pos, blocks[b].second);
      while (pos + 8 <= end && out.size() < nmsg) {
        Msg m{u16(pos), u8(pos + 4), pos + 8, u16(pos + 2)};
        need(m.off, m.size);
        pos += 8 + m.size;
        if (m.type == 0x10) blocks.push_back({base + u64(m.off), u64(m.off + 8)});
        out.push_back(m);
      }
      if (blocks.size() > 4096) throw FormatError("object header continuation loop");
    }
    return out;
  }

  // ---- groups
  std::string heap_name(uint64_t heap, uint64_t off) const {
    if (!sig(heap, "HEAP")) throw FormatError("bad local heap");
    const uint64_t s = base + u64(heap + 24) + off;
    need(s, 1);
    const void* e = memchr(&d[s], 0, d.size() - s);
    if (!e) throw FormatError("unterminated name");
    return std::string(reinterpret_cast<const char*>(&d[s]), reinterpret_cast<const char*>(e));
  }
  // B-tree walks: a node of a damaged file may point back at itself or at an ancestor; depth alone would allow used^32 visits.
  // No tree of a real file has more nodes than the file has 24-byte pieces.
  void visit(uint64_t* budget) const {
    if (*budget == 0) throw FormatError("B-tree has more nodes than the file can hold (a loop)");
    --*budget;
  }
  uint64_t node_budget() const { return d.size() / 24 + 16; }
  void walk_group_btree(uint64_t node, uint64_t heap, std::map<std::string, uint64_t>& out, uint64_t* budget, int depth = 0) const {
    if (depth > 32) throw FormatError("group B-tree too deep");
    visit(budget);
    if (sig(node, "SNOD")) {
      const unsigned n = u16(node + 6);
      for (unsigned i = 0; i < n; ++i) {
        const uint64_t e = node + 8 + 40ull * i;
        out[heap_name(heap, u64(e))] = base + u64(e + 8);
      }
      return;
    }
    if (!sig(node, "TREE") || u8(node + 4) != 0) throw FormatError("bad group B-tree node");
    const unsigned used = u16(node + 6);
    uint64_t p = node + 24;
    for (unsigned i = 0; i < used; ++i, p += 16) walk_group_btree(base + u64(p + 8), heap, out, budget, depth + 1);   // key(8) child(8) ...
  }
  std::map<std::string, uint64_t> links(uint64_t addr) const {
    std::map<std::string, uint64_t> out;
    for (const Msg& m : messages(addr)) {
      if (m.type == 0x11) {
        uint64_t budget = node_budget();
        walk_group_btree(base + u64(m.off), base + u64(m.off + 8), out, &budget);
      } else if (m.type == 0x02) {   // Link Info: dense storage lives in a fractal heap
        const uint8_t flags = u8(m.off + 1);
        if (u64(m.off + 2 + ((flags & 1) ? 8 : 0)) != UNDEF) throw FormatError("dense (fractal heap) link storage is not supported");
      } else if (m.type == 0x06) {   // Link message
        const uint8_t flags = u8(m.off + 1);
        uint64_t p = m.off + 2;
        int ltype = 0;
        if (flags & 0x08) ltype = u8(p++);
        if (flags & 0x04) p += 8;
        if (flags & 0x10) p += 1;
        const int nsz = 1 << (flags & 3);
        uint64_t nlen = 0;
        for (int i = 0; i < nsz; ++i) nlen |= (uint64_t)u8(p + i) << (8 * i);
        p += nsz;
        need(p, nlen);
        std::string name(reinterpret_cast<const char*>(&d[p]), (size_t)nlen);
        p += nlen;
        if (ltype == 0) out[name] = base + u64(p);
      }
    }
    return out;
  }
  bool resolve(const std::string& path, uint64_t start, uint64_t* out) const {
    uint64_t addr = start;
    size_t i = 0;
    while (i < path.size()) {
      size_t j = path.find('/', i);
      if (j == std::string::npos) j = path.size();
      if (j > i) {
        const auto ch = links(addr);
        const auto it = ch.find(path.substr(i, j - i));
        if (it == ch.end()) return false;
        addr = it->second;
      }
      i = j + 1;
    }
    *out = addr;
    return true;
  }

  // ---- datatypes / dataspaces
  DType dtype(uint64_t off) const {
    DType t;
    const int cls = u8(off) & 0x0F;
    const uint8_t bits0 = u8(off + 1);
    t.size = u32(off + 4);
    t.kind = cls;
    // the element size comes from the file and divides / multiplies below (shuffle filter, byte counts): a damaged file must not
    // reach that arithmetic with 0 or an absurd value
    if (t.size == 0 || t.size > (1u << 20)) throw FormatError("datatype size " + std::to_string(t.size));
    if (cls == 0) {
      if (bits0 & 1) throw FormatError("big-endian integers are not supported");
      if (t.size != 1 && t.size != 2 && t.size != 4 && t.size != 8) throw FormatError("integer size " + std::to_string(t.size));
      t.sign = (bits0 & 8) != 0;
    } else if (cls == 1) {
      if (bits0 & 1) throw FormatError("big-endian floats are not supported");
      if (t.size != 4 && t.size != 8) throw FormatError("float size " + std::to_string(t.size));
    } else if (cls == 9) {
      if ((bits0 & 0x0F) != 1) throw FormatError("only variable-length strings are supported");
    } else if (cls != 3) {
      throw FormatError("datatype class " + std::to_string(cls) + " is not supported");
    }
    return t;
  }
  std::vector<uint64_t> dims(uint64_t off) const {
    const int ver = u8(off), rank = u8(off + 1);
    std::vector<uint64_t> v(rank);
    for (int i = 0; i < rank; ++i) v[i] = u64(off + (ver == 1 ? 8 : 4) + 8ull * i);
    return v;
  }
  // element count of a dataspace; every product checked, and bounded so that count x (element size <= 8) and a cast to int64
  // cannot wrap (a dimension of 2^63 in a damaged file used to pass the size guards after wrapping)
  static uint64_t element_count(const std::vector<uint64_t>& dm) {
    uint64_t n = 1;
    for (uint64_t v : dm)
      if (__builtin_mul_overflow(n, v, &n) || n > (1ull << 40)) throw FormatError("dataspace too large");
    return n;
  }
  std::string vlen(const uint8_t* raw) const {   // 16-byte descriptor: length, collection address, object index
    uint32_t ln, idx;
    uint64_t coll;
    memcpy(&ln, raw, 4);
    memcpy(&coll, raw + 4, 8);
    memcpy(&idx, raw + 12, 4);
    const uint64_t c = base + coll;
    if (!sig(c, "GCOL")) throw FormatError("bad global heap collection");
    const uint64_t end = c + u64(c + 8);
    for (uint64_t p = c + 16; p + 16 <= end;) {
      const unsigned oi = u16(p);
      const uint64_t osz = u64(p + 8);
      if (oi == 0) break;
      if (osz > end - (p + 16)) throw FormatError("global heap object runs past its collection");   // also keeps p increasing
      if (oi == idx) {
        if (ln > osz) throw FormatError("variable-length string longer than its heap object");
        need(p + 16, ln);
        return std::string(reinterpret_cast<const char*>(&d[p + 16]), ln);
      }
      p += 16 + ((osz + 7) & ~7ull);
    }
    throw FormatError("global heap object not found");
  }
  std::string decode_string(const DType& t, const std::vector<uint8_t>& raw) const {
    if (t.kind == 3) {
      const size_t n = std::min<size_t>(t.size, raw.size());
      if (n == 0) return std::string();            // (memchr on an empty vector's null data() is undefined, however harmless)
      const void* e = memchr(raw.data(), 0, n);
      return std::string(reinterpret_cast<const char*>(raw.data()), e ? (size_t)(reinterpret_cast<const uint8_t*>(e) - raw.data()) : n);
    }
    if (t.kind == 9) {
      if (raw.size() < 16) throw FormatError("short variable-length descriptor");
      return vlen(raw.data());
    }
    throw FormatError("not a string");
  }

  // ---- attributes: value of the string attribute `want`, or "" when absent
  std::string string_attr(uint64_t addr, const char* want) const {
    for (const Msg& m : messages(addr)) {
      if (m.type != 0x0C) continue;
      const int ver = u8(m.off);
      if (ver < 1 || ver > 3) throw FormatError("attribute message version " + std::to_string(ver));
      const unsigned nsz = u16(m.off + 2), tsz = u16(m.off + 4), ssz = u16(m.off + 6);
      uint64_t p = m.off + 8 + (ver == 3 ? 1 : 0);
      auto step = [&](unsigned n) { return ver == 1 ? ((n + 7u) & ~7u) : n; };
      need(p, nsz);
      const std::string name(reinterpret_cast<const char*>(&d[p]), strnlen(reinterpret_cast<const char*>(&d[p]), nsz));
      p += step(nsz);
      const uint64_t toff = p;
      p += step(tsz);
      p += step(ssz);
      if (name != want) continue;
      const DType t = dtype(toff);
      if (t.kind != 3 && t.kind != 9) return std::string();
      const uint64_t left = m.off + m.size - p;
      const uint64_t n = t.kind == 3 ? std::min<uint64_t>(t.size, left) : 16;
      need(p, n);
      return decode_string(t, std::vector<uint8_t>(d.begin() + p, d.begin() + p + n));
    }
    return std::string();
  }

  // ---- datasets
  struct Chunk {
    uint64_t off0;
    uint32_t csize, fmask;
    uint64_t addr;
  };
  void chunks(uint64_t node, int ndim, std::vector<Chunk>& out, uint64_t* budget, int depth = 0) const {
    if (depth > 32) throw FormatError("chunk B-tree too deep");
    visit(budget);
    if (!sig(node, "TREE") || u8(node + 4) != 1) throw FormatError("bad chunk B-tree node");
    const int level = u8(node + 5);
    const unsigned used = u16(node + 6);
    const uint64_t ksz = 8 + 8ull * ndim;
    uint64_t p = node + 24;
    for (unsigned i = 0; i < used; ++i, p += ksz + 8) {
      const uint64_t child = base + u64(p + ksz);
      if (level == 0) {
        visit(budget);                       // a chunk record is 24+ bytes of the file as well
        out.push_back(Chunk{u64(p + 8), u32(p), u32(p + 4), child});
      } else {
        chunks(child, ndim, out, budget, depth + 1);
      }
    }
  }
  // raw bytes of a dataset (+ its datatype and element count)
  std::vector<uint8_t> dataset(uint64_t addr, DType* t_out, uint64_t* count) const {
    uint64_t layout = 0, lsize = 0;
    bool have_dt = false, have_dims = false;
    DType t;
    std::vector<uint64_t> dm;
    std::vector<int> filters;
    for (const Msg& m : messages(addr)) {
      if (m.type == 0x01) dm = dims(m.off), have_dims = true;
      else if (m.type == 0x03) t = dtype(m.off), have_dt = true;
      else if (m.type == 0x08) layout = m.off, lsize = m.size;
      else if (m.type == 0x0B) {
        const int ver = u8(m.off), nf = u8(m.off + 1);
        uint64_t p = m.off + (ver == 1 ? 8 : 2);
        for (int i = 0; i < nf; ++i) {
          const unsigned fid = u16(p);
          unsigned nlen, ncd;
          if (ver == 1 || fid >= 256) nlen = u16(p + 2), ncd = u16(p + 6);
          else nlen = 0, ncd = u16(p + 4);
          if (ver == 1) p += 8 + ((nlen + 7u) & ~7u) + 4ull * ncd + ((ncd % 2) ? 4 : 0);
          else p += (fid >= 256 ? 8 + nlen : 6) + 4ull * ncd;
          filters.push_back((int)fid);
        }
      }
    }
    if (!layout || !have_dt || !have_dims) throw FormatError("not a dataset");
    (void)lsize;
    if (u8(layout) != 3) throw FormatError("data layout message version " + std::to_string(u8(layout)));
    const uint64_t n = element_count(dm);
    uint64_t total;
    if (__builtin_mul_overflow(n, (uint64_t)t.size, &total) || total > (1ull << 34)) throw FormatError("dataset too large");
    const int cls = u8(layout + 1);
    std::vector<uint8_t> raw;
    if (cls == 0) {
      const unsigned sz = u16(layout + 2);
      need(layout + 4, sz);
      raw.assign(d.begin() + layout + 4, d.begin() + layout + 4 + sz);
    } else if (cls == 1) {
      const uint64_t a = u64(layout + 2), sz = u64(layout + 10);
      if (a != UNDEF) {
        need(base + a, sz);
        raw.assign(d.begin() + base + a, d.begin() + base + a + sz);
      }
    } else if (cls == 2) {
      const int ndim = u8(layout + 2);
      const uint64_t btree = u64(layout + 3);
      if (dm.size() != 1 || ndim != 2) throw FormatError("only 1-D chunked datasets are supported");
      const uint64_t cbytes = (uint64_t)u32(layout + 11) * t.size;
      for (int f : filters)
        if (f != 1 && f != 2) throw FormatError("unsupported filter " + std::to_string(f));
      raw.assign(total, 0);
      std::vector<Chunk> cks;
      uint64_t budget = node_budget();
      if (btree != UNDEF) chunks(base + btree, ndim, cks, &budget);
      std::vector<uint8_t> a, b;
      for (const Chunk& c : cks) {
        need(c.addr, c.csize);
        a.assign(d.begin() + c.addr, d.begin() + c.addr + c.csize);
        for (int k = (int)filters.size() - 1; k >= 0; --k) {
          if (c.fmask & (1u << k)) continue;
          if (filters[k] == 1) {   // deflate: a chunk inflates to the chunk size (more only if the file lies)
            if (fast_inflate(a.data(), a.size(), b, cbytes)) {
              a.swap(b);
              continue;
            }
            b.resize(std::max<uint64_t>(cbytes, 64));
            for (;;) {
              uLongf dl = (uLongf)b.size();
              const int rc = uncompress(b.data(), &dl, a.data(), (uLong)a.size());
              if (rc == Z_OK) {
                b.resize(dl);
                break;
              }
              if (rc != Z_BUF_ERROR || b.size() > (1u << 30)) throw FormatError("deflate stream is corrupt");
              b.resize(b.size() * 2);
            }
            a.swap(b);
          } else {                 // byte shuffle
            const size_t ne = a.size() / t.size;
            b.resize(a.size());
            for (size_t e = 0; e < ne; ++e)
              for (uint32_t j = 0; j < t.size; ++j) b[e * t.size + j] = a[j * ne + e];
            for (size_t r = ne * t.size; r < a.size(); ++r) b[r] = a[r];
            a.swap(b);
          }
        }
        uint64_t s;
        if (__builtin_mul_overflow(c.off0, (uint64_t)t.size, &s) || s >= total) continue;
        const uint64_t room = std::min<uint64_t>(cbytes, total - s);     // never s + cbytes: that sum can wrap
        memcpy(raw.data() + s, a.data(), std::min<uint64_t>(room, a.size()));
      }
    } else {
      throw FormatError("layout class " + std::to_string(cls));
    }
    if ((t.kind == 0 || t.kind == 1) && raw.size() < total) throw FormatError("dataset shorter than its dataspace");
    *t_out = t;
    *count = n;
    return raw;
  }
  int64_t dataset_count(uint64_t addr) const {   // element count without reading the data
    for (const Msg& m : messages(addr))
      if (m.type == 0x01) {
        return (int64_t)element_count(dims(m.off));
      }
    throw FormatError("not a dataset");
  }
  std::string string_dataset(uint64_t addr) const {
    DType t;
    uint64_t n;
    const std::vector<uint8_t> raw = dataset(addr, &t, &n);
    return decode_string(t, raw);
  }
};

}  // namespace

struct chiron_fast5 {
  H5 h5;
  std::vector<ReadRec> reads;
};

namespace {

ReadRec read_record(const H5& h5, uint64_t raw_group, uint64_t analyses_root, const std::string& suffix) {
  ReadRec r;
  r.suffix = suffix;
  if (!h5.resolve("Signal", raw_group, &r.signal_addr)) throw FormatError("no Signal dataset");
  r.n_samples = h5.dataset_count(r.signal_addr);
  r.read_id = h5.string_attr(raw_group, "read_id");
  static const char* paths[2] = {"Analyses/Basecall_1D_000/BaseCalled_template/Fastq", "Analyses/Alignment_000/Aligned_template/Fasta"};
  for (const char* p : paths) {
    uint64_t a;
    try {
      if (h5.resolve(p, analyses_root, &a)) {
        r.fastq = h5.string_dataset(a);
        break;
      }
    } catch (const FormatError&) {
    }
  }
  return r;
}

template <typename F>
chiron_status guarded(F&& f) {
  try {
    return f();
  } catch (const FormatError& e) {
    return chiron::set_error(CHIRON_ERR_INVALID, "fast5: %s", e.what());
  } catch (const std::bad_alloc&) {
    return chiron::set_error(CHIRON_ERR_INVALID, "fast5: out of memory");
  } catch (const std::exception& e) {
    return chiron::set_error(CHIRON_ERR_INVALID, "fast5: %s", e.what());
  }
}

void copy_text(const std::string& s, char* out, size_t cap) {
  if (!out || cap == 0) return;
  const size_t n = std::min(s.size(), cap - 1);
  memcpy(out, s.data(), n);
  out[n] = 0;
}

}  // namespace

extern "C" chiron_status chiron_fast5_open(const char* path, chiron_fast5** out) {
  if (!path || !out) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_fast5_open: null argument");
  *out = nullptr;
  return guarded([&]() -> chiron_status {
    chiron_fast5* f = new chiron_fast5();
    try {
      f->h5.open(path);
      const H5& h5 = f->h5;
      const auto top = h5.links(h5.root);
      if (top.count("Raw")) {
        // single-read file (extract_file, extract_sig_ref.py:149-175): the first group under /Raw/Reads
        uint64_t rr;
        if (!h5.resolve("Raw/Reads", h5.root, &rr)) throw FormatError("no /Raw/Reads");
        const auto reads = h5.links(rr);
        if (reads.empty()) throw FormatError("no read under /Raw/Reads");
        f->reads.push_back(read_record(h5, reads.begin()->second, h5.root, ""));   // std::map iterates in sorted order
      } else {
        // multi-read file (extract_file_v2, :178-193): one record per top-level read group, sorted by name
        for (const auto& kv : top) {
          const auto sub = h5.links(kv.second);
          const auto it = sub.find("Raw");
          if (it == sub.end()) continue;
          f->reads.push_back(read_record(h5, it->second, kv.second, kv.first));
        }
      }
    } catch (...) {
      delete f;
      throw;
    }
    *out = f;
    return CHIRON_OK;
  });
}

extern "C" void chiron_fast5_close(chiron_fast5* f) { delete f; }

extern "C" int32_t chiron_fast5_read_count(const chiron_fast5* f) { return f ? (int32_t)f->reads.size() : 0; }

extern "C" chiron_status chiron_fast5_read_info(const chiron_fast5* f, int32_t i, char* suffix, size_t suffix_cap, char* read_id, size_t id_cap,
                                                int64_t* n_samples, int64_t* fastq_len) {
  if (!f || i < 0 || i >= (int32_t)f->reads.size()) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_fast5_read_info: no such read");
  const ReadRec& r = f->reads[i];
  copy_text(r.suffix, suffix, suffix_cap);
  copy_text(r.read_id, read_id, id_cap);
  if (n_samples) *n_samples = r.n_samples;
  if (fastq_len) *fastq_len = (int64_t)r.fastq.size();
  return CHIRON_OK;
}

extern "C" chiron_status chiron_fast5_fastq(const chiron_fast5* f, int32_t i, char* out, int64_t cap) {
  if (!f || i < 0 || i >= (int32_t)f->reads.size() || !out || cap < 1) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_fast5_fastq: bad argument");
  if ((int64_t)f->reads[i].fastq.size() + 1 > cap) return chiron::set_error(CHIRON_ERR_OVERFLOW, "chiron_fast5_fastq: %zu bytes needed", f->reads[i].fastq.size() + 1);
  copy_text(f->reads[i].fastq, out, (size_t)cap);
  return CHIRON_OK;
}

extern "C" chiron_status chiron_fast5_signal(const chiron_fast5* f, int32_t i, float* out, int64_t cap, int32_t reverse) {
  if (!f || i < 0 || i >= (int32_t)f->reads.size() || !out) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_fast5_signal: bad argument");
  return guarded([&]() -> chiron_status {
    DType t;
    uint64_t n = 0;
    const std::vector<uint8_t> raw = f->h5.dataset(f->reads[i].signal_addr, &t, &n);
    if (cap < 0 || n > (uint64_t)cap) return chiron::set_error(CHIRON_ERR_OVERFLOW, "chiron_fast5_signal: %llu samples, capacity %lld", (unsigned long long)n, (long long)cap);
    if (t.kind != 0 && t.kind != 1) throw FormatError("the Signal dataset is not numeric");
    if (raw.size() / t.size < n) throw FormatError("dataset shorter than its dataspace");
    auto put = [&](uint64_t k, float v) { out[reverse ? n - 1 - k : k] = v; };
    const uint8_t* p = raw.data();
    if (t.kind == 0) {
      for (uint64_t k = 0; k < n; ++k) {
        float v;
        switch (t.size) {
          case 1: v = t.sign ? (float)(int8_t)p[k] : (float)p[k]; break;
          case 2: { uint16_t x; memcpy(&x, p + 2 * k, 2); v = t.sign ? (float)(int16_t)x : (float)x; break; }
          case 4: { uint32_t x; memcpy(&x, p + 4 * k, 4); v = t.sign ? (float)(int32_t)x : (float)x; break; }
          case 8: { uint64_t x; memcpy(&x, p + 8 * k, 8); v = t.sign ? (float)(int64_t)x : (float)x; break; }
          default: throw FormatError("integer size " + std::to_string(t.size));
        }
        put(k, v);
      }
    } else if (t.kind == 1 && t.size == 4) {
      for (uint64_t k = 0; k < n; ++k) { float x; memcpy(&x, p + 4 * k, 4); put(k, x); }
    } else if (t.kind == 1 && t.size == 8) {
      for (uint64_t k = 0; k < n; ++k) { double x; memcpy(&x, p + 8 * k, 8); put(k, (float)x); }
    } else {
      throw FormatError("the Signal dataset is not numeric");
    }
    return CHIRON_OK;
  });
}

// extract_sig_ref.py:122-123: f.write(delimiter.join(str(v) for v in raw_signal)) for the integer DAC values `chiron call`
// extracts (unit = False, entry.py:36): decimal integers, no trailing delimiter.  A value that is not an integer in
// float32 (never the case for int16 DAC counts) is refused -- Python's repr of a float is not reproduced here.
extern "C" chiron_status chiron_write_signal_text(const char* path, const float* v, int64_t n, const char* delimiter) {
  if (!path || (!v && n > 0) || n < 0) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_write_signal_text: bad argument");
  const char* dl = delimiter ? delimiter : "\n";
  const size_t dn = strlen(dl);
  // 100 000 samples per read and one such file per read: this formatter is on the host's critical path behind the fp16 engine
  // (pointer writes into one buffer, two digits per division by 100)
  static const char pairs[] =
      "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869"
      "707172737475767778798081828384858687888990919293949596979899";
  std::string buf;
  buf.resize((size_t)n * (20 + dn) + 16);
  char* out = &buf[0];
  for (int64_t k = 0; k < n; ++k) {
    const float x = v[k];
    long long iv;
    if (x > -2.0e9f && x < 2.0e9f) {
      const int i32 = (int)x;              // every real signal
      if ((float)i32 != x) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_write_signal_text: sample %lld (%g) is not an integer", (long long)k, (double)x);
      iv = i32;
    } else {
      iv = (long long)x;
      if ((float)iv != x || x > 9.0e15f || x < -9.0e15f || x != x) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_write_signal_text: sample %lld (%g) is not an integer", (long long)k, (double)x);
    }
    if (k) {
      if (dn == 1) {
        *out++ = dl[0];
      } else {
        memcpy(out, dl, dn);
        out += dn;
      }
    }
    unsigned long long a = iv < 0 ? (unsigned long long)(-iv) : (unsigned long long)iv;
    if (iv < 0) *out++ = '-';
    char tmp[24];
    int len = 0;
    if (a < 4000000000ull) {          // every real signal: 32-bit arithmetic
      unsigned u = (unsigned)a;
      while (u >= 100) {
        const unsigned r = u % 100;
        u /= 100;
        tmp[len++] = pairs[2 * r + 1];
        tmp[len++] = pairs[2 * r];
      }
      if (u >= 10) {
        tmp[len++] = pairs[2 * u + 1];
        tmp[len++] = pairs[2 * u];
      } else {
        tmp[len++] = (char)('0' + u);
      }
    } else {
      do { tmp[len++] = (char)('0' + a % 10); a /= 10; } while (a);
    }
    while (len) *out++ = tmp[--len];
  }
  buf.resize((size_t)(out - &buf[0]));
  FILE* fo = fopen(path, "wb");
  if (!fo) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_write_signal_text: cannot open %s", path);
  const size_t w = buf.empty() ? 0 : fwrite(buf.data(), 1, buf.size(), fo);
  const int rc = fclose(fo);
  if (w != buf.size() || rc != 0) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_write_signal_text: short write to %s", path);
  return CHIRON_OK;
}
