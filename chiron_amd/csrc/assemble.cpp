// Overlap-consensus vote (host C++): chiron/utils/easy_assembler.py glue_kernal :276-294, stick_kernal :296-300,
// simple_assembly_kernal :212-250 (difflib matching blocks + Poisson-like offset prior), simple_assembly(_qs)
// :302-335 / :393-432, add_count(_qs) :381-387 / :435-442.
// Reproduces the reference's quirks deliberately: the consensus length only accounts for segments 1..n-1 (the
// `continue` for segment 0 skips the length update), so a single-segment read yields an empty consensus; ties in the
// glue score keep the first (smallest) overlap; a negative start clips the segment's head.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/chiron_amd.h"

namespace chiron {
chiron_status set_error(chiron_status st, const char* fmt, ...);
}

namespace {

int64_t glue_disp(const uint8_t* cur, int64_t n, const uint8_t* prev, int64_t prev_n) {
  // max_overlap = min(math.floor(0.1 * prev_n), n)   -- same IEEE double product as the reference
  int64_t max_overlap = (int64_t)std::floor(0.1 * (double)prev_n);
  if (n < max_overlap) max_overlap = n;
  int64_t best_i = 0, best_score = 0;
  for (int64_t i = 1; i < max_overlap; ++i) {
    int64_t same = 0;
    const uint8_t* tail = prev + (prev_n - i);
    for (int64_t j = 0; j < i; ++j) same += (cur[j] == tail[j]);
    const int64_t score = 2 * same - i;
    if (score > best_score) {
      best_score = score;
      best_i = i;
    }
  }
  return prev_n - best_i;
}

// ---- difflib.SequenceMatcher(None, a, b).get_matching_blocks() for byte sequences (Ratcliff-Obershelp as CPython
// implements it: longest match in the whole rectangle first, then the pieces left and right of it).  Details that
// change results and are therefore kept: `b2j` lists the positions of every element of b EXCEPT "popular" ones when
// len(b) >= 200 (autojunk: an element occurring more than len(b)//100 + 1 times is dropped from the index, though it
// still extends a match); the longest-match scan keeps the FIRST best (lowest i, then lowest j); the matched block is
// then extended left and right over equal elements.
struct Block {
  int64_t i, j, k;
};

class Matcher {
 public:
  Matcher(const uint8_t* a, int64_t la, const uint8_t* b, int64_t lb) : a_(a), b_(b), la_(la), lb_(lb) {
    for (int64_t j = 0; j < lb; ++j) b2j_[b[j]].push_back(j);
    if (lb >= 200) {
      const int64_t ntest = lb / 100 + 1;
      for (auto& idx : b2j_)
        if ((int64_t)idx.size() > ntest) idx.clear();
    }
    len_cur_.assign(lb + 1, 0);
    len_new_.assign(lb + 1, 0);
  }

  Block longest(int64_t alo, int64_t ahi, int64_t blo, int64_t bhi) {
    int64_t besti = alo, bestj = blo, bestsize = 0;
    // j2len[j] = length of the longest match ending with a[i-1], b[j]; stored at index j + 1 so that j - 1 = -1 is slot 0
    touched_cur_.clear();
    for (int64_t i = alo; i < ahi; ++i) {
      touched_new_.clear();
      for (int64_t j : b2j_[a_[i]]) {
        if (j < blo) continue;
        if (j >= bhi) break;
        const int64_t k = len_cur_[j] + 1;   // len_cur_[j] is j2len[j - 1]
        len_new_[j + 1] = k;
        touched_new_.push_back(j + 1);
        if (k > bestsize) {
          besti = i - k + 1;
          bestj = j - k + 1;
          bestsize = k;
        }
      }
      for (int64_t t : touched_cur_) len_cur_[t] = 0;
      len_cur_.swap(len_new_);
      touched_cur_.swap(touched_new_);
    }
    for (int64_t t : touched_cur_) len_cur_[t] = 0;
    while (besti > alo && bestj > blo && a_[besti - 1] == b_[bestj - 1]) {
      --besti;
      --bestj;
      ++bestsize;
    }
    while (besti + bestsize < ahi && bestj + bestsize < bhi && a_[besti + bestsize] == b_[bestj + bestsize]) ++bestsize;
    return Block{besti, bestj, bestsize};
  }

  // sorted by (i, j); the terminating (la, lb, 0) entry of difflib is appended by the caller
  std::vector<Block> blocks() {
    struct Rect {
      int64_t alo, ahi, blo, bhi;
    };
    std::vector<Rect> todo{{0, la_, 0, lb_}};
    std::vector<Block> out;
    while (!todo.empty()) {
      const Rect r = todo.back();
      todo.pop_back();
      const Block m = longest(r.alo, r.ahi, r.blo, r.bhi);
      if (m.k == 0) continue;
      out.push_back(m);
      if (r.alo < m.i && r.blo < m.j) todo.push_back({r.alo, m.i, r.blo, m.j});
      if (m.i + m.k < r.ahi && m.j + m.k < r.bhi) todo.push_back({m.i + m.k, r.ahi, m.j + m.k, r.bhi});
    }
    std::sort(out.begin(), out.end(), [](const Block& x, const Block& y) { return x.i != y.i ? x.i < y.i : (x.j != y.j ? x.j < y.j : x.k < y.k); });
    return out;
  }

 private:
  const uint8_t *a_, *b_;
  int64_t la_, lb_;
  std::vector<int64_t> b2j_[256];
  std::vector<int64_t> len_cur_, len_new_, touched_cur_, touched_new_;
};

// easy_assembler.py:212-250.  For every diagonal offset (position in prev minus position in cur) that carries a
// matching block -- plus the offset lb - la of difflib's terminating empty block -- the score is
//   |off| * ln(rate) - ln(|off|!) + matched * ln(p_same / 0.25) + 0 * ln(p_diff / 0.25),
// rate = N * jump_step_ratio for off >= 0 and back_ratio * N * jump_step_ratio for off < 0 (N = len(cur)); the
// displacement is the first offset (in order of first appearance among the sorted blocks) with the largest score.
// Constants exactly as the reference writes them.
int64_t simple_disp(const uint8_t* cur, int64_t n, const uint8_t* prev, int64_t prev_n, double error_rate, double jump_step_ratio,
                    double* best_log_px) {
  const double back_ratio = 6.5 * 10e-4;
  const double p_same = 1 - 2 * error_rate + 26.0 / 25 * std::pow(error_rate, 2.0);
  const double p_diff = 1 - p_same;
  const double l_same = std::log(p_same / 0.25), l_diff = std::log(p_diff / 0.25);
  Matcher m(cur, n, prev, prev_n);
  std::vector<Block> bl = m.blocks();
  bl.push_back(Block{n, prev_n, 0});
  std::vector<int64_t> offs;      // insertion order of the reference's dicts
  std::vector<int64_t> matched;
  for (const Block& b : bl) {
    const int64_t off = b.j - b.i;
    size_t q = 0;
    while (q < offs.size() && offs[q] != off) ++q;
    if (q == offs.size()) {
      offs.push_back(off);
      matched.push_back(0);
    }
    matched[q] += b.k;
  }
  const double l_fwd = std::log((double)n * jump_step_ratio);
  const double l_back = std::log(back_ratio * (double)n * jump_step_ratio);
  int64_t best = 0;
  double best_v = 0;
  for (size_t q = 0; q < offs.size(); ++q) {
    const int64_t off = offs[q];
    const int64_t k = off < 0 ? -off : off;
    double lfact = 0;   // sum(log(x + 1) for x in range(k)), summed in that order
    for (int64_t x = 0; x < k; ++x) lfact += std::log((double)(x + 1));
    double v = (double)k * (off < 0 ? l_back : l_fwd) - lfact;
    v += (double)matched[q] * l_same;
    v += 0.0 * l_diff;
    if (q == 0 || v > best_v) {
      best_v = v;
      best = off;
    }
  }
  if (best_log_px) *best_log_px = best_v;
  return best;
}

int64_t displacement(int32_t kernal, const uint8_t* cur, int64_t n, const uint8_t* prev, int64_t pn, double error_rate,
                     double jump_step_ratio, double* log_px) {
  if (log_px) *log_px = 0;
  if (kernal == CHIRON_KERNAL_GLUE) return glue_disp(cur, n, prev, pn);
  if (kernal == CHIRON_KERNAL_STICK) return pn;
  return simple_disp(cur, n, prev, pn, error_rate, jump_step_ratio, log_px);
}

bool known_kernal(int32_t k) { return k == CHIRON_KERNAL_GLUE || k == CHIRON_KERNAL_STICK || k == CHIRON_KERNAL_SIMPLE; }

// pass 1 of the vote: where every segment starts (running position; may be negative with the simple kernel) -> length
int64_t segment_starts(const uint8_t* bases, const int64_t* seg_off, int64_t n_seg, int32_t kernal, double error_rate, double jump_step_ratio,
                       std::vector<int64_t>& start) {
  start.assign((size_t)std::max<int64_t>(n_seg, 0), 0);
  int64_t pos = 0, length = 0;
  for (int64_t s = 1; s < n_seg; ++s) {
    const int64_t n = seg_off[s + 1] - seg_off[s], pn = seg_off[s] - seg_off[s - 1];
    pos += displacement(kernal, bases + seg_off[s], n, bases + seg_off[s - 1], pn, error_rate, jump_step_ratio, nullptr);
    start[(size_t)s] = pos;
    length = std::max(length, pos + n);
  }
  return length;
}
void cast_votes(const uint8_t* bases, const int64_t* seg_off, int64_t n_seg, const double* seg_qs, const std::vector<int64_t>& start, int64_t length,
                double* counts, double* qs_sum, int64_t cap);

}  // namespace

extern "C" chiron_status chiron_overlap_displacement(const uint8_t* cur, int64_t n, const uint8_t* prev, int64_t prev_n, int32_t kernal,
                                                     double error_rate, double jump_step_ratio, int64_t* disp, double* log_px) {
  if (!disp || n < 0 || prev_n < 0 || (n > 0 && !cur) || (prev_n > 0 && !prev))
    return chiron::set_error(CHIRON_ERR_INVALID, "chiron_overlap_displacement: bad arguments");
  if (!known_kernal(kernal)) return chiron::set_error(CHIRON_ERR_INVALID, "assembly kernal %d (1 = glue, 2 = stick, 3 = simple)", kernal);
  *disp = displacement(kernal, cur, n, prev, prev_n, error_rate, jump_step_ratio, log_px);
  return CHIRON_OK;
}

extern "C" chiron_status chiron_assemble(const uint8_t* bases, const int64_t* seg_off, int64_t n_seg, const double* seg_qs, int32_t kernal,
                                         double error_rate, double jump_step_ratio, double* counts, double* qs_sum, int64_t cap,
                                         int64_t* out_len) {
  if (!seg_off || !out_len || n_seg < 0) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_assemble: bad arguments");
  if (!known_kernal(kernal)) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_assemble: kernal %d (1 = glue, 2 = stick, 3 = simple)", kernal);
  std::vector<int64_t> start;
  const int64_t length = segment_starts(bases, seg_off, n_seg, kernal, error_rate, jump_step_ratio, start);
  *out_len = length;
  if (length > cap) return chiron::set_error(CHIRON_ERR_OVERFLOW, "chiron_assemble: consensus needs %lld columns, capacity %lld", (long long)length, (long long)cap);
  if (length == 0) return CHIRON_OK;
  if (!counts) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_assemble: null counts");
  for (int r = 0; r < 4; ++r) {
    memset(counts + r * cap, 0, sizeof(double) * length);
    if (qs_sum) memset(qs_sum + r * cap, 0, sizeof(double) * length);
  }
  cast_votes(bases, seg_off, n_seg, seg_qs, start, length, counts, qs_sum, cap);
  return CHIRON_OK;
}

namespace {
// pass 2 of the vote.  A segment starting left of column 0 loses its head (add_count); columns beyond `length` are
// dropped exactly like concensus[:, :length] (only segment 0 can reach past it)
void cast_votes(const uint8_t* bases, const int64_t* seg_off, int64_t n_seg, const double* seg_qs, const std::vector<int64_t>& start, int64_t length,
                double* counts, double* qs_sum, int64_t cap) {
  for (int64_t s = 0; s < n_seg; ++s) {
    const int64_t n = seg_off[s + 1] - seg_off[s];
    const uint8_t* seg = bases + seg_off[s];
    const double q = (seg_qs && qs_sum) ? seg_qs[s] : 0.0;
    const int64_t st = start[(size_t)s];
    for (int64_t j = st < 0 ? -st : 0; j < n; ++j) {
      const int64_t colx = st + j;
      if (colx >= length) break;
      const int b = seg[j] & 3;
      counts[b * cap + colx] += 1.0;
      if (qs_sum && seg_qs) qs_sum[b * cap + colx] += q;
    }
  }
}
}  // namespace

// ---- one read from decoded windows to its files, without the interpreter (chiron_eval.py:446-462 + write_output
// :176-228): index2base of every window, the vote above, np.argmax (first maximum, :457), qs (:152-174; the same
// decisions as chiron_amd/eval.py qs(): the quality belongs to the LAST of equal top counts, a column nobody voted for
// scores 0), and the result / segments files in the reference's formats.  ctypes releases the GIL for the whole call, so
// the finishing threads of `chiron call` run in parallel; behind the fp16 engine they were what bounded the pipeline.
namespace {
bool write_file(const char* path, const std::string& text) {
  FILE* f = fopen(path, "wb");
  if (!f) return false;
  const size_t w = text.empty() ? 0 : fwrite(text.data(), 1, text.size(), f);
  return (fclose(f) == 0) && w == text.size();
}
}  // namespace

extern "C" chiron_status chiron_finish_read(const uint8_t* bases, const int64_t* seg_off, int64_t n_seg, const double* seg_qs, int32_t kernal,
                                            double error_rate, double jump_step_ratio, const char* name, const char* result_path,
                                            const char* segments_path, int32_t fastq, int32_t rna, char* consensus_out, int64_t consensus_cap,
                                            int64_t* consensus_len) {
  if (!seg_off || n_seg < 0 || !name || !result_path || !consensus_len || (n_seg > 0 && seg_off[n_seg] > 0 && !bases))
    return chiron::set_error(CHIRON_ERR_INVALID, "chiron_finish_read: bad arguments");
  if (!known_kernal(kernal)) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_finish_read: kernal %d (1 = glue, 2 = stick, 3 = simple)", kernal);
  static const char ACGT[4] = {'A', 'C', 'G', 'T'};
  const bool want_q = fastq != 0 && seg_qs != nullptr;
  std::vector<int64_t> start;
  const int64_t length = segment_starts(bases, seg_off, n_seg, kernal, error_rate, jump_step_ratio, start);
  // the vote of cast_votes() with the four bases of a column next to each other ([column][base]: one cache line serves
  // four columns; the [base][column] matrices of chiron_assemble are what numpy wants, not what this loop wants)
  std::vector<float> counts;      // vote counts are small integers: exact in float
  std::vector<double> qsum;
  if (length > 0) {
    counts.assign((size_t)4 * length, 0.f);
    if (want_q) qsum.assign((size_t)4 * length, 0.0);
    for (int64_t s = 0; s < n_seg; ++s) {
      const int64_t n = seg_off[s + 1] - seg_off[s];
      const uint8_t* seg = bases + seg_off[s];
      const double q = want_q ? seg_qs[s] : 0.0;
      const int64_t st = start[(size_t)s];
      for (int64_t j = st < 0 ? -st : 0; j < n; ++j) {
        const int64_t colx = st + j;
        if (colx >= length) break;
        const size_t at = (size_t)colx * 4 + (seg[j] & 3);
        counts[at] += 1.f;
        if (want_q) qsum[at] += q;      // added in segment order, like add_count_qs
      }
    }
  }
  std::string seq((size_t)length, 'A'), qual;
  if (want_q) qual.assign((size_t)length, '!');
  const double ln10 = std::log(10.0);
  constexpr int QTAB = 64;
  struct QTable {
    double v[QTAB][QTAB];
    QTable() {
      for (int a = 0; a < QTAB; ++a)
        for (int b = 0; b < QTAB; ++b) v[a][b] = 10.0 * std::log10(((double)a + 1.0) / ((double)b + 1.0));
    }
  };
  static const QTable qtab;
  for (int64_t c = 0; c < length; ++c) {
    const float* cc = &counts[(size_t)c * 4];
    double n1 = cc[0];
    int arg = 0, top = 0;
    for (int b = 1; b < 4; ++b)
      if (cc[b] > n1) n1 = cc[b], arg = b;   // np.argmax: first maximum
    seq[(size_t)c] = ACGT[arg];
    if (!want_q) continue;
    for (int b = 0; b < 4; ++b)
      if (cc[b] == n1) top = b;              // last of the maxima (stable argsort)
    double n2 = -1.0;
    for (int b = 0; b < 4; ++b)
      if (b != top && cc[b] > n2) n2 = cc[b];
    long long q = 0;
    if (n1 > 0) {
      // vote counts are small integers: 10 log10((n1 + 1) / (n2 + 1)) comes from a table of the same expression
      const double ratio_db = (n1 < QTAB && n2 >= 0 && n2 < QTAB) ? qtab.v[(int)n1][(int)n2] : 10.0 * std::log10((n1 + 1.0) / (n2 + 1.0));
      q = (long long)(ratio_db + qsum[(size_t)c * 4 + top] / n1 / ln10);
    }
    const long long code = q + 33;
    if (code < 0 || code > 127)
      return chiron::set_error(CHIRON_ERR_INVALID, "chiron_finish_read: quality code %lld of column %lld is not one ASCII byte", code, (long long)c);
    qual[(size_t)c] = (char)code;
  }
  // result/<name>.fastq = @name / sequence / + / quality, each line terminated; .fasta = >name / sequence, no final newline
  std::string text;
  text.reserve((size_t)2 * length + 64);
  text.append(want_q ? "@" : ">").append(name).append("\n");
  const size_t seq_at = text.size();
  text.append(seq);
  if (rna)   // chiron_eval.py:204-205: the written consensus only (the caller's string keeps T, as finish_read returns it)
    for (size_t i = seq_at; i < text.size(); ++i)
      if (text[i] == 'T') text[i] = 'U';
  if (want_q) text.append("\n+\n").append(qual).append("\n");
  if (!write_file(result_path, text)) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_finish_read: cannot write %s", result_path);
  if (segments_path) {
    // segments/<name>.<ext>: one FASTA-style record per window, named <name><window index> (never T -> U: the reference
    // converts the consensus only, chiron_eval.py:204-205)
    std::string segs;
    segs.reserve((size_t)(seg_off[n_seg] - seg_off[0]) + (size_t)n_seg * (strlen(name) + 12));
    char num[24];
    for (int64_t s = 0; s < n_seg; ++s) {
      segs.append(">").append(name);
      snprintf(num, sizeof(num), "%lld", (long long)s);
      segs.append(num).append("\n");
      for (int64_t j = seg_off[s]; j < seg_off[s + 1]; ++j) segs.push_back(ACGT[bases[j] & 3]);
      segs.push_back('\n');
    }
    if (!write_file(segments_path, segs)) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_finish_read: cannot write %s", segments_path);
  }
  *consensus_len = length;
  if (consensus_out) {
    if (length > consensus_cap) return chiron::set_error(CHIRON_ERR_OVERFLOW, "chiron_finish_read: consensus of %lld bases, capacity %lld", (long long)length, (long long)consensus_cap);
    memcpy(consensus_out, seq.data(), (size_t)length);
  }
  return CHIRON_OK;
}

// chiron_input.py:527-539 read_signal(): whitespace separated numbers -> float32 (through double, as Python does)
extern "C" chiron_status chiron_parse_signal_text(const char* text, size_t len, float* out, size_t cap, size_t* n_out) {
  if (!text || !out || !n_out) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_parse_signal_text: null argument");
  const char* p = text;
  const char* end = text + len;
  size_t n = 0;
  auto is_ws = [](char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; };
  while (true) {
    while (p < end && is_ws(*p)) ++p;
    if (p >= end) break;
    const char* tok = p;
    // fast path: plain (signed) decimal integers, which is what extract_sig_ref writes
    bool neg = false;
    if (*p == '-' || *p == '+') {
      neg = *p == '-';
      ++p;
    }
    long long v = 0;
    int digits = 0;
    while (p < end && *p >= '0' && *p <= '9' && digits < 15) {
      v = v * 10 + (*p - '0');
      ++p;
      ++digits;
    }
    double d;
    if (digits > 0 && (p == end || is_ws(*p))) {
      d = neg ? -(double)v : (double)v;
    } else {
      // general number: copy the token (strtod needs a terminator) and let the C library decide
      const char* q = tok;
      while (q < end && !is_ws(*q)) ++q;
      char buf[64];
      const size_t tl = (size_t)(q - tok);
      if (tl >= sizeof(buf)) return chiron::set_error(CHIRON_ERR_INVALID, "could not convert string to float: token of %zu characters", tl);
      memcpy(buf, tok, tl);
      buf[tl] = 0;
      char* stop = nullptr;
      // strtod also reads C99 hexadecimal floats ("0x1p3"), which Python's float() / np.float32() refuse
      const bool hex = memchr(buf, 'x', tl) || memchr(buf, 'X', tl);
      d = strtod(buf, &stop);
      if (hex || stop != buf + tl || tl == 0) return chiron::set_error(CHIRON_ERR_INVALID, "could not convert string to float: '%s'", buf);
      p = q;
    }
    if (n >= cap) return chiron::set_error(CHIRON_ERR_OVERFLOW, "chiron_parse_signal_text: more than %zu values", cap);
    out[n++] = (float)d;
  }
  *n_out = n;
  return CHIRON_OK;
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8: what TF's tensor bundle stores per tensor
// (BundleEntryProto.crc32c, masked; tensor_bundle.cc checks it on restore, chiron_amd/tf_bundle.py does the same).
extern "C" chiron_status chiron_crc32c(const void* data, size_t len, uint32_t* out) {
  if ((!data && len) || !out) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_crc32c: null argument");
  // built once: C++11 initialises a function-local static exactly once, with the stores ordered before any reader
  // (ctypes releases the GIL, so first calls do race)
  struct Table {
    uint32_t t[8][256];
    Table() {
      for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
        t[0][i] = c;
      }
      for (int k = 1; k < 8; ++k)
        for (uint32_t i = 0; i < 256; ++i) t[k][i] = t[0][t[k - 1][i] & 0xFF] ^ (t[k - 1][i] >> 8);
    }
  };
  static const Table tab;
  const uint32_t (*table)[256] = tab.t;
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t crc = 0xFFFFFFFFu;
  while (len >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= crc;
    crc = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
          table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    p += 8;
    len -= 8;
  }
  while (len--) crc = table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  *out = crc ^ 0xFFFFFFFFu;
  return CHIRON_OK;
}
