// Overlap-consensus vote (host C++): chiron/utils/easy_assembler.py glue_kernal :276-294,
// stick_kernal :296-300, simple_assembly(_qs) :302-335 / :393-432, add_count(_qs) :381-387 / :435-442.
// Reproduces the reference's quirks deliberately: the consensus length only accounts for segments
// 1..n-1 (the `continue` for segment 0 skips the length update), so a single-segment read yields an
// empty consensus; ties in the glue score keep the first (smallest) overlap.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/chiron_amd.h"

namespace chiron {
chiron_status set_error(chiron_status st, const char* fmt, ...);
}

static int64_t glue_disp(const uint8_t* cur, int64_t n, const uint8_t* prev, int64_t prev_n) {
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

extern "C" chiron_status chiron_assemble(const uint8_t* bases, const int64_t* seg_off, int64_t n_seg,
                                         const double* seg_qs, int32_t kernal, double* counts, double* qs_sum,
                                         int64_t cap, int64_t* out_len) {
  if (!seg_off || !out_len || n_seg < 0) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_assemble: bad arguments");
  if (kernal != CHIRON_KERNAL_GLUE && kernal != CHIRON_KERNAL_STICK)
    return chiron::set_error(CHIRON_ERR_INVALID, "chiron_assemble: kernal %d (1=glue, 2=stick)", kernal);
  // pass 1: displacements -> length
  int64_t pos = 0, length = 0, extent = 0;
  for (int64_t s = 0; s < n_seg; ++s) {
    const int64_t n = seg_off[s + 1] - seg_off[s];
    if (s > 0) {
      const int64_t pn = seg_off[s] - seg_off[s - 1];
      const int64_t disp = kernal == CHIRON_KERNAL_GLUE ? glue_disp(bases + seg_off[s], n, bases + seg_off[s - 1], pn) : pn;
      pos += disp;
      if (pos + n > length) length = pos + n;
    }
    if (pos + n > extent) extent = pos + n;
  }
  *out_len = length;
  if (length > cap) return chiron::set_error(CHIRON_ERR_OVERFLOW, "chiron_assemble: consensus needs %lld columns, capacity %lld", (long long)length, (long long)cap);
  if (length == 0) return CHIRON_OK;
  if (!counts) return chiron::set_error(CHIRON_ERR_INVALID, "chiron_assemble: null counts");
  for (int r = 0; r < 4; ++r) {
    memset(counts + r * cap, 0, sizeof(double) * length);
    if (qs_sum) memset(qs_sum + r * cap, 0, sizeof(double) * length);
  }
  // pass 2: votes (columns beyond `length` are dropped exactly like concensus[:, :length])
  pos = 0;
  for (int64_t s = 0; s < n_seg; ++s) {
    const int64_t n = seg_off[s + 1] - seg_off[s];
    if (s > 0) {
      const int64_t pn = seg_off[s] - seg_off[s - 1];
      pos += kernal == CHIRON_KERNAL_GLUE ? glue_disp(bases + seg_off[s], n, bases + seg_off[s - 1], pn) : pn;
    }
    const uint8_t* seg = bases + seg_off[s];
    const double q = (seg_qs && qs_sum) ? seg_qs[s] : 0.0;
    for (int64_t j = 0; j < n; ++j) {
      const int64_t colx = pos + j;
      if (colx >= length) break;
      const int b = seg[j] & 3;
      counts[b * cap + colx] += 1.0;
      if (qs_sum && seg_qs) qs_sum[b * cap + colx] += q;
    }
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
      d = strtod(buf, &stop);
      if (stop != buf + tl || tl == 0) return chiron::set_error(CHIRON_ERR_INVALID, "could not convert string to float: '%s'", buf);
      p = q;
    }
    if (n >= cap) return chiron::set_error(CHIRON_ERR_OVERFLOW, "chiron_parse_signal_text: more than %zu values", cap);
    out[n++] = (float)d;
  }
  *n_out = n;
  return CHIRON_OK;
}
