// chiron_pipeline_run: the host side of `chiron call` on its direct fast5 path as ONE native call -- no interpreter, no GIL.
//
// The reference overlaps reading, inference and decoding with TF queue runners and six decode threads (chiron_eval.py:304-368,
// :378-463); chiron_amd/eval.py:evaluation restates that with two Python thread pools around native calls (fast5 decode, text
// writer, submit_pieces, finish_read), and what binds it at eight ranks behind the fp16 engines is the interpreter lock INSIDE each
// rank's pools (round 5: profiles/r05_host_ceiling_*).  This file is the same pipeline with the glue in C++ too:
//
//   reader threads   fast5 -> samples (chiron_fast5_*: extract_sig_ref.py:92-147), raw/<name>.signal and reference/<stem>_ref.fastq
//                    as the reference's extraction writes them, one zero-padded signal buffer per read whose rows at stride `jump`
//                    ARE the windows (chiron_input.py:276-286)
//   the caller's     files consumed IN ORDER (results never depend on timing): cross-read packing into batches of batch_size rows
//   thread           (chiron_eval.py:321-334), half-even seq_len (:337), submit_pieces / collect over the engine's slots -- a slot gets
//                    its next batch before the finished one is regrouped --, per-read regroup of the compact decode (:403-446)
//   finisher threads chiron_finish_read (bases, vote, argmax, quality string, result/ and segments/ files: :446-462, :176-228)
//                    + meta/<name>.meta (:229-242)
//
// Same files as the Python pipeline, byte for byte (tests/test_gpu_parity.py, tests/test_pipeline_native.py), except the per-read
// timings inside meta/*.meta.  Scope: fast5 input or `.signal` text files (read_signal, chiron_input.py:527-539: nothing is written to raw/
// then, names keep their sub-folder), population BN (a partial last batch is submitted as it is: rows are independent of their batch
// there), host vote for every read.  Everything else stays with eval.evaluation.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

#include "../../include/chiron_amd.h"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }

struct Read {
  std::string name;            // <stem><suffix>: file_pre of every output
  std::vector<float> buf;      // zero padded: (n_win - 1) * jump + segment_len samples
  int64_t n = 0;               // samples behind `start`
  int32_t n_win = 0;
  double t0 = 0, t_read = 0;
  // regroup
  std::vector<uint8_t> flat;
  std::vector<int64_t> seg_len;
  std::vector<double> qs;
  int32_t rows_seen = 0;
};

struct Loaded {                // one input file, filled by a reader thread
  std::vector<std::shared_ptr<Read>> reads;
  std::vector<std::string> errors;
  std::string fatal;           // a `.signal` file that cannot be read or parsed ends the run (the reference raises)
  bool ready = false;
};

struct Run {                   // a contiguous chunk of one read inside a batch
  std::shared_ptr<Read> read;
  int32_t pos, rows;
};

struct Batch {
  std::vector<Run> runs;
  std::vector<const float*> pieces;
  std::vector<int32_t> piece_rows;
  std::vector<int64_t> stride;
  std::vector<int32_t> seq_len;
  int32_t n = 0;
};

struct Scratch {               // a collected batch's decode, copied out of the slot before the slot is reused
  std::vector<uint8_t> flat;
  std::vector<int32_t> counts;
  std::vector<float> prob;
};

bool write_file(const std::string& path, const std::string& text) {
  FILE* f = fopen(path.c_str(), "w");
  if (!f) return false;
  const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
  return fclose(f) == 0 && ok;
}

std::string stem_of(const std::string& path) {
  const size_t s = path.find_last_of('/');
  std::string base = s == std::string::npos ? path : path.substr(s + 1);
  const size_t d = base.find_last_of('.');
  return d == std::string::npos ? base : base.substr(0, d);
}

bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

void make_parent_dirs(const std::string& path) {      // mkdir -p dirname(path)
  for (size_t i = 1; i < path.size(); ++i)
    if (path[i] == '/') mkdir(path.substr(0, i).c_str(), 0777);
}

std::string fmt(const char* f, double v) {
  char b[64];
  snprintf(b, sizeof b, f, v);
  return b;
}

}  // namespace

extern "C" chiron_status chiron_pipeline_run(chiron_engine* e, const char* const* paths, int64_t n_paths, const chiron_pipeline_opts* o,
                                             chiron_pipeline_stats* st_out) {
  if (!paths || !o || n_paths < 0) return CHIRON_ERR_INVALID;
  if (!o->null_engine && !e) return CHIRON_ERR_INVALID;
  const int B = o->batch_size, L = o->segment_len, J = o->jump;
  if (B < 1 || L < 1 || J < 1 || o->n_slots < 1 || !o->output) return CHIRON_ERR_INVALID;
  double ratio = o->null_ratio > 0 ? o->null_ratio : 1.0;
  if (!o->null_engine) {
    int32_t T;
    chiron_status s = chiron_engine_dims(e, &T, &ratio);
    if (s) return s;
  }
  const int n_threads = std::max(1, o->n_threads);
  const bool fastq = o->fastq != 0;
  const std::string root = o->output, delim = o->delimiter ? o->delimiter : "\n";
  const int kernal = J >= L ? CHIRON_KERNAL_STICK : (J > 0.9 * L ? CHIRON_KERNAL_GLUE : CHIRON_KERNAL_SIMPLE);   // chiron_eval.py:138-150
  const double t_begin = now_s();

  // ---------------------------------------------------------------- readers: files in order, at most 2 * n_threads ahead of the consumer
  std::vector<Loaded> loaded((size_t)n_paths);
  std::mutex lm;
  std::condition_variable lcv;
  std::atomic<int64_t> next_file{0};
  int64_t consumed = 0;            // under lm
  int64_t ahead_windows = 0;       // under lm: windows of the files loaded and not yet consumed
  std::atomic<bool> stop{false};
  auto load = [&](int64_t fi) {
    Loaded& out = loaded[(size_t)fi];
    const std::string full = paths[fi], stem = stem_of(full);
    double t0 = now_s();
    if (ends_with(full, ".signal")) {                 // read_signal (chiron_input.py:527-539): whitespace-separated numbers -> float32
      auto r = std::make_shared<Read>();
      std::string rel = full;
      if (o->name_root) {
        const std::string rootdir = o->name_root;
        if (rel.compare(0, rootdir.size(), rootdir) == 0) rel = rel.substr(rootdir.size());
        while (!rel.empty() && rel[0] == '/') rel.erase(0, 1);
      } else {
        const size_t sl = rel.find_last_of('/');
        if (sl != std::string::npos) rel = rel.substr(sl + 1);
      }
      r->name = rel.substr(0, rel.size() - 7);
      std::string text;
      FILE* fi_ = fopen(full.c_str(), "rb");
      bool ok = fi_ != nullptr;
      if (ok) {
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, fi_)) > 0) text.append(chunk, got);
        ok = ferror(fi_) == 0;
        fclose(fi_);
      }
      std::vector<float> sig(text.size() / 2 + 1);
      size_t ns = 0;
      if (!ok) out.fatal = "cannot read " + full;
      else if (chiron_parse_signal_text(text.data(), text.size(), sig.data(), sig.size(), &ns) != CHIRON_OK) out.fatal = full + ": " + chiron_last_error();
      if (!out.fatal.empty()) return;
      r->n = std::max<int64_t>(0, (int64_t)ns - o->start);
      r->n_win = r->n > 0 ? (int32_t)((r->n + J - 1) / J) : 0;
      if (r->n_win > 0) {
        r->buf.assign((size_t)(r->n_win - 1) * J + L, 0.0f);
        const int64_t m = std::min<int64_t>(r->n, (int64_t)r->buf.size());
        memcpy(r->buf.data(), sig.data() + o->start, (size_t)m * sizeof(float));
      }
      r->t0 = t0;
      r->t_read = now_s() - t0;
      out.reads.push_back(std::move(r));
      return;
    }
    chiron_fast5* f = nullptr;
    if (chiron_fast5_open(full.c_str(), &f) != CHIRON_OK) {
      out.errors.push_back("Cannot extract file " + full + ". " + chiron_last_error());
      return;
    }
    const int nr = chiron_fast5_read_count(f);
    // every record first (extract_sig_ref.py:97-117: a file that fails anywhere is logged and skipped as a whole), then the writes
    struct Rec {
      std::string suffix, ref;
      std::vector<float> sig;
    };
    std::vector<Rec> recs;
    bool ok = nr > 0;
    if (!ok) out.errors.push_back("Cannot extract file " + full + ". Fail in extracting raw signal.");
    for (int i = 0; i < nr && ok; ++i) {
      char suffix[256], rid[256];
      int64_t ns = 0, fq = 0;
      ok = chiron_fast5_read_info(f, i, suffix, sizeof suffix, rid, sizeof rid, &ns, &fq) == CHIRON_OK;
      Rec rc;
      if (ok) {
        rc.suffix = suffix;
        rc.sig.resize((size_t)ns);
        if (ns > 0) ok = chiron_fast5_signal(f, i, rc.sig.data(), ns, o->rna) == CHIRON_OK;
      }
      if (ok && fq > 0) {
        rc.ref.assign((size_t)fq + 1, '\0');          // room for the terminator the reader writes
        ok = chiron_fast5_fastq(f, i, &rc.ref[0], fq + 1) == CHIRON_OK;
        rc.ref.resize((size_t)fq);
      }
      if (!ok) out.errors.push_back("Cannot extract file " + full + ". " + chiron_last_error());
      else recs.push_back(std::move(rc));
    }
    if (!ok) recs.clear();
    for (Rec& rc : recs) {
      const int64_t ns = (int64_t)rc.sig.size();
      if (ns == 0) {
        out.errors.push_back("Cannot extract file " + full + ". Got empty raw signal");
        continue;
      }
      auto r = std::make_shared<Read>();
      r->name = stem + rc.suffix;
      if (!o->no_raw && chiron_write_signal_text((root + "/raw/" + r->name + ".signal").c_str(), rc.sig.data(), ns, delim.c_str()) != CHIRON_OK) {
        out.errors.push_back("Cannot write raw/" + r->name + ".signal: " + chiron_last_error());
        continue;
      }
      if (!rc.ref.empty()) {                         // extract_sig_ref.py:125-127: "@<stem>\n" + the reference minus its first line
        const size_t nl = rc.ref.find('\n');
        write_file(root + "/reference/" + stem + "_ref.fastq", "@" + stem + "\n" + (nl == std::string::npos ? std::string() : rc.ref.substr(nl + 1)));
      }
      // chiron_input.py:276-286: windows signal[start:][i : i + L] for i in range(0, n, jump), zero padded
      r->n = std::max<int64_t>(0, ns - o->start);
      r->n_win = r->n > 0 ? (int32_t)((r->n + J - 1) / J) : 0;
      if (r->n_win > 0) {
        r->buf.assign((size_t)(r->n_win - 1) * J + L, 0.0f);
        const int64_t m = std::min<int64_t>(r->n, (int64_t)r->buf.size());
        memcpy(r->buf.data(), rc.sig.data() + o->start, (size_t)m * sizeof(float));
      }
      r->t0 = t0;
      r->t_read = now_s() - t0;
      t0 = now_s();
      out.reads.push_back(std::move(r));
    }
    chiron_fast5_close(f);
  };
  std::vector<std::thread> readers;
  for (int t = 0; t < n_threads; ++t)
    readers.emplace_back([&] {
      for (;;) {
        const int64_t fi = next_file.fetch_add(1);
        if (fi >= n_paths) return;
        {
          std::unique_lock<std::mutex> lk(lm);
          // read-ahead: 2 * n_threads files as the Python pools -- or, when that is less than three batches' worth of windows (batch 4096
          // of 257-window reads takes 16 files per batch: twelve files ahead starve the engine), until three batches are waiting
          lcv.wait(lk, [&] { return stop.load() || fi < consumed + 2 * n_threads || (ahead_windows < 3 * (int64_t)B && fi < consumed + 4096); });
          if (stop.load()) return;
        }
        load(fi);
        {
          std::lock_guard<std::mutex> lk(lm);
          loaded[(size_t)fi].ready = true;
          for (auto& r : loaded[(size_t)fi].reads) ahead_windows += r->n_win;
        }
        lcv.notify_all();
      }
    });

  // ---------------------------------------------------------------- finishers
  std::deque<std::shared_ptr<Read>> fq;
  std::mutex fm;
  std::condition_variable fcv;
  bool fdone = false;
  std::atomic<int64_t> n_bases{0}, n_finished{0}, n_failed{0};
  std::string first_error;
  std::mutex em;
  auto finish = [&](Read& r) {
    const double basecall_time = now_s() - r.t0;
    const int64_t n_seg = (int64_t)r.seg_len.size();
    std::vector<int64_t> off((size_t)n_seg + 1, 0);
    for (int64_t i = 0; i < n_seg; ++i) off[(size_t)i + 1] = off[(size_t)i] + r.seg_len[(size_t)i];
    const double zero = 0.0;
    const double* qs = fastq ? (n_seg ? r.qs.data() : &zero) : nullptr;
    const uint8_t none = 0;
    int64_t clen = 0;
    const std::string ext = fastq ? "fastq" : "fasta";
    const std::string res = root + "/result/" + r.name + "." + ext, seg = root + "/segments/" + r.name + "." + ext;
    if (r.name.find('/') != std::string::npos) {     // a recursive `.signal` input: <sub-folder>/<read> (chiron_eval.py:280-283)
      make_parent_dirs(res);
      if (!o->concise) {
        make_parent_dirs(seg);
        make_parent_dirs(root + "/meta/" + r.name + ".meta");
      }
    }
    const chiron_status s = chiron_finish_read(r.flat.empty() ? &none : r.flat.data(), off.data(), n_seg, qs, kernal, 0.2, (double)J / (double)L, r.name.c_str(),
                                               res.c_str(), o->concise ? nullptr : seg.c_str(), fastq ? 1 : 0, o->rna, nullptr, 0, &clen);
    if (s != CHIRON_OK) {
      ++n_failed;
      std::lock_guard<std::mutex> lk(em);
      if (first_error.empty()) first_error = std::string("finishing ") + r.name + ": " + chiron_last_error();
      return;
    }
    const double assembly_time = now_s() - r.t0;
    if (!o->concise) {                               // chiron_eval.py:229-242
      const double total = now_s() - r.t0;
      const double spans[5] = {r.t_read, basecall_time - r.t_read, assembly_time - basecall_time, total - assembly_time, total};
      std::string m = "# Reading Basecalling assembly output total rate(bp/s)\n";
      for (double v : spans) m += fmt("%5.3f", v) + " ";
      m += fmt("%5.3f", (double)clen / total) + "\n# read_len batch_size segment_len jump start_pos\n";
      char b[160];
      snprintf(b, sizeof b, "%lld %d %d %d %d\n", (long long)clen, B, L, J, o->start);
      m += b;
      m += std::string("# input_name model_name\n") + (o->input_name ? o->input_name : "") + " " + (o->model_name ? o->model_name : "") + "\n";
      write_file(root + "/meta/" + r.name + ".meta", m);
    }
    n_bases += clen;
    ++n_finished;
  };
  std::vector<std::thread> finishers;
  for (int t = 0; t < n_threads; ++t)
    finishers.emplace_back([&] {
      for (;;) {
        std::shared_ptr<Read> r;
        {
          std::unique_lock<std::mutex> lk(fm);
          fcv.wait(lk, [&] { return fdone || !fq.empty(); });
          if (fq.empty()) return;
          r = std::move(fq.front());
          fq.pop_front();
        }
        finish(*r);
      }
    });
  auto to_finisher = [&](std::shared_ptr<Read> r) {
    {
      std::lock_guard<std::mutex> lk(fm);
      fq.push_back(std::move(r));
    }
    fcv.notify_one();
  };

  // ---------------------------------------------------------------- the null engine (host ceiling measurements): a canned decode
  std::vector<int32_t> null_counts;
  std::vector<uint8_t> null_flat;
  std::vector<float> null_prob;
  if (o->null_engine) {
    unsigned x = 12345u;
    auto rnd = [&] { x = x * 1664525u + 1013904223u; return x >> 8; };
    null_counts.resize((size_t)B);
    for (int i = 0; i < B; ++i) null_counts[(size_t)i] = 38 + (int)(rnd() % 13);       // ~44 bases per window
    size_t tot = 0;
    for (int v : null_counts) tot += (size_t)v;
    null_flat.resize(tot);
    for (auto& v : null_flat) v = (uint8_t)(rnd() & 3);
    null_prob.resize((size_t)B);
    for (auto& v : null_prob) v = 1.0f + (float)(rnd() % 5000) * 1e-3f;
  }

  // ---------------------------------------------------------------- packing, engine, regroup (this thread)
  chiron_status status = CHIRON_OK;
  const int n_slots = o->n_slots;
  std::vector<std::unique_ptr<Batch>> inflight((size_t)n_slots);
  int64_t step = 0, n_batches = 0, n_windows = 0, n_reads = 0;
  Scratch sc;
  const uint32_t flags = CHIRON_COMPACT_DECODE | (fastq ? CHIRON_WANT_PROB : 0u);
  auto collect = [&](int slot, Batch& b) -> chiron_status {   // -> sc
    if (o->null_engine) {
      sc.counts.assign(null_counts.begin(), null_counts.begin() + b.n);
      size_t tot = 0;
      for (int v : sc.counts) tot += (size_t)v;
      sc.flat.assign(null_flat.begin(), null_flat.begin() + (long)tot);
      sc.prob.assign(null_prob.begin(), null_prob.begin() + b.n);
      return CHIRON_OK;
    }
    chiron_decoded d;
    const chiron_status s = chiron_engine_collect(e, slot, &d);
    if (s) return s;
    if (!d.flat_labels || !d.row_counts || d.batch != b.n) return CHIRON_ERR_STATE;
    sc.counts.assign(d.row_counts, d.row_counts + b.n);
    sc.flat.assign(d.flat_labels, d.flat_labels + d.nnz);
    sc.prob.assign(d.prob_logits, d.prob_logits + b.n);
    return CHIRON_OK;
  };
  auto regroup = [&](Batch& b) {
    std::vector<int64_t> off((size_t)b.n + 1, 0);
    for (int i = 0; i < b.n; ++i) off[(size_t)i + 1] = off[(size_t)i] + sc.counts[(size_t)i];
    for (Run& run : b.runs) {
      Read& r = *run.read;
      r.flat.insert(r.flat.end(), sc.flat.begin() + off[(size_t)run.pos], sc.flat.begin() + off[(size_t)(run.pos + run.rows)]);
      for (int i = run.pos; i < run.pos + run.rows; ++i)
        if (sc.counts[(size_t)i] > 0) {                   // rows with an empty decode vanish (sparse2dense, chiron_eval.py:36-66)
          r.seg_len.push_back(sc.counts[(size_t)i]);
          if (fastq) r.qs.push_back((double)sc.prob[(size_t)i]);
        }
      r.rows_seen += run.rows;
      if (r.rows_seen == r.n_win) to_finisher(run.read);
    }
  };
  auto launch = [&](std::unique_ptr<Batch> b) -> chiron_status {
    const int slot = (int)(step % n_slots);
    ++step;
    std::unique_ptr<Batch> done = std::move(inflight[(size_t)slot]);
    if (done) {
      const chiron_status s = collect(slot, *done);
      if (s) return s;
    }
    if (!o->null_engine) {
      const chiron_status s = chiron_engine_submit_pieces(e, slot, b->pieces.data(), b->piece_rows.data(), b->stride.data(), (int32_t)b->pieces.size(),
                                                          b->seq_len.data(), b->n, o->beam, flags);
      if (s) return s;
    }
    ++n_batches;
    n_windows += b->n;
    inflight[(size_t)slot] = std::move(b);
    if (done) regroup(*done);
    return CHIRON_OK;
  };
  std::unique_ptr<Batch> cur(new Batch);
  std::vector<std::string> errors;
  for (int64_t fi = 0; fi < n_paths && status == CHIRON_OK; ++fi) {
    {
      std::unique_lock<std::mutex> lk(lm);
      lcv.wait(lk, [&] { return loaded[(size_t)fi].ready; });
      consumed = fi + 1;
      for (auto& r : loaded[(size_t)fi].reads) ahead_windows -= r->n_win;
    }
    lcv.notify_all();
    Loaded& ld = loaded[(size_t)fi];
    for (auto& m : ld.errors) errors.push_back(m);
    if (!ld.fatal.empty()) {
      errors.push_back(ld.fatal);
      status = CHIRON_ERR_INVALID;
      break;
    }
    for (auto& rp : ld.reads) {
      ++n_reads;
      if (rp->n_win == 0) {            // nothing behind `start`: an empty consensus, as the reference writes it
        to_finisher(rp);
        continue;
      }
      int32_t i = 0;
      while (i < rp->n_win && status == CHIRON_OK) {
        const int32_t take = std::min(B - cur->n, rp->n_win - i);
        cur->runs.push_back(Run{rp, cur->n, take});
        cur->pieces.push_back(rp->buf.data() + (size_t)i * J);
        cur->piece_rows.push_back(take);
        cur->stride.push_back(J);
        for (int32_t k = i; k < i + take; ++k) {
          const int64_t len = std::min<int64_t>(rp->n - (int64_t)J * k, L);
          cur->seq_len.push_back((int32_t)std::nearbyint((double)len / ratio));      // chiron_eval.py:337, round half even
        }
        cur->n += take;
        i += take;
        if (cur->n == B) {
          status = launch(std::move(cur));
          cur.reset(new Batch);
        }
      }
    }
    ld.reads.clear();
  }
  if (status == CHIRON_OK && cur->n > 0) status = launch(std::move(cur));
  for (int k = 0; k < n_slots && status == CHIRON_OK; ++k) {
    const int slot = (int)((step + k) % n_slots);
    if (inflight[(size_t)slot]) {
      status = collect(slot, *inflight[(size_t)slot]);
      if (status == CHIRON_OK) regroup(*inflight[(size_t)slot]);
      inflight[(size_t)slot].reset();
    }
  }
  // ---------------------------------------------------------------- shut down
  {
    std::lock_guard<std::mutex> lk(lm);
    stop = true;
    consumed = n_paths;
  }
  lcv.notify_all();
  for (auto& t : readers) t.join();
  {
    std::lock_guard<std::mutex> lk(fm);
    fdone = true;
  }
  fcv.notify_all();
  for (auto& t : finishers) t.join();
  std::string engine_error;
  if (status != CHIRON_OK && !o->null_engine) {
    // a failed engine call: keep its reason, then return every slot that still holds a batch to idle (collect drains the slot's stream and
    // resets its state) -- the caller's engine stays usable
    engine_error = chiron_last_error();
    for (int k = 0; k < n_slots; ++k)
      if (inflight[(size_t)k]) {
        chiron_decoded d;
        (void)chiron_engine_collect(e, k, &d);
        inflight[(size_t)k].reset();
      }
    chiron_engine_sync(e);
  }
  if (st_out) {
    memset(st_out, 0, sizeof *st_out);
    st_out->reads = n_reads;
    st_out->reads_finished = n_finished.load();
    st_out->windows = n_windows;
    st_out->batches = n_batches;
    st_out->consensus_bases = n_bases.load();
    st_out->files_failed = (int64_t)errors.size();
    st_out->seconds = now_s() - t_begin;
    std::string all;
    for (auto& m : errors) all += m + "\n";
    if (!first_error.empty()) all += first_error + "\n";
    if (!engine_error.empty()) all += "engine: " + engine_error + "\n";
    snprintf(st_out->messages, sizeof st_out->messages, "%s", all.c_str());
  }
  if (status == CHIRON_OK && n_failed.load() > 0) return CHIRON_ERR_INVALID;
  return status;
}
