// chiron_pipeline_run (chiron_amd/csrc/pipeline.cpp) under ThreadSanitizer / AddressSanitizer, built by tests/test_host_sanitizers.py
// together with fast5.cpp and assemble.cpp: reader threads, the packing thread and finisher threads over a folder of fast5 files, the
// NULL engine in place of the GPU (no engine entry point is reached; the four the file references are stubbed).  The contract: no data
// race, no leak, no out-of-bounds access; every read finished.
//   usage: tsan_pipeline <output folder> <threads> <file.fast5>...
#include <cstdarg>
#include <cstdio>
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
extern "C" {
const char* chiron_last_error(void) { return "stub"; }
chiron_status chiron_engine_dims(const chiron_engine*, int32_t*, double*) { return CHIRON_ERR_STATE; }
chiron_status chiron_engine_collect(chiron_engine*, int32_t, chiron_decoded*) { return CHIRON_ERR_STATE; }
chiron_status chiron_engine_sync(chiron_engine*) { return CHIRON_ERR_STATE; }
chiron_status chiron_engine_submit_pieces(chiron_engine*, int32_t, const float* const*, const int32_t*, const int64_t*, int32_t, const int32_t*, int32_t,
                                          int32_t, uint32_t) {
  return CHIRON_ERR_STATE;
}
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  chiron_pipeline_opts o;
  memset(&o, 0, sizeof o);
  o.batch_size = 64, o.segment_len = 400, o.jump = 390, o.fastq = 1, o.n_threads = atoi(argv[2]), o.n_slots = 3, o.null_engine = 1, o.null_ratio = 1.0;
  o.output = argv[1], o.delimiter = "\n", o.input_name = "in", o.model_name = "null";
  std::vector<const char*> paths(argv + 3, argv + argc);
  for (int rep = 0; rep < 3; ++rep) {
    chiron_pipeline_stats st;
    const chiron_status s = chiron_pipeline_run(nullptr, paths.data(), (int64_t)paths.size(), &o, &st);
    if (s != CHIRON_OK || st.reads != st.reads_finished || st.reads < 1) {
      fprintf(stderr, "tsan_pipeline: status %d, %lld reads, %lld finished: %s\n", (int)s, (long long)st.reads, (long long)st.reads_finished, st.messages);
      return 1;
    }
    o.concise = rep & 1;
  }
  printf("clean\n");
  return 0;
}
