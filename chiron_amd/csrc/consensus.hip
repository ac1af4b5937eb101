// Overlap consensus of one read on the device: what chiron_assemble + the argmax of chiron_eval.py:457
// + qs (chiron_eval.py:152-174) compute on the host -- glue_kernal / stick_kernal displacements (easy_assembler.py:276-300),
// running start positions, the vote of simple_assembly_qs (:393-432) -- as three HBM-bound kernels, for reads long
// enough (hundreds of thousands of windows) that the vote is worth a launch.  Integer work: bit-identical to the host
// form by construction (counts are exact, quality sums are added in segment order like the host loop).
//
//   displacement_kernel   one thread per segment pair: score(i) = 2 * matches - i over the overlaps the reference tries
//   scan_starts_kernel    one workgroup: inclusive scan of the displacements -> start column of every segment, and the
//                         consensus length max(start + n) over segments 1.. (segment 0 is skipped, as the reference's
//                         `continue` skips it)
//   vote_kernel           one thread per consensus column: the segments covering it are found by bisection on the
//                         (non-decreasing) starts, votes and quality sums accumulate in segment order, then the base
//                         (first maximum, chiron_eval.py:457) and the three numbers qs() needs of a column -- n1, n2 and
//                         the quality sum behind the winning base -- are written; no [4][len] matrices ever exist.  The
//                         Phred character itself is computed by the HOST's formula from those (eval.qs_from_votes): a
//                         device log10 may differ from NumPy's in the last bit, and int() truncation turns that into a
//                         different character at integer boundaries
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/chiron_amd.h"

namespace chiron {
chiron_status set_error(chiron_status st, const char* fmt, ...);

namespace {

__global__ __launch_bounds__(256) void displacement_kernel(const uint8_t* __restrict__ bases, const int64_t* __restrict__ off, int64_t n_seg,
                                                           int glue, int64_t* __restrict__ disp) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  if (s == 0) {
    disp[0] = 0;
    return;
  }
  const int64_t n = off[s + 1] - off[s], pn = off[s] - off[s - 1];
  int64_t best_i = 0;
  if (glue) {
    // max_overlap = min(floor(0.1 * prev_n), n): the same IEEE double product as the reference
    int64_t max_overlap = (int64_t)floor(0.1 * (double)pn);
    if (n < max_overlap) max_overlap = n;
    const uint8_t* cur = bases + off[s];
    const uint8_t* prev = bases + off[s - 1];
    int64_t best_score = 0;
    for (int64_t i = 1; i < max_overlap; ++i) {
      int64_t same = 0;
      const uint8_t* tail = prev + (pn - i);
      for (int64_t j = 0; j < i; ++j) same += cur[j] == tail[j];
      const int64_t score = 2 * same - i;
      if (score > best_score) {   // first strict maximum
        best_score = score;
        best_i = i;
      }
    }
  }
  disp[s] = pn - best_i;
}

// one workgroup of 1024 threads walks the array in chunks of 1024: Hillis-Steele scan in LDS + running carry
__global__ __launch_bounds__(1024) void scan_starts_kernel(const int64_t* __restrict__ disp, const int64_t* __restrict__ off, int64_t n_seg,
                                                           int64_t* __restrict__ start, int64_t* __restrict__ out_len_maxn) {
  __shared__ int64_t buf[2][1024];
  __shared__ int64_t carry, len_s, maxn_s;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0, len_s = 0, maxn_s = 0;
  __syncthreads();
  int64_t my_len = 0, my_maxn = 0;
  for (int64_t base = 0; base < n_seg; base += 1024) {
    const int64_t s = base + tid;
    int cur = 0;
    buf[0][tid] = s < n_seg ? disp[s] : 0;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      buf[cur ^ 1][tid] = buf[cur][tid] + (tid >= d ? buf[cur][tid - d] : 0);
      cur ^= 1;
      __syncthreads();
    }
    const int64_t st = carry + buf[cur][tid];
    if (s < n_seg) {
      start[s] = st;
      const int64_t n = off[s + 1] - off[s];
      if (s > 0 && st + n > my_len) my_len = st + n;
      if (n > my_maxn) my_maxn = n;
    }
    __syncthreads();
    if (tid == 1023) carry = st;
    __syncthreads();
  }
  atomicMax(reinterpret_cast<unsigned long long*>(&len_s), (unsigned long long)my_len);   // values are >= 0
  atomicMax(reinterpret_cast<unsigned long long*>(&maxn_s), (unsigned long long)my_maxn);
  __syncthreads();
  if (tid == 0) {
    out_len_maxn[0] = len_s;
    out_len_maxn[1] = maxn_s;
  }
}

__global__ __launch_bounds__(256) void vote_kernel(const uint8_t* __restrict__ bases, const int64_t* __restrict__ off, const int64_t* __restrict__ start,
                                                   const double* __restrict__ seg_qs, int64_t n_seg, int64_t length, int64_t maxn,
                                                   uint8_t* __restrict__ consensus, int32_t* __restrict__ out_n1, int32_t* __restrict__ out_n2,
                                                   double* __restrict__ out_qtop) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= length) return;
  // last segment whose start is <= col (starts are non-decreasing: every displacement is >= 0)
  int64_t lo = 0, hi = n_seg;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (start[mid] <= col) lo = mid + 1; else hi = mid;
  }
  const int64_t last = lo - 1;
  // no segment is longer than maxn, so nothing that starts before col - maxn + 1 can reach col
  int64_t first = last;
  while (first > 0 && start[first - 1] > col - maxn) --first;
  double cnt[4] = {0, 0, 0, 0}, qsum[4] = {0, 0, 0, 0};
  for (int64_t s = first; s <= last; ++s) {   // ascending: the order in which the host loop adds
    const int64_t j = col - start[s];
    if (j < off[s + 1] - off[s]) {
      const int b = bases[off[s] + j] & 3;
      cnt[b] += 1.0;
      if (seg_qs) qsum[b] += seg_qs[s];
    }
  }
  int arg = 0;        // np.argmax: first maximum (chiron_eval.py:457)
  int top = 0;        // the quality belongs to the LAST of the maxima (ascending argsort of NumPy 1.x, eval.qs)
  double n1 = cnt[0];
  for (int b = 1; b < 4; ++b) {
    if (cnt[b] > n1) n1 = cnt[b], arg = b;
  }
  for (int b = 0; b < 4; ++b)
    if (cnt[b] == n1) top = b;
  double n2 = -1;     // second largest value with multiplicity
  {
    bool skipped = false;
    for (int b = 0; b < 4; ++b) {
      if (cnt[b] == n1 && !skipped) {
        skipped = true;
        continue;
      }
      if (cnt[b] > n2) n2 = cnt[b];
    }
  }
  consensus[col] = (uint8_t)arg;
  if (out_n1) {
    out_n1[col] = (int32_t)n1;      // counts are small exact integers
    out_n2[col] = (int32_t)n2;
    out_qtop[col] = qsum[top];
  }
}

// Stream-ordered allocations: hipFree would synchronise the whole device and stall the engine's in-flight slots
struct DeviceBuffers {
  static constexpr int N = 10;
  void* p[N] = {};
  hipStream_t stream = nullptr;
  ~DeviceBuffers() {
    if (!stream) return;
    for (void* q : p)
      if (q) hipFreeAsync(q, stream);
    hipStreamSynchronize(stream);
    hipStreamDestroy(stream);
  }
};

}  // namespace
}  // namespace chiron

#define CONS_TRY(expr)                                                                                                     \
  do {                                                                                                                      \
    hipError_t _e = (expr);                                                                                                 \
    if (_e != hipSuccess) return chiron::set_error(CHIRON_ERR_DEVICE, "%s failed: %s (consensus.hip:%d)", #expr, hipGetErrorString(_e), __LINE__); \
  } while (0)

extern "C" chiron_status chiron_consensus_device(int32_t device_id, const uint8_t* bases, const int64_t* seg_off, int64_t n_seg, const double* seg_qs,
                                                 int32_t kernal, uint8_t* consensus, int32_t* n1, int32_t* n2, double* q_top, int64_t cap,
                                                 int64_t* out_len) {
  using namespace chiron;
  if (!seg_off || !out_len || n_seg < 0) return set_error(CHIRON_ERR_INVALID, "chiron_consensus_device: bad arguments");
  if (kernal != CHIRON_KERNAL_GLUE && kernal != CHIRON_KERNAL_STICK)
    return set_error(CHIRON_ERR_INVALID, "chiron_consensus_device: kernal %d (1 = glue, 2 = stick; the simple kernel's displacements are host code)", kernal);
  const bool votes = n1 || n2 || q_top;
  if (votes && !(n1 && n2 && q_top)) return set_error(CHIRON_ERR_INVALID, "chiron_consensus_device: n1, n2 and q_top come together");
  if (votes && !seg_qs) return set_error(CHIRON_ERR_INVALID, "chiron_consensus_device: the vote summary was requested without seg_qs");
  *out_len = 0;
  if (n_seg < 2) return CHIRON_OK;   // a single segment yields an empty consensus (the reference's `continue`)
  const int64_t n_bases = seg_off[n_seg];
  for (int64_t s = 0; s < n_seg; ++s)
    if (seg_off[s + 1] < seg_off[s]) return set_error(CHIRON_ERR_INVALID, "chiron_consensus_device: seg_off must not decrease");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_error(CHIRON_ERR_DEVICE, "no HIP device visible: libchiron_amd has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return set_error(CHIRON_ERR_INVALID, "device_id %d out of range (%d devices)", device_id, ndev);
  CONS_TRY(hipSetDevice(device_id));
  DeviceBuffers d;
  CONS_TRY(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
  uint8_t*& d_bases = reinterpret_cast<uint8_t*&>(d.p[0]);
  int64_t*& d_off = reinterpret_cast<int64_t*&>(d.p[1]);
  int64_t*& d_disp = reinterpret_cast<int64_t*&>(d.p[2]);
  int64_t*& d_start = reinterpret_cast<int64_t*&>(d.p[3]);
  int64_t*& d_len = reinterpret_cast<int64_t*&>(d.p[4]);
  double*& d_qs = reinterpret_cast<double*&>(d.p[5]);
  uint8_t*& d_cons = reinterpret_cast<uint8_t*&>(d.p[6]);
  int32_t*& d_n1 = reinterpret_cast<int32_t*&>(d.p[7]);
  int32_t*& d_n2 = reinterpret_cast<int32_t*&>(d.p[8]);
  double*& d_qtop = reinterpret_cast<double*&>(d.p[9]);
  CONS_TRY(hipMallocAsync(&d.p[0], (size_t)n_bases + 16, d.stream));
  CONS_TRY(hipMallocAsync(&d.p[1], (size_t)(n_seg + 1) * 8, d.stream));
  CONS_TRY(hipMallocAsync(&d.p[2], (size_t)n_seg * 8, d.stream));
  CONS_TRY(hipMallocAsync(&d.p[3], (size_t)n_seg * 8, d.stream));
  CONS_TRY(hipMallocAsync(&d.p[4], 16, d.stream));
  if (seg_qs) CONS_TRY(hipMallocAsync(&d.p[5], (size_t)n_seg * 8, d.stream));
  CONS_TRY(hipMemcpyAsync(d_bases, bases, (size_t)n_bases, hipMemcpyHostToDevice, d.stream));
  CONS_TRY(hipMemcpyAsync(d_off, seg_off, (size_t)(n_seg + 1) * 8, hipMemcpyHostToDevice, d.stream));
  if (seg_qs) CONS_TRY(hipMemcpyAsync(d_qs, seg_qs, (size_t)n_seg * 8, hipMemcpyHostToDevice, d.stream));
  hipLaunchKernelGGL(displacement_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, d.stream, d_bases, d_off, n_seg,
                     kernal == CHIRON_KERNAL_GLUE ? 1 : 0, d_disp);
  hipLaunchKernelGGL(scan_starts_kernel, dim3(1), dim3(1024), 0, d.stream, d_disp, d_off, n_seg, d_start, d_len);
  int64_t len_maxn[2] = {0, 0};
  CONS_TRY(hipMemcpyAsync(len_maxn, d_len, 16, hipMemcpyDeviceToHost, d.stream));
  CONS_TRY(hipStreamSynchronize(d.stream));
  const int64_t length = len_maxn[0];
  *out_len = length;
  if (length > cap) return set_error(CHIRON_ERR_OVERFLOW, "chiron_consensus_device: consensus needs %lld columns, capacity %lld", (long long)length, (long long)cap);
  if (length == 0) return CHIRON_OK;
  if (!consensus) return set_error(CHIRON_ERR_INVALID, "chiron_consensus_device: null consensus");
  CONS_TRY(hipMallocAsync(&d.p[6], (size_t)length, d.stream));
  if (votes) {
    CONS_TRY(hipMallocAsync(&d.p[7], (size_t)length * 4, d.stream));
    CONS_TRY(hipMallocAsync(&d.p[8], (size_t)length * 4, d.stream));
    CONS_TRY(hipMallocAsync(&d.p[9], (size_t)length * 8, d.stream));
  }
  hipLaunchKernelGGL(vote_kernel, dim3((unsigned)((length + 255) / 256)), dim3(256), 0, d.stream, d_bases, d_off, d_start, seg_qs ? d_qs : nullptr,
                     n_seg, length, len_maxn[1], d_cons, votes ? d_n1 : nullptr, votes ? d_n2 : nullptr, votes ? d_qtop : nullptr);
  CONS_TRY(hipGetLastError());
  CONS_TRY(hipMemcpyAsync(consensus, d_cons, (size_t)length, hipMemcpyDeviceToHost, d.stream));
  if (votes) {
    CONS_TRY(hipMemcpyAsync(n1, d_n1, (size_t)length * 4, hipMemcpyDeviceToHost, d.stream));
    CONS_TRY(hipMemcpyAsync(n2, d_n2, (size_t)length * 4, hipMemcpyDeviceToHost, d.stream));
    CONS_TRY(hipMemcpyAsync(q_top, d_qtop, (size_t)length * 8, hipMemcpyDeviceToHost, d.stream));
  }
  CONS_TRY(hipStreamSynchronize(d.stream));
  return CHIRON_OK;
}
