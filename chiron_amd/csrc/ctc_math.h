// exp / log for the CTC beam search, specified down to the operation so that every implementation of the
// decoder produces the same bits.
//
// TF's CTCBeamSearchDecoder (ctc_beam_search.h Step(), ctc_loss_util.h LogSumExp) calls float expf / logf /
// log1pf of whatever libm it was built against; those differ between platforms in the last ulp, and beam search
// is discontinuous in them (a TopN eviction decided by one ulp changes the string).  The decoder here therefore
// uses its own pair, written only in operations IEEE-754 defines exactly (multiply, add, fused multiply-add,
// round-to-nearest-even, integer arithmetic on the exponent field): Cody-Waite reduction + the Cephes single
// precision polynomials (about 1 ulp).  The same operations in the same order in any language give the same
// bits -- oracle/chiron_oracle.c restates them, tests/test_gpu_parity.py compares bit for bit.
//
// All callers pass d <= 0 to ctc_exp_neg and x >= 1 to ctc_log_pos.
#pragma once
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define CHIRON_CTC_FN __host__ __device__ static __forceinline__
#else
#define CHIRON_CTC_FN static inline
#endif

CHIRON_CTC_FN int ctc_f2i(float x) {
  int i;
  memcpy(&i, &x, 4);
  return i;
}
CHIRON_CTC_FN float ctc_i2f(int i) {
  float x;
  memcpy(&x, &i, 4);
  return x;
}

// e^d for d <= 0; exactly 0 below -86 (e^-86 = 4.5e-38 is still a normal float, no subnormal is ever produced)
CHIRON_CTC_FN float ctc_exp_neg(float d) {
  if (!(d >= -86.0f)) return 0.0f;
  const float n = rintf(d * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, d);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  const float y = fmaf(p, r2, r) + 1.0f;
  return ctc_i2f(ctc_f2i(y) + ((int)n << 23));
}

// ln x for a positive normal x
CHIRON_CTC_FN float ctc_log_pos(float x) {
  const int bits = ctc_f2i(x);
  int e = (bits >> 23) - 127;
  float m = ctc_i2f((bits & 0x007fffff) | 0x3f800000);  // [1, 2)
  if (m > 1.41421356237309505f) {
    m = m * 0.5f;
    e += 1;
  }
  const float f = m - 1.0f;
  const float z = f * f;
  float y = 7.0376836292e-2f;
  y = fmaf(y, f, -1.1514610310e-1f);
  y = fmaf(y, f, 1.1676998740e-1f);
  y = fmaf(y, f, -1.2420140846e-1f);
  y = fmaf(y, f, 1.4249322787e-1f);
  y = fmaf(y, f, -1.6668057665e-1f);
  y = fmaf(y, f, 2.0000714765e-1f);
  y = fmaf(y, f, -2.4999993993e-1f);
  y = fmaf(y, f, 3.3333331174e-1f);
  y = y * f;
  y = y * z;
  const float fe = (float)e;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(z, -0.5f, y);
  float r = f + y;
  r = fmaf(fe, 0.693359375f, r);
  return r;
}

// ctc_loss_util.h LogSumExp: log(e^a + e^b), -inf is the log of zero
CHIRON_CTC_FN float ctc_log_sum_exp(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const float m = a > b ? a : b;
  const float d = (a > b ? b : a) - m;
  return m + ctc_log_pos(1.0f + ctc_exp_neg(d));
}

// log-softmax of one frame of K classes (ctc_beam_search.h Step(): raw activations are normalised per frame):
// out[k] = (x[k] - max) - ln sum_j e^(x[j] - max), the sum taken in index order
template <int K>
CHIRON_CTC_FN void ctc_log_softmax(const float* x, float* out) {
  float mx = x[0];
#pragma unroll
  for (int k = 1; k < K; ++k) mx = x[k] > mx ? x[k] : mx;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) s += ctc_exp_neg(x[k] - mx);
  const float lse = ctc_log_pos(s);
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = (x[k] - mx) - lse;
}
