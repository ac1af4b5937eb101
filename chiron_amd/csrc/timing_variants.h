// Timing-only build switches -- ONE place for all of them.
//
// tools/variants.sh builds instrumented copies of a kernel source in which a PART of the kernel is switched off (no gate math,
// half of the MFMAs, no stores ...) to measure what that part costs.  Such a build computes GARBAGE.  The switches below are the
// only ones of that kind in the product sources; each is 0 in the product, and a non-zero value without CHIRON_TIMING_BUILD is a
// compile error, so that a stray -D can never ship wrong answers silently.  An object compiled with CHIRON_TIMING_BUILD registers
// itself at load time: chiron_build_flags() then reports CHIRON_BUILD_TIMING and chiron_amd/_lib.py refuses the library unless
// CHIRON_ALLOW_TIMING_BUILD=1 is set (the measurement tools set it).
//
//   CHIRON_SENS         bit mask, the sensitivity builds of DESIGN section 8: 1 recurrence without gate math (lstm.hip),
//                       2 Winograd without input transform, 4 without output transform (wino.hip), 8 recurrence with half of its
//                       MFMAs (lstm.hip), 16 DMA GEMM with half of its MFMAs, 32 the projection (ZOUT) form without its epilogue, 64 the
//                       convolution form without its epilogue, 1024 cycle counters (per-chunk barrier waits, epilogue, whole
//                       tiles) printed by four workgroups (gemm.hip), 2048 the same for the F(4,3) kernel (wino.hip), 4096 (with 32) the projection's 20 stores
//                       spread over the next tile's main loop: what a second accumulator set would buy (gemm.hip)
//   CHIRON_W32_VARIANT  lstm32w_kernel: 1 no gate math, 2 no MFMAs, 3 no transpose
//   CHIRON_F16F_VARIANT lstm16f_kernel, bit mask: 1 no gate math, 2 no MFMAs, 4 no output stores, 8 no x prefetch, 16 no x tile
//                       reads, 32 no h tile reads, 64 no h tile writes
//   CHIRON_S16_VARIANT  stream16.hip, bit mask: 1 no output stores, 2 no input DMA, 4 no MFMAs
#pragma once

#ifndef CHIRON_SENS
#define CHIRON_SENS 0
#endif
#ifndef CHIRON_W32_VARIANT
#define CHIRON_W32_VARIANT 0
#endif
#ifndef CHIRON_F16F_VARIANT
#define CHIRON_F16F_VARIANT 0
#endif
#ifndef CHIRON_S16_VARIANT
#define CHIRON_S16_VARIANT 0
#endif

#if (CHIRON_SENS != 0 || CHIRON_W32_VARIANT != 0 || CHIRON_F16F_VARIANT != 0 || CHIRON_S16_VARIANT != 0) && !defined(CHIRON_TIMING_BUILD)
#error "a timing-only kernel variant (CHIRON_SENS / CHIRON_*_VARIANT) computes garbage: build it with -DCHIRON_TIMING_BUILD (tools/variants.sh), never as the product"
#endif

namespace chiron {
void mark_timing_build(const char* source);   // engine.hip
}
#ifdef CHIRON_TIMING_BUILD
namespace {
struct ChironTimingMark {
  ChironTimingMark() { chiron::mark_timing_build(__FILE__); }
} chiron_timing_mark_instance;
}  // namespace
#endif
