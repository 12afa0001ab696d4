// Fast build of the kernels: the march on the fast policy (rm_device.hpp FM:
// v_fma / v_rcp / v_sqrt / v_log at hardware rate, trig-free power-8
// Mandelbulb); normals, shading and the random stream as in the parity build.
#define RM_BUILD_FAST 1
#include <type_traits>
#include "rm_device.hpp"
#include "rm_kernels.inc"
#include "rm_frame_kernels.inc"
#ifndef RM_WITH_WAVEFRONT
#define RM_WITH_WAVEFRONT 0  // 1: the tests' cross-check build (rm_api.hip "the wavefront pipeline")
#endif
#if RM_WITH_WAVEFRONT
#include "rm_wavefront.inc"
#endif

#ifdef RM_LANE_STATS
// diagnostic build only: Mandelbulb evaluations of this TU, {lane-rounds used, lane-rounds issued, lanes active, lanes issued}
extern "C" int rm_fast_lane_stats(unsigned long long* out4, int reset) {
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(rm::g_lane_stats), 32) != hipSuccess) return 1;
  if (reset) {
    const unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rm::g_lane_stats), z, 32) != hipSuccess) return 1;
  }
  return 0;
}
#endif
