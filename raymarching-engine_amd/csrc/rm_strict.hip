// Parity build of the kernels: every operation on the precise policy (rm_device.hpp PM).
#define RM_BUILD_FAST 0
#undef RM_NORMAL_POLICY  // experiment builds override it for the fast TU only
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
