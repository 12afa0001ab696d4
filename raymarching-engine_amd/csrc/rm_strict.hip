// Parity build of the kernels: every operation on the precise policy (rm_device.hpp PM).
#define RM_BUILD_FAST 0
#undef RM_NORMAL_POLICY  // experiment builds override it for the fast TU only
#include <type_traits>
#include "rm_device.hpp"
#include "rm_kernels.inc"
#include "rm_wavefront.inc"
