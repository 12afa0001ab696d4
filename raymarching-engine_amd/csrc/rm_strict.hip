// Parity build of the kernels: every operation on the precise policy (rm_device.hpp PM).
#define RM_BUILD_FAST 0
#include <type_traits>
#include "rm_device.hpp"
#include "rm_kernels.inc"
#include "rm_wavefront.inc"
