// Fast build of the kernels: the march on the fast policy (rm_device.hpp FM:
// v_fma / v_rcp / v_sqrt / v_log at hardware rate, trig-free power-8
// Mandelbulb); normals, shading and the random stream as in the parity build.
#define RM_BUILD_FAST 1
#include <type_traits>
#include "rm_device.hpp"
#include "rm_kernels.inc"
#include "rm_wavefront.inc"
