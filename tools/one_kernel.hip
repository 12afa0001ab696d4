// One pixel kernel, alone in a translation unit: a few seconds to compile instead of minutes, for a look at its registers,
// spills and code (tools/one_kernel.sh).  -DOK_KIND=9 -DOK_PREVIEW=false -DOK_FAST=true -DOK_EPS=false -DOK_SHAPE=1
#ifndef OK_FAST
#define OK_FAST true
#endif
#define RM_BUILD_FAST (OK_FAST ? 1 : 0)
#define RM_ONLY_KERNELS 1
#include <type_traits>
#include "../raymarching-engine_amd/csrc/rm_device.hpp"
#include "../raymarching-engine_amd/csrc/rm_kernels.inc"
#ifndef OK_KIND
#define OK_KIND 9
#endif
#ifndef OK_PREVIEW
#define OK_PREVIEW false
#endif
#ifndef OK_EPS
#define OK_EPS false
#endif
#ifndef OK_SHAPE
#define OK_SHAPE 1
#endif
template __global__ void rm::rm_pixel_kernel<OK_KIND, OK_PREVIEW, OK_FAST, OK_EPS, OK_SHAPE>(const KParams);
