// Parity build of the kernels in the arithmetic of the GL stack the reference's golden images were rendered under
// (rm_device.hpp RM_GL_STACK: that stack's transcendental functions, rm_ss_math.hpp, and its min / max / fract / unorm
// conventions; none of the IEEE-derived exact shortcuts).  Selected per context by rm_ctx_set_gl_stack; with it
// rm_render_sample / rm_probe* / rm_present* on the strict flags reproduce tests/golden/ bit for bit on the GPU.
// Everything of this unit lives in namespace rm_gl (the same sources as rm_strict.hip, renamed), behind C entry points.
#define RM_BUILD_FAST 0
#define RM_GL_STACK 1
#undef RM_NORMAL_POLICY
#define rm rm_gl
#include <type_traits>
#include "rm_device.hpp"
#include "rm_kernels.inc"
#include "rm_frame_kernels.inc"
#undef rm

extern "C" {
hipError_t rm_gl_launch_pixels(const void* kparams, hipStream_t stream) { return rm_gl::launch_pixels_strict(*static_cast<const KParams*>(kparams), stream); }
hipError_t rm_gl_launch_probe(const void* probe_params, hipStream_t stream) { return rm_gl::launch_probe_strict(*static_cast<const ProbeParams*>(probe_params), stream); }
hipError_t rm_gl_launch_camera_rng(const RmUniforms* u, int W, int H, int what, int count, float* out, hipStream_t stream) {
  return rm_gl::launch_camera_rng(*u, W, H, what, count, out, stream);
}
hipError_t rm_gl_launch_present(const float4* color, const float4* normal_dof, int W, int H, float brightness, uchar4* out, hipStream_t stream) {
  return rm_gl::launch_present(color, normal_dof, W, H, brightness, out, stream);
}
hipError_t rm_gl_launch_present_striped(const float4* color, const float4* normal_dof, int W, int H, float brightness, uchar4* out, int stripe_rows, int parts,
                                        int part, int local_rows, hipStream_t stream) {
  return rm_gl::launch_present_striped(color, normal_dof, W, H, brightness, out, stripe_rows, parts, part, local_rows, stream);
}
hipError_t rm_gl_launch_math_probe(int fn, const float* a, const float* b, int n, float* out, hipStream_t stream) {
  return rm_gl::launch_math_probe(fn, a, b, n, out, stream);
}
hipError_t rm_gl_set_native_tan(int on, hipStream_t) {  // synchronous: the source is the caller's stack
  return hipMemcpyToSymbol(HIP_SYMBOL(rm_gl::g_native_tan), &on, sizeof on, 0, hipMemcpyHostToDevice);
}
hipError_t rm_gl_launch_present_rows(const float4* color, long long pixels, float brightness, uchar4* out, hipStream_t stream) {
  return rm_gl::launch_present_rows(color, pixels, brightness, out, stream);
}
}
