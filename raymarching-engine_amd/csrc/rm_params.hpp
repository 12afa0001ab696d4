// rm_params.hpp -- kernel argument blocks shared by the API and both kernel builds.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hip_raymarch.h"

struct DevScene {
  int kind;
  int nprims;
  const RmPrim* prims;  // device pointer (RM_SCENE_TABLE)
  float p[16];
  RmMaterial mat;
};

struct KParams {
  RmUniforms u;
  DevScene scene;
  float4* color;
  float4* normal_dof;    // nullptr = colour only
  float4* albedo_depth;
  int W, H;            // full image (textureSize(previousColor), raymarcher.frag:183)
  int row_begin;       // first image row held by the planes
  int tx, ty, tw, th;  // tile, already clipped to the window
  float retire_eps;    // 0 = exact (fixed-point) retire only
};

struct ProbeParams {
  DevScene scene;
  const float* in;
  float* out;
  int n;
  int what;
  float param;
};

namespace rm {
hipError_t launch_pixels_strict(const KParams& P, hipStream_t stream);
hipError_t launch_pixels_fast(const KParams& P, hipStream_t stream);
hipError_t launch_probe_strict(const ProbeParams& P, hipStream_t stream);
hipError_t launch_probe_fast(const ProbeParams& P, hipStream_t stream);
hipError_t launch_camera_rng(const RmUniforms& u, int W, int H, int what, int count, float* out, hipStream_t stream);
}  // namespace rm
