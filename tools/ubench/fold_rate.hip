// The 64-row smooth-union sphere fold of CSG-64 (rm_device.hpp eval_spheres_one_k) ALONE: every wave folds the whole table from LDS for a
// fixed number of evaluations -- no march, no divergence, no repack, no culling -- at 8 / 4 / 2 / 1 waves per SIMD.  What fraction of the
// chip's issue slots does the fold's instruction mix take when nothing else is in the way?  (C4 / C5 run at 0.66 / 0.68: is the missing
// third the mix -- one quarter-rate v_sqrt per 13 instructions, a dependent chain of 5 per row, one LDS read per row -- or the kernel around it?)
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o fold_rate fold_rate.hip && ./fold_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int MODE>
__device__ __forceinline__ float sphere_row1(const float4 r, float x, float y, float z) {
  const float qx = x - r.x, qy = y - r.y, qz = z - r.z;
  const float q2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
  return ((MODE & 1) ? q2 * 0.37f : __builtin_amdgcn_sqrtf(q2)) - r.w;  // MODE & 1: an ordinary multiply in the square root's place
}
__device__ __forceinline__ float smooth_row(float d, float di, float k, float hik) {
  const float t = di - d;
  const float h = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(hik, t, 0.5f), 0.0f), 1.0f);
  return __builtin_fmaf(-h, __builtin_fmaf(k, 1.0f - h, t), di);
}

template <int MODE>
__global__ __launch_bounds__(512, 8) void fold(const float4* table, int n, int evals, float k, float* out, unsigned long long* cyc) {
  __shared__ float4 rows[256];
  for (int i = threadIdx.x; i < n; i += blockDim.x) rows[i] = table[i];
  __syncthreads();
  float x = 0.001f * threadIdx.x - 0.3f, y = 0.002f * blockIdx.x - 0.5f, z = -2.0f;
  const float hik = 0.5f / k;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float acc = 0.0f;
  for (int e = 0; e < evals; e++) {
    float d = sphere_row1<MODE>(rows[0], x, y, z);
    int i = 1;
    float4 k0 = rows[1], k1 = rows[2], k2 = rows[3], k3 = rows[4];  // MODE & 2: the same four rows every trip, read once per evaluation (no LDS read in the loop)
    for (; i + 3 < n; i += 4) {
      float4 r0, r1, r2, r3;
      if (MODE & 2) { r0 = k0; r1 = k1; r2 = k2; r3 = k3; asm volatile("" : "+v"(k0.x), "+v"(k1.x), "+v"(k2.x), "+v"(k3.x)); }
      else { r0 = rows[i]; r1 = rows[i + 1]; r2 = rows[i + 2]; r3 = rows[i + 3]; }
      if (MODE & 4) {  // the four square roots of a trip back to back (one statement), then the four unions
        auto q2 = [&](const float4 r) { const float qx = x - r.x, qy = y - r.y, qz = z - r.z; return __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx)); };
        float s0 = q2(r0), s1 = q2(r1), s2 = q2(r2), s3 = q2(r3);
        asm volatile("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_sqrt_f32 %2, %2\n\tv_sqrt_f32 %3, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        d = smooth_row(d, s0 - r0.w, k, hik);
        d = smooth_row(d, s1 - r1.w, k, hik);
        d = smooth_row(d, s2 - r2.w, k, hik);
        d = smooth_row(d, s3 - r3.w, k, hik);
        continue;
      }
      const float d0 = sphere_row1<MODE>(r0, x, y, z), d1 = sphere_row1<MODE>(r1, x, y, z);
      d = smooth_row(d, d0, k, hik);
      d = smooth_row(d, d1, k, hik);
      const float d2 = sphere_row1<MODE>(r2, x, y, z), d3 = sphere_row1<MODE>(r3, x, y, z);
      d = smooth_row(d, d2, k, hik);
      d = smooth_row(d, d3, k, hik);
    }
    for (; i < n; i++) d = smooth_row(d, sphere_row1<MODE>(rows[i], x, y, z), k, hik);
    z += 0.01f * d;  // the march's dependence of the next point on this distance
    acc += d;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the rows through the SCALAR data cache instead of LDS (round 2 measured a fold like this at 1.9x the LDS one's time; here with the roots grouped, and 8 rows per trip
// so that the loads of the next rows are in flight): rows in __constant__ memory, uniform index -> s_load_dwordx4/x8/x16, SGPR operands
__constant__ float4 c_rows[256];
template <int TRIP>
__global__ __launch_bounds__(512, 8) void fold_scalar(int n, int evals, float k, float* out) {
  float x = 0.001f * threadIdx.x - 0.3f, y = 0.002f * blockIdx.x - 0.5f, z = -2.0f;
  const float hik = 0.5f / k;
  float acc = 0.0f;
  auto q2 = [&](const float4 r) { const float qx = x - r.x, qy = y - r.y, qz = z - r.z; return __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx)); };
  for (int e = 0; e < evals; e++) {
    const float4 f = c_rows[0];
    float d = __builtin_amdgcn_sqrtf(q2(f)) - f.w;
    int i = 1;
    for (; i + TRIP - 1 < n; i += TRIP) {
#pragma unroll
      for (int h = 0; h < TRIP; h += 4) {
        const float4 r0 = c_rows[i + h], r1 = c_rows[i + h + 1], r2 = c_rows[i + h + 2], r3 = c_rows[i + h + 3];
        float s0 = q2(r0), s1 = q2(r1), s2 = q2(r2), s3 = q2(r3);
        asm volatile("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_sqrt_f32 %2, %2\n\tv_sqrt_f32 %3, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        d = smooth_row(d, s0 - r0.w, k, hik);
        d = smooth_row(d, s1 - r1.w, k, hik);
        d = smooth_row(d, s2 - r2.w, k, hik);
        d = smooth_row(d, s3 - r3.w, k, hik);
      }
    }
    for (; i < n; i++) { const float4 r = c_rows[i]; d = smooth_row(d, __builtin_amdgcn_sqrtf(q2(r)) - r.w, k, hik); }
    z += 0.01f * d;
    acc += d;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  const int n = 64, evals = 2000;
  std::vector<float4> t(n);
  for (int i = 0; i < n; i++) t[i] = make_float4(0.9f * ((i & 3) - 1.5f), 0.9f * (((i >> 2) & 3) - 1.5f), 0.9f * ((i >> 4) - 1.5f), 0.3f + 0.002f * i);
  float4* d_t; float* d_out; unsigned long long* d_cyc;
  hipMalloc(&d_t, n * sizeof(float4)); hipMemcpy(d_t, t.data(), n * sizeof(float4), hipMemcpyHostToDevice);
  const int max_blocks = 256 * 4;
  hipMalloc(&d_out, (size_t)max_blocks * 512 * 4 * 4); hipMalloc(&d_cyc, max_blocks * 8 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  // instructions per evaluation and wave: 13 VALU per row (8 sphere + 5 union) + 1 v_mov per 4 rows; v_sqrt counted at 3.2 slots
  const double valu_per_eval = 64 * 13.0 + 16;
  auto run = [&](int mode, int blocks, int threads, int ev) {
    switch (mode) {
      case 0: fold<0><<<blocks, threads>>>(d_t, n, ev, 0.2f, d_out, d_cyc); break;
      case 1: fold<1><<<blocks, threads>>>(d_t, n, ev, 0.2f, d_out, d_cyc); break;
      case 2: fold<2><<<blocks, threads>>>(d_t, n, ev, 0.2f, d_out, d_cyc); break;
      case 3: fold<3><<<blocks, threads>>>(d_t, n, ev, 0.2f, d_out, d_cyc); break;
      case 4: fold<4><<<blocks, threads>>>(d_t, n, ev, 0.2f, d_out, d_cyc); break;
      default: fold<6><<<blocks, threads>>>(d_t, n, ev, 0.2f, d_out, d_cyc); break;
    }
  };
  const char* names[6] = {"the fold as the kernels run it", "an ordinary multiply in the square root's place", "no LDS read in the loop (the same four rows every trip)", "neither",
                          "the four square roots of a trip back to back", "... and no LDS read in the loop"};
  for (int mode = 0; mode < 6; mode++) {
    const double slots = valu_per_eval + ((mode & 1) ? 0.0 : 64 * 2.2);
    printf("== %s\n", names[mode]);
    for (int blocks_per_cu : {4, 2, 1}) {  // 8-wave workgroups: 8 / 4 / 2 waves per SIMD
      const int blocks = 256 * blocks_per_cu;
      run(mode, blocks, 512, 10);
      hipEventRecord(a);
      run(mode, blocks, 512, evals);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double waves = blocks * 8.0;
      printf("%d waves per SIMD: %.3f ms; %.3g wave-level VALU instructions/s, %.3g issue slots/s = %.2f of the ~1.0e12 a dense fp32 stream gets\n", blocks_per_cu * 2, ms,
             waves * evals * valu_per_eval / (ms * 1e-3), waves * evals * slots / (ms * 1e-3), waves * evals * slots / (ms * 1e-3) / 1.0e12);
    }
    hipEventRecord(a);
    run(mode, 256, 256, evals);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("1 wave per SIMD: %.3f ms; %.3g issue slots/s = %.2f\n", ms, 256 * 4.0 * evals * slots / (ms * 1e-3), 256 * 4.0 * evals * slots / (ms * 1e-3) / 1.0e12);
  }
  hipMemcpyToSymbol(HIP_SYMBOL(c_rows), t.data(), n * sizeof(float4));
  for (int trip : {4, 8}) {
    printf("== the rows through the scalar data cache (s_load, SGPR operands), roots grouped, %d rows per trip\n", trip);
    for (int blocks_per_cu : {4, 2, 1}) {
      const int blocks = 256 * blocks_per_cu;
      if (trip == 4) fold_scalar<4><<<blocks, 512>>>(n, 10, 0.2f, d_out); else fold_scalar<8><<<blocks, 512>>>(n, 10, 0.2f, d_out);
      hipEventRecord(a);
      if (trip == 4) fold_scalar<4><<<blocks, 512>>>(n, evals, 0.2f, d_out); else fold_scalar<8><<<blocks, 512>>>(n, evals, 0.2f, d_out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("%d waves per SIMD: %.3f ms\n", blocks_per_cu * 2, ms);
    }
  }
  return 0;
}
