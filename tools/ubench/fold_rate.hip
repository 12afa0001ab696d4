// The 64-row smooth-union sphere fold of CSG-64 (rm_device.hpp eval_spheres_one_k) ALONE: every wave folds the whole table from LDS for a
// fixed number of evaluations -- no march, no divergence, no repack, no culling -- at 8 / 4 / 2 / 1 waves per SIMD.  What fraction of the
// chip's issue slots does the fold's instruction mix take when nothing else is in the way?  (C4 / C5 run at 0.66 / 0.68: is the missing
// third the mix -- one quarter-rate v_sqrt per 13 instructions, a dependent chain of 5 per row, one LDS read per row -- or the kernel around it?)
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o fold_rate fold_rate.hip && ./fold_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__device__ __forceinline__ float sphere_row1(const float4 r, float x, float y, float z) {
  const float qx = x - r.x, qy = y - r.y, qz = z - r.z;
  return __builtin_amdgcn_sqrtf(__builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx))) - r.w;
}
__device__ __forceinline__ float smooth_row(float d, float di, float k, float hik) {
  const float t = di - d;
  const float h = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(hik, t, 0.5f), 0.0f), 1.0f);
  return __builtin_fmaf(-h, __builtin_fmaf(k, 1.0f - h, t), di);
}

template <int ROWS_PER_TRIP>
__global__ __launch_bounds__(512, 8) void fold(const float4* table, int n, int evals, float k, float* out, unsigned long long* cyc) {
  __shared__ float4 rows[256];
  for (int i = threadIdx.x; i < n; i += blockDim.x) rows[i] = table[i];
  __syncthreads();
  float x = 0.001f * threadIdx.x - 0.3f, y = 0.002f * blockIdx.x - 0.5f, z = -2.0f;
  const float hik = 0.5f / k;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float acc = 0.0f;
  for (int e = 0; e < evals; e++) {
    float d = sphere_row1(rows[0], x, y, z);
    int i = 1;
    for (; i + 3 < n; i += 4) {
      const float4 r0 = rows[i], r1 = rows[i + 1], r2 = rows[i + 2], r3 = rows[i + 3];
      const float d0 = sphere_row1(r0, x, y, z), d1 = sphere_row1(r1, x, y, z);
      d = smooth_row(d, d0, k, hik);
      d = smooth_row(d, d1, k, hik);
      const float d2 = sphere_row1(r2, x, y, z), d3 = sphere_row1(r3, x, y, z);
      d = smooth_row(d, d2, k, hik);
      d = smooth_row(d, d3, k, hik);
    }
    for (; i < n; i++) d = smooth_row(d, sphere_row1(rows[i], x, y, z), k, hik);
    z += 0.01f * d;  // the march's dependence of the next point on this distance
    acc += d;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
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
  const double valu_per_eval = 64 * 13.0 + 16, slots_per_eval = valu_per_eval + 64 * 2.2;
  for (int blocks_per_cu : {4, 2, 1}) {  // 8-wave workgroups: 8 / 4 / 2 waves per SIMD
    const int blocks = 256 * blocks_per_cu;
    fold<4><<<blocks, 512>>>(d_t, n, 10, 0.2f, d_out, d_cyc);
    hipEventRecord(a);
    fold<4><<<blocks, 512>>>(d_t, n, evals, 0.2f, d_out, d_cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waves = blocks * 8.0;
    printf("%d waves per SIMD: %.3f ms; %.3g wave-level VALU instructions/s, %.3g issue slots/s (a dense fp32 stream: ~1.0e12) = %.2f of them\n", blocks_per_cu * 2, ms,
           waves * evals * valu_per_eval / (ms * 1e-3), waves * evals * slots_per_eval / (ms * 1e-3), waves * evals * slots_per_eval / (ms * 1e-3) / 1.0e12);
  }
  // 256-thread workgroups, one per SIMD quadruple: 1 wave per SIMD
  {
    const int blocks = 256;
    hipEventRecord(a);
    fold<4><<<blocks, 256>>>(d_t, n, evals, 0.2f, d_out, d_cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waves = blocks * 4.0;
    printf("1 wave per SIMD: %.3f ms; %.3g issue slots/s = %.2f\n", ms, waves * evals * slots_per_eval / (ms * 1e-3), waves * evals * slots_per_eval / (ms * 1e-3) / 1.0e12);
  }
  return 0;
}
