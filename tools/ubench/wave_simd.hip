// Where do the waves of a workgroup sit?  One wave per 256-thread workgroup runs a dependent FMA chain, the other
// three wait at a barrier (what a compacted workgroup of the pixel kernel looks like).  If wave w always sat on
// SIMD w, "always wave 0" would use one SIMD of four.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const int worker = MODE == 0 ? 0 : MODE == 1 ? (blockIdx.x & 3) : MODE == 2 ? ((blockIdx.x >> 8) & 3) : -1;
  float a = threadIdx.x;
  const float m = 0.999f, c = 0.001f;
  if (worker < 0 || wave == worker)
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 32; u++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
    }
  __syncthreads();
  if (a == 12345.6f) out[0] = a;
}
template <int MODE>
void run(const char* name, float* d) {
  const int blocks = 256 * 8, iters = 4096;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 16);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves = MODE == 3 ? blocks * 4.0 : blocks;
  printf("%-52s %8.3f ms  %.3e wave-instr/s\n", name, ms, waves * iters * 32.0 / (ms * 1e-3));
}
int main() {
  float* d; (void)hipMalloc(&d, 4);
  run<3>("all four waves work", d);
  run<0>("wave 0 of every workgroup works", d);
  run<1>("wave (blockIdx & 3) works", d);
  run<2>("wave ((blockIdx >> 8) & 3) works", d);
  return 0;
}
