// Issue rate of v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950 (one MI355X).
// hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float m = 0.999f, c = 0.001f;
  const float2v mm = {m, m}, cc = {c, c};
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(m), "v"(c));
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(mm), "v"(cc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(mm), "v"(cc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(mm), "v"(cc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(mm), "v"(cc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(mm), "v"(cc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(mm), "v"(cc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(mm), "v"(cc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(mm), "v"(cc));
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(mm));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(cc));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2) : "v"(mm));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(cc));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p4) : "v"(mm));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p5) : "v"(cc));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p6) : "v"(mm));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p7) : "v"(cc));
      }
    } else if (MODE == 3) {  // quarter-rate candidates
#pragma unroll
      for (int u = 0; u < 4; u++) {
        asm volatile("v_sqrt_f32 %0, %0" : "+v"(a0));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(a1));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(a2));
        asm volatile("v_log_f32 %0, %0" : "+v"(a3));
        asm volatile("v_sqrt_f32 %0, %0" : "+v"(a4));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(a5));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(a6));
        asm volatile("v_log_f32 %0, %0" : "+v"(a7));
      }
    } else if (MODE == 5) {  // one dependent chain per wave
#pragma unroll
      for (int u = 0; u < 32; u++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
    } else if (MODE == 6) {  // two chains
#pragma unroll
      for (int u = 0; u < 16; u++) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
      }
    } else if (MODE == 7) {  // four chains
#pragma unroll
      for (int u = 0; u < 8; u++) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(m), "v"(c));
      }
    } else if (MODE == 8) {  // one chain, VOP2 forms
#pragma unroll
      for (int u = 0; u < 16; u++) {
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(m));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c));
      }
    } else {  // v_mul / v_add / v_cndmask / v_max mix
#pragma unroll
      for (int u = 0; u < 4; u++) {
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(m));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(c));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a2) : "v"(c));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(m));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(c));
        asm volatile("v_min_f32 %0, %0, %1" : "+v"(a5) : "v"(m));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(m));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(c));
      }
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* d, double per_instr_flops, int blocks = 256 * 8) {
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 16, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instrs = (double)blocks * 4 /*waves*/ * iters * 32.0;
  printf("%-44s %8.3f ms  %.3e wave-instr/s  %.1f cycles/instr/SIMD @2.4GHz  %.1f TFLOP/s\n", name, ms, instrs / (ms * 1e-3),
         1024 * 2.4e9 / (instrs / (ms * 1e-3)), instrs * 64 * per_instr_flops / (ms * 1e-3) / 1e12);
}

int main() {
  float* d; hipMalloc(&d, 4);
  run<0>("v_fma_f32", d, 2);
  run<1>("v_pk_fma_f32", d, 4);
  run<2>("v_pk_mul_f32 / v_pk_add_f32", d, 2);
  run<3>("v_sqrt/rsq/rcp/log_f32", d, 1);
  run<4>("v_mul/add/max/min_f32", d, 1);
  run<5>("v_fma_f32, 1 dependent chain per wave", d, 2);
  run<6>("v_fma_f32, 2 chains", d, 2);
  run<7>("v_fma_f32, 4 chains", d, 2);
  run<8>("v_mul/v_add, 1 dependent chain", d, 1);
  run<0>("v_fma_f32 8 chains, 4 waves/SIMD", d, 2, 256 * 4);
  run<5>("v_fma_f32 1 chain, 4 waves/SIMD", d, 2, 256 * 4);
  run<0>("v_fma_f32 8 chains, 2 waves/SIMD", d, 2, 256 * 2);
  run<5>("v_fma_f32 1 chain, 2 waves/SIMD", d, 2, 256 * 2);
  run<0>("v_fma_f32 8 chains, 1 wave/SIMD", d, 2, 256);
  run<5>("v_fma_f32 1 chain, 1 wave/SIMD", d, 2, 256);
  return 0;
}
