// Issue interval of fp32 VALU instructions on gfx950 in SHADER CYCLES (s_memtime), with the clock the chip actually
// holds under that load (s_memtime / s_memrealtime x 100 MHz), so that "cycles per instruction" is not confused with
// a lowered clock.  A second look at profiles/r01_valu_issue_rate.txt (which divided wall time by the nominal 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o issue_clock issue_clock.hip && ./issue_clock
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

struct Stamp {
  unsigned long long cyc, real;
};

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int MODE>
__global__ __launch_bounds__(256) void k(Stamp* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float m = 0.999f + seed * 1e-9f, c = 0.001f + seed * 1e-9f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 chains, three VGPR sources (the r01 microbenchmark)
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (MODE == 1) {  // 8 chains, multiplier and addend from SGPR / inline constant (one register-file read per instruction)
      REP8(asm volatile("v_fma_f32 %0, %0, %8, 1.0\n v_fma_f32 %1, %1, %8, 1.0\n v_fma_f32 %2, %2, %8, 1.0\n v_fma_f32 %3, %3, %8, 1.0\n"
                        "v_fma_f32 %4, %4, %8, 1.0\n v_fma_f32 %5, %5, %8, 1.0\n v_fma_f32 %6, %6, %8, 1.0\n v_fma_f32 %7, %7, %8, 1.0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));)
    } else if (MODE == 2) {  // VOP2 v_mul_f32, two VGPR sources
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                        "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (MODE == 3) {  // VOP2 v_mul_f32 with an SGPR source
      REP8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                        "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));)
    } else if (MODE == 4) {  // VOP2 v_fmac_f32 (a += b*c): two VGPR sources + accumulator
      REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                        "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (MODE == 5) {  // one dependent chain
      REP32(asm volatile("v_fma_f32 %0, %0, %1, 1.0\n v_fma_f32 %0, %0, %1, 1.0" : "+v"(a0) : "s"(m));)
    } else if (MODE == 6) {  // transcendentals, 8 chains
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rsq_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_log_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rsq_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_log_f32 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 8 || MODE == 9) {  // round 4: fp64 v_fma_f64 (8) / v_mul_f64 + v_add_f64 (9), 4 chains on register pairs
      double d0 = a0, d1 = a1, d2 = a2, d3 = a3, dm = m, dc = c;
      if (MODE == 8) {
        REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                          "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm), "v"(dc));)
      } else {
        REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %5\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %5\n"
                          "v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %5\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %5"
                          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm), "v"(dc));)
      }
      a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
    } else if (MODE == 7) {  // v_pk_fma_f32 on register pairs, SGPR-free
      typedef float float2v __attribute__((ext_vector_type(2)));
      float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));)
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if ((threadIdx.x & 63) == 0) {
    Stamp st{t1 - t0, r1 - r0};
    if (s == 12345.678f) st.cyc = 0;
    out[blockIdx.x * 4 + (threadIdx.x >> 6)] = st;
  }
}

template <int MODE>
void run(const char* name, Stamp* d, int blocks_per_cu, int instr_per_iter = 64) {
  const int iters = 4096, blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 64, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> h(blocks * 4);
  hipMemcpy(h.data(), d, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost);
  std::vector<double> cyc, ghz;
  for (auto& s : h) { cyc.push_back((double)s.cyc); ghz.push_back((double)s.cyc / (double)s.real * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
  const double n = (double)iters * instr_per_iter;
  const double waves_per_simd = blocks_per_cu;  // 4 waves per block, 4 SIMDs per CU
  printf("%-58s %d waves/SIMD  %7.3f ms  clock %.2f GHz  %.2f cycles/instr/wave = %.2f cycles/instr/SIMD  (%.3e wave-instr/s)\n", name,
         blocks_per_cu, ms, ghz[ghz.size() / 2], cyc[cyc.size() / 2] / n, cyc[cyc.size() / 2] / n / waves_per_simd,
         (double)blocks * 4 * n / (ms * 1e-3));
}

int main() {
  Stamp* d; hipMalloc(&d, sizeof(Stamp) * 256 * 8 * 4);
  for (int w : {8, 4, 2, 1}) {
    run<0>("v_fma_f32 v,v,v,v  8 chains", d, w);
    run<1>("v_fma_f32 v,v,s,1.0  8 chains", d, w);
    run<2>("v_mul_f32 v,v,v  8 chains", d, w);
    run<3>("v_mul_f32 v,s,v  8 chains", d, w);
    run<4>("v_fmac_f32 v,v,v  8 chains", d, w);
    run<5>("v_fma_f32 1 dependent chain", d, w);
    run<6>("v_rcp/rsq/sqrt/log_f32  8 chains", d, w);
    run<7>("v_pk_fma_f32  4 pair chains", d, w);
    run<8>("v_fma_f64  4 chains", d, w);
    run<9>("v_mul_f64 / v_add_f64  4 chains", d, w);
  }
  return 0;
}
