// Do VOP3 output modifiers (mul:2) take effect once a wave has cleared MODE.IEEE and the fp32 denormal bits with
// s_setreg?  (The kernel descriptor of a HIP kernel starts with IEEE = 1 and denormals kept; the ISA ignores omod in
// that state.)   hipcc --offload-arch=gfx950 -O3 tools/ubench/omod_probe.hip -o tools/ubench/omod_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float* in, float* out, int set_mode) {
  if (set_mode & 1) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);  // MODE[5:4] fp32 denormals: flush
  if (set_mode & 2) __builtin_amdgcn_s_setreg(1 | (9 << 6) | (0 << 11), 0);  // MODE[9] IEEE: off
  const float a = in[threadIdx.x], b = in[threadIdx.x + 64];
  float m, f;
  asm volatile("v_mul_f32_e64 %0, %1, %2 mul:2" : "=v"(m) : "v"(a), "v"(b));
  asm volatile("v_fma_f32 %0, %1, %1, -%2 mul:2" : "=v"(f) : "v"(a), "v"(b));
  out[threadIdx.x] = m;
  out[threadIdx.x + 64] = f;
  out[threadIdx.x + 128] = a * 1e-30f * 1e-10f;  // a denormal result when kept
  out[threadIdx.x + 192] = fminf(a, __builtin_nanf(""));
}
int main() {
  float h[128], r[256], *din, *dout;
  for (int i = 0; i < 128; i++) h[i] = 0.37f + 0.011f * i;
  hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof r);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 4; mode++) {
    probe<<<1, 64>>>(din, dout, mode);
    hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    int ok_m = 0, ok_f = 0;
    for (int i = 0; i < 64; i++) {
      ok_m += r[i] == 2.0f * (h[i] * h[i + 64]);
      ok_f += r[i + 64] == 2.0f * __builtin_fmaf(h[i], h[i], -h[i + 64]);
    }
    printf("mode %d (1 = denormals flushed, 2 = IEEE off): mul:2 applied on %d/64, fma mul:2 on %d/64; sample %g vs unscaled %g; denormal product %g; min(a, NaN) %g\n",
           mode, ok_m, ok_f, r[0], h[0] * h[64], r[128], r[192]);
  }
  return 0;
}
