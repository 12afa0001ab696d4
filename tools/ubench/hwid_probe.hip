// Where do the waves of a workgroup sit?  Every wave of 512-thread workgroups (the headline kernel's shape, 64 VGPRs,
// 24 KB of LDS: 4 per CU) records HW_REG_HW_ID and XCC_ID; the host prints, for a few workgroups, wave -> (SIMD, CU, SE)
// and the histogram of "SIMD of wave w" over all workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/hwid_probe.hip -o tools/ubench/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512, 8) void probe(unsigned int* out, int spin) {
  __shared__ float pad[6144];  // 24 KB like the pixel kernel
  unsigned int hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = threadIdx.x * 0.001f;
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;  // stay resident so that 4 workgroups share a CU
  pad[threadIdx.x] = a;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = (xcc & 0xf) | (pad[(threadIdx.x + 7) & 511] > 1e30f ? 16 : 0);
  }
}
int main() {
  const int blocks = 2048;
  unsigned int* d;
  hipMalloc(&d, blocks * 8 * 2 * 4);
  probe<<<blocks, 512>>>(d, 20000);
  std::vector<unsigned int> h(blocks * 16);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  for (int b : {0, 1, 2, 3, 500, 1500}) {
    printf("workgroup %4d:", b);
    for (int w = 0; w < 8; w++) {
      const unsigned int hw = h[(b * 8 + w) * 2], x = h[(b * 8 + w) * 2 + 1];
      printf("  w%d simd %u cu %2u sh %u se %u xcc %u waveid %2u |", w, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, x & 15, hw & 15);
    }
    printf("\n");
  }
  int hist[8][4] = {};
  int pattern_same = 0;
  for (int b = 0; b < blocks; b++) {
    bool same = true;
    for (int w = 0; w < 8; w++) {
      const int s = (h[(b * 8 + w) * 2] >> 4) & 3;
      hist[w][s]++;
      same = same && s == (w & 3);
    }
    pattern_same += same;
  }
  for (int w = 0; w < 8; w++) printf("wave %d on SIMD 0..3: %5d %5d %5d %5d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  printf("workgroups with wave w on SIMD w %% 4: %d of %d\n", pattern_same, blocks);
  printf("raw hw_id of workgroup 0: ");
  for (int w = 0; w < 8; w++) printf("%08x ", h[w * 2]);
  printf("\n");
  return 0;
}
