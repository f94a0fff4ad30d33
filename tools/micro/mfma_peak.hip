// Practical fp32 MFMA ceiling of the device: every wave issues independent v_mfma_f32_32x32x2_f32 back to back
// (no memory traffic), WAVES_PER_SIMD waves per SIMD.  Prints TFLOP/s and the shader clock derived from
// clock64() (shader cycles) against wall_clock64() (100 MHz).   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_peak(float* out, long long* clk, int iters) {
  f32x16 c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 1.f; c2[i] = 2.f; c3[i] = 3.f; }
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  const int iters = 20000;
  for (int wgs_per_cu : {1, 2}) {
    const int blocks = 256 * wgs_per_cu;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_peak<<<blocks, 256>>>(out, clk, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_peak<<<blocks, 256>>>(out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 16.0 * 4096.0;
    printf("%d workgroup(s) of 4 waves per CU: %.1f ms  %.1f TFLOP/s;  shader cycles %lld over %.1f us -> %.0f MHz; cycles per MFMA per wave %.1f\n",
           wgs_per_cu, ms, flops / (ms * 1e-3) / 1e12, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0), (double)h[0] / (iters * 16.0));
    hipFree(out); hipFree(clk);
  }
  return 0;
}
