// fp32 MFMA ceiling with REAL operand values: like mfma_peak.hip (independent v_mfma_f32_32x32x2_f32 back to back, no memory
// traffic in the loop), but the operands are 16 + 16 register-resident values per lane taken from a random buffer (uniform in
// [-1,1) / [-0.1,0.1), as the GEMM operands of the verify step are), cycled through — so the multipliers toggle as they do in
// a GEMM.  Compares against the same loop on constant operands.   hipcc --offload-arch=gfx950 -O3 mfma_peak_data.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool RANDOM>
__global__ __launch_bounds__(256) void k_peak(const float* src, float* out, long long* clk, int iters) {
  f32x16 c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
  float a[16], b[16];
  for (int i = 0; i < 16; ++i) {
    a[i] = RANDOM ? src[(threadIdx.x * 16 + i) & 65535] : 1e-3f;
    b[i] = RANDOM ? 0.1f * src[(threadIdx.x * 16 + i + 4096 + blockIdx.x) & 65535] : 1e-4f;
  }
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u + 1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1], b[u], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1], b[u + 1], c3, 0, 0, 0);
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <bool RANDOM>
void run(const float* src, int wgs_per_cu, int iters) {
  const int blocks = 256 * wgs_per_cu;
  float* out; long long* clk;
  hipMalloc(&out, blocks * 256 * 4);
  hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_peak<RANDOM><<<blocks, 256>>>(src, out, clk, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_peak<RANDOM><<<blocks, 256>>>(src, out, clk, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double flops = (double)blocks * 4 * iters * 32.0 * 4096.0;
  printf("%s operands, %d workgroup(s) of 4 waves per CU: %.1f ms  %.1f TFLOP/s; shader clock %.0f MHz; cycles per MFMA per wave %.1f\n",
         RANDOM ? "random  " : "constant", wgs_per_cu, ms, flops / (ms * 1e-3) / 1e12, h[0] / (h[1] / 100.0), (double)h[0] / (iters * 32.0));
  hipFree(out); hipFree(clk);
}

int main() {
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  float* src; hipMalloc(&src, h.size() * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep)
    for (int w : {1, 2}) { run<false>(src, w, 20000); run<true>(src, w, 20000); }
  return 0;
}
