// Latency of a chain of dependent v_mfma_f32_32x32x2_f32 on one wave, on an otherwise idle GPU and right after a kernel that
// loads every CU — does a launch of a few waves run at a lower clock?  (tools/micro: probes, not product code.)
// hipcc --offload-arch=gfx950 -O3 -o mfma_chain.bin mfma_chain.hip && ./mfma_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void chain(float* out, unsigned long long* ticks, int n, int indep) {
  f32x16 a, b, c, d;
  for (int i = 0; i < 16; ++i) { a[i] = 0.f; b[i] = 0.f; c[i] = 0.f; d[i] = 0.f; }
  const float x = (float)threadIdx.x * 1e-3f, y = 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (indep == 1) {
    for (int i = 0; i < n; ++i) a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
  } else {
    for (int i = 0; i < n; i += 4) {
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
      b = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, b, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, d, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + b[i] + c[i] + d[i];
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 4 * 64 * 4096); hipMalloc(&ticks, 8 * 4096);
  const int n = 4096;
  unsigned long long h[4];
  auto run = [&](const char* what, int blocks, int indep) {
    hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, out, ticks, n, indep);
    hipDeviceSynchronize();
    hipMemcpy(h, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.1f ns per MFMA (100 MHz realtime ticks: %llu for %d MFMAs)\n", what, 10.0 * h[0] / n, h[0], n);
  };
  run("one wave, dependent chain, cold", 1, 1);
  run("one wave, dependent chain, again", 1, 1);
  run("one wave, four independent chains", 1, 4);
  run("4096 waves, dependent chain (loads the chip)", 4096, 1);
  run("one wave, dependent chain, right after the loaded launch", 1, 1);
  run("one wave, four independent chains, after", 1, 4);
  for (int i = 0; i < 3; ++i) run("one wave, dependent chain, repeated", 1, 1);
  return 0;
}
