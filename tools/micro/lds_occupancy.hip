// Do two workgroups with 73 728 B of static LDS each share a CU on this device?  Every workgroup just waits ~100 us;
// 512 workgroups finish in ~100 us if two are resident per CU, ~200 us if not.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int FLOATS>
__global__ __launch_bounds__(256) void k_wait(float* out, int ticks) {
  __shared__ float lds[FLOATS];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  out[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x * 7) % FLOATS];
}
template <int FLOATS>
void run(const char* name) {
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  for (int blocks : {256, 512, 768, 1024}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_wait<FLOATS><<<blocks, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_wait<FLOATS><<<blocks, 256>>>(out, 10000);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s LDS, %4d workgroups: %.1f us\n", name, blocks, ms * 1e3);
  }
  hipFree(out);
}
int main() { run<18432>("73728 B"); run<17408>("69632 B"); run<9216>("36864 B"); return 0; }
