// Standalone check and timing of the W-stationary K = 256 GEMM body (tools/micro/gemm_ws_body.hip.h):
// bit-for-bit against a plain one-wave-per-tile kernel that contracts the k's in the family's canonical order, then
// microseconds and TFLOP/s per shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/micro/gemm_ws.hip -o tools/micro/gemm_ws.bin
//   tools/micro/gemm_ws.bin [M ...]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ttx {
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct GemmArgs {            // as in ttx_kernels.hip.h
  const float* X; int ldx;
  const float* W; int ldw;
  const float* bias;
  float* Y; int ldy;
  const int* m_ptr;
  int M, N, K;
  int k_per_split;
  int relu;
  int raw;
  long long slab_stride;
  unsigned long long* dbg;
  int big_min_tiles;
  int big_wide_tiles;
};
}  // namespace ttx
#include "gemm_ws_body.hip.h"
using namespace ttx;

#ifndef WS_WAVES
#define WS_WAVES 4
#endif
#ifndef WS_DEEP
#define WS_DEEP true
#endif
__global__ __launch_bounds__(64 * WS_WAVES, 2) void k_ws(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  gws_body<WS_WAVES, WS_DEEP>(a, M, blockIdx.x, gridDim.x, smem);
}

// canonical order, nothing else: one wave per 32x32 tile, operands straight from global memory
__global__ __launch_bounds__(64) void k_ref(GemmArgs a) {
  const int M = a.M;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const float* xp = a.X + (size_t)min(m0 + r, M - 1) * a.ldx + 4 * h;
  const float* wp = a.W + (size_t)(n0 + r) * a.ldw + 4 * h;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k = 0; k < a.K; k += 8) {
    const float4 av = *reinterpret_cast<const float4*>(xp + k);
    const float4 bv = *reinterpret_cast<const float4*>(wp + k);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
  }
  const float bv = (!a.raw && a.bias) ? a.bias[n0 + r] : 0.f;
  const float lo = a.relu ? 0.f : -INFINITY;
  for (int v = 0; v < 16; ++v) {
    const int row = m0 + (v & 3) + 8 * (v >> 2) + 4 * h;
    if (row < M) a.Y[(size_t)row * a.ldy + n0 + r] = fmaxf(acc[v] + bv, lo);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  std::vector<int> Ms;
  for (int i = 1; i < argc; ++i) Ms.push_back(atoi(argv[i]));
  if (Ms.empty()) Ms = {15872, 7936, 4960, 2480, 1000, 33};
  const int K = 256;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ws), hipFuncAttributeMaxDynamicSharedMemorySize, GWS_SMEM_FLOATS * 4));
  const int maxM = 16384, maxN = 2048;
  float *X, *W, *B, *Y0, *Y1;
  CK(hipMalloc(&X, (size_t)maxM * K * 4)); CK(hipMalloc(&W, (size_t)maxN * K * 4)); CK(hipMalloc(&B, maxN * 4));
  CK(hipMalloc(&Y0, (size_t)maxM * maxN * 4)); CK(hipMalloc(&Y1, (size_t)maxM * maxN * 4));
  {
    std::vector<float> hx((size_t)maxM * K), hw((size_t)maxN * K), hb(maxN);
    srand(1);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f);
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int M : Ms) {
    if (M > maxM) continue;
    for (int N : {256, 768, 2048}) {
      for (int mode = 0; mode < 2; ++mode) {        // 0: bias + relu, 1: raw
        GemmArgs a{};
        a.X = X; a.ldx = K; a.W = W; a.ldw = K; a.bias = B; a.ldy = N; a.M = M; a.N = N; a.K = K; a.k_per_split = K;
        a.relu = mode == 0; a.raw = mode == 1;
        CK(hipMemset(Y0, 0xff, (size_t)M * N * 4)); CK(hipMemset(Y1, 0xff, (size_t)M * N * 4));
        a.Y = Y0;
        k_ref<<<dim3(N / 32, (M + 31) / 32), 64>>>(a);
        a.Y = Y1;
        const int n_strips = N / 64, n_rb = (M + 31) / 32;
        const int n_grp = std::max(1, std::min((n_rb + WS_WAVES - 1) / WS_WAVES, 512 / n_strips));
        const int n_wgs = n_strips * n_grp;
        k_ws<<<n_wgs, 64 * WS_WAVES, GWS_SMEM_FLOATS * 4>>>(a);
        CK(hipDeviceSynchronize());
        std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
        CK(hipMemcpy(h0.data(), Y0, h0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), Y1, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) bad += std::memcmp(&h0[i], &h1[i], 4) != 0;
        const int iters = 50;
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) k_ws<<<n_wgs, 64 * WS_WAVES, GWS_SMEM_FLOATS * 4>>>(a);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("M=%5d N=%4d %s: %zu of %zu values differ; %d workgroups; %.1f us  %.1f TFLOP/s\n", M, N, mode ? "raw " : "relu", bad, h0.size(),
               n_wgs, us, 2.0 * M * N * K / (us * 1e-6) / 1e12);
        fflush(stdout);
      }
    }
  }
  return 0;
}
