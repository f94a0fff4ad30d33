// How much would a head-major K/V cache buy the verify-step attention?  One wave per (sequence, head) streams the K and V
// pieces of its keys as the attention kernel does (float4 per lane, 32-key tiles) and sums them; position-major layout
// [seq][pos][H*32] (one 128-B piece per 1-KB row, today's) against head-major [seq][head][pos][32] (4 KB contiguous per tile).
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/kv_layout_bw.hip -o tools/micro/kv_layout_bw.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool HEAD_MAJOR>
__global__ __launch_bounds__(256) void k_read(const float* __restrict__ k, const float* __restrict__ v, float* out, int n_units, int H, int Lc,
                                              int keys) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (unit >= n_units) return;
  const int seq = unit / H, head = unit % H;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k0 = 0; k0 < keys; k0 += 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                      // 32 keys x 128 B = 4 x (64 lanes x 16 B)
      const int key = k0 + 8 * i + (lane >> 3), piece = (lane & 7) * 4;
      const size_t off = HEAD_MAJOR ? (((size_t)seq * H + head) * Lc + key) * 32 + piece
                                    : ((size_t)seq * Lc + key) * (H * 32) + head * 32 + piece;
      const float4 a = *reinterpret_cast<const float4*>(k + off);
      const float4 b = *reinterpret_cast<const float4*>(v + off);
      acc.x += a.x * b.x; acc.y += a.y * b.y; acc.z += a.z * b.z; acc.w += a.w * b.w;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[unit] = 1.f;
}

int main() {
  const int H = 8, Lc = 202, S = 640, layers = 16;     // four pools x four layers of caches are alive between two uses
  const size_t per = (size_t)S * Lc * H * 32;
  float *k, *v, *out;
  if (hipMalloc(&k, per * 4 * layers) != hipSuccess || hipMalloc(&v, per * 4 * layers) != hipSuccess || hipMalloc(&out, S * H * 4) != hipSuccess) return 1;
  (void)hipMemset(k, 0, per * 4 * layers); (void)hipMemset(v, 0, per * 4 * layers);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int keys : {64, 96, 160}) {
    for (int hm = 0; hm < 2; ++hm) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int l = 0; l < layers; ++l) {               // cycle through all caches so that none stays in the Infinity Cache
          if (hm) k_read<true><<<S * H / 4, 256>>>(k + l * per, v + l * per, out, S * H, H, Lc, keys);
          else k_read<false><<<S * H / 4, 256>>>(k + l * per, v + l * per, out, S * H, H, Lc, keys);
        }
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const double bytes = (double)layers * S * H * keys * 128 * 2;
      printf("%3d keys, %s: %.1f us per launch, %.2f TB/s\n", keys, hm ? "head-major    " : "position-major", best * 1e3 / layers, bytes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
