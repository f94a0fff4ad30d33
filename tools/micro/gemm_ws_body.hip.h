// W-stationary tiling of the K = 256 GEMMs of the verify step (QKV, d x d, FFN1):  Y[m, n] = sum_k X[m, k] * W[n, k].
//
// A workgroup keeps ONE 64-column strip of W (64 x 256 fp32 = 64 KB, row stride 260 floats) in LDS for its whole life and
// its four waves stream 32-row blocks of X past it: a wave loads its own block straight from global memory into the
// registers the MFMA reads (lane (r, h) owns row r and the k's 8g+4h .. 8g+4h+3 — one float4 per 8 k's, exactly the
// operand pairing of the LDS-tiled kernels), takes the W fragments from LDS, and stores its 32 x 64 result.  No X tile is
// shared between waves, so after the one barrier that publishes the strip there is no synchronisation at all: the waves of
// a CU drift apart and one wave's loads and stores overlap another's MFMAs.  A wave that has more than one block (long
// row counts, wide N) requests the next block's first chunk before it stores the current result.
//
// Same numbers as every other tiling of the family (ttx_kernels.hip.h g2_body / g4_body): per output the k's are
// contracted in ascending groups of eight, pairs (0,4) (1,5) (2,6) (3,7) per v_mfma_f32_32x32x2_f32, one accumulator
// from k = 0 to 255, then fmaxf(acc + bias, relu ? 0 : -inf).
//
// An experiment: bit-identical to the production tilings and no faster, so it is not part of the library.
// Expects ttx::GemmArgs and ttx::f32x16 to be defined (the micro harness tools/micro/gemm_ws.hip copies them).
#pragma once

namespace ttx {

constexpr int GWS_K = 256, GWS_LDW = GWS_K + 4, GWS_BN = 64;
constexpr int GWS_SMEM_FLOATS = GWS_BN * GWS_LDW;       // 66 560 B

typedef float gws_f4 __attribute__((ext_vector_type(4)));

// `lin` = linear index of this workgroup among the `n_wgs` (256-thread) workgroups of the launch that take part.
// Requires K == 256 == k_per_split, N % 64 == 0, ldx/ldw % 4 == 0 and n_wgs >= N / 64.
template <int WAVES = 4, bool DEEP = true>
__device__ __forceinline__ void gws_body(const GemmArgs& a, const int M, const int lin, const int n_wgs, float* wl) {
  const int n_strips = a.N >> 6;
  const int n_grp = n_wgs / n_strips;                   // workgroups per strip
  if (lin >= n_grp * n_strips) return;
#ifdef GWS_NO_XCD_MAP
  const int strip = lin % n_strips, grp = lin / n_strips;
#else
  // Workgroups lin, lin + 8, ... share an XCD and its L2: give one XCD all strips of the SAME row blocks, so that a block
  // of X comes into that L2 once and is read from there by the other strips (W is small and every XCD reads all of it).
  int strip, grp;
  {
    const int n_all = n_grp * n_strips, xcd = lin & 7, i = lin >> 3;
    const int g8 = n_grp >> 3;                          // whole groups of eight row-block groups
    if (lin < g8 * 8 * n_strips) { strip = i % n_strips; grp = (i / n_strips) * 8 + xcd; }
    else { const int rest = lin - g8 * 8 * n_strips; strip = rest % n_strips; grp = g8 * 8 + rest / n_strips; (void)n_all; }
  }
#endif
  const int n_rb = (M + 31) >> 5;                       // 32-row blocks
  if (grp * WAVES >= n_rb) return;                      // nothing for any wave of this workgroup (uniform)
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = strip * GWS_BN;
  const int stride = n_grp * WAVES;
  int rb = grp * WAVES + wave;

  // The four 64-k chunks of a block live in four fixed register buffers; the moment a chunk has been consumed its buffer is
  // refilled with the same chunk of the wave's NEXT block, so every load is three chunks (about 10 us) ahead of its use —
  // X comes from HBM / the Infinity Cache (16 MB do not fit an XCD's L2) — and the stores of a block are never waited
  // for: vmcnt retires in order, and everything the next chunk needs was requested BEFORE those stores.
  gws_f4 c0[8], c1[8], c2[8], c3[8];
  auto xload = [&](gws_f4 (&c)[8], const float* xp, int k0) {
#ifdef GWS_TIMING_NO_XLOAD         // timing experiment only (wrong results): X operands from registers
#pragma unroll
    for (int g = 0; g < 8; ++g) asm volatile("" : "+v"(c[g]));
    (void)xp; (void)k0;
    return;
#endif
#pragma unroll
    for (int g = 0; g < 8; ++g) c[g] = *reinterpret_cast<const gws_f4*>(xp + k0 + 8 * g);
  };
  const float* xp = a.X + (size_t)min(min(rb, n_rb - 1) * 32 + r, M - 1) * a.ldx + 4 * h;   // idle waves read a valid block
  xload(c0, xp, 0);
  if constexpr (DEEP) { xload(c1, xp, 64); xload(c2, xp, 128); xload(c3, xp, 192); }

  // the strip: 64 columns x 64 float4, consecutive threads along k (coalesced), 16 float4 per thread
  {
    const float* wsrc = a.W + (size_t)n0 * a.ldw;
#pragma unroll
    for (int i = 0; i < 64 / WAVES; ++i) {
      const int idx = t + 64 * WAVES * i;
      const int col = idx >> 6, k4 = idx & 63;
      *reinterpret_cast<gws_f4*>(&wl[col * GWS_LDW + 4 * k4]) = *reinterpret_cast<const gws_f4*>(wsrc + (size_t)col * a.ldw + 4 * k4);
    }
  }
  __syncthreads();

  const char* lds_base = reinterpret_cast<const char*>(wl);
  const unsigned w0_off = (unsigned)(r * GWS_LDW + 4 * h) * 4u;
  const unsigned w1_off = (unsigned)((32 + r) * GWS_LDW + 4 * h) * 4u;
  const float lo = a.relu ? 0.f : -INFINITY;
  const float bv0 = (!a.raw && a.bias) ? a.bias[n0 + r] : 0.f;
  const float bv1 = (!a.raw && a.bias) ? a.bias[n0 + 32 + r] : 0.f;
  float* Y = a.Y;                                       // raw partial sums of an unsplit K go to slab 0

  auto chunk_mma = [&](f32x16& acc0, f32x16& acc1, const gws_f4 (&c)[8], int k0) {
    // The strip never changes after the barrier, so to the compiler every W fragment is loop-invariant: it would read all
    // 256 values per lane ahead of the loop and spill them.  An address it cannot see through pins the reads to their chunk.
    unsigned o0 = w0_off + 4 * k0, o1 = w1_off + 4 * k0;        // LDS byte addresses, made opaque per chunk
    asm volatile("" : "+v"(o0), "+v"(o1) :: "memory");
    const float* p0 = reinterpret_cast<const float*>(lds_base + o0);
    const float* p1 = reinterpret_cast<const float*>(lds_base + o1);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#ifdef GWS_TIMING_NO_LDS           // timing experiment only (wrong results): no W fragment reads
      const gws_f4 b0 = c[(g + 1) & 7], b1 = c[(g + 2) & 7];
      (void)p0; (void)p1;
#else
      const gws_f4 b0 = *reinterpret_cast<const gws_f4*>(p0 + 8 * g);
      const gws_f4 b1 = *reinterpret_cast<const gws_f4*>(p1 + 8 * g);
#endif
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c[g].x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c[g].x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c[g].y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c[g].y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c[g].z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c[g].z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c[g].w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c[g].w, b1.w, acc1, 0, 0, 0);
    }
  };

  if constexpr (!DEEP) {
    // first version of the experiment: two buffers, one chunk ahead, stores straight after the last chunk (fewer registers:
    // four waves per SIMD fit)
    for (; rb < n_rb; rb += stride) {
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
      const int m0 = rb * 32;
      const float* xc = a.X + (size_t)min(m0 + r, M - 1) * a.ldx + 4 * h;
      if (rb == grp * WAVES + wave) { /* c0 holds chunk 0 of the first block already */ } else xload(c0, xc, 0);
      xload(c1, xc, 64);
      chunk_mma(acc0, acc1, c0, 0);
      xload(c0, xc, 128);
      chunk_mma(acc0, acc1, c1, 64);
      xload(c1, xc, 192);
      chunk_mma(acc0, acc1, c0, 128);
      chunk_mma(acc0, acc1, c1, 192);
      float* yp = Y + (size_t)(m0 + 4 * h) * a.ldy + n0 + r;
      const int rows_left = M - (m0 + 4 * h);
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int dr = (v & 3) + 8 * (v >> 2);
        if (dr < rows_left) {
          yp[(size_t)dr * a.ldy] = fmaxf(acc0[v] + bv0, lo);
          yp[(size_t)dr * a.ldy + 32] = fmaxf(acc1[v] + bv1, lo);
        }
      }
    }
    return;
  }
  // Rows this launch may write: with a device-side row count the buffers hold a.M rows (the capacity), so a block that
  // starts below the live count is stored whole when it fits the capacity — rows past the live count are dead to every
  // consumer.  Only such blocks run in the pipelined loop: a second, predicated store path inside it would make the
  // compiler count the outstanding stores conservatively and drain them before every block.
  const int cap = a.m_ptr ? a.M : M;
  const int n_whole = min(n_rb, cap >> 5);
  // vmcnt counts loads and stores together and retires in order, and the compiler's wait for a buffer is "at most as many
  // operations outstanding as were issued after its load" on the cheapest path into that point.  `arrived` makes it wait
  // for chunk 0 of the NEXT block just BEFORE a block's 32 stores go out (and before the loop), when that costs nothing;
  // the first wait after the stores is then for chunk 1, one chunk of MFMAs (about 3 us) later.
  auto arrived = [&](gws_f4 (&c)[8]) {
    asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
  };
  arrived(c0);
  for (; rb < n_whole; rb += stride) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int m0 = rb * 32;
    const int nrb = min(rb + stride, n_rb - 1);         // past the last block: re-read a valid one (unconditional loads
    const float* xn = a.X + (size_t)min(nrb * 32 + r, M - 1) * a.ldx + 4 * h;   //  keep the waits counted, not drained)
    chunk_mma(acc0, acc1, c0, 0);
    xload(c0, xn, 0);
    chunk_mma(acc0, acc1, c1, 64);
    xload(c1, xn, 64);
    chunk_mma(acc0, acc1, c2, 128);
    xload(c2, xn, 128);
    chunk_mma(acc0, acc1, c3, 192);
    xload(c3, xn, 192);
    float* yp = Y + (size_t)(m0 + 4 * h) * a.ldy + n0 + r;
    arrived(c0);
#ifdef GWS_TIMING_NO_STORE         // timing experiment only: one value per lane leaves
    if (acc0[0] + acc1[5] == 12345.678f) yp[0] = 1.f;
    continue;
#endif
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const size_t off = (size_t)((v & 3) + 8 * (v >> 2)) * a.ldy;
      yp[off] = fmaxf(acc0[v] + bv0, lo);
      yp[off + 32] = fmaxf(acc1[v] + bv1, lo);
    }
  }
  if (rb < n_rb) {                                      // the one block that crosses the end of the buffer: row by row
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int m0 = rb * 32;
    const float* xl = a.X + (size_t)min(m0 + r, M - 1) * a.ldx + 4 * h;
    xload(c0, xl, 0); xload(c1, xl, 64); xload(c2, xl, 128); xload(c3, xl, 192);
    chunk_mma(acc0, acc1, c0, 0);
    chunk_mma(acc0, acc1, c1, 64);
    chunk_mma(acc0, acc1, c2, 128);
    chunk_mma(acc0, acc1, c3, 192);
    float* yp = Y + (size_t)(m0 + 4 * h) * a.ldy + n0 + r;
    const int rows_left = M - (m0 + 4 * h);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int dr = (v & 3) + 8 * (v >> 2);
      if (dr < rows_left) {
        yp[(size_t)dr * a.ldy] = fmaxf(acc0[v] + bv0, lo);
        yp[(size_t)dr * a.ldy + 32] = fmaxf(acc1[v] + bv1, lo);
      }
    }
  }
}

}  // namespace ttx
