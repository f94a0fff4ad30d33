// GEMM translation unit of libttx_hip.so: the fp32-MFMA GEMM family (v_mfma_f32_32x32x2_f32), the slab-sum + bias +
// residual + LayerNorm finisher, and their host-side launchers.  SURVEY.md §2.3 K2/K5/K6/K7/K8.
//
// ONE ARITHMETIC FOR EVERY TILING (the canonical slice order).  Y[m, n] = sum_k X[m, k] W[n, k] is DEFINED as
//     Y = (((S_0 + S_1) + S_2) + ... ) + S_{K/s - 1},      S_j = the fp32 MFMA chain over k in [j s, (j + 1) s) started from 0,
// with the slice length s = gemm_slice_k(K): 64 for the K = d contractions (QKV, the three d x d projections, FFN1, the
// classifier), 256 for FFN2 (K = F >= 2048).  Inside a slice every kernel feeds the MFMA the same k pairing (lanes 0-31 the
// k's 8g..8g+3, lanes 32-63 the k's 8g+4..8g+7, in ascending g), so S_j is the same number whichever kernel forms it.
// What differs between the kernels is only WHO adds the slices:
//   * k_gemm24 (128x128 / 128x64 / 64x64 tiles picked from the live row count): one workgroup walks all slices of its tile,
//     finishing each in a fresh accumulator and folding it into the running total with fp32 adds, in slice order;
//   * k_gemm3 (32x32 tiles, step GEMMs of a few hundred rows): the four waves of a workgroup take the four slices of
//     K = 256 — four 0.85 us chains side by side instead of one of 3.4 us — and the totals are added in slice order;
//   * FFN2 at few rows: one workgroup per slice (grid z), raw slabs, k_finish_ln adds the slabs in slab order.
// All of them therefore return identical bits, and the choice (GemmVariant, tile shape) follows the live row count of a step
// without touching batch invariance: a row's result does not depend on how many other rows the launch has.
#include "ttx_internal.h"

#include <algorithm>
#include <cmath>

namespace ttx {

// ------------------------------------------------------------------------------------------------
// Generic fallback (K a multiple of 32 that is neither 64, 128 nor a multiple of 256): 32-deep LDS-staged tiles, one chain per output element.  Such K have
// no canonical slices (gemm_slice_k = 0) and every variant sends them here, so they too have one arithmetic.
template <int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void k_gemm_tn(GemmArgs a) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BM = 32 * WGM, BN = 32 * WGN, BK = 32, LDT = BK + 4;
  constexpr int RPP = NT / 8;                 // tile rows filled per pass (8 float4 per 32-float row)
  constexpr int AP = BM / RPP, BP = BN / RPP;
  static_assert(AP >= 1 && BP >= 1, "tile too small for the thread count");
  __shared__ __attribute__((aligned(16))) float As[BM * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];

  const int M = a.m_ptr ? *a.m_ptr : a.M;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= M) return;
  const int kbeg = blockIdx.z * a.k_per_split;
  const int kend = min(a.K, kbeg + a.k_per_split);

  const int t = threadIdx.x;
  const int lr = t >> 3, lc = (t & 7) * 4;
  const int wave = t >> 6, lane = t & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;

  float4 ra[AP], rb[BP];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int row = m0 + p * RPP + lr;
      ra[p] = (row < M) ? *reinterpret_cast<const float4*>(a.X + (size_t)row * a.ldx + k0 + lc)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int col = n0 + p * RPP + lr;
      rb[p] = (col < a.N) ? *reinterpret_cast<const float4*>(a.W + (size_t)col * a.ldw + k0 + lc)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
    for (int p = 0; p < AP; ++p) *reinterpret_cast<float4*>(&As[(p * RPP + lr) * LDT + lc]) = ra[p];
#pragma unroll
    for (int p = 0; p < BP; ++p) *reinterpret_cast<float4*>(&Bs[(p * RPP + lr) * LDT + lc]) = rb[p];
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);       // next tile's HBM/L2 latency hides under this tile's MFMAs
    const float* ap = &As[(wm * 32 + r) * LDT + 4 * h];
    const float* bp = &Bs[(wn * 32 + r) * LDT + 4 * h];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      // lanes 0-31 feed k = kk..kk+3, lanes 32-63 feed k = kk+4..kk+7 (same permutation on both
      // operands, so each MFMA sums two matching k's)
      const float4 av = *reinterpret_cast<const float4*>(ap + kk);
      const float4 bv = *reinterpret_cast<const float4*>(bp + kk);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)
  float* Y = a.Y + (a.raw ? (size_t)blockIdx.z * a.slab_stride : 0);
  const int col = n0 + wn * 32 + r;
  // all values are finished before the first (predicated) store: a pending load inside the store
  // branches would make the compiler drain vmcnt — and with it the previous store — sixteen times
  const float bv = (!a.raw && a.bias) ? a.bias[min(col, a.N - 1)] : 0.f;
  const float lo = a.relu ? 0.f : -INFINITY;
  float val[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) val[v] = fmaxf(acc[v] + bv, lo);
  if (col < a.N) {
    float* yp = Y + (size_t)(m0 + wm * 32 + 4 * h) * a.ldy + col;
    const int rows_left = M - (m0 + wm * 32 + 4 * h);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int dr = (v & 3) + 8 * (v >> 2);
      if (dr < rows_left) yp[(size_t)dr * a.ldy] = val[v];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 64x64 tiles (launches of up to a few hundred workgroups, where one workgroup's time is load latency, not bandwidth): each
// thread keeps a ring of four 64-deep K tiles in flight in registers (all of K = 256 is requested from L2 / Infinity Cache
// before the first MFMA), LDS is double-buffered so one barrier per tile suffices, and the output tile is four 32x32
// fp32-MFMA accumulators, one per wave.  A 64-deep tile is one canonical slice of the K = d contractions: the accumulator is
// folded into the running total after every tile (slice_k = 64) or after every ring pass of four (slice_k = 256).
struct G2Frag { float4 a0, a1, a2, a3, b0, b1, b2, b3; };
struct G2Ptrs { const float* x0; const float* x1; const float* x2; const float* x3;
                const float* w0; const float* w1; const float* w2; const float* w3; };

// Unconditional loads (row/column indices are clamped by the caller): rows >= M and columns >= N only
// ever feed accumulator rows/columns that the epilogue does not store, and branch-free loads let the
// compiler keep counted vmcnt waits instead of draining everything.
__device__ __forceinline__ G2Frag g2_load(const G2Ptrs& p, int koff) {
  G2Frag f;
  f.a0 = *reinterpret_cast<const float4*>(p.x0 + koff);
  f.a1 = *reinterpret_cast<const float4*>(p.x1 + koff);
  f.a2 = *reinterpret_cast<const float4*>(p.x2 + koff);
  f.a3 = *reinterpret_cast<const float4*>(p.x3 + koff);
  f.b0 = *reinterpret_cast<const float4*>(p.w0 + koff);
  f.b1 = *reinterpret_cast<const float4*>(p.w1 + koff);
  f.b2 = *reinterpret_cast<const float4*>(p.w2 + koff);
  f.b3 = *reinterpret_cast<const float4*>(p.w3 + koff);
  return f;
}

template <int LDT>
__device__ __forceinline__ void g2_store(const G2Frag& f, float* as, float* bs, int lr, int lc) {
  *reinterpret_cast<float4*>(&as[(lr) * LDT + lc]) = f.a0;
  *reinterpret_cast<float4*>(&as[(16 + lr) * LDT + lc]) = f.a1;
  *reinterpret_cast<float4*>(&as[(32 + lr) * LDT + lc]) = f.a2;
  *reinterpret_cast<float4*>(&as[(48 + lr) * LDT + lc]) = f.a3;
  *reinterpret_cast<float4*>(&bs[(lr) * LDT + lc]) = f.b0;
  *reinterpret_cast<float4*>(&bs[(16 + lr) * LDT + lc]) = f.b1;
  *reinterpret_cast<float4*>(&bs[(32 + lr) * LDT + lc]) = f.b2;
  *reinterpret_cast<float4*>(&bs[(48 + lr) * LDT + lc]) = f.b3;
}

template <int BK, int LDT>
__device__ __forceinline__ void g2_mma(f32x16& acc, const float* ap, const float* bp) {
#pragma unroll
  for (int kk = 0; kk < BK; kk += 8) {
    const float4 av = *reinterpret_cast<const float4*>(ap + kk);
    const float4 bv = *reinterpret_cast<const float4*>(bp + kk);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
  }
}

// Fold of one finished slice into the running total (`first`: the total IS the slice — no "0 +" in front of it, so that the
// sum starts exactly like the small-row kernels' part[0] + part[1] + ...), and a fresh accumulator for the next slice.
__device__ __forceinline__ void fold_slice(f32x16& tot, f32x16& acc, bool first) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    tot[i] = first ? acc[i] : tot[i] + acc[i];
    acc[i] = 0.f;
  }
}

// The fold without a bubble in the matrix pipe: consecutive slices ALTERNATE between two accumulator sets.  A slice starts
// with an MFMA whose C operand is the constant 0 (no register zeroing), and while its MFMAs run the finished slice in the other
// set is added to the total with plain VALU adds (mode 1: the total IS that slice; 2: total += slice; 0: nothing to fold yet).
// Same additions in the same order as fold_slice.
__device__ __forceinline__ void fold_values(f32x16& tot, const f32x16& y, int mode, int lo, int hi) {
  if (mode == 0) return;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i >= lo && i < hi) tot[i] = (mode == 1) ? y[i] : tot[i] + y[i];
}

// One 64-deep tile = one canonical slice into the fresh accumulator `x`, folding `y` meanwhile.
template <int BK, int LDT>
__device__ __forceinline__ void g2_mma_slice(f32x16& x, const f32x16& y, f32x16& tot, int mode, const float* ap, const float* bp) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < BK; kk += 8) {
    const float4 av = *reinterpret_cast<const float4*>(ap + kk);
    const float4 bv = *reinterpret_cast<const float4*>(bp + kk);
    x = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, kk == 0 ? zero : x, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, x, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, x, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, x, 0, 0, 0);
    if (kk == 0) fold_values(tot, y, mode, 0, 16);
  }
}

constexpr int G24_SMEM_FLOATS = 2 * 2 * 64 * 68;       // 69 632 B: the larger of the two tilings' LDS images (64x64: two buffers of A and B, 64 x 68 each; 128x64: 2 x (128 + 64) x 36)

// NT = number of 64-deep K tiles per workgroup when it is 1, 2 or 4 (straight-line code, every tile
// requested up front); NT = 0: any multiple of 4 tiles, ring slots refilled as they drain.
template <int NT>
__device__ __forceinline__ void g2_body(const GemmArgs& a, const int M, const int bx, const int by, const int bz, float* smem) {
  constexpr int BM = 64, BN = 64, BK = 64, LDT = BK + 4, RING = 4;
  typedef float (*TileBufs)[BM * LDT];
  TileBufs As = reinterpret_cast<TileBufs>(smem);
  TileBufs Bs = reinterpret_cast<TileBufs>(smem + 2 * BM * LDT);

  const int m0 = by * BM, n0 = bx * BN;
  if (m0 >= M) return;
  const int kbeg = bz * a.k_per_split;
  const int kend = min(a.K, kbeg + a.k_per_split);
  const int ntiles = (kend - kbeg) / BK;

  const int t = threadIdx.x;
  const int lr = t >> 4, lc = (t & 15) * 4;      // 16 float4 per 64-float row, 16 rows per pass
  const int wave = t >> 6, lane = t & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  G2Ptrs p;
  {
    const int r0 = m0 + lr, c0 = n0 + lr;
    const float* xb = a.X + kbeg + lc;
    const float* wb = a.W + kbeg + lc;
    p.x0 = xb + (size_t)min(r0, M - 1) * a.ldx;
    p.x1 = xb + (size_t)min(r0 + 16, M - 1) * a.ldx;
    p.x2 = xb + (size_t)min(r0 + 32, M - 1) * a.ldx;
    p.x3 = xb + (size_t)min(r0 + 48, M - 1) * a.ldx;
    p.w0 = wb + (size_t)min(c0, a.N - 1) * a.ldw;
    p.w1 = wb + (size_t)min(c0 + 16, a.N - 1) * a.ldw;
    p.w2 = wb + (size_t)min(c0 + 32, a.N - 1) * a.ldw;
    p.w3 = wb + (size_t)min(c0 + 48, a.N - 1) * a.ldw;
  }
  f32x16 acc, tot;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; tot[i] = 0.f; }
  const int aoff = (wm * 32 + r) * LDT + 4 * h, boff = (wn * 32 + r) * LDT + 4 * h;
  // slices of this workgroup's K range: one per tile (slice_k = 64), one per four tiles (slice_k = 256), or the whole range
  // as one chain (slice_k = 0, or a range no longer than a slice: `sliced` false, the accumulator is the result)
  const int tiles_per_slice = a.slice_k > 0 ? a.slice_k / BK : 0;
  const bool sliced = tiles_per_slice > 0 && tiles_per_slice < ntiles;
  int done = 0;                                   // tiles finished
  auto tile_mma = [&](const float* ap, const float* bp) {
    g2_mma<BK, LDT>(acc, ap, bp);
    ++done;
    if (sliced && done % tiles_per_slice == 0) fold_slice(tot, acc, done == tiles_per_slice);
  };
  // every tile its own slice (the K = d contractions): slices alternate between `acc` and `alt` (g2_mma_slice)
  const bool per_tile = NT != 0 && sliced && tiles_per_slice == 1;     // (the open-ended ring keeps the plain fold: registers)
  f32x16 alt;
#pragma unroll
  for (int i = 0; i < 16; ++i) alt[i] = 0.f;

  if constexpr (NT == 1) {
    const G2Frag f0 = g2_load(p, 0);
    g2_store<LDT>(f0, As[0], Bs[0], lr, lc);
    __syncthreads();
    tile_mma(As[0] + aoff, Bs[0] + boff);
  } else if constexpr (NT == 2) {
    const G2Frag f0 = g2_load(p, 0);
    const G2Frag f1 = g2_load(p, BK);
    g2_store<LDT>(f0, As[0], Bs[0], lr, lc);
    g2_store<LDT>(f1, As[1], Bs[1], lr, lc);
    __syncthreads();
    if (per_tile) {
      g2_mma_slice<BK, LDT>(acc, alt, tot, 0, As[0] + aoff, Bs[0] + boff);
      g2_mma_slice<BK, LDT>(alt, acc, tot, 1, As[1] + aoff, Bs[1] + boff);
      fold_values(tot, alt, 2, 0, 16);
    } else {
      tile_mma(As[0] + aoff, Bs[0] + boff);
      tile_mma(As[1] + aoff, Bs[1] + boff);
    }
  } else if (per_tile) {
    // the same ring with the four tiles of a pass going to acc, alt, acc, alt
    G2Frag f0 = g2_load(p, 0);
    G2Frag f1 = g2_load(p, BK);
    G2Frag f2 = g2_load(p, 2 * BK);
    G2Frag f3 = g2_load(p, 3 * BK);
    const int last = ntiles - 1;
    for (int base = 0; base < ntiles; base += RING) {
      g2_store<LDT>(f0, As[0], Bs[0], lr, lc);
      if constexpr (NT == 0) f0 = g2_load(p, min(base + RING, last) * BK);
      __syncthreads();
      g2_mma_slice<BK, LDT>(acc, alt, tot, base == 0 ? 0 : 2, As[0] + aoff, Bs[0] + boff);
      g2_store<LDT>(f1, As[1], Bs[1], lr, lc);
      if constexpr (NT == 0) f1 = g2_load(p, min(base + RING + 1, last) * BK);
      __syncthreads();
      g2_mma_slice<BK, LDT>(alt, acc, tot, base == 0 ? 1 : 2, As[1] + aoff, Bs[1] + boff);
      g2_store<LDT>(f2, As[0], Bs[0], lr, lc);
      if constexpr (NT == 0) f2 = g2_load(p, min(base + RING + 2, last) * BK);
      __syncthreads();
      g2_mma_slice<BK, LDT>(acc, alt, tot, 2, As[0] + aoff, Bs[0] + boff);
      g2_store<LDT>(f3, As[1], Bs[1], lr, lc);
      if constexpr (NT == 0) f3 = g2_load(p, min(base + RING + 3, last) * BK);
      __syncthreads();
      g2_mma_slice<BK, LDT>(alt, acc, tot, 2, As[1] + aoff, Bs[1] + boff);
    }
    fold_values(tot, alt, 2, 0, 16);
  } else {
    // ring of four register tiles, two LDS buffers; slot indices are compile-time (no register moves:
    // moving a pending load's destination would force a wait on it)
    G2Frag f0 = g2_load(p, 0);
    G2Frag f1 = g2_load(p, BK);
    G2Frag f2 = g2_load(p, 2 * BK);
    G2Frag f3 = g2_load(p, 3 * BK);
    const int last = ntiles - 1;
    for (int base = 0; base < ntiles; base += RING) {
      g2_store<LDT>(f0, As[0], Bs[0], lr, lc);
      if constexpr (NT == 0) f0 = g2_load(p, min(base + RING, last) * BK);      // clamped: branch-free refill
      __syncthreads();
      tile_mma(As[0] + aoff, Bs[0] + boff);
      g2_store<LDT>(f1, As[1], Bs[1], lr, lc);
      if constexpr (NT == 0) f1 = g2_load(p, min(base + RING + 1, last) * BK);
      __syncthreads();
      tile_mma(As[1] + aoff, Bs[1] + boff);
      g2_store<LDT>(f2, As[0], Bs[0], lr, lc);
      if constexpr (NT == 0) f2 = g2_load(p, min(base + RING + 2, last) * BK);
      __syncthreads();
      tile_mma(As[0] + aoff, Bs[0] + boff);
      g2_store<LDT>(f3, As[1], Bs[1], lr, lc);
      if constexpr (NT == 0) f3 = g2_load(p, min(base + RING + 3, last) * BK);
      __syncthreads();
      tile_mma(As[1] + aoff, Bs[1] + boff);
    }
  }
  if (sliced) acc = tot;

  float* Y = a.Y + (a.raw ? (size_t)bz * a.slab_stride : 0);
  const int col = n0 + wn * 32 + r;
  // all values are finished before the first (predicated) store: a pending load inside the store
  // branches would make the compiler drain vmcnt — and with it the previous store — sixteen times
  const float bv = (!a.raw && a.bias) ? a.bias[min(col, a.N - 1)] : 0.f;
  const float lo = a.relu ? 0.f : -INFINITY;
  float val[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) val[v] = fmaxf(acc[v] + bv, lo);
  if (m0 + BM <= M && n0 + BN <= a.N) {             // interior workgroup (uniform): straight-line stores
    float* yp = Y + (size_t)(m0 + wm * 32 + 4 * h) * a.ldy + col;
#pragma unroll
    for (int v = 0; v < 16; ++v) yp[(size_t)((v & 3) + 8 * (v >> 2)) * a.ldy] = val[v];
  } else if (col < a.N) {
    float* yp = Y + (size_t)(m0 + wm * 32 + 4 * h) * a.ldy + col;
    const int rows_left = M - (m0 + wm * 32 + 4 * h);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int dr = (v & 3) + 8 * (v >> 2);
      if (dr < rows_left) yp[(size_t)dr * a.ldy] = val[v];
    }
  }
}

template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm2(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 64 * 68];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  g2_body<NT>(a, M, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// ------------------------------------------------------------------------------------------------
// 128x64 tiles for launches with thousands of rows (slot pools: M up to ~16 000 step rows; encoder / cross-K/V bulk
// passes): 2x2 waves of 64x32 (two independent 32x32 fp32-MFMA accumulators per wave, so back-to-back MFMAs never wait on
// each other), 32-deep K tiles, LDS double buffer (stride 36 floats: conflict-free ds_read_b128), one barrier per tile, two
// workgroups per CU so that one's prologue / epilogue overlaps the other's MFMAs.  Against the 64x64 kernel: fewer L2->LDS
// and LDS->register bytes per MFMA.  The next tile's global loads are issued before the MFMA block of the current one and
// written to the other LDS buffer BETWEEN the MFMAs of the following tile (loads are unconditional, tile index clamped:
// counted vmcnt waits).  (A 128x128 tiling — four accumulators per wave — was the fastest for FFN1 / FFN2 at >= 8 000 rows
// while one chain per element was the arithmetic; with the running slice AND the folded total it needs more than the 256
// registers two workgroups per CU leave each wave and loses to this one: profiles/r03_gemm_bench_canonical_slices.txt, v4.)
// Canonical slices: a pair of 32-deep tiles is one 64-k slice; the accumulators are folded into the totals after every
// slice_k / 64 pairs.
struct G4Frag { float4 a0, a1, a2, a3, b0, b1; };

template <bool ALT>
__device__ __forceinline__ void g4_body(const GemmArgs& a, const int M, const int bx, const int by, const int bz, float* smem) {
  constexpr int BM = 128, BN = 64, BK = 32, LDT = BK + 4;
  constexpr int WN = BN / 2;                      // columns per wave
  typedef float (*TileBufsA)[BM * LDT];
  typedef float (*TileBufsB)[BN * LDT];
  TileBufsA As = reinterpret_cast<TileBufsA>(smem);
  TileBufsB Bs = reinterpret_cast<TileBufsB>(smem + 2 * BM * LDT);
  const int m0 = by * BM, n0 = bx * BN;
  if (m0 >= M) return;
  const int kbeg = bz * a.k_per_split;
  const int kend = min(a.K, kbeg + a.k_per_split);
  const int ntiles = (kend - kbeg) / BK;
  const int t = threadIdx.x;
  const int lr = t >> 3, lc = (t & 7) * 4;       // 8 float4 per 32-float row, 32 rows per pass, 4 passes
  const int wave = t >> 6, lane = t & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  const float* xp[4];
  const float* wp[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) xp[i] = a.X + (size_t)min(m0 + lr + 32 * i, M - 1) * a.ldx + kbeg + lc;
#pragma unroll
  for (int i = 0; i < 2; ++i) wp[i] = a.W + (size_t)min(n0 + lr + 32 * i, a.N - 1) * a.ldw + kbeg + lc;
  auto gload = [&](int tile) {
    G4Frag f;
    const int ko = tile * BK;
    f.a0 = *reinterpret_cast<const float4*>(xp[0] + ko);
    f.a1 = *reinterpret_cast<const float4*>(xp[1] + ko);
    f.a2 = *reinterpret_cast<const float4*>(xp[2] + ko);
    f.a3 = *reinterpret_cast<const float4*>(xp[3] + ko);
    f.b0 = *reinterpret_cast<const float4*>(wp[0] + ko);
    f.b1 = *reinterpret_cast<const float4*>(wp[1] + ko);
    return f;
  };
  auto lstore = [&](const G4Frag& f, int buf) {
    float* as = As[buf] + lr * LDT + lc;
    float* bs = Bs[buf] + lr * LDT + lc;
    *reinterpret_cast<float4*>(as) = f.a0;
    *reinterpret_cast<float4*>(as + 32 * LDT) = f.a1;
    *reinterpret_cast<float4*>(as + 64 * LDT) = f.a2;
    *reinterpret_cast<float4*>(as + 96 * LDT) = f.a3;
    *reinterpret_cast<float4*>(bs) = f.b0;
    *reinterpret_cast<float4*>(bs + 32 * LDT) = f.b1;
  };

  // Two accumulator sets (a0x / a1x: even / odd slices) and the folded total.  A slice starts with MFMAs whose C operand is the
  // constant 0; while it runs, the slice finished in the other set is folded into the total between the MFMAs (fold_values).
  f32x16 a00, a10, b00, b10, t00, t10;
#pragma unroll
  for (int i = 0; i < 16; ++i) { a00[i] = 0.f; a10[i] = 0.f; b00[i] = 0.f; b10[i] = 0.f; t00[i] = 0.f; t10[i] = 0.f; }
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int aoff = (wm * 64 + r) * LDT + 4 * h, boff = (wn * WN + r) * LDT + 4 * h;

  // One 32-deep K tile into the accumulators (x0, x1): 8 MFMAs per 8 k's.  `fresh`: the tile opens a slice (C = 0 for its
  // first MFMA pair).  `during(q)` (q = 0..3) runs after the first MFMA pair of every 8-k step: the LDS writes of the NEXT
  // tile and a quarter of the pending fold go there, between MFMAs, so that they cost no MFMA time.
  auto mma = [&](f32x16& x0, f32x16& x1, int buf, bool fresh, auto&& during) {
    const float* ap = As[buf] + aoff;
    const float* bp = Bs[buf] + boff;
    // fragments of the 8-k step after the current one are read from LDS while the current step's MFMAs run
    float4 a0 = *reinterpret_cast<const float4*>(ap);
    float4 a1 = *reinterpret_cast<const float4*>(ap + 32 * LDT);
    float4 b0 = *reinterpret_cast<const float4*>(bp);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      float4 na0 = a0, na1 = a1, nb0 = b0;
      if (kk + 8 < BK) {
        na0 = *reinterpret_cast<const float4*>(ap + kk + 8);
        na1 = *reinterpret_cast<const float4*>(ap + 32 * LDT + kk + 8);
        nb0 = *reinterpret_cast<const float4*>(bp + kk + 8);
      }
      if (kk == 0 && fresh) {                             // uniform branch: two straight-line versions of the first pair
        x0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, zero, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, zero, 0, 0, 0);
      } else {
        x0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, x1, 0, 0, 0);
      }
      during(kk >> 3);
      x0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, x1, 0, 0, 0);
      x0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, x1, 0, 0, 0);
      x0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, x1, 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0;
    }
  };
  // quarter q of a tile's LDS writes (one float4 of A's 128 rows per quarter, B's 64 rows in the first two)
  auto lstore_q = [&](const G4Frag& f, int buf, int q) {
    float* as = As[buf] + lr * LDT + lc;
    float* bs = Bs[buf] + lr * LDT + lc;
    if (q == 0) { *reinterpret_cast<float4*>(as) = f.a0; *reinterpret_cast<float4*>(bs) = f.b0; }
    else if (q == 1) { *reinterpret_cast<float4*>(as + 32 * LDT) = f.a1; *reinterpret_cast<float4*>(bs + 32 * LDT) = f.b1; }
    else if (q == 2) *reinterpret_cast<float4*>(as + 64 * LDT) = f.a2;
    else *reinterpret_cast<float4*>(as + 96 * LDT) = f.a3;
  };
  // two register tiles in flight (static slots, clamped refills: same shape as the 64x64 kernel's ring): a tile's loads are
  // issued two MFMA blocks before its LDS write.  The K range is a multiple of 64: ntiles is even.
  const int last = ntiles - 1;
  const int bcol = n0 + wn * WN + r;
  // the bias value is requested before the K loop (a load still pending in the epilogue would make every predicated store
  // wait for vmcnt(0), i.e. for the previous store: 32 serialised round trips)
  const float bias0 = (!a.raw && a.bias) ? a.bias[min(bcol, a.N - 1)] : 0.f;
  const int pairs_per_slice = (a.slice_k > 0 && a.slice_k < kend - kbeg) ? a.slice_k / (2 * BK) : ntiles / 2;   // not sliced: one slice
  const int n_slices = (ntiles / 2) / pairs_per_slice;
  asm volatile("" ::: "memory");
  G4Frag f0 = gload(0);
  asm volatile("" ::: "memory");        // issue order f0 then f1 also ahead of the loop: the header waits with a counted vmcnt, not 0
  G4Frag f1 = gload(1);
  // Tile i is computed from one LDS buffer while tile i + 1 is written into the other BETWEEN the MFMAs (after the
  // barrier that ends a phase every wave has finished reading the buffer the next phase overwrites), and the registers
  // just emptied are refilled from global memory for tile i + 2: LDS writes, global loads and MFMAs overlap inside every
  // wave instead of only across the two workgroups of a CU.
  lstore(f0, 0);
  asm volatile("" ::: "memory");
  f0 = gload(min(2, last));
  __syncthreads();
  int i = 0;                                            // next tile
  // one canonical slice into (x0, x1); the slice in (y0, y1) is folded into the total during its first tile (mode: fold_values)
  auto slice = [&](f32x16& x0, f32x16& x1, const f32x16& y0, const f32x16& y1, int mode) {
    for (int pr = 0; pr < pairs_per_slice; ++pr) {
      const int m0f = pr == 0 ? mode : 0;
      mma(x0, x1, 0, pr == 0, [&](int q) { lstore_q(f1, 1, q); fold_values(t00, y0, m0f, 4 * q, 4 * q + 4); fold_values(t10, y1, m0f, 4 * q, 4 * q + 4); });
      asm volatile("" ::: "memory");
      f1 = gload(min(i + 3, last));
      __syncthreads();
      mma(x0, x1, 1, false, [&](int q) { lstore_q(f0, 0, q); });      // tile i + 2 (a clamped repeat of the last tile at the end: never read)
      asm volatile("" ::: "memory");
      f0 = gload(min(i + 4, last));
      __syncthreads();
      i += 2;
    }
  };
  int s = 0;
  if constexpr (ALT) {
    for (; s + 1 < n_slices; s += 2) {
      slice(a00, a10, b00, b10, s == 0 ? 0 : 2);
      slice(b00, b10, a00, a10, s == 0 ? 1 : 2);
    }
  }
  f32x16 c00, c10;
  if constexpr (!ALT) {
    // long slices (K >= 2048: eight 32-deep tiles each): the fold is rare, one accumulator set and a plain fold between slices
    // leave more registers to the loads in flight
    for (; s < n_slices; ++s) {
      slice(a00, a10, b00, b10, 0);
      fold_values(t00, a00, s == 0 ? 1 : 2, 0, 16); fold_values(t10, a10, s == 0 ? 1 : 2, 0, 16);
    }
    c00 = t00; c10 = t10;
  } else if (n_slices == 1) {                                  // one chain over the whole range (raw slab of one slice, or no slicing)
    slice(a00, a10, b00, b10, 0);
    c00 = a00; c10 = a10;
  } else {
    if (s < n_slices) {                                 // odd count: the last slice goes to the first set
      slice(a00, a10, b00, b10, 2);
      fold_values(t00, a00, 2, 0, 16); fold_values(t10, a10, 2, 0, 16);
    } else {
      fold_values(t00, b00, 2, 0, 16); fold_values(t10, b10, 2, 0, 16);
    }
    c00 = t00; c10 = t10;
  }

  float* Y = a.Y + (a.raw ? (size_t)bz * a.slab_stride : 0);
  const float lo = a.relu ? 0.f : -INFINITY;
  auto store_tile = [&](const f32x16& c, int tm) {
    const int col = n0 + wn * WN + r;
    float val[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) val[v] = fmaxf(c[v] + bias0, lo);
    const int row0 = m0 + wm * 64 + tm * 32 + 4 * h;
    float* yp = Y + (size_t)row0 * a.ldy + col;
    if (m0 + BM <= M && n0 + BN <= a.N) {            // interior workgroup (uniform): straight-line stores
#pragma unroll
      for (int v = 0; v < 16; ++v) yp[(size_t)((v & 3) + 8 * (v >> 2)) * a.ldy] = val[v];
    } else if (col < a.N) {
      const int rows_left = M - row0;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int dr = (v & 3) + 8 * (v >> 2);
        if (dr < rows_left) yp[(size_t)dr * a.ldy] = val[v];
      }
    }
  };
  store_tile(c00, 0);
  store_tile(c10, 1);
}

// The first tiles of the whole grid in dispatch order take the work (all K slabs included): the dispatcher hands
// consecutive workgroups to consecutive CUs, so a contiguous block spreads one per CU / XCD; actives separated by idle
// workgroups ended up two to a CU with other CUs empty (2x the time, measured).  XCD-aware order (speed only): workgroups
// lin and lin + 8 share an XCD and its L2, so the workgroups of one XCD take a CONTIGUOUS run of tiles — the column tiles
// of a row block (same X rows) then hit one L2 instead of eight.  Returns false for a workgroup without a tile.
__device__ __forceinline__ bool tile_of_workgroup(int n_tiles, int nbx, int nby, int& bx, int& by, int& slab) {
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (lin >= n_tiles) return false;
  const int xb = n_tiles >> 3, xr = n_tiles & 7, xcd = lin & 7;
  const int v = xcd * xb + min(xcd, xr) + (lin >> 3);
  slab = v / (nbx * nby);
  const int rem = v - slab * (nbx * nby);
  bx = rem % nbx; by = rem / nbx;
  return true;
}

// One launch, two tilings: the grid is laid out for 64x64 tiles; when the row count read from the device gives the 128x64
// tiling at least `big_min_tiles` workgroups, the first workgroups in dispatch order each compute such a tile and the others
// leave at once, otherwise all compute their 64x64 tile.  Both evaluate the canonical slice sum: bit-identical results.
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm24(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[G24_SMEM_FLOATS];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  const int nby = (M + 127) >> 7, nbx = (a.N + 63) >> 6;
  const int mid_tiles = nby * nbx * (int)gridDim.z;                        // 128x64 tiles
  if (a.big_min_tiles > 0 && mid_tiles >= a.big_min_tiles) {
    int bx, by, slab;
    if (!tile_of_workgroup(mid_tiles, nbx, nby, bx, by, slab)) return;
    g4_body<NT == 4>(a, M, bx, by, slab, smem);
  } else {
    g2_body<NT>(a, M, blockIdx.x, blockIdx.y, blockIdx.z, smem);
  }
}

// ------------------------------------------------------------------------------------------------
// 32x32 tiles for step GEMMs that are short of workgroups (GV_SMALL): the K range of the workgroup is split over its 4
// waves — with K = 256 every wave forms exactly one canonical 64-k slice — MFMA operands are loaded straight from global
// memory into the registers the MFMA reads (lane (r,h) owns row r / column r and the k's 8g+4h..8g+4h+3, which is exactly
// one float4 per 8 k's) — no LDS staging, no barrier before the math; the four slices meet in LDS (16.5 KB), are added in
// slice order and leave as whole 128-B rows.  4x the workgroups of the 64x64 kernel and many of them resident per CU.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_gemm3(GemmArgs a) {
  constexpr int KW = 64;                                   // k's per wave = the canonical slice of the K = 256 contractions
  __shared__ __attribute__((aligned(16))) float part[4][32 * 33];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  if (m0 >= M) return;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int r = lane & 31, h = lane >> 5;
  const int kbeg = blockIdx.z * a.k_per_split + wave * KW;
  const float* xp = a.X + (size_t)min(m0 + r, M - 1) * a.ldx + kbeg + 4 * h;
  const float* wp = a.W + (size_t)min(n0 + r, a.N - 1) * a.ldw + kbeg + 4 * h;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 av[KW / 8], bv[KW / 8];
#pragma unroll
  for (int g = 0; g < KW / 8; ++g) {
    av[g] = *reinterpret_cast<const f32x4*>(xp + 8 * g);
    bv[g] = *reinterpret_cast<const f32x4*>(wp + 8 * g);
  }
  // The empty asm reads every destination register: all sixteen requests are out before the first MFMA (otherwise the
  // scheduler pairs each load with its MFMAs and the wave eats one memory latency per pair).
  asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]),
                    "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
#pragma unroll
  for (int g = 0; g < KW / 8; ++g) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].x, bv[g].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].y, bv[g].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].z, bv[g].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].w, bv[g].w, acc, 0, 0, 0);
  }
#pragma unroll
  for (int v = 0; v < 16; ++v) part[wave][((v & 3) + 8 * (v >> 2) + 4 * h) * 33 + r] = acc[v];
  __syncthreads();
  // 256 threads -> 32 rows x 32 columns, 4 values each along a row (one 128-B row per 8 threads)
  const int row = t >> 3, c0 = (t & 7) * 4;
  float4 o;
  float* op = &o.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = row * 33 + c0 + i;
    op[i] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];       // slice order
  }
  const int grow = m0 + row, gcol = n0 + c0;
  if (grow < M && gcol < a.N) {
    float* Y = a.Y + (a.raw ? (size_t)blockIdx.z * a.slab_stride : 0) + (size_t)grow * a.ldy + gcol;
    const float lo = a.relu ? 0.f : -INFINITY;
    if (gcol + 3 < a.N && (a.ldy & 3) == 0) {
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!a.raw && a.bias) bb = *reinterpret_cast<const float4*>(a.bias + gcol);
      o.x = fmaxf(o.x + bb.x, lo); o.y = fmaxf(o.y + bb.y, lo); o.z = fmaxf(o.z + bb.z, lo); o.w = fmaxf(o.w + bb.w, lo);
      *reinterpret_cast<float4*>(Y) = o;
    } else {
      for (int i = 0; i < 4 && gcol + i < a.N; ++i) {
        const float bb = (!a.raw && a.bias) ? a.bias[gcol + i] : 0.f;
        Y[i] = fmaxf(op[i] + bb, lo);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// y = LN2?( LN( (resid + bias) + (slab[0] + slab[1] + ...) ) ): the slabs are added to each other first, in slab order — the
// same sum a single workgroup walking all slices leaves in its one slab (canonical slice order, top of this file).
template <int VPL>
__device__ __forceinline__ void ln_inplace(float (&x)[VPL], const float* g, const float* b, int c0, int d, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) s += x[i];
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) { const float t = x[i] - mean; q += t * t; }
  const float var = wave_sum(q) / (float)d;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) x[i] = (x[i] - mean) * rstd * g[c0 + i] + b[c0 + i];
}

template <int VPL>
__global__ __launch_bounds__(256) void k_finish_ln(FinishArgs a) {
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const int c0 = lane * VPL;
  float x[VPL];
  const size_t off = (size_t)row * a.d + c0;
  if constexpr (VPL == 4) {
    if (a.n_slabs == 1 || a.n_slabs == 8) {
      // the common shapes (one slab; the eight FFN2 slices of a small step): every operand is requested before the first
      // use — the loop version below waits for each array in turn.  Same additions in the same order.
      const float4 r4 = *reinterpret_cast<const float4*>(a.resid + off);
      const float4 b4 = *reinterpret_cast<const float4*>(a.bias + c0);
      float4 tsum = *reinterpret_cast<const float4*>(a.slabs + off);
      if (a.n_slabs == 8) {
        float4 p[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) p[s] = *reinterpret_cast<const float4*>(a.slabs + (size_t)(s + 1) * a.slab_stride + off);
#pragma unroll
        for (int s = 0; s < 7; ++s) { tsum.x += p[s].x; tsum.y += p[s].y; tsum.z += p[s].z; tsum.w += p[s].w; }
      }
      const float4 g1 = *reinterpret_cast<const float4*>(a.g1 + c0);
      const float4 e1 = *reinterpret_cast<const float4*>(a.b1 + c0);
      const float* g2p = a.g2 ? a.g2 : a.g1;         // clamped: loaded either way, used only if a.g2
      const float* e2p = a.g2 ? a.b2 : a.b1;
      const float4 g2 = *reinterpret_cast<const float4*>(g2p + c0);
      const float4 e2 = *reinterpret_cast<const float4*>(e2p + c0);
      const uint8_t valid = a.row_valid ? a.row_valid[row] : (uint8_t)1;
      x[0] = (r4.x + b4.x) + tsum.x; x[1] = (r4.y + b4.y) + tsum.y; x[2] = (r4.z + b4.z) + tsum.z; x[3] = (r4.w + b4.w) + tsum.w;
      const float ga[4] = {g1.x, g1.y, g1.z, g1.w}, ea[4] = {e1.x, e1.y, e1.z, e1.w};
      ln_inplace<4>(x, ga, ea, 0, a.d, a.eps);
      if (a.g2) {
        const float gb[4] = {g2.x, g2.y, g2.z, g2.w}, eb[4] = {e2.x, e2.y, e2.z, e2.w};
        ln_inplace<4>(x, gb, eb, 0, a.d, a.eps);
      }
      const bool keep = valid != 0;
      *reinterpret_cast<float4*>(a.Y + off) = make_float4(keep ? x[0] : 0.f, keep ? x[1] : 0.f, keep ? x[2] : 0.f, keep ? x[3] : 0.f);
      return;
    }
  }
  float tsum[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) tsum[i] = a.slabs[off + i];
  for (int s = 1; s < a.n_slabs; ++s) {
    const float* p = a.slabs + (size_t)s * a.slab_stride + off;
#pragma unroll
    for (int i = 0; i < VPL; ++i) tsum[i] += p[i];
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) x[i] = (a.resid[off + i] + a.bias[c0 + i]) + tsum[i];
  ln_inplace<VPL>(x, a.g1, a.b1, c0, a.d, a.eps);
  if (a.g2) ln_inplace<VPL>(x, a.g2, a.b2, c0, a.d, a.eps);
  const bool keep = a.row_valid ? (a.row_valid[row] != 0) : true;
#pragma unroll
  for (int i = 0; i < VPL; ++i) a.Y[off + i] = keep ? x[i] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Host side

int gemm_slice_k(int K) {
  if (K == 64 || K == 128) return 64;
  if (K % 256) return 0;                       // no canonical slices: k_gemm_tn's single chain in every variant
  return K >= 2048 ? 256 : 64;
}

// the 32x32 one-wave-per-slice kernel serves the K = 256 step GEMMs up to 768 columns (QKV, the d x d projections, the
// classifier): exactly four canonical slices per workgroup
static bool use_gemm3(bool step, int variant, int N, int K) {
  return step && variant == GV_SMALL && K == 256 && N <= 768;
}

int gemm_splits(int N, int K, bool step, int variant) {
  (void)N;
  // FFN2 of a small step: one workgroup per canonical slice (K / 256 slabs, at most 16: the slab workspace), summed by k_finish_ln
  if (step && variant == GV_SMALL && gemm_slice_k(K) == 256 && K / 256 <= 16) return K / 256;
  return 1;
}

int launch_gemm(ttx_session* s, hipStream_t st, const float* X, int ldx, const float* W, int ldw, const float* bias,
                float* Y, int ldy, const int* m_ptr, int Mmax, int N, int K, bool relu, int splits, long long slab_stride,
                int variant) {
  if (Mmax <= 0) return TTX_OK;
  if (K % 32) return fail(TTX_ERR_INVALID, "GEMM K must be a multiple of 32");
  GemmArgs a;
  a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.bias = bias; a.Y = Y; a.ldy = ldy; a.m_ptr = m_ptr;
  a.M = Mmax; a.N = N; a.K = K; a.relu = relu ? 1 : 0;
  a.raw = splits > 0 ? 1 : 0;
  const int S = splits > 0 ? splits : 1;
  a.k_per_split = K / S;
  a.slab_stride = slab_stride;
  a.slice_k = gemm_slice_k(K);
  if (a.slice_k && a.k_per_split % a.slice_k) return fail(TTX_ERR_INVALID, "split-K slabs must be whole canonical slices");
  a.big_min_tiles = 0;
  hipEvent_t e1 = nullptr;
  if (s->profile) {
    if (s->ev_used == s->ev_pool.size()) {
      hipEvent_t a0, a1;
      HIP_TRY(hipEventCreate(&a0));
      HIP_TRY(hipEventCreate(&a1));
      s->ev_pool.push_back({a0, a1});
    }
    HIP_TRY(hipEventRecord(s->ev_pool[s->ev_used].first, st));
    e1 = s->ev_pool[s->ev_used].second;
    s->ev_used++;
  }
  const bool step = (m_ptr != nullptr);
  if (use_gemm3(step, variant, N, K) && S == 1) {
    hipLaunchKernelGGL(k_gemm3, dim3(cdiv(N, 32), cdiv(Mmax, 32), 1), dim3(256), 0, st, a);
  } else if (a.slice_k == 0) {
    hipLaunchKernelGGL((k_gemm_tn<2, 2>), dim3(cdiv(N, 64), cdiv(Mmax, 64), S), dim3(256), 0, st, a);
  } else if (a.k_per_split % 256 == 0 && !(step && variant == GV_SMALL)) {
    // one launch that picks the tiling (128x64 / 64x64) from the live row count.  Where the 128-row tiling starts to pay
    // depends on how long a tile runs (profiles/r03_gemm_bench_canonical_slices.txt): deep contractions (FFN2, K = 2048) from
    // about a third of the base count on, the 2048-wide FFN1 only from about twice the base count
    a.big_min_tiles = K >= 2048 ? std::max(1, s->big_min_tiles / 3) : (N >= 2048 ? 2 * s->big_min_tiles : s->big_min_tiles);
    if (s->big_min_tiles == 0) a.big_min_tiles = 0;
    dim3 grid(cdiv(N, 64), cdiv(Mmax, 64), S);
    if (a.k_per_split == 256) hipLaunchKernelGGL((k_gemm24<4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemm24<0>), grid, dim3(256), 0, st, a);
  } else {
    dim3 grid(cdiv(N, 64), cdiv(Mmax, 64), S);
    switch (a.k_per_split) {
      case 64: hipLaunchKernelGGL((k_gemm2<1>), grid, dim3(256), 0, st, a); break;
      case 128: hipLaunchKernelGGL((k_gemm2<2>), grid, dim3(256), 0, st, a); break;
      case 256: hipLaunchKernelGGL((k_gemm2<4>), grid, dim3(256), 0, st, a); break;
      default: hipLaunchKernelGGL((k_gemm2<0>), grid, dim3(256), 0, st, a); break;     // a multiple of 256 (gemm_slice_k)
    }
  }
  if (e1) HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

int launch_finish(ttx_session* s, hipStream_t st, const float* slabs, int n_slabs, long long slab_stride, const float* bias,
                  const float* resid, const float* g1, const float* b1, const float* g2, const float* b2,
                  const uint8_t* row_valid, float* Y, const int* m_ptr, int Mmax) {
  if (Mmax <= 0) return TTX_OK;
  const ttx_config& c = s->m->cfg;
  FinishArgs a;
  a.slabs = slabs; a.n_slabs = n_slabs; a.slab_stride = slab_stride; a.bias = bias; a.resid = resid;
  a.g1 = g1; a.b1 = b1; a.g2 = g2; a.b2 = b2; a.row_valid = row_valid; a.Y = Y; a.m_ptr = m_ptr; a.M = Mmax;
  a.d = c.embedding_dim; a.eps = c.layer_norm_eps;
  dim3 grid(cdiv(Mmax, 4));
  switch (c.embedding_dim / 64) {
    case 1: hipLaunchKernelGGL((k_finish_ln<1>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_finish_ln<2>), grid, dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((k_finish_ln<4>), grid, dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((k_finish_ln<8>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((k_finish_ln<16>), grid, dim3(256), 0, st, a); break;
  }
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

// Development aid (tools/bench_gemm.py, ttx_debug_gemm_bench): one GEMM shape in isolation on random operands.
// variant: 2 = 64x64 tiles (k_gemm24 with the 128-row tiling off), 46 = 128x64 tiles, 24 = k_gemm24's own choice,
// 3 = the one-wave-per-slice 32x32 kernel (K = 256), 8 = one workgroup per slice (raw slabs).  `splits` > 0 asks for that many raw slabs.  Reports microseconds per launch over `reps` back-to-back launches
// and the largest absolute difference of the (slab-summed, in slab order) result to the 64x64 tiling's.
int gemm_bench(ttx_session* s, int M, int N, int K, int splits, int variant, int reps, double* us_per_launch, double* max_abs_diff) {
  if (!s || M <= 0 || N <= 0 || K <= 0 || gemm_slice_k(K) == 0 || reps <= 0) return fail(TTX_ERR_INVALID, "bad argument to ttx_debug_gemm_bench");
  HIP_TRY(hipSetDevice(s->m->device));
  const int S = splits > 0 ? splits : 1;
  const int slice = gemm_slice_k(K);
  if (K % S || (K / S) % std::max(slice, 64)) return fail(TTX_ERR_INVALID, "K / splits must be a whole number of canonical slices");
  if (variant == 3 && (K != 256 || S != 1)) return fail(TTX_ERR_INVALID, "variant 3 serves K = 256 without slabs");
  std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N);
  uint64_t z = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return (float)((z >> 40) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() * 0.125f;
  for (auto& v : hb) v = rnd();
  float *dx = nullptr, *dw = nullptr, *db = nullptr, *dy = nullptr, *dref = nullptr;
  int* dm = nullptr;
  HIP_TRY(hipMalloc(&dx, hx.size() * 4));
  HIP_TRY(hipMalloc(&dw, hw.size() * 4));
  HIP_TRY(hipMalloc(&db, hb.size() * 4));
  HIP_TRY(hipMalloc(&dy, (size_t)S * M * N * 4));
  HIP_TRY(hipMalloc(&dref, (size_t)M * N * 4));
  HIP_TRY(hipMalloc(&dm, 4));
  HIP_TRY(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dm, &M, 4, hipMemcpyHostToDevice));
  hipStream_t st = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const bool was_profile = s->profile;
  s->profile = false;
  const int keep_min = s->big_min_tiles;
  auto launch = [&](int var, float* y, int n_slabs) -> int {
    int gv = GV_BIG;
    s->big_min_tiles = keep_min;
    if (var == 2) s->big_min_tiles = 0;
    else if (var == 46) s->big_min_tiles = 1;
    else if (var == 3 || var == 8) gv = GV_SMALL;
    return launch_gemm(s, st, dx, K, dw, K, n_slabs > 0 ? nullptr : db, y, N, dm, M, N, K, false, n_slabs, (long long)M * N, gv);
  };
  int rc = launch(2, dref, 0);                                     // reference: 64x64 tiles, one workgroup walks all slices
  const int slabs = (variant == 8) ? gemm_splits(N, K, true, GV_SMALL) : splits;
  const int Sx = slabs > 0 ? slabs : 1;
  if (rc == TTX_OK && Sx > S) rc = fail(TTX_ERR_INVALID, "variant 8 needs splits >= K / 256 slabs of output room");
  for (int i = 0; i < 3 && rc == TTX_OK; ++i) rc = launch(variant, dy, slabs);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (rc == TTX_OK) {
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < reps && rc == TTX_OK; ++i) rc = launch(variant, dy, slabs);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (us_per_launch) *us_per_launch = 1e3 * ms / reps;
  }
  if (rc == TTX_OK && max_abs_diff) {
    std::vector<float> y((size_t)Sx * M * N), yr((size_t)M * N);
    HIP_TRY(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(yr.data(), dref, yr.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (size_t i = 0; i < (size_t)M * N; ++i) {
      float u = y[i];
      for (int k = 1; k < Sx; ++k) u += y[(size_t)k * M * N + i];             // fp32, slab order: what k_finish_ln does
      if (slabs > 0) u += hb[i % N];                                            // the reference launch added the bias in its epilogue
      worst = std::max(worst, (double)std::fabs(u - yr[i]));
    }
    *max_abs_diff = worst;
  }
  s->big_min_tiles = keep_min; s->profile = was_profile;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(st);
  (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(db); (void)hipFree(dy); (void)hipFree(dref); (void)hipFree(dm);
  return rc;
}

}  // namespace ttx
