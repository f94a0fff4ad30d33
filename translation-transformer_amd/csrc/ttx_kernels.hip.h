// Device code of libttx_hip.so — hand-written HIP for gfx950 (CDNA4, wave64).  fp32 throughout:
// the reference runs with `precision: null` (configs/cfg_standard_product_prediction.yaml:7) and the
// parity bar is token identity, so the dense contractions use the f32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain), not a reduced-precision format.
//
// Kernel inventory:
//   k_gemm_tn        K2/K5/K6/K8  Y = act(X·Wᵀ + b) or raw split-K slabs; LDS-staged 32-deep K tiles,
//                                 one 32x32 MFMA accumulator tile per wave
//   k_finish_ln      K5/K6/K7     sum of split-K slabs + bias + residual + LayerNorm (+ final stack norm)
//   k_attn<mode>     K3/K4        small-sequence attention, one wave per (row, head, <=16 queries),
//                                 wavefront-shuffle softmax; KV-cache + in-flight draft keys
//   k_embed_*        K1           token embedding + sinusoid row (pos + 1)
//   k_argmax         K8           wavefront-shuffle argmax over the vocabulary
//   k_make_drafts    K9           sliding-window drafts with the reference's fp32 index spacing
//   k_accept / k_kvcopy  K10      verify + accept + retire + active-row compaction on the device
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ttx_select.h"

namespace ttx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// fp32 product from bf16 pieces (opt-in experiment, TTX_FFN_BF16X6=1): an fp32 value splits EXACTLY into three
// bf16 numbers (its 24 mantissa bits in groups of 8: hi = top 16 bits of x, mid = top 16 bits of x - hi, lo = top 16 bits of
// the rest), a bf16 x bf16 product is exact in fp32, and of the nine partial products of x * w the three smallest
// (mid*lo, lo*mid, lo*lo: <= 2^-24 relative) are dropped — six v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) instead of
// eight v_mfma_f32_32x32x2_f32 (64 cycles each): 2.7x the matrix-pipe rate at fp32-level error.  The operands stay fp32 in
// memory and in LDS; a wave splits its own fragments in registers (about 5.5 VALU operations per value, issued beside the MFMAs).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct B6Frag { bf16x8 hi, mid, lo; };     // 8 consecutive k of one row / column

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ B6Frag b6_split(const float4& u, const float4& v) {
  // pairs of values: the two exact subtractions per value are packed fp32 operations (v_pk_add_f32), the masks plain ANDs, and
  // one v_perm_b32 per pair and piece packs the two upper halves into a bf16x2 register
  const f32x2 x[4] = {{u.x, u.y}, {u.z, u.w}, {v.x, v.y}, {v.z, v.w}};
  union { unsigned w[4]; bf16x8 v; } H, M, L;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32x2 xb = __builtin_bit_cast(u32x2, x[j]);
    const f32x2 r1 = x[j] - __builtin_bit_cast(f32x2, xb & 0xffff0000u);          // exact
    const u32x2 r1b = __builtin_bit_cast(u32x2, r1);
    const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, r1b & 0xffff0000u);           // exact
    const u32x2 r2b = __builtin_bit_cast(u32x2, r2);
    H.w[j] = __builtin_amdgcn_perm(xb.y, xb.x, 0x07060302u);                       // element 2j in the low half, 2j+1 in the high half
    M.w[j] = __builtin_amdgcn_perm(r1b.y, r1b.x, 0x07060302u);
    L.w[j] = __builtin_amdgcn_perm(r2b.y, r2b.x, 0x07060302u);
  }
  return B6Frag{H.v, M.v, L.v};
}

// c += a * b over 16 k's: smallest partial products first
__device__ __forceinline__ void b6_mma(f32x16& c, const B6Frag& a, const B6Frag& b) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Device-resident loop state of one generate call (one per session).
struct DecState {
  int n_active;        // Bc: rows still decoding
  int r_rows;          // Bc * N
  int m_rows;          // Bc * N * (D+1): rows of this step's GEMMs
  int stop;            // loop finished
  int width;           // the reference's generated_tokens.size(1)
  int steps;           // decoder calls so far (model_calls_num)
  int error;           // 1: a row finished at width > max_len (the reference raises there)
  int n_copy;          // rows whose accepted K/V must be copied into the cache after this step
  long long accepted, produced, verified_positions, kv_prefix_positions, src_positions;
};

struct CopyRec { int b, best, nacc, front_old, flags; };   // flags: 1 finished this step, 2 retired without finishing

// Host-mapped (pinned) words the accept kernels publish after every step; the host polls them instead of
// synchronising the stream.
struct HostInfo { int stop; int steps_done; int width; int n_active; };

// ------------------------------------------------------------------------------------------------
// GEMM:  Y[m, n] = sum_k X[m, k] * W[n, k]   (torch.nn.Linear layout: both operands K-contiguous)
struct GemmArgs {
  const float* X; int ldx;
  const float* W; int ldw;
  const float* bias;         // may be null
  float* Y; int ldy;
  const int* m_ptr;          // device-resident row count (null: use M)
  int M, N, K;
  int k_per_split;           // K range handled by one blockIdx.z
  int relu;                  // epilogue
  int raw;                   // 1: write un-biased partial sums to slab blockIdx.z
  long long slab_stride;     // floats between slabs
  unsigned long long* dbg;   // diagnostic builds only: per-workgroup phase stamps (100 MHz realtime clock)
  int big_min_tiles;         // k_gemm24: smallest 128-row-tile count (x splits) at which a 128-row tiling is used
  int big_wide_tiles;        // k_gemm24: smallest 128x128-tile count at which that tiling is preferred over 128x64
};

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
// GEMM v2 for the decode-step shapes (M <= ~1000 rows, K = 256 per workgroup): every launch is a few
// hundred workgroups at most, so one workgroup's time is load latency, not bandwidth.  Each thread
// therefore keeps a ring of four 64-deep K tiles in flight in registers (all of K = 256 is requested
// from L2/Infinity Cache before the first MFMA), LDS is double-buffered so one barrier per tile
// suffices, and the 64x64 output tile is four 32x32 fp32-MFMA accumulators, one per wave.
#ifndef TTX_G2_BUFS
#define TTX_G2_BUFS 2
#endif
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

// the same tile from bf16 pieces: ap / bp point at the lane's row / column with offset 8h (not 4h): 8 consecutive k per lane
template <int BK, int LDT>
__device__ __forceinline__ void g2_mma_b6(f32x16& acc, const float* ap, const float* bp) {
  B6Frag a = b6_split(*reinterpret_cast<const float4*>(ap), *reinterpret_cast<const float4*>(ap + 4));
  B6Frag b = b6_split(*reinterpret_cast<const float4*>(bp), *reinterpret_cast<const float4*>(bp + 4));
#pragma unroll
  for (int kk = 0; kk < BK; kk += 16) {
    // the next step's fragments are split between the six MFMAs of this one (a 32x32 tile per wave has twice the split
    // work per MFMA of the 128-row tilings: this path is VALU-bound)
    const int kn = (kk + 16 < BK) ? kk + 16 : kk;
    const B6Frag na = b6_split(*reinterpret_cast<const float4*>(ap + kn), *reinterpret_cast<const float4*>(ap + kn + 4));
    const B6Frag nb = b6_split(*reinterpret_cast<const float4*>(bp + kn), *reinterpret_cast<const float4*>(bp + kn + 4));
    b6_mma(acc, a, b);
#pragma unroll
    for (int i_ = 0; i_ < 6; ++i_) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 15, 0);
    }
    a = na; b = nb;
  }
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

// NT = number of 64-deep K tiles per workgroup when it is 1, 2 or 4 (straight-line code, every tile
// requested up front); NT = 0: any multiple of 4 tiles, ring slots refilled as they drain.
constexpr int G24_SMEM_FLOATS = 2 * 2 * 128 * 36;      // 73 728 B: the larger of the two tilings' LDS images

template <int NT, bool B6 = false>
__device__ __forceinline__ void g2_body(const GemmArgs& a, const int M, const int bx, const int by, const int bz, float* smem) {
  constexpr int BM = 64, BN = 64, BK = 64, LDT = BK + 4, RING = 4;
  typedef float (*TileBufs)[BM * LDT];
  TileBufs As = reinterpret_cast<TileBufs>(smem);
  TileBufs Bs = reinterpret_cast<TileBufs>(smem + TTX_G2_BUFS * BM * LDT);

  unsigned long long* dbg = a.dbg ? a.dbg + 8 * (size_t)(bx + gridDim.x * (by + gridDim.y * bz)) : nullptr;
#define TTX_GSTAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
  TTX_GSTAMP(0);
  const int m0 = by * BM, n0 = bx * BN;
  if (m0 >= M) return;
  TTX_GSTAMP(1);
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
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int aoff = (wm * 32 + r) * LDT + (B6 ? 8 : 4) * h, boff = (wn * 32 + r) * LDT + (B6 ? 8 : 4) * h;
  auto tile_mma = [&](const float* ap, const float* bp) {
    if constexpr (B6) g2_mma_b6<BK, LDT>(acc, ap, bp);
    else g2_mma<BK, LDT>(acc, ap, bp);
  };

  if constexpr (NT == 1) {
    const G2Frag f0 = g2_load(p, 0);
    g2_store<LDT>(f0, As[0], Bs[0], lr, lc);
    __syncthreads();
    tile_mma(As[0] + aoff, Bs[0] + boff);
      if (TTX_G2_BUFS == 1) __syncthreads();
  } else if constexpr (NT == 2) {
    const G2Frag f0 = g2_load(p, 0);
    const G2Frag f1 = g2_load(p, BK);
    g2_store<LDT>(f0, As[0], Bs[0], lr, lc);
    if (TTX_G2_BUFS == 2) g2_store<LDT>(f1, As[TTX_G2_BUFS - 1], Bs[TTX_G2_BUFS - 1], lr, lc);
    __syncthreads();
    tile_mma(As[0] + aoff, Bs[0] + boff);
    if (TTX_G2_BUFS == 1) {
      __syncthreads();
      g2_store<LDT>(f1, As[0], Bs[0], lr, lc);
      __syncthreads();
    }
    tile_mma(As[TTX_G2_BUFS - 1] + aoff, Bs[TTX_G2_BUFS - 1] + boff);
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
      if (base == 0) TTX_GSTAMP(2);
      tile_mma(As[0] + aoff, Bs[0] + boff);
      if (TTX_G2_BUFS == 1) __syncthreads();
      g2_store<LDT>(f1, As[TTX_G2_BUFS - 1], Bs[TTX_G2_BUFS - 1], lr, lc);
      if constexpr (NT == 0) f1 = g2_load(p, min(base + RING + 1, last) * BK);
      __syncthreads();
      tile_mma(As[TTX_G2_BUFS - 1] + aoff, Bs[TTX_G2_BUFS - 1] + boff);
      if (TTX_G2_BUFS == 1) __syncthreads();
      g2_store<LDT>(f2, As[0], Bs[0], lr, lc);
      if constexpr (NT == 0) f2 = g2_load(p, min(base + RING + 2, last) * BK);
      __syncthreads();
      tile_mma(As[0] + aoff, Bs[0] + boff);
      if (TTX_G2_BUFS == 1) __syncthreads();
      g2_store<LDT>(f3, As[TTX_G2_BUFS - 1], Bs[TTX_G2_BUFS - 1], lr, lc);
      if constexpr (NT == 0) f3 = g2_load(p, min(base + RING + 3, last) * BK);
      __syncthreads();
      tile_mma(As[TTX_G2_BUFS - 1] + aoff, Bs[TTX_G2_BUFS - 1] + boff);
      if (TTX_G2_BUFS == 1) __syncthreads();
    }
  }

  TTX_GSTAMP(3);
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
  if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TTX_GSTAMP(4); }
#undef TTX_GSTAMP
}

template <int NT>
__global__ __launch_bounds__(256) void k_gemm2(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[2 * TTX_G2_BUFS * 64 * 68];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  g2_body<NT>(a, M, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// ------------------------------------------------------------------------------------------------
// GEMM v4 for launches with thousands of rows (row schedule: M up to ~8 000 step rows; encoder / cross-K/V bulk
// passes): 128x128 output tile per workgroup, 2x2 waves of 64x64 (four independent 32x32 fp32-MFMA accumulators
// per wave, so back-to-back MFMAs never wait on each other), 32-deep K tiles, LDS double buffer (stride 36 floats:
// conflict-free ds_read_b128), one barrier per tile.  Against the 64x64 kernel: half the L2->LDS bytes and half the
// LDS->register bytes per MFMA, 4 096 MFMA cycles per wave between barriers instead of 2 048.  The next tile's
// global loads are issued before the MFMA block of the current one and written to the other LDS buffer after the
// following barrier (loads are unconditional, tile index clamped: counted vmcnt waits).
struct G4Frag { float4 a0, a1, a2, a3, b0, b1, b2, b3; };

// BN = 128: 2x2 waves of 64x64 (four accumulators per wave).  BN = 64: 2x2 waves of 64x32 (two accumulators per wave, one B
// fragment): twice the workgroups for the launches that are short of them (N <= 768 at a few thousand rows), so that two
// are resident per CU and one's prologue / epilogue overlaps the other's MFMAs.  Same K order in one accumulator per
// element as every other tiling: bit-identical results.
template <int BN, bool B6 = false>
__device__ __forceinline__ void g4_body(const GemmArgs& a, const int M, const int bx, const int by, const int bz, float* smem) {
  constexpr int BM = 128, BK = 32, LDT = BK + 4;
  constexpr int WN = BN / 2;                      // columns per wave
  typedef float (*TileBufsA)[BM * LDT];
  typedef float (*TileBufsB)[BN * LDT];
  TileBufsA As = reinterpret_cast<TileBufsA>(smem);
  TileBufsB Bs = reinterpret_cast<TileBufsB>(smem + 2 * BM * LDT);
  unsigned long long* dbg = a.dbg ? a.dbg + 8 * (size_t)(bx + gridDim.x * (by + gridDim.y * bz)) : nullptr;
#ifdef TTX_G4_STAMP_OUTER
  if (dbg && threadIdx.x == 0) dbg[0] = clock64();
#endif
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
  const float* wp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xp[i] = a.X + (size_t)min(m0 + lr + 32 * i, M - 1) * a.ldx + kbeg + lc;
    wp[i] = a.W + (size_t)min(n0 + min(lr + 32 * i, BN - 1), a.N - 1) * a.ldw + kbeg + lc;
  }
  auto gload = [&](int tile) {
    G4Frag f;
    const int ko = tile * BK;
    f.a0 = *reinterpret_cast<const float4*>(xp[0] + ko);
    f.a1 = *reinterpret_cast<const float4*>(xp[1] + ko);
    f.a2 = *reinterpret_cast<const float4*>(xp[2] + ko);
    f.a3 = *reinterpret_cast<const float4*>(xp[3] + ko);
    f.b0 = *reinterpret_cast<const float4*>(wp[0] + ko);
    f.b1 = *reinterpret_cast<const float4*>(wp[1] + ko);
    if constexpr (BN == 128) {
      f.b2 = *reinterpret_cast<const float4*>(wp[2] + ko);
      f.b3 = *reinterpret_cast<const float4*>(wp[3] + ko);
    }
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
    if constexpr (BN == 128) {
      *reinterpret_cast<float4*>(bs + 64 * LDT) = f.b2;
      *reinterpret_cast<float4*>(bs + 96 * LDT) = f.b3;
    }
  };

  f32x16 c00, c01, c10, c11;
#pragma unroll
  for (int i = 0; i < 16; ++i) { c00[i] = 0.f; c01[i] = 0.f; c10[i] = 0.f; c11[i] = 0.f; }
  const int aoff = (wm * 64 + r) * LDT + (B6 ? 8 : 4) * h, boff = (wn * WN + r) * LDT + (B6 ? 8 : 4) * h;
#ifdef TTX_G4_STAGGER
  // Two workgroups share a CU.  Dispatched together they run in phase: both in their prologue, both in their epilogue at
  // the same time, the MFMA pipe idle then.  The second half of the first wave of workgroups (the ones that land beside
  // workgroups 0..255) starts half a tile late; later workgroups inherit the offset from the slots they take over.
  {
    const int lin0 = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (lin0 >= 256 && lin0 < 512)
      for (int q = 0; q < TTX_G4_STAGGER; ++q) __builtin_amdgcn_s_sleep(100);      // 100 x 64 cycles each
  }
#endif
#ifdef TTX_G4_PRIO
  // Two workgroups share a CU (one wave of each per SIMD).  With equal priority the SIMD alternates between their
  // MFMAs, both advance in lockstep and reach their LDS/barrier phases together, leaving the MFMA pipe idle then.
  // Giving the workgroup whose first wave sits in an odd wave slot a higher priority makes the phases complementary.
  __shared__ int s_prio;
  if (t == 0) s_prio = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1;     // HW_ID.WAVE_ID
  __syncthreads();
  if (s_prio) __builtin_amdgcn_s_setprio(3);
#endif

  // One 32-deep K tile: 16 MFMAs per 8 k's.  `during(q)` (q = 0..3) runs after the first MFMA group of every 8-k
  // step: the LDS writes of the NEXT tile go there, between MFMAs, so that they cost no MFMA time (the matrix pipe runs
  // on while the wave issues them).
  auto mma = [&](int buf, auto&& during) {
    const float* ap = As[buf] + aoff;
    const float* bp = Bs[buf] + boff;
    if constexpr (B6) {
      // bf16 pieces: two 16-k steps per tile, 8 consecutive k per lane; same K order in one accumulator per element for
      // every tiling (g2_mma_b6): bit-identical across tilings like the fp32 path.  The split of the NEXT fragment (about 44
      // VALU operations) is issued between the six MFMAs of the current accumulator update: one MFMA, eight VALU, ...
      auto ldA = [&](int blk, int kk) { return b6_split(*reinterpret_cast<const float4*>(ap + blk * 32 * LDT + kk),
                                                         *reinterpret_cast<const float4*>(ap + blk * 32 * LDT + kk + 4)); };
      auto ldB = [&](int blk, int kk) { return b6_split(*reinterpret_cast<const float4*>(bp + blk * 32 * LDT + kk),
                                                         *reinterpret_cast<const float4*>(bp + blk * 32 * LDT + kk + 4)); };
#define TTX_B6_INTERLEAVE()                                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         \
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                         \
      }
      B6Frag fa0 = ldA(0, 0), fb0 = ldB(0, 0);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 16) {
        const B6Frag fa1 = ldA(1, kk);
        b6_mma(c00, fa0, fb0);
        TTX_B6_INTERLEAVE();
        during(kk >> 3);
        if constexpr (BN == 128) {
          const B6Frag fb1 = ldB(1, kk);
          b6_mma(c10, fa1, fb0);
          TTX_B6_INTERLEAVE();
          during((kk >> 3) + 1);
          const B6Frag na0 = (kk + 16 < BK) ? ldA(0, kk + 16) : fa0;
          b6_mma(c01, fa0, fb1);
          TTX_B6_INTERLEAVE();
          const B6Frag nb0 = (kk + 16 < BK) ? ldB(0, kk + 16) : fb0;
          b6_mma(c11, fa1, fb1);
          TTX_B6_INTERLEAVE();
          fa0 = na0; fb0 = nb0;
        } else {
          const B6Frag na0 = (kk + 16 < BK) ? ldA(0, kk + 16) : fa0;
          const B6Frag nb0 = (kk + 16 < BK) ? ldB(0, kk + 16) : fb0;
          b6_mma(c10, fa1, fb0);
          TTX_B6_INTERLEAVE();
          TTX_B6_INTERLEAVE();
          during((kk >> 3) + 1);
          fa0 = na0; fb0 = nb0;
        }
      }
#undef TTX_B6_INTERLEAVE
      return;
    }
    // fragments of the 8-k step after the current one are read from LDS while the current step's MFMAs run
    float4 a0 = *reinterpret_cast<const float4*>(ap);
    float4 a1 = *reinterpret_cast<const float4*>(ap + 32 * LDT);
    float4 b0 = *reinterpret_cast<const float4*>(bp);
    float4 b1 = b0;
    if constexpr (BN == 128) b1 = *reinterpret_cast<const float4*>(bp + 32 * LDT);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      float4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (kk + 8 < BK) {
        na0 = *reinterpret_cast<const float4*>(ap + kk + 8);
        na1 = *reinterpret_cast<const float4*>(ap + 32 * LDT + kk + 8);
        nb0 = *reinterpret_cast<const float4*>(bp + kk + 8);
        if constexpr (BN == 128) nb1 = *reinterpret_cast<const float4*>(bp + 32 * LDT + kk + 8);
      }
      if constexpr (BN == 128) {
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, c11, 0, 0, 0);
        during(kk >> 3);
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b1.z, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b1.w, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, c11, 0, 0, 0);
      } else {
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, c10, 0, 0, 0);
        during(kk >> 3);
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, c10, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, c10, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, c10, 0, 0, 0);
      }
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
  };
  // quarter q of a tile's LDS writes (two float4 of A's 128 rows, one or two of B's rows)
  auto lstore_q = [&](const G4Frag& f, int buf, int q) {
    float* as = As[buf] + lr * LDT + lc;
    float* bs = Bs[buf] + lr * LDT + lc;
    if (q == 0) { *reinterpret_cast<float4*>(as) = f.a0; *reinterpret_cast<float4*>(bs) = f.b0; }
    else if (q == 1) { *reinterpret_cast<float4*>(as + 32 * LDT) = f.a1; *reinterpret_cast<float4*>(bs + 32 * LDT) = f.b1; }
    else if (q == 2) { *reinterpret_cast<float4*>(as + 64 * LDT) = f.a2; if constexpr (BN == 128) *reinterpret_cast<float4*>(bs + 64 * LDT) = f.b2; }
    else { *reinterpret_cast<float4*>(as + 96 * LDT) = f.a3; if constexpr (BN == 128) *reinterpret_cast<float4*>(bs + 96 * LDT) = f.b3; }
  };
  // two register tiles in flight (static slots, clamped refills: same shape as k_gemm2's ring): a tile's loads are
  // issued two MFMA blocks (~3.4 us) before its LDS write.  The K range is a multiple of 64: ntiles is even.
  const int last = ntiles - 1;
  const int bcol = n0 + wn * WN + r;
  const float bias0 = (!a.raw && a.bias) ? a.bias[min(bcol, a.N - 1)] : 0.f;
  const float bias1 = (!a.raw && a.bias && BN == 128) ? a.bias[min(bcol + 32, a.N - 1)] : 0.f;
  asm volatile("" ::: "memory");
  G4Frag f0 = gload(0);
  asm volatile("" ::: "memory");        // issue order f0 then f1 also ahead of the loop: the header waits with vmcnt(8), not 0
  G4Frag f1 = gload(1);
#ifdef TTX_G4_STAMP_OUTER
#define TTX_G4STAMP(k) do { } while (0)
#define TTX_G4OUTER(k) do { if (dbg && t == 0) dbg[k] = clock64(); } while (0)
#else
#define TTX_G4STAMP(k) do { if (dbg && i == 2 && t == 0) dbg[k] = clock64(); } while (0)
#define TTX_G4OUTER(k) do { } while (0)
#endif
  TTX_G4OUTER(1);
#ifdef TTX_G4_NO_OVERLAP
  for (int i = 0; i < ntiles; i += 2) {
    TTX_G4STAMP(0);
    lstore(f0, 0);
    if (i == 0) TTX_G4OUTER(2);
    asm volatile("" ::: "memory");      // LDS writes first, then the refill into the same registers (no copies, counted vmcnt)
    TTX_G4STAMP(1);
    f0 = gload(min(i + 2, last));
    TTX_G4STAMP(2);
    __syncthreads();
    TTX_G4STAMP(3);
    mma(0, [](int) {});
    TTX_G4STAMP(4);
    lstore(f1, 1);
    asm volatile("" ::: "memory");
    f1 = gload(min(i + 3, last));
    TTX_G4STAMP(5);
    __syncthreads();
    TTX_G4STAMP(6);
    mma(1, [](int) {});
    TTX_G4STAMP(7);
  }
#else
  // Tile i is computed from one LDS buffer while tile i + 1 is written into the other BETWEEN the MFMAs (after the
  // barrier that ends a phase every wave has finished reading the buffer the next phase overwrites), and the registers
  // just emptied are refilled from global memory for tile i + 2: LDS writes, global loads and MFMAs overlap inside every
  // wave instead of only across the two workgroups of a CU.
  lstore(f0, 0);
  asm volatile("" ::: "memory");
  f0 = gload(min(2, last));
  TTX_G4OUTER(2);
  __syncthreads();
  for (int i = 0; i < ntiles; i += 2) {
    mma(0, [&](int q) { lstore_q(f1, 1, q); });
    asm volatile("" ::: "memory");
    f1 = gload(min(i + 3, last));
    __syncthreads();
    mma(1, [&](int q) { lstore_q(f0, 0, q); });          // tile i + 2 (a clamped repeat of the last tile at the end: never read)
    asm volatile("" ::: "memory");
    f0 = gload(min(i + 4, last));
    __syncthreads();
  }
#endif
  TTX_G4OUTER(3);
#undef TTX_G4STAMP

  // Epilogue.  The bias values were requested before the K loop (a load still pending here would make every
  // predicated store below wait for vmcnt(0), i.e. for the previous store: 64 serialised round trips).
  float* Y = a.Y + (a.raw ? (size_t)bz * a.slab_stride : 0);
  const float lo = a.relu ? 0.f : -INFINITY;
  auto store_tile = [&](const f32x16& c, int tm, int tn, float bv) {
    const int col = n0 + wn * WN + tn * 32 + r;
    float val[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) val[v] = fmaxf(c[v] + bv, lo);
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
  store_tile(c00, 0, 0, bias0);
  if constexpr (BN == 128) store_tile(c01, 0, 1, bias1);
  store_tile(c10, 1, 0, bias0);
  if constexpr (BN == 128) store_tile(c11, 1, 1, bias1);
#ifdef TTX_G4_STAMP_OUTER
  if (dbg && t == 0) {
    dbg[4] = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dbg[5] = clock64(); dbg[6] = dbg[5]; dbg[7] = dbg[5];
  }
#endif
#undef TTX_G4OUTER
}

__global__ __launch_bounds__(256) void k_gemm4(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[G24_SMEM_FLOATS];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  g4_body<128>(a, M, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// 128x64 tiles in isolation (tools/bench_gemm.py variant 46).
__global__ __launch_bounds__(256) void k_gemm46(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[G24_SMEM_FLOATS];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  g4_body<64>(a, M, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// One launch, two tilings: the grid is laid out for 64x64 tiles; when the row count read from the device gives the
// 128x128 tiling at least `big_min_tiles` workgroups (GemmArgs), the first workgroups in dispatch order each compute
// a 128x128 tile and the others leave at once, otherwise all compute their 64x64 tile.  Both tilings accumulate a
// K range in the same order in one accumulator per output element: the results are bit-identical, so the choice may
// follow the live row count without touching batch invariance.
template <int NT>
__global__ __launch_bounds__(256) void k_gemm24(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[G24_SMEM_FLOATS];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  const int nby = (M + 127) >> 7;
  const int big_tiles = nby * ((a.N + 127) >> 7) * (int)gridDim.z;
  const int mid_tiles = nby * ((a.N + 63) >> 6) * (int)gridDim.z;         // 128x64 tiles
  // 128x128 tiles when they fill the chip about twice over (big_wide_tiles), else 128x64 tiles when THOSE reach
  // big_min_tiles, else 64x64.  All three accumulate a K range in the same order: bit-identical results.
  const bool wide = big_tiles >= a.big_wide_tiles;
  if (wide || mid_tiles >= a.big_min_tiles) {
    // The first tiles of the whole grid in dispatch order take the work (all K slices included): the dispatcher hands
    // consecutive workgroups to consecutive CUs, so a contiguous block spreads one per CU / XCD; actives separated by idle
    // workgroups ended up two to a CU with other CUs empty (2x the time, measured).
    const int n_tiles = wide ? big_tiles : mid_tiles;
    const int nbx = wide ? (a.N + 127) >> 7 : (a.N + 63) >> 6;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (lin >= n_tiles) return;
#ifndef TTX_NO_XCD_REMAP
    // XCD-aware order (speed only): workgroups lin and lin + 8 share an XCD and its L2, so the workgroups of one XCD
    // take a CONTIGUOUS run of tiles — the column tiles of a row block (same X rows) then hit one L2 instead of eight.
    const int xb = n_tiles >> 3, xr = n_tiles & 7, xcd = lin & 7;
    const int v = xcd * xb + min(xcd, xr) + (lin >> 3);
#else
    const int v = lin;
#endif
    const int slice = v / (nbx * nby), rem = v - slice * (nbx * nby);
    if (wide) g4_body<128>(a, M, rem % nbx, rem / nbx, slice, smem);
    else g4_body<64>(a, M, rem % nbx, rem / nbx, slice, smem);
  } else {
    g2_body<NT>(a, M, blockIdx.x, blockIdx.y, blockIdx.z, smem);
  }
}

// The same launch with every product formed from bf16 pieces (b6_split / b6_mma): opt-in experiment for the FFN pair.
template <int NT>
__global__ __launch_bounds__(256, 2) void k_gemm24_b6(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[G24_SMEM_FLOATS];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  const int nby = (M + 127) >> 7;
  const int big_tiles = nby * ((a.N + 127) >> 7) * (int)gridDim.z;
  const int mid_tiles = nby * ((a.N + 63) >> 6) * (int)gridDim.z;
  const bool wide = a.big_min_tiles > 0 && big_tiles >= a.big_wide_tiles;
  if (wide || (a.big_min_tiles > 0 && mid_tiles >= a.big_min_tiles)) {
    const int n_tiles = wide ? big_tiles : mid_tiles;
    const int nbx = wide ? (a.N + 127) >> 7 : (a.N + 63) >> 6;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (lin >= n_tiles) return;
    const int xb = n_tiles >> 3, xr = n_tiles & 7, xcd = lin & 7;
    const int v = xcd * xb + min(xcd, xr) + (lin >> 3);
    const int slice = v / (nbx * nby), rem = v - slice * (nbx * nby);
    if (wide) g4_body<128, true>(a, M, rem % nbx, rem / nbx, slice, smem);
    else g4_body<64, true>(a, M, rem % nbx, rem / nbx, slice, smem);
  } else {
    g2_body<NT, true>(a, M, blockIdx.x, blockIdx.y, blockIdx.z, smem);
  }
}

// ------------------------------------------------------------------------------------------------
// GEMM v3 for launches that are short of workgroups: a 32x32 output tile per workgroup, its K range split
// over the 4 waves, MFMA operands loaded straight from global memory into the registers the MFMA reads
// (lane (r,h) owns row r / column r and the k's 8g+4h..8g+4h+3, which is exactly one float4 per 8 k's) — no
// LDS staging, no barrier before the math; the four partial tiles meet in LDS (16.5 KB) and leave as
// whole 128-B rows.  4x the workgroups of the 64x64 kernel and many of them resident per CU.
template <int KW>   // k's per wave (K range of the workgroup / 4): 16, 64 or 128; 0 = runtime loop in steps of 64
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_gemm3(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float part[4][32 * 33];
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  if (m0 >= M) return;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int r = lane & 31, h = lane >> 5;
  const int kw = (KW > 0) ? KW : a.k_per_split / 4;
  const int kbeg = blockIdx.z * a.k_per_split + wave * kw;
  const float* xp = a.X + (size_t)min(m0 + r, M - 1) * a.ldx + kbeg + 4 * h;
  const float* wp = a.W + (size_t)min(n0 + r, a.N - 1) * a.ldw + kbeg + 4 * h;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  constexpr int CH = (KW > 0 && KW < 64) ? KW : 64;      // k's requested at once per wave
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  auto load = [&](f32x4 (&av)[CH / 8], f32x4 (&bv)[CH / 8], int k0) {
#pragma unroll
    for (int g = 0; g < CH / 8; ++g) {
      av[g] = *reinterpret_cast<const f32x4*>(xp + k0 + 8 * g);
      bv[g] = *reinterpret_cast<const f32x4*>(wp + k0 + 8 * g);
    }
  };
  // The empty asm reads every destination register of a chunk: all of its requests are out before the first MFMA
  // (otherwise the scheduler pairs each load with its MFMAs and the wave eats one memory latency per pair).
  auto pin = [&](f32x4 (&av)[CH / 8], f32x4 (&bv)[CH / 8]) {
    if constexpr (CH == 64)
      asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]),
                        "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
    else
      asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(bv[0]), "+v"(bv[1]));
  };
  auto mma = [&](const f32x4 (&av)[CH / 8], const f32x4 (&bv)[CH / 8]) {
#pragma unroll
    for (int g = 0; g < CH / 8; ++g) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].x, bv[g].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].y, bv[g].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].z, bv[g].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].w, bv[g].w, acc, 0, 0, 0);
    }
  };
  if constexpr (KW > 0 && KW <= 64) {
    f32x4 a0[CH / 8], b0[CH / 8];
    load(a0, b0, 0);
    pin(a0, b0);
    mma(a0, b0);
  } else {
    // two chunks in flight: chunk i+1 is requested before chunk i is consumed (kw is a multiple of 128 here)
    f32x4 a0[CH / 8], b0[CH / 8], a1[CH / 8], b1[CH / 8];
    load(a0, b0, 0);
    for (int k0 = 0; k0 < kw; k0 += 2 * CH) {
      load(a1, b1, k0 + CH);
      pin(a0, b0);
      mma(a0, b0);
      if (k0 + 2 * CH < kw) load(a0, b0, k0 + 2 * CH);
      pin(a1, b1);
      mma(a1, b1);
    }
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
    op[i] = part[0][e] + part[1][e] + part[2][e] + part[3][e];
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
// Wave reductions (64 lanes).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------------------------------------
// y = LN2?( LN( resid + bias + sum_s slab[s] ) ) — one wave per row, VPL contiguous values per lane.
struct FinishArgs {
  const float* slabs; int n_slabs; long long slab_stride;
  const float* bias;
  const float* resid;          // [M, d]
  const float* g1; const float* b1;
  const float* g2; const float* b2;   // optional second LayerNorm (final stack norm), may be null
  const uint8_t* row_valid;    // optional: rows with 0 are written as zeros
  float* Y;
  const int* m_ptr; int M; int d; float eps;
};

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
    if (a.n_slabs <= 2) {
      // the common shapes (one or two slabs): every operand is requested before the first use — the loop version below
      // waited for each array in turn (five dependent round trips per row).  Same additions in the same order.
      const float4 r4 = *reinterpret_cast<const float4*>(a.resid + off);
      const float4 b4 = *reinterpret_cast<const float4*>(a.bias + c0);
      const float4 p0 = *reinterpret_cast<const float4*>(a.slabs + off);
      const float4 p1 = *reinterpret_cast<const float4*>(a.slabs + (size_t)(a.n_slabs - 1) * a.slab_stride + off);   // = p0 if one slab
      const float4 g1 = *reinterpret_cast<const float4*>(a.g1 + c0);
      const float4 e1 = *reinterpret_cast<const float4*>(a.b1 + c0);
      const float* g2p = a.g2 ? a.g2 : a.g1;         // clamped: loaded either way, used only if a.g2
      const float* e2p = a.g2 ? a.b2 : a.b1;
      const float4 g2 = *reinterpret_cast<const float4*>(g2p + c0);
      const float4 e2 = *reinterpret_cast<const float4*>(e2p + c0);
      const uint8_t valid = a.row_valid ? a.row_valid[row] : (uint8_t)1;
      x[0] = r4.x + b4.x; x[1] = r4.y + b4.y; x[2] = r4.z + b4.z; x[3] = r4.w + b4.w;
      x[0] += p0.x; x[1] += p0.y; x[2] += p0.z; x[3] += p0.w;
      if (a.n_slabs == 2) { x[0] += p1.x; x[1] += p1.y; x[2] += p1.z; x[3] += p1.w; }
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
#pragma unroll
  for (int i = 0; i < VPL; ++i) x[i] = a.resid[off + i] + a.bias[c0 + i];
  for (int s = 0; s < a.n_slabs; ++s) {
    const float* p = a.slabs + (size_t)s * a.slab_stride + off;
#pragma unroll
    for (int i = 0; i < VPL; ++i) x[i] += p[i];
  }
  ln_inplace<VPL>(x, a.g1, a.b1, c0, a.d, a.eps);
  if (a.g2) ln_inplace<VPL>(x, a.g2, a.b2, c0, a.d, a.eps);
  const bool keep = a.row_valid ? (a.row_valid[row] != 0) : true;
#pragma unroll
  for (int i = 0; i < VPL; ++i) a.Y[off + i] = keep ? x[i] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Attention.  One wave per (row, head, tile of <= MAXQ queries).  Keys come in two segments:
//   A: `nA` keys every query may see (KV cache prefix / encoder memory), individually maskable;
//   B: `nB` keys with the causal rule  key j visible to query i  <=>  j <= qpos0 + i.
// Scores live in LDS as S[key][MAXQ+1]; softmax by wavefront shuffles; P·V with lane = head dim.
constexpr int ATT_DH = 32;
constexpr int ATT_MAXQ = 16;
constexpr int ATT_SQ = ATT_MAXQ + 1;

__host__ __device__ inline size_t attn_lds_bytes(int max_keys) {
  return sizeof(float) * ((size_t)ATT_MAXQ * ATT_DH + (size_t)max_keys * ATT_SQ + ATT_MAXQ);
}

// keyptr(key, kp, vp): K/V row pointers of key; vis(i, key): may query i (0..nq) see key?
template <class KeyPtr, class Vis>
__device__ __forceinline__ void attn_core(const float* __restrict__ q, int ldq, int nq, int nk, KeyPtr keyptr, Vis vis,
                                          float* __restrict__ out, int ldo, float scale, float* lds) {
  const int lane = threadIdx.x & 63;
  float* Qs = lds;                               // [MAXQ][DH]
  float* S = lds + ATT_MAXQ * ATT_DH;            // [nk][SQ]
  float* inv = S + (size_t)nk * ATT_SQ;          // [MAXQ]

  for (int e = lane * 4; e < nq * ATT_DH; e += 256) {
    const int i = e / ATT_DH, c = e % ATT_DH;
    *reinterpret_cast<float4*>(&Qs[i * ATT_DH + c]) = *reinterpret_cast<const float4*>(q + (size_t)i * ldq + c);
  }
  __syncthreads();

  // phase 1: scores, lane <-> key
  for (int c0 = 0; c0 < nk; c0 += 64) {
    const int key = c0 + lane;
    if (key < nk) {
      const float* kp;
      const float* vp;
      keyptr(key, kp, vp);
      float4 kr[ATT_DH / 4];
#pragma unroll
      for (int c = 0; c < ATT_DH / 4; ++c) kr[c] = *reinterpret_cast<const float4*>(kp + 4 * c);
      for (int i = 0; i < nq; ++i) {
        const float4* qv = reinterpret_cast<const float4*>(&Qs[i * ATT_DH]);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < ATT_DH / 4; ++c) {
          const float4 qq = qv[c];
          dot = fmaf(kr[c].x, qq.x, dot); dot = fmaf(kr[c].y, qq.y, dot);
          dot = fmaf(kr[c].z, qq.z, dot); dot = fmaf(kr[c].w, qq.w, dot);
        }
        S[(size_t)key * ATT_SQ + i] = vis(i, key) ? dot * scale : -INFINITY;
      }
    }
  }
  __syncthreads();

  // phase 2: softmax over keys, one query at a time
  for (int i = 0; i < nq; ++i) {
    float m = -INFINITY;
    for (int key = lane; key < nk; key += 64) m = fmaxf(m, S[(size_t)key * ATT_SQ + i]);
    m = wave_max(m);
    float sum = 0.f;
    for (int key = lane; key < nk; key += 64) {
      const float sv = S[(size_t)key * ATT_SQ + i];
      const float p = (sv == -INFINITY) ? 0.f : expf(sv - m);
      S[(size_t)key * ATT_SQ + i] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) inv[i] = sum > 0.f ? 1.0f / sum : 0.f;
  }
  __syncthreads();

  // phase 3: out[i][d] = sum_key P[i][key] V[key][d]; lane = d + 32 * (key parity)
  const int d = lane & 31, half = lane >> 5;
  float acc[ATT_MAXQ];
#pragma unroll
  for (int i = 0; i < ATT_MAXQ; ++i) acc[i] = 0.f;
#pragma unroll 4
  for (int key = half; key < nk; key += 2) {
    const float* kp;
    const float* vp;
    keyptr(key, kp, vp);
    const float v = vp[d];
    const float* pr = &S[(size_t)key * ATT_SQ];
#pragma unroll
    for (int i = 0; i < ATT_MAXQ; ++i)
      if (i < nq) acc[i] = fmaf(pr[i], v, acc[i]);
  }
#pragma unroll
  for (int i = 0; i < ATT_MAXQ; ++i) acc[i] += __shfl_xor(acc[i], 32, 64);
  if (half == 0) {
#pragma unroll
    for (int i = 0; i < ATT_MAXQ; ++i)
      if (i < nq) out[(size_t)i * ldo + d] = acc[i] * inv[i];
  }
}

enum AttnMode { ATT_ENC = 0, ATT_FULL_SELF = 1, ATT_FULL_CROSS = 2, ATT_STEP_SELF = 3, ATT_STEP_CROSS = 4 };

// Step-mode row layout (one verify step): a running sequence ("slot") owns RPS = 1 + N*D consecutive rows:
//   row 0                      the token at the row's front (position f) — identical for all N drafts, computed once
//   row 1 + n*D + (j-1)        token j (1..D) of draft n, at position f + j
__host__ __device__ inline int step_rps(int N, int D) { return 1 + N * D; }

struct AttnArgs {
  const float* q; int ldq;         // query rows (packed QKV buffer or a plain [M,d] buffer)
  const float* k; const float* v; int ldkv;   // step/encoder keys (packed QKV buffer) or cross K/V
  float* out; int d;               // [M, d]
  float scale;
  int L;                           // ENC: Ls; FULL_*: Lt (queries per row)
  int Lk;                          // FULL_CROSS / STEP_CROSS: Ls
  const int* tok; int pad;         // ENC: src tokens [B,Ls]; FULL_SELF: tgt tokens [R,Lt]; STEP_SELF: gen [B, gen_ld]
  const uint8_t* key_pad;          // FULL_CROSS: [Rm, Ls] 1 = PAD key; STEP_CROSS/ENC: src_valid (1 = real token)
  const int* mem_row;              // FULL_CROSS: decoder row -> memory row (null: identity)
  // step modes
  const DecState* st; const int* act_idx; const int* front;
  const int* src_of;               // step modes: running row -> source row of the encoder memory (null: identity)
  const int* src_len;              // STEP_CROSS: keys of each row's source (null: Lk for every row) — slot pool
  const float* kcache; const float* vcache; long long cache_seq_stride;  // floats per sequence in the cache
  int gen_ld; int N; int D;
  unsigned long long* dbg;         // diagnostic builds only: per-block phase stamps (100 MHz realtime clock)
};

// Streaming fallback for key counts beyond the LDS images of k_attn2: one wave per (group, head, <=16 queries).
template <int MODE>
__global__ __launch_bounds__(64) void k_attn(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int hd = blockIdx.y * ATT_DH;
  const int q0 = blockIdx.z * ATT_MAXQ;
  const int g = blockIdx.x;

  if constexpr (MODE == ATT_ENC || MODE == ATT_FULL_SELF || MODE == ATT_FULL_CROSS) {
    const int nq = min(ATT_MAXQ, a.L - q0);
    if (nq <= 0) return;
    const size_t row0 = (size_t)g * a.L;
    if constexpr (MODE == ATT_FULL_CROSS) {
      const int mr = a.mem_row ? a.mem_row[g] : g;
      const size_t mrow0 = (size_t)mr * a.Lk;
      const uint8_t* kpad = a.key_pad + mrow0;
      const float* kb = a.k + mrow0 * a.ldkv + hd;
      const float* vb = a.v + mrow0 * a.ldkv + hd;
      const int ld = a.ldkv;
      attn_core(a.q + (row0 + q0) * a.ldq + hd, a.ldq, nq, a.Lk,
                [=](int key, const float*& kp, const float*& vp) { kp = kb + (size_t)key * ld; vp = vb + (size_t)key * ld; },
                [=](int, int key) { return kpad[key] == 0; },
                a.out + (row0 + q0) * a.d + hd, a.d, a.scale, lds);
    } else {
      const int* tk = a.tok + row0;
      const int pad = a.pad;
      const float* kb = a.k + row0 * a.ldkv + hd;
      const float* vb = a.v + row0 * a.ldkv + hd;
      const int ld = a.ldkv;
      const bool causal = (MODE == ATT_FULL_SELF);
      attn_core(a.q + (row0 + q0) * a.ldq + hd, a.ldq, nq, causal ? min(a.L, q0 + nq) : a.L,
                [=](int key, const float*& kp, const float*& vp) { kp = kb + (size_t)key * ld; vp = vb + (size_t)key * ld; },
                [=](int i, int key) { return tk[key] != pad && (!causal || key <= q0 + i); },
                a.out + (row0 + q0) * a.d + hd, a.d, a.scale, lds);
    }
  } else {
    if (g >= a.st->n_active) return;
    const int RPS = step_rps(a.N, a.D), D = a.D;
    const int nq = min(ATT_MAXQ, RPS - q0);
    if (nq <= 0) return;
    const int b = a.act_idx[g];
    const size_t srow0 = (size_t)g * RPS;              // first step row of this slot
    if constexpr (MODE == ATT_STEP_SELF) {
      const int f = a.front[b];
      const int* tk = a.tok + (size_t)b * a.gen_ld;
      const int pad = a.pad;
      const float* kc = a.kcache + (size_t)b * a.cache_seq_stride + hd;
      const float* vc = a.vcache + (size_t)b * a.cache_seq_stride + hd;
      const float* kb = a.k + srow0 * a.ldkv + hd;
      const float* vb = a.v + srow0 * a.ldkv + hd;
      const int ld = a.ldkv, dd = a.d;
      const bool front_ok = tk[f] != pad;
      // keys: cached prefix [0,f), then every step row of the slot (row 0 = position f, draft rows after it)
      attn_core(a.q + (srow0 + q0) * a.ldq + hd, a.ldq, nq, f + RPS,
                [=](int key, const float*& kp, const float*& vp) {
                  if (key < f) { kp = kc + (size_t)key * dd; vp = vc + (size_t)key * dd; }
                  else { kp = kb + (size_t)(key - f) * ld; vp = vb + (size_t)(key - f) * ld; }
                },
                [=](int i, int key) {
                  if (key < f) return tk[key] != pad;
                  const int kr = key - f, qr = q0 + i;
                  if (kr == 0) return front_ok;               // position f is visible to every step row
                  if (qr == 0) return false;
                  const int kn = (kr - 1) / D, qn = (qr - 1) / D;
                  return kn == qn && kr <= qr;                // same draft, not later
                },
                a.out + (srow0 + q0) * a.d + hd, a.d, a.scale, lds);
    } else {
      const size_t mrow0 = (size_t)(a.src_of ? a.src_of[b] : b) * a.Lk;
      const uint8_t* kv = a.key_pad + mrow0;
      const float* kb = a.k + mrow0 * a.ldkv + hd;
      const float* vb = a.v + mrow0 * a.ldkv + hd;
      const int ld = a.ldkv;
      // slot pool: only the slot's own source positions hold data (the rest of its row is stale or uninitialised,
      // and a masked key's V still enters 0 * V)
      attn_core(a.q + (srow0 + q0) * a.ldq + hd, a.ldq, nq, a.src_len ? a.src_len[b] : a.Lk,
                [=](int key, const float*& kp, const float*& vp) { kp = kb + (size_t)key * ld; vp = vb + (size_t)key * ld; },
                [=](int, int key) { return kv[key] != 0; },
                a.out + (srow0 + q0) * a.d + hd, a.d, a.scale, lds);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Attention v2 (the fast path; k_attn above stays as the long-sequence fallback).
// One 256-thread workgroup per (group, head, tile of <= 32 queries) where a group is one source /
// decoder row / running sequence, so the N drafts of a sequence share one pass over the cached prefix
// and over the encoder memory.  K and V of every visible key are staged in LDS by coalesced 128-B row
// loads issued back to back (one latency, not one per key chunk); S = Q·Kᵀ and O = P·V run on the fp32
// MFMA (32x32x2), key tiles / key ranges split over the 4 waves; softmax by wavefront shuffles.
constexpr int A2_MT = 32;                  // rows of one MFMA tile
constexpr int A2_QT = 64;                  // queries per workgroup (two MFMA row tiles share the staged K/V)
constexpr int A2_LDQ = ATT_DH + 4;         // LDS row stride of Q and K rows (conflict-free ds_read_b128)

__host__ __device__ inline int a2_nkp(int nk) { return (nk + 31) & ~31; }
constexpr int A2_OP = 4 * A2_MT * 33;      // floats of the four waves' partial output tiles (aliased onto the score image)
constexpr int A2_VSTEPS = 12;              // V rows held in registers: 8 keys per step per wave -> up to 384 keys
__host__ __device__ inline int a2_qcap(int q_per_group) { return q_per_group <= A2_MT ? A2_MT : A2_QT; }
__host__ __device__ inline size_t a2_score_floats(int nkp) {
  const size_t sf = (size_t)A2_MT * (nkp + 4);
  return sf > (size_t)A2_OP ? sf : (size_t)A2_OP;
}
__host__ __device__ inline size_t attn2_lds_bytes(int max_keys, int qcap) {
  const size_t nkp = a2_nkp(max_keys);
  return sizeof(float) * ((size_t)qcap * A2_LDQ + nkp * A2_LDQ + a2_score_floats((int)nkp) + A2_MT) +
         sizeof(int) * (nkp + A2_QT);
}
__host__ __device__ inline bool attn2_fits(int max_keys) { return a2_nkp(max_keys) <= 32 * A2_VSTEPS; }

// Visibility is decided from one int per key and one per query, computed once while staging:
//   key flag A2_MASKED   masked (PAD key / padding row)
//   key flag A2_ALL      visible to every query (cached prefix, encoder memory)
//   otherwise a2_flag(group, position): visible to queries of the same group at position >= key position.
// Groups are spaced 2^17 apart and positions are < 2^16, so "same group and kpos <= qpos" is the single unsigned
// comparison (qf - kf) < 2^16.
constexpr int A2_ALL = 0x7fffffff;
constexpr int A2_MASKED = 0x7ffffffe;
__host__ __device__ inline int a2_flag(int group, int pos) { return (group << 17) | pos; }
__device__ __forceinline__ bool a2_visible(int qf, int kf) {
  return kf == A2_ALL || (unsigned)(qf - kf) < 65536u;
}

// keyptr(key, kp, vp): branch-free K/V row pointers of key (0 <= key < nk); keyflag(key), qflag(qi): see above.
template <class KeyPtr, class KeyFlag, class QFlag>
__device__ __forceinline__ void attn2_core(const float* __restrict__ q, int ldq, int nq, int nk, KeyPtr keyptr, KeyFlag keyflag,
                                           QFlag qflag, float* __restrict__ out, int ldo, float scale, float* lds, int qcap,
                                           unsigned long long* dbg = nullptr) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int r = lane & 31, h = lane >> 5;
  const int nkp = a2_nkp(nk);
#define TTX_STAMP(i) do { if (dbg && t == 0) dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
  TTX_STAMP(0);
  const int lds_s = nkp + 4;
  float* Qs = lds;                                   // [qcap][36]
  float* Ks = Qs + qcap * A2_LDQ;                    // [nkp][36]
  float* S = Ks + (size_t)nkp * A2_LDQ;              // [32][nkp+4]   scores of the current row tile
  float* Op = S;                                     // [4 waves][32][33] partial outputs reuse the score image
  float* inv = S + a2_score_floats(nkp);             // [32]
  int* kfl = reinterpret_cast<int*>(inv + A2_MT);    // [nkp]
  int* qfl = kfl + nkp;                              // [64]

  // ---- stage Q, K, V: a row is 32 floats = 8 lanes x float4.  Loads are unconditional (indices are clamped;
  // rows past nq / nk are masked through the flags) and all of them are requested before the first LDS write.
  const int lr = t >> 3, lc = (t & 7) * 4;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 qv0 = *reinterpret_cast<const f32x4*>(q + (size_t)min(lr, nq - 1) * ldq + lc);
  f32x4 qv1 = qv0;
  if (qcap > A2_MT) qv1 = *reinterpret_cast<const f32x4*>(q + (size_t)min(lr + 32, nq - 1) * ldq + lc);
  constexpr int A2_U = 8;                            // passes of 32 keys in flight
  // V never touches LDS: in O = P·V lane (dh, h) needs V[key][dh] for its wave's keys only, so each wave keeps its
  // nkp/4 value rows in registers (requested here, consumed after the softmax).  Wave w owns the 8-key groups
  // w, w+4, w+8, ...: a fixed interleave, so the order in which a row's keys are summed does not depend on how far
  // the batch's padding extends (masked keys add exact zeros) — results are the same in any batch.
  const int kq = nkp / 4;
  float vr[A2_VSTEPS][4];
  for (int k0 = 0; k0 < nkp; k0 += 32 * A2_U) {
    f32x4 kv[A2_U];
#pragma unroll
    for (int u = 0; u < A2_U; ++u) {
      const float* kp;
      const float* vp;
      keyptr(min(k0 + u * 32 + lr, nk - 1), kp, vp);
      kv[u] = *reinterpret_cast<const f32x4*>(kp + lc);
    }
    if (k0 == 0) {
#pragma unroll
      for (int si = 0; si < A2_VSTEPS; ++si) {
        if (si * 8 < kq) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float* kp;
            const float* vp;
            keyptr(min((wave + 4 * si) * 8 + 4 * h + jj, nk - 1), kp, vp);
            vr[si][jj] = vp[r];
          }
        }
      }
      // the visibility flags' token loads ride behind the K/V requests
      for (int key = t; key < nkp; key += 256) kfl[key] = (key < nk) ? keyflag(key) : A2_MASKED;
      if (t < A2_QT) qfl[t] = qflag(t);
    }
    // every K request is in flight before the first value is consumed: the empty asm reads all eight registers,
    // so the scheduler cannot sink a load down to its LDS write
    asm volatile("" : "+v"(kv[0]), "+v"(kv[1]), "+v"(kv[2]), "+v"(kv[3]), "+v"(kv[4]), "+v"(kv[5]), "+v"(kv[6]), "+v"(kv[7]));
#pragma unroll
    for (int u = 0; u < A2_U; ++u) {
      const int key = k0 + u * 32 + lr;
      if (k0 + u * 32 < nkp) *reinterpret_cast<f32x4*>(&Ks[(size_t)key * A2_LDQ + lc]) = kv[u];
    }
  }
  *reinterpret_cast<f32x4*>(&Qs[lr * A2_LDQ + lc]) = qv0;
  if (qcap > A2_MT) *reinterpret_cast<f32x4*>(&Qs[(lr + 32) * A2_LDQ + lc]) = qv1;
  __syncthreads();
  TTX_STAMP(1);

  const int n_mt = (nq + A2_MT - 1) / A2_MT;         // 1 or 2 row tiles

  // The two row tiles (queries 0-31, 32-63) run one after the other through the same score buffer: K and V
  // stay staged, the S image is reused.
  const int n_kt = nkp / 32;
  for (int mt = 0; mt < n_mt; ++mt) {
    const int q0 = mt * A2_MT;
    // ---- S = scale * Q Kᵀ with masking; key tiles are dealt to the 4 waves
    for (int kt = wave; kt < n_kt; kt += 4) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ap = &Qs[(q0 + r) * A2_LDQ + 4 * h];
      const float* bp = &Ks[(size_t)(kt * 32 + r) * A2_LDQ + 4 * h];
#pragma unroll
      for (int kk = 0; kk < ATT_DH; kk += 8) {
        const float4 av = *reinterpret_cast<const float4*>(ap + kk);
        const float4 bv = *reinterpret_cast<const float4*>(bp + kk);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
      }
      const int key = kt * 32 + r;
      const int kf = kfl[key];
      const int* qf = qfl + q0 + 4 * h;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int dq = (v & 3) + 8 * (v >> 2);        // rows past nq carry a query flag that sees nothing but A2_ALL
        S[(size_t)(dq + 4 * h) * lds_s + key] = a2_visible(qf[dq], kf) ? acc[v] * scale : -INFINITY;
      }
    }
    __syncthreads();
    if (mt == 0) TTX_STAMP(2);

    // ---- softmax: every wave owns 8 query rows, 8 lanes per row (each lane nkp/8 keys, 3-step shuffles).
    // Rows past nq hold finite or -inf scores of no consequence: their outputs are never stored.
    {
      const int ql = wave * 8 + (lane >> 3), l8 = lane & 7;
      float* row = S + (size_t)ql * lds_s;
      float m = -INFINITY;
      for (int key = l8; key < nkp; key += 8) m = fmaxf(m, row[key]);
      m = fmaxf(m, __shfl_xor(m, 4, 8));
      m = fmaxf(m, __shfl_xor(m, 2, 8));
      m = fmaxf(m, __shfl_xor(m, 1, 8));
      const float mm = (m == -INFINITY) ? 0.f : m;   // fully masked row: every p becomes exp(-inf) = 0
      float sum = 0.f;
      for (int key = l8; key < nkp; key += 8) {
        const float p = __expf(row[key] - mm);
        row[key] = p;
        sum += p;
      }
      sum += __shfl_xor(sum, 4, 8);
      sum += __shfl_xor(sum, 2, 8);
      sum += __shfl_xor(sum, 1, 8);
      if (l8 == 0) inv[ql] = sum > 0.f ? 1.0f / sum : 0.f;
    }
    __syncthreads();
    if (mt == 0) TTX_STAMP(3);

    // ---- O = P V: wave w takes keys [w*nkp/4, (w+1)*nkp/4) with its V rows from registers; partial sums meet in LDS
    {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* prow = S + (size_t)r * lds_s + 4 * h + wave * 8;
#pragma unroll
      for (int si = 0; si < A2_VSTEPS; ++si) {
        if (si * 8 < kq) {
          const float4 pv = *reinterpret_cast<const float4*>(prow + si * 32);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pv.x, vr[si][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pv.y, vr[si][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pv.z, vr[si][2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pv.w, vr[si][3], acc, 0, 0, 0);
        }
      }
      __syncthreads();                                // every wave has read its P columns: the image becomes Op
      float* part = Op + (size_t)wave * (A2_MT * 33);
#pragma unroll
      for (int v = 0; v < 16; ++v) part[((v & 3) + 8 * (v >> 2) + 4 * h) * 33 + r] = acc[v];
    }
    __syncthreads();
    if (mt == 0) TTX_STAMP(4);
    for (int e = t; e < A2_MT * ATT_DH; e += 256) {
      const int ql = e >> 5, c = e & 31;
      const float o = Op[ql * 33 + c] + Op[A2_MT * 33 + ql * 33 + c] + Op[2 * A2_MT * 33 + ql * 33 + c] +
                      Op[3 * A2_MT * 33 + ql * 33 + c];
      if (q0 + ql < nq) out[(size_t)(q0 + ql) * ldo + c] = o * inv[ql];
    }
    if (mt == 0) TTX_STAMP(5);
    if (mt + 1 < n_mt) __syncthreads();               // S, inv and Op are rewritten by the next row tile
  }
  TTX_STAMP(6);
#undef TTX_STAMP
}

// amdgpu_waves_per_eu(1, 2): the LDS images allow at most two workgroups per CU, so let the compiler keep the
// staging loads in registers (with the default occupancy target it spills them to scratch to stay under 64 VGPRs)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_attn2(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int hd = blockIdx.y * ATT_DH;
  const int tile = blockIdx.z;
  auto q_any = [](int) { return 0; };

  if constexpr (MODE == ATT_ENC || MODE == ATT_FULL_CROSS) {
    const int g = blockIdx.x;
    const int q0 = tile * A2_QT;
    const int nq = min(A2_QT, a.L - q0);
    if (nq <= 0) return;
    const size_t row0 = (size_t)g * a.L;
    if constexpr (MODE == ATT_ENC) {
      const int* tk = a.tok + row0;
      const int pad = a.pad;
      const float* kb = a.k + row0 * a.ldkv + hd;
      const float* vb = a.v + row0 * a.ldkv + hd;
      const int ld = a.ldkv;
      attn2_core(a.q + (row0 + q0) * a.ldq + hd, a.ldq, nq, a.L,
                 [=](int key, const float*& kp, const float*& vp) { kp = kb + (size_t)key * ld; vp = vb + (size_t)key * ld; },
                 [=](int key) { return tk[key] != pad ? A2_ALL : A2_MASKED; }, q_any,
                 a.out + (row0 + q0) * a.d + hd, a.d, a.scale, lds, a2_qcap(a.L));
    } else {
      const int mr = a.mem_row ? a.mem_row[g] : g;
      const size_t mrow0 = (size_t)mr * a.Lk;
      const uint8_t* kpad = a.key_pad + mrow0;
      const float* kb = a.k + mrow0 * a.ldkv + hd;
      const float* vb = a.v + mrow0 * a.ldkv + hd;
      const int ld = a.ldkv;
      attn2_core(a.q + (row0 + q0) * a.ldq + hd, a.ldq, nq, a.Lk,
                 [=](int key, const float*& kp, const float*& vp) { kp = kb + (size_t)key * ld; vp = vb + (size_t)key * ld; },
                 [=](int key) { return kpad[key] == 0 ? A2_ALL : A2_MASKED; }, q_any,
                 a.out + (row0 + q0) * a.d + hd, a.d, a.scale, lds, a2_qcap(a.L));
    }
  } else if constexpr (MODE == ATT_FULL_SELF) {
    const int g = blockIdx.x;
    const int q0 = tile * A2_QT;
    const int nq = min(A2_QT, a.L - q0);
    if (nq <= 0) return;
    const size_t row0 = (size_t)g * a.L;
    const int* tk = a.tok + row0;
    const int pad = a.pad;
    const float* kb = a.k + row0 * a.ldkv + hd;
    const float* vb = a.v + row0 * a.ldkv + hd;
    const int ld = a.ldkv;
    attn2_core(a.q + (row0 + q0) * a.ldq + hd, a.ldq, nq, min(a.L, q0 + nq),
               [=](int key, const float*& kp, const float*& vp) { kp = kb + (size_t)key * ld; vp = vb + (size_t)key * ld; },
               [=](int key) { return tk[key] != pad ? a2_flag(0, key) : A2_MASKED; },
               [=](int qi) { return a2_flag(0, q0 + qi); },
               a.out + (row0 + q0) * a.d + hd, a.d, a.scale, lds, a2_qcap(a.L));
  } else {
    // step modes: group = running sequence (slot); a workgroup takes 64 of the slot's RPS step rows
    const int slot = blockIdx.x;
    if (slot >= a.st->n_active) return;
    const int D = a.D, RPS = step_rps(a.N, a.D);
    const int r0 = tile * A2_QT;                       // first step row of this tile
    const int nq = min(A2_QT, RPS - r0);
    if (nq <= 0) return;
    const int b = a.act_idx[slot];
    const size_t srow0 = (size_t)slot * RPS;
    if constexpr (MODE == ATT_STEP_SELF) {
      const int f = a.front[b];
      const int* tk = a.tok + (size_t)b * a.gen_ld;
      const int pad = a.pad;
      const float* kc = a.kcache + (size_t)b * a.cache_seq_stride + hd;
      const float* vc = a.vcache + (size_t)b * a.cache_seq_stride + hd;
      const float* kb = a.k + srow0 * a.ldkv + hd;
      const float* vb = a.v + srow0 * a.ldkv + hd;
      const int ld = a.ldkv, dd = a.d;
      // keys: cached prefix [0,f) | step row 0 (position f) | the rows of every draft that has a query in this tile
      const int rlast = r0 + nq - 1;
      const int n_lo = (r0 == 0) ? 0 : (r0 - 1) / D;
      const int n_hi = (rlast == 0) ? -1 : (rlast - 1) / D;
      const int kr0 = 1 + n_lo * D;                    // first draft row staged
      const int n_draft_keys = (n_hi >= n_lo && D > 0) ? (n_hi - n_lo + 1) * D : 0;
      attn2_core(a.q + (srow0 + r0) * a.ldq + hd, a.ldq, nq, f + 1 + n_draft_keys,
                 [=](int key, const float*& kp, const float*& vp) {
                   const bool cached = key < f;
                   const int srow = (key == f) ? 0 : kr0 + (key - f - 1);
                   const size_t off = cached ? (size_t)key * dd : (size_t)srow * ld;
                   kp = (cached ? kc : kb) + off;
                   vp = (cached ? vc : vb) + off;
                 },
                 [=](int key) {
                   if (key <= f) return tk[key] != pad ? A2_ALL : A2_MASKED;   // prefix and the front token
                   const int kr = kr0 + (key - f - 1);
                   const int kn = (kr - 1) / D;
                   return a2_flag(kn - n_lo, kr - 1 - kn * D);
                 },
                 [=](int qi) {
                   const int qr = r0 + qi;
                   if (qr == 0 || D == 0) return a2_flag(0x3fff, 0);           // sees prefix + front token only
                   const int qn = (qr - 1) / D;
                   return a2_flag(qn - n_lo, qr - 1 - qn * D);
                 },
                 a.out + (srow0 + r0) * a.d + hd, a.d, a.scale, lds, a2_qcap(RPS),
                 a.dbg ? a.dbg + 8 * (size_t)(blockIdx.x * gridDim.y + blockIdx.y) : nullptr);
    } else {
      const size_t mrow0 = (size_t)(a.src_of ? a.src_of[b] : b) * a.Lk;
      const uint8_t* kv = a.key_pad + mrow0;
      const float* kb = a.k + mrow0 * a.ldkv + hd;
      const float* vb = a.v + mrow0 * a.ldkv + hd;
      const int ld = a.ldkv;
      attn2_core(a.q + (srow0 + r0) * a.ldq + hd, a.ldq, nq, a.src_len ? a.src_len[b] : a.Lk,   // see k_attn: slot pool
                 [=](int key, const float*& kp, const float*& vp) { kp = kb + (size_t)key * ld; vp = vb + (size_t)key * ld; },
                 [=](int key) { return kv[key] != 0 ? A2_ALL : A2_MASKED; }, q_any,
                 a.out + (srow0 + r0) * a.d + hd, a.d, a.scale, lds, a2_qcap(RPS),
                 a.dbg ? a.dbg + 8 * (size_t)(blockIdx.x * gridDim.y + blockIdx.y) : nullptr);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GEMM + bias + residual + LayerNorm in one launch for the d-wide projections of the verify step at large row counts
// (d = 256): a workgroup owns 64 FULL rows (4 waves side by side, each 64 rows x 64 columns = the four-accumulator
// wave tile of k_gemm4, same two-tiles-in-flight pipeline), parks the raw products in LDS and then finishes its rows
// exactly the way k_finish_ln does — one wave per row, lane c owns columns 4c..4c+3, x = (resid + bias) + product,
// same shuffle reductions — so the result is bit-identical to k_gemm2 / k_gemm4 (one slab) followed by k_finish_ln,
// and the choice between the two may follow the launch's row capacity.  Saves the finish launch and the 2 x M x d x 4
// bytes of slab traffic.
struct GemmLnArgs {
  GemmArgs g;                 // X, W, K, m_ptr, M (row capacity), N = d; bias/Y/raw unused
  FinishArgs f;               // bias, resid, g1/b1 (, g2/b2), row_valid, Y, d, eps; slabs unused
};

__global__ __launch_bounds__(256) void k_gemm_ln256(GemmLnArgs args) {
  // 16-deep K tiles: the double-buffered operand images (50 KB) stay below the row image (66.5 KB), so two workgroups
  // share a CU (with 32-deep tiles it was one per CU and 46 us per launch against 24 + 14 for the separate kernels)
  constexpr int BM = 64, BN = 256, BK = 16, LDT = BK + 4, LDX = BN + 4;
  constexpr int OPER = 2 * BM * LDT + 2 * BN * LDT, IMG = BM * LDX;
  __shared__ __attribute__((aligned(16))) float smem[OPER > IMG ? OPER : IMG];
  const GemmArgs& a = args.g;
  const FinishArgs& fa = args.f;
  float (*As)[BM * LDT] = reinterpret_cast<float (*)[BM * LDT]>(smem);
  float (*Bs)[BN * LDT] = reinterpret_cast<float (*)[BN * LDT]>(smem + 2 * BM * LDT);
  const int M = a.m_ptr ? *a.m_ptr : a.M;
  const int m0 = blockIdx.x * BM;
  if (m0 >= M) return;
  const int ntiles = a.K / BK;
  const int t = threadIdx.x;
  const int lr = t >> 2, lc = (t & 3) * 4;       // 4 float4 per 16-float row, 64 rows per pass
  const int wave = t >> 6, lane = t & 63;
  const int r = lane & 31, h = lane >> 5;
  const float* xp = a.X + (size_t)min(m0 + lr, M - 1) * a.ldx + lc;
  const float* wp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wp[i] = a.W + (size_t)(lr + 64 * i) * a.ldw + lc;
  struct Frag { float4 a0, b0, b1, b2, b3; };
  auto gload = [&](int tile) {
    Frag f;
    const int ko = tile * BK;
    f.a0 = *reinterpret_cast<const float4*>(xp + ko);
    f.b0 = *reinterpret_cast<const float4*>(wp[0] + ko);
    f.b1 = *reinterpret_cast<const float4*>(wp[1] + ko);
    f.b2 = *reinterpret_cast<const float4*>(wp[2] + ko);
    f.b3 = *reinterpret_cast<const float4*>(wp[3] + ko);
    return f;
  };
  auto lstore = [&](const Frag& f, int buf) {
    float* as = As[buf] + lr * LDT + lc;
    float* bs = Bs[buf] + lr * LDT + lc;
    *reinterpret_cast<float4*>(as) = f.a0;
    *reinterpret_cast<float4*>(bs) = f.b0;
    *reinterpret_cast<float4*>(bs + 64 * LDT) = f.b1;
    *reinterpret_cast<float4*>(bs + 128 * LDT) = f.b2;
    *reinterpret_cast<float4*>(bs + 192 * LDT) = f.b3;
  };
  f32x16 c00, c01, c10, c11;
#pragma unroll
  for (int i = 0; i < 16; ++i) { c00[i] = 0.f; c01[i] = 0.f; c10[i] = 0.f; c11[i] = 0.f; }
  const int aoff = r * LDT + 4 * h, boff = (wave * 64 + r) * LDT + 4 * h;
  auto mma = [&](int buf) {
    const float* ap = As[buf] + aoff;
    const float* bp = Bs[buf] + boff;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      const float4 a0 = *reinterpret_cast<const float4*>(ap + kk);
      const float4 a1 = *reinterpret_cast<const float4*>(ap + 32 * LDT + kk);
      const float4 b0 = *reinterpret_cast<const float4*>(bp + kk);
      const float4 b1 = *reinterpret_cast<const float4*>(bp + 32 * LDT + kk);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, c11, 0, 0, 0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, c11, 0, 0, 0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b1.z, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, c11, 0, 0, 0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b1.w, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, c11, 0, 0, 0);
    }
  };
  const int last = ntiles - 1;
  Frag f0 = gload(0);
  asm volatile("" ::: "memory");
  Frag f1 = gload(min(1, last));
  for (int i = 0; i < ntiles; i += 2) {           // K is a multiple of 32: ntiles is even
    lstore(f0, 0);
    asm volatile("" ::: "memory");
    f0 = gload(min(i + 2, last));
    __syncthreads();
    mma(0);
    lstore(f1, 1);
    asm volatile("" ::: "memory");
    f1 = gload(min(i + 3, last));
    __syncthreads();
    mma(1);
  }
  __syncthreads();                                 // everybody is done with the operand buffers: reuse them as the row image
  float* T = smem;                                 // [64][LDX]
  auto park = [&](const f32x16& c, int tm, int tn) {
    const int col = wave * 64 + tn * 32 + r;
    const int row0 = tm * 32 + 4 * h;
#pragma unroll
    for (int v = 0; v < 16; ++v) T[(row0 + (v & 3) + 8 * (v >> 2)) * LDX + col] = c[v];
  };
  park(c00, 0, 0);
  park(c01, 0, 1);
  park(c10, 1, 0);
  park(c11, 1, 1);
  __syncthreads();
  // the k_finish_ln<4> arithmetic, 16 rows per wave
  const int c0 = lane * 4;
  const float4 bias4 = *reinterpret_cast<const float4*>(fa.bias + c0);
  for (int rr = wave; rr < BM; rr += 4) {
    const int row = m0 + rr;
    if (row >= M) break;
    const size_t off = (size_t)row * fa.d + c0;
    const float4 rs = *reinterpret_cast<const float4*>(fa.resid + off);
    const float4 pv = *reinterpret_cast<const float4*>(T + rr * LDX + c0);
    float x[4] = {rs.x + bias4.x, rs.y + bias4.y, rs.z + bias4.z, rs.w + bias4.w};
    x[0] += pv.x; x[1] += pv.y; x[2] += pv.z; x[3] += pv.w;
    ln_inplace<4>(x, fa.g1, fa.b1, c0, fa.d, fa.eps);
    if (fa.g2) ln_inplace<4>(x, fa.g2, fa.b2, c0, fa.d, fa.eps);
    const bool keep = fa.row_valid ? (fa.row_valid[row] != 0) : true;
    float4 y = {keep ? x[0] : 0.f, keep ? x[1] : 0.f, keep ? x[2] : 0.f, keep ? x[3] : 0.f};
    *reinterpret_cast<float4*>(fa.Y + off) = y;
  }
}

// ------------------------------------------------------------------------------------------------
// Attention v3 for the verify step: ONE WAVE per (running sequence, head, 32 step rows), no LDS, no barrier.
//
// Everything stays in the layout the fp32 MFMA produces.  Scores are computed TRANSPOSED, S^T = K Q^T (A operand =
// 32 keys of the tile, B operand = the 32 queries), so a lane (r, h) ends up with 16 scores of ONE query r (keys
// (v&3) + 8(v>>2) + 4h of the tile): the online-softmax statistics of a query live in its own two lanes (one
// exchange with lane^32), and the probabilities are already the B operand of the second product
// O^T = V^T P^T in exactly the key pairing (own register t of the h = 0 lane with own register t of the h = 1 lane)
// the MFMA contracts; its A operand V^T is 16 coalesced 128-B row reads per tile.  O^T again keeps one query per
// lane, so rescaling by exp(m_old - m_new) and the final 1/l are per-lane scalars.  Keys are visited in tiles of
// 32 in a fixed order: a row's arithmetic does not depend on the batch it sits in.
// A3Tile = what one 32-key tile needs from memory, per lane.
struct A3Tile {
  float4 k0, k1, k2, k3;     // A operand of S^T: key key0 + r, dims 8g + 4h .. +3
  float v[16];               // A operand of O^T: V[key(t, h)][r], key(t, h) = key0 + (t&3) + 8(t>>2) + 4h
  int own;                   // validity word of key key0 + r (token / source-valid byte), balloted below
};

// `n_plain` = number of leading keys that every query sees whenever they are real tokens (cached prefix and front
// token, or all encoder positions): tiles made of such keys skip the per-key flag arithmetic altogether.
// `lin_limit`, `klin`, `vlin`, `lin_ld`: keys below lin_limit sit at klin/vlin + key * lin_ld (the cache, or the encoder
// memory): a tile made of such keys takes its addresses from one base instead of sixteen per-key selections.
//
// Arithmetic (the same whichever wave computes a tile): every 32-key tile i yields a partial (m_i, l_i, O_i) with its
// own maximum; the partials are folded IN TILE ORDER into (M, L, O) by  M' = max(M, m_i),  L' = L e^(M-M') + l_i e^(m_i-M'),
// O' likewise.  SPLIT = false: one wave does all tiles of its (sequence, head) and folds as it goes.  SPLIT = true: the four
// waves of a workgroup take tiles w, w+4, ... of ONE (sequence, head), park the partials in LDS and then fold them in
// tile order (wave w finishing dims 8w + 4h .. +3): bit-identical results, four times the parallelism — used when few
// sequences are decoded (a 32-row batch), chosen by the host from the launch size.
__device__ __forceinline__ void a3_fold(float& M, float& L, float mi, float li, float& a, float& b) {
  const float Mn = fmaxf(M, mi);
  a = (M == -INFINITY) ? 0.f : __expf(M - Mn);
  b = (mi == -INFINITY) ? 0.f : __expf(mi - Mn);
  L = __fmaf_rn(L, a, __fmul_rn(li, b));
  M = Mn;
}
constexpr int A3_PART = 16 * 64 + 64;          // floats of one parked tile partial: O_i [16][64], m_i [32], l_i [32]

template <int MODE, bool SPLIT, typename KeyPtr, typename KeyOwn, typename KeyFlag, typename QFlag>
__device__ __forceinline__ void attn3_core(const float* q, int ldq, int nq, int nk, int n_plain, int lin_limit, const float* klin,
                                           const float* vlin, int lin_ld, KeyPtr keyptr, KeyOwn keyown, KeyFlag keyflag,
                                           QFlag qflag, float* out, int ldo, float scale, float* lds) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  // B operand of S^T: query r, dims 8g + 4h .. +3 (rows past nq repeat the last query; they are never stored)
  f32x4 qv[4];
  {
    const float* qp = q + (size_t)min(r, nq - 1) * ldq + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      qv[g] = *reinterpret_cast<const f32x4*>(qp + 8 * g);
      qv[g] *= scale;
    }
  }
  const int qf = qflag(min(r, nq - 1));
  float m = -INFINITY, l = 0.f;
  f32x16 o;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.f;

  // every load of a tile is unconditional (key indices clamped to nk - 1; such keys are masked): a conditional load
  // costs a branch and a full vmcnt(0) round trip each
  auto load_tile = [&](int key0) {
    A3Tile tl;
    if (key0 + 32 <= lin_limit) {                   // uniform: all 32 keys exist and are laid out linearly
      const float* kp = klin + (size_t)(key0 + r) * lin_ld + 4 * h;
      tl.k0 = *reinterpret_cast<const float4*>(kp);
      tl.k1 = *reinterpret_cast<const float4*>(kp + 8);
      tl.k2 = *reinterpret_cast<const float4*>(kp + 16);
      tl.k3 = *reinterpret_cast<const float4*>(kp + 24);
      tl.own = keyown(key0 + r);
      const float* vp = vlin + (size_t)(key0 + 4 * h) * lin_ld + r;
#pragma unroll
      for (int t = 0; t < 16; ++t) tl.v[t] = vp[(size_t)((t & 3) + 8 * (t >> 2)) * lin_ld];
      return tl;
    }
    const float *kp, *vp;
    const int kown = min(key0 + r, nk - 1);
    keyptr(kown, kp, vp);
    tl.k0 = *reinterpret_cast<const float4*>(kp + 4 * h);
    tl.k1 = *reinterpret_cast<const float4*>(kp + 4 * h + 8);
    tl.k2 = *reinterpret_cast<const float4*>(kp + 4 * h + 16);
    tl.k3 = *reinterpret_cast<const float4*>(kp + 4 * h + 24);
    tl.own = keyown(kown);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float *kq, *vq;
      keyptr(min(key0 + (t & 3) + 8 * (t >> 2) + 4 * h, nk - 1), kq, vq);
      tl.v[t] = vq[r];
    }
    return tl;
  };

  const int ntiles = (nk + 31) >> 5;
  const int wave = SPLIT ? (int)(threadIdx.x >> 6) : 0;
  constexpr int TSTEP = SPLIT ? 4 : 1;
  A3Tile cur = load_tile(min(wave, ntiles - 1) * 32);
  for (int it = wave; it < ntiles; it += TSTEP) {
    const int key0 = it * 32;
    // the next tile's loads go out before this tile's arithmetic (the empty asm keeps them above it); the copy at the
    // bottom of the loop is where they are waited for
#ifndef TTX_A3_NOPREFETCH
    A3Tile nxt = load_tile(min(it + TSTEP, ntiles - 1) * 32);
    asm volatile("" ::: "memory");
#endif
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k0.x, qv[0].x, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k0.y, qv[0].y, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k0.z, qv[0].z, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k0.w, qv[0].w, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k1.x, qv[1].x, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k1.y, qv[1].y, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k1.z, qv[1].z, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k1.w, qv[1].w, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k2.x, qv[2].x, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k2.y, qv[2].y, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k2.z, qv[2].z, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k2.w, qv[2].w, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k3.x, qv[3].x, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k3.y, qv[3].y, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k3.z, qv[3].z, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k3.w, qv[3].w, sacc, 0, 0, 0);
    // validity of the tile's 32 keys as a bit mask (lanes 0..31 hold keys key0 .. key0+31), keys past nk cleared
    unsigned valid = (unsigned)__ballot(cur.own != 0);
    if (nk - key0 < 32) valid &= (1u << (nk - key0)) - 1u;
    float mx = -INFINITY;
    if (key0 + 32 <= n_plain) {                       // uniform: a tile of plain keys (most tiles)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int j = (t & 3) + 8 * (t >> 2) + 4 * h;
        sacc[t] = ((valid >> j) & 1u) ? sacc[t] : -INFINITY;
        mx = fmaxf(mx, sacc[t]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int j = (t & 3) + 8 * (t >> 2) + 4 * h;
        const int kf = keyflag(min(key0 + j, nk - 1), (valid >> j) & 1u);     // keys past nk: valid bit 0 -> see below
        const bool vis = (key0 + j < nk) && a2_visible(qf, kf);
        sacc[t] = vis ? sacc[t] : -INFINITY;
        mx = fmaxf(mx, sacc[t]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));                               // m_i
    const float base = (mx == -INFINITY) ? 0.f : mx;                  // nothing visible in this tile: every exp below is 0
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      sacc[t] = __expf(sacc[t] - base);
      rs += sacc[t];
    }
    rs += __shfl_xor(rs, 32);                                          // l_i
    f32x16 oi;
#pragma unroll
    for (int i = 0; i < 16; ++i) oi[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) oi = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.v[t], sacc[t], oi, 0, 0, 0);
    if constexpr (!SPLIT) {
      float fa, fb;
      a3_fold(m, l, mx, rs, fa, fb);
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = __fmaf_rn(o[i], fa, __fmul_rn(oi[i], fb));
    } else {
      float* part = lds + (size_t)it * A3_PART;
#pragma unroll
      for (int i = 0; i < 16; ++i) part[i * 64 + lane] = oi[i];
      if (h == 0) { part[16 * 64 + r] = mx; part[16 * 64 + 32 + r] = rs; }
    }
#ifndef TTX_A3_NOPREFETCH
    cur = nxt;
#else
    if (it + TSTEP < ntiles) cur = load_tile((it + TSTEP) * 32);
#endif
  }
  if constexpr (SPLIT) {
    __syncthreads();
    float o4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < ntiles; ++it) {            // the same fold, in tile order; this wave owns values 4w .. 4w+3
      const float* part = lds + (size_t)it * A3_PART;
      float fa, fb;
      a3_fold(m, l, part[16 * 64 + r], part[16 * 64 + 32 + r], fa, fb);
#pragma unroll
      for (int c = 0; c < 4; ++c) o4[c] = __fmaf_rn(o4[c], fa, __fmul_rn(part[(4 * wave + c) * 64 + lane], fb));
    }
    if (r < nq) {
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      f32x4 w = {o4[0] * inv, o4[1] * inv, o4[2] * inv, o4[3] * inv};
      *reinterpret_cast<f32x4*>(out + (size_t)r * ldo + 4 * h + 8 * wave) = w;
    }
    return;
  }
  // o[v] = O[query r][dim (v&3) + 8(v>>2) + 4h]: four float4 per lane
  if (r < nq) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float* op = out + (size_t)r * ldo + 4 * h;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 w = {o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv};
      *reinterpret_cast<f32x4*>(op + 8 * c) = w;
    }
  }
}

constexpr int A3_QT = 32;            // step rows per wave
template <int MODE, bool SPLIT>
__global__ __launch_bounds__(256) void k_attn3(AttnArgs a) {
  static_assert(MODE == ATT_STEP_SELF || MODE == ATT_STEP_CROSS, "k_attn3 serves the verify step");
  extern __shared__ __attribute__((aligned(16))) float a3_lds[];      // SPLIT: one A3_PART per key tile
  const int slot = blockIdx.x;
  if (slot >= a.st->n_active) return;
  const int head = SPLIT ? (int)blockIdx.y : (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  const int hd = head * ATT_DH;
  const int D = a.D, RPS = step_rps(a.N, a.D);
  const int r0 = blockIdx.z * A3_QT;
  const int nq = min(A3_QT, RPS - r0);
  if (nq <= 0) return;
  const int b = a.act_idx[slot];
  const size_t srow0 = (size_t)slot * RPS;
  if constexpr (MODE == ATT_STEP_SELF) {
    const int f = a.front[b];
    const int* tk = a.tok + (size_t)b * a.gen_ld;
    const int pad = a.pad;
    const float* kc = a.kcache + (size_t)b * a.cache_seq_stride + hd;
    const float* vc = a.vcache + (size_t)b * a.cache_seq_stride + hd;
    const float* kb = a.k + srow0 * a.ldkv + hd;
    const float* vb = a.v + srow0 * a.ldkv + hd;
    const int ld = a.ldkv, dd = a.d;
    // keys: cached prefix [0,f) | step row 0 (position f) | the rows of every draft that has a query in this tile
    const int rlast = r0 + nq - 1;
    const int n_lo = (r0 == 0) ? 0 : (r0 - 1) / D;
    const int n_hi = (rlast == 0) ? -1 : (rlast - 1) / D;
    const int kr0 = 1 + n_lo * D;
    const int n_draft_keys = (n_hi >= n_lo && D > 0) ? (n_hi - n_lo + 1) * D : 0;
    attn3_core<MODE, SPLIT>(a.q + (srow0 + r0) * a.ldq + hd, a.ldq, nq, f + 1 + n_draft_keys, f + 1, f, kc, vc, dd,
                     [=](int key, const float*& kp, const float*& vp) {
                       const bool cached = key < f;
                       const int srow = (key == f) ? 0 : kr0 + (key - f - 1);
                       const size_t off = cached ? (size_t)key * dd : (size_t)srow * ld;
                       kp = (cached ? kc : kb) + off;
                       vp = (cached ? vc : vb) + off;
                     },
                     [=](int key) { return tk[min(key, f)] != pad ? 1 : 0; },       // prefix / front token is a real token
                     [=](int key, unsigned real) {
                       const int kr = kr0 + max(key - f - 1, 0);
                       const int kn = (kr - 1) / max(D, 1);
                       const int draft_flag = a2_flag(kn - n_lo, kr - 1 - kn * D);
                       return key <= f ? (real ? A2_ALL : A2_MASKED) : draft_flag;
                     },
                     [=](int qi) {
                       const int qr = r0 + qi;
                       if (qr == 0 || D == 0) return a2_flag(0x3fff, 0);
                       const int qn = (qr - 1) / D;
                       return a2_flag(qn - n_lo, qr - 1 - qn * D);
                     },
                     a.out + (srow0 + r0) * a.d + hd, a.d, a.scale, a3_lds);
  } else {
    const size_t mrow0 = (size_t)(a.src_of ? a.src_of[b] : b) * a.Lk;
    const uint8_t* kvalid = a.key_pad + mrow0;
    const float* kb = a.k + mrow0 * a.ldkv + hd;
    const float* vb = a.v + mrow0 * a.ldkv + hd;
    const int ld = a.ldkv;
    const int nkeys = a.src_len ? a.src_len[b] : a.Lk;
    attn3_core<MODE, SPLIT>(a.q + (srow0 + r0) * a.ldq + hd, a.ldq, nq, nkeys, ((nkeys + 31) & ~31), nkeys, kb, vb, ld,
                     [=](int key, const float*& kp, const float*& vp) { kp = kb + (size_t)key * ld; vp = vb + (size_t)key * ld; },
                     [=](int key) { return (int)kvalid[key]; },
                     [=](int, unsigned real) { return real ? A2_ALL : A2_MASKED; },
                     [](int) { return 0; },
                     a.out + (srow0 + r0) * a.d + hd, a.d, a.scale, a3_lds);
  }
}

// ------------------------------------------------------------------------------------------------
// Embedding + positional row (pos + 1); one wave per token row, float4 per lane when d == 256.
struct EmbedArgs {
  const float* table; const float* pe; float* X; int d;
  int V;                           // rows of `table`: ids outside [0, V) are looked up as id 0 (memory safety only —
                                   // the Python layer rejects such inputs like torch's embedding does)
  // full mode: tokens int32 [rows], position = row % L
  const int* tok; int rows; int L;
  // step mode
  const DecState* st; const int* act_idx; const int* front; const int* gen; int gen_ld;
  const int* drafts; int N; int D;   // drafts int32 [B, N, D]
};

template <bool STEP>
__global__ __launch_bounds__(256) void k_embed(EmbedArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int tok, pos;
  if constexpr (STEP) {
    if (row >= a.st->m_rows) return;
    const int RPS = step_rps(a.N, a.D);
    const int rs = row % RPS;                      // row inside the slot (layout: see step_rps)
    const int b = a.act_idx[row / RPS];
    const int f = a.front[b];
    if (rs == 0) {
      tok = a.gen[(size_t)b * a.gen_ld + f];
      pos = f;
    } else {
      const int n = (rs - 1) / a.D, j = (rs - 1) % a.D;          // draft n, token j (0-based) at position f + 1 + j
      tok = a.drafts[((size_t)b * a.N + n) * a.D + j];
      pos = f + 1 + j;
    }
  } else {
    if (row >= a.rows) return;
    tok = a.tok[row];
    pos = row % a.L;
  }
  if ((unsigned)tok >= (unsigned)a.V) tok = 0;
  const float* e = a.table + (size_t)tok * a.d;
  const float* p = a.pe + (size_t)(pos + 1) * a.d;
  float* x = a.X + (size_t)row * a.d;
  for (int c = lane * 4; c < a.d; c += 256) {
    const float4 ev = *reinterpret_cast<const float4*>(e + c);
    const float4 pv = *reinterpret_cast<const float4*>(p + c);
    *reinterpret_cast<float4*>(x + c) = make_float4(ev.x + pv.x, ev.y + pv.y, ev.z + pv.z, ev.w + pv.w);
  }
}

// int64 -> int32 tokens, plus the "real token" byte mask
__global__ void k_prepare_tokens(const int64_t* in, int* out, uint8_t* valid, int n, int pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int t = (int)in[i];
    out[i] = t;
    if (valid) valid[i] = (t != pad) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// argmax over the vocabulary, one wave per row; first maximum wins (torch.argmax on CPU).
__global__ __launch_bounds__(256) void k_argmax(const float* logits, int V, int* pred, const int* m_ptr, int M) {
  const int rows = m_ptr ? *m_ptr : M;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = logits + (size_t)row * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    const float v = p[c];
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  // a row of NaNs compares false everywhere: never hand an out-of-range token id to the embedding lookup
  if (lane == 0) pred[row] = (bi < V) ? bi : 0;
}

// ------------------------------------------------------------------------------------------------
// make_drafts (src/utils/drafting.py:45-67).  One block per source row; `off` skips leading tokens
// (the generators pass src[:, 1:]).  The window index is  (int)( float(i) * (float(take-1) / float(max(N-1,1))) )
// evaluated in fp32 with round-to-nearest multiplies/divides and no contraction, as torch does.
template <typename OutT>
__global__ __launch_bounds__(64) void k_make_drafts(const int* src, int src_ld, int off, int L, int N, int D,
                                                    int eos, int pad, int repl, OutT* out) {
  extern __shared__ int pre[];                   // service-token prefix sums over the padded row, [Lp + 1]
  const int b = blockIdx.x;
  const int need = N + D - 1;
  const int Lp = L > need ? L : need;
  const int W = Lp - D + 1;
  const int* s = src + (size_t)b * src_ld + off;
  __shared__ int n_clean;
  if (threadIdx.x == 0) {
    int c = 0;
    pre[0] = 0;
    for (int i = 0; i < Lp; ++i) {
      const int t = (i < L) ? s[i] : pad;
      c += (t == eos || t == pad) ? 1 : 0;
      pre[i + 1] = c;
    }
    int clean = 0;
    for (int w = 0; w < W; ++w) clean += (pre[w + D] - pre[w] == 0) ? 1 : 0;
    n_clean = clean;
  }
  __syncthreads();
  const int take = n_clean > N ? n_clean : N;
  const float ratio = __fdiv_rn((float)(take - 1), (float)(N - 1 > 1 ? N - 1 : 1));
  for (int e = threadIdx.x; e < N * D; e += blockDim.x) {
    const int i = e / D, j = e % D;
    int w = (int)__fmul_rn((float)i, ratio);
    if (w > W - 1) w = W - 1;
    const int p = w + j;
    int t = (p < L) ? s[p] : pad;
    if (t == eos || t == pad) t = repl;
    out[((size_t)b * N + i) * D + e % D] = (OutT)t;
  }
}

// ------------------------------------------------------------------------------------------------
// Greedy-speculative bookkeeping (speculative_decoding.py:93-171) on the device.
struct LoopArgs {
  DecState* st; int* act_idx; int* front; int* gen; int gen_ld;
  const int* drafts; const int* pred;
  CopyRec* rec; int64_t* out; HostInfo* host; int* haspad;
  // per-row width rule (ttx_gen_params.row_rule): every row decodes as if it were alone in its batch and its front after
  // every step is recorded, so a scheduler may regroup rows freely and still reproduce each original batch exactly
  int row_rule; short* traj; int traj_ld; int* fin_step;
  // slot pool (continuous batching, implies row_rule): a slot is re-used by a new row as soon as its row retires, so
  // a row's step count, its position in the caller's arrays and the output pointers live beside the slot state
  int pool; int* rstep; int* row_of; const struct PoolIo* io;
  int B, N, D, Ls, max_len, pad, bos, eos;
};
struct PoolIo { int64_t* out; short* traj; int* fin_step; int traj_ld; int pad_; };

__global__ void k_loop_init(LoopArgs a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = a.B * a.gen_ld;
  for (int i = tid; i < total; i += gridDim.x * blockDim.x) a.gen[i] = (i % a.gen_ld == 0) ? a.bos : a.pad;
  for (int i = tid; i < a.B * a.max_len; i += gridDim.x * blockDim.x) a.out[i] = a.pad;
  for (int i = tid; i < a.B; i += gridDim.x * blockDim.x) {
    a.act_idx[i] = i; a.front[i] = 0; a.haspad[i] = 0;
    if (a.row_rule) a.fin_step[i] = 0;
  }
  if (a.row_rule)
    for (int i = tid; i < a.B * a.traj_ld; i += gridDim.x * blockDim.x) a.traj[i] = (i % a.traj_ld == 0) ? 0 : -1;
  if (tid == 0) {
    DecState s;
    s.n_active = a.B; s.r_rows = a.B * a.N; s.m_rows = a.B * step_rps(a.N, a.D);
    s.width = 1; s.steps = 0; s.error = 0; s.n_copy = 0;
    s.stop = (1 >= a.max_len) ? 1 : 0;          // `while generated_tokens.size(1) < max_len` (:93)
    if (s.stop) { s.n_active = 0; s.r_rows = 0; s.m_rows = 0; }
    s.accepted = s.produced = s.verified_positions = s.kv_prefix_positions = s.src_positions = 0;
    *a.st = s;
    a.host->width = 1;
    a.host->steps_done = 0;
    a.host->stop = s.stop;
    __threadfence_system();
  }
}

// One block.  Verify each draft against the argmax tokens, keep the longest accepted prefix plus one
// bonus token, retire rows that produced EOS, compact the active list, decide whether the loop goes on.
constexpr int ACCEPT_THREADS = 1024;                // one slot per thread up to 1 024 slots per round; 16 waves copy finished rows
__global__ __launch_bounds__(ACCEPT_THREADS) void k_accept(LoopArgs a) {
  __shared__ int s_maxfront, s_anyfin, s_suspect, s_nn, s_maxf_new, s_nfin;
  __shared__ int s_finlist[256];                    // finished rows of this step (their output copy is shared out below)
  __shared__ long long s_acc, s_prefix;
  __shared__ int s_scan[ACCEPT_THREADS];
  DecState* st = a.st;
  const int Bc = st->n_active;
  if (Bc == 0) return;
  const int D1 = a.D + 1, RPS = step_rps(a.N, a.D);
  if (threadIdx.x == 0) { s_maxfront = 0; s_anyfin = 0; s_acc = 0; s_prefix = 0; s_suspect = 0; s_nn = 0; s_maxf_new = 0; s_nfin = 0; }
  __syncthreads();
  for (int slot = threadIdx.x; slot < Bc; slot += blockDim.x) {
    const int b = a.act_idx[slot];
    const int f = a.front[b];
    const int* ps = a.pred + (size_t)slot * RPS;       // predictions of the slot's step rows
    // prediction made at position f + j on draft n: row 0 for j = 0, else row 1 + n*D + (j-1)
    int best = 0, bacc = -1;
    for (int n = 0; n < a.N; ++n) {
      const int* dr = a.drafts + ((size_t)b * a.N + n) * a.D;
      const int* pr = ps + 1 + n * a.D - 1;            // pr[j] = prediction at position f + j for j >= 1
      // first mismatch without an early exit: the D + D loads are independent and go out back to back (the
      // early-exit loop was a chain of dependent global loads, ~30 round trips per row)
      int acc = a.D;
      for (int j = a.D - 1; j >= 0; --j)
        if (dr[j] != (j == 0 ? ps[0] : pr[j])) acc = j;
      if (acc > bacc) { bacc = acc; best = n; }
    }
    const int* pr = ps + 1 + best * a.D - 1;
    int* g = a.gen + (size_t)b * a.gen_ld;
    bool fin = false, sawpad = false;
    for (int j = 0; j <= bacc; ++j) {
      const int t = (j == 0) ? ps[0] : pr[j];
      g[f + 1 + j] = t;
      fin |= (t == a.eos);
      sawpad |= (t == a.pad);                          // a PAD inside the generated part (reference quirk 2)
    }
    if (sawpad) a.haspad[b] = 1;
    a.front[b] = f + bacc + 1;
    int flags = fin ? 1 : 0;
    if (a.row_rule) {
      int it = st->steps + 1;                          // all rows of a device batch start together ...
      short* trow = a.traj + (size_t)b * a.traj_ld;
      int* finp = a.fin_step + b;
      if (a.pool) {                                    // ... rows of a slot pool do not
        it = a.rstep[b] + 1;
        a.rstep[b] = it;
        trow = a.io->traj + (size_t)a.row_of[b] * a.traj_ld;
        finp = a.io->fin_step + a.row_of[b];
      }
      if (it < a.traj_ld) trow[it] = (short)(f + bacc + 1);
      if (fin) *finp = it;
      // alone in a batch this row would see width f + D + 2 after this step and stop once that reaches max_len (:93)
      else if (f + D1 + 1 >= a.max_len) flags = 2;
    }
    a.rec[slot] = CopyRec{b, best, bacc, f, flags};
    atomicMax(&s_maxfront, f);
    atomicAdd((unsigned long long*)&s_acc, (unsigned long long)bacc);
    atomicAdd((unsigned long long*)&s_prefix, (unsigned long long)f);
    if (fin) {
      s_anyfin = 1;
      const int k = atomicAdd(&s_nfin, 1);
      if (k < 256) s_finlist[k] = b;
    }
  }
  __syncthreads();
  const int width = s_maxfront + 1 + D1;          // columns of generated_tokens after this step (:97-102,:145)
  const int wcopy = width < a.max_len ? width : a.max_len;
  // finished rows -> output (:158); compaction of the running list by a block-wide ordered scan
  int nn_before = 0;
  for (int base = 0; base < Bc; base += blockDim.x) {
    const int slot = base + threadIdx.x;
    const int code = slot < Bc ? a.rec[slot].b : -1;
    const int keep = (slot < Bc && a.rec[slot].flags == 0) ? 1 : 0;
    s_scan[threadIdx.x] = keep;
    __syncthreads();
    for (int off = 1; off < blockDim.x; off <<= 1) {   // inclusive Hillis-Steele scan over <= 256 flags
      const int v = (threadIdx.x >= off) ? s_scan[threadIdx.x - off] : 0;
      __syncthreads();
      s_scan[threadIdx.x] += v;
      __syncthreads();
    }
    if (keep) {
      a.act_idx[nn_before + s_scan[threadIdx.x] - 1] = code;
      if (a.haspad[code]) s_suspect = 1;
      atomicMax(&s_maxf_new, a.front[code]);
    }
    nn_before += s_scan[blockDim.x - 1];
    __syncthreads();
  }
  const int wout = a.row_rule ? a.max_len : wcopy;    // columns past a row's front are PAD either way
  auto copy_row = [&](int b, int first, int stride) {
    const int* g = a.gen + (size_t)b * a.gen_ld;
    int64_t* orow = a.pool ? a.io->out + (size_t)a.row_of[b] * a.max_len : a.out + (size_t)b * a.max_len;
    for (int c = first; c < wout; c += stride) orow[c] = g[c];
  };
  if (s_nfin <= 256) {                               // one wave per finished row
    for (int k = threadIdx.x >> 6; k < s_nfin; k += (int)(blockDim.x >> 6)) copy_row(s_finlist[k], threadIdx.x & 63, 64);
  } else {                                           // more rows finished at once than the list holds: scan all slots
    for (int slot = 0; slot < Bc; ++slot)
      if (a.rec[slot].flags == 1) copy_row(a.rec[slot].b, threadIdx.x, blockDim.x);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nn = nn_before;
    st->n_copy = Bc;
    st->steps += 1;
    st->accepted += s_acc;
    st->produced += s_acc + Bc;
    st->verified_positions += (long long)Bc * RPS;
    st->kv_prefix_positions += s_prefix;
    st->src_positions += (long long)Bc * a.Ls;
    st->width = width;
    if (s_anyfin && width > a.max_len && !a.row_rule) st->error = 1;
    int stop = (nn == 0 || (!a.row_rule && width >= a.max_len)) ? 1 : 0;
    if (a.row_rule && s_suspect) st->error = 3;     // a PAD inside a sequence: per-batch quirks cannot be replayed from rows
    if (!stop && s_suspect && !a.row_rule) {
      // Reference quirk 2 (speculative_decoding.py:97,111-115): if some column up to the longest running row's
      // front is PAD in every running row, the reference under-sizes its padded tensor and the draft scatter
      // raises.  Only possible when a running row holds a PAD token, so this scan almost never runs.
      const int maxf = s_maxf_new;
      for (int c = 0; c <= maxf && !stop; ++c) {
        bool allpad = true;
        for (int i = 0; i < nn && allpad; ++i) {
          const int b = a.act_idx[i];
          if (a.front[b] >= c && a.gen[(size_t)b * a.gen_ld + c] != a.pad) allpad = false;
        }
        if (allpad) { st->error = 2; stop = 1; }
      }
    }
    st->stop = stop;
    st->n_active = stop ? 0 : nn;
    st->r_rows = stop ? 0 : nn * a.N;
    st->m_rows = stop ? 0 : nn * RPS;
    a.host->width = width;
    a.host->n_active = stop ? 0 : nn;
    a.host->stop = stop;
    __threadfence_system();
    a.host->steps_done = st->steps;                  // last: the host reads the other words once it sees this one move
    __threadfence_system();
  }
}

// ------------------------------------------------------------------------------------------------
// Slot pool (continuous batching under the per-row rule).  The pool has B slots; k_pool_init empties it, k_pool_admit
// hands free slots to R new rows (encoder output, cross K/V and drafts of those rows were just computed into
// compact staging buffers), k_pool_fill moves the staged data into the slots.  The verify step and k_accept are the
// ones of the batch path: they only ever see `act_idx` and per-slot state.
__global__ void k_pool_init(LoopArgs a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = tid; i < a.B; i += gridDim.x * blockDim.x) { a.act_idx[i] = 0; a.front[i] = 0; a.haspad[i] = 0; a.rstep[i] = 0; a.row_of[i] = 0; }
  if (tid == 0) {
    DecState s;
    s.n_active = 0; s.r_rows = 0; s.m_rows = 0; s.width = 1; s.steps = 0; s.error = 0; s.n_copy = 0; s.stop = 0;
    s.accepted = s.produced = s.verified_positions = s.kv_prefix_positions = s.src_positions = 0;
    *a.st = s;
    a.host->width = 1; a.host->steps_done = 0; a.host->stop = 0; a.host->n_active = 0;
    __threadfence_system();
  }
}

struct PoolAdmitArgs {
  DecState* st; int* act_idx; int* front; int* haspad; int* rstep; int* row_of; int* src_len; int* new_slot; HostInfo* host;
  int B, N, D, R, first_row, Ls_new;
};

// One block.  Free slots = those not in act_idx[0, n_active); the R new rows take the lowest free ones in order.
__global__ __launch_bounds__(256) void k_pool_admit(PoolAdmitArgs a) {
  extern __shared__ int s_used[];                   // [B]
  DecState* st = a.st;
  const int n = st->n_active;
  for (int i = threadIdx.x; i < a.B; i += blockDim.x) s_used[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) s_used[a.act_idx[i]] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    int got = 0;
    for (int b = 0; b < a.B && got < a.R; ++b) {
      if (s_used[b]) continue;
      a.new_slot[got] = b;
      a.act_idx[n + got] = b;
      a.front[b] = 0; a.haspad[b] = 0; a.rstep[b] = 0;
      a.row_of[b] = a.first_row + got;
      a.src_len[b] = a.Ls_new;
      ++got;
    }
    // the host only admits as many rows as it knows to be free, so got == R
    const int nn = n + got;
    st->n_active = nn; st->r_rows = nn * a.N; st->m_rows = nn * step_rps(a.N, a.D);
    st->stop = 0;
    if (got != a.R) st->error = 4;
    a.host->stop = 0;
    a.host->n_active = nn;
    __threadfence_system();
  }
}

struct PoolFillArgs {
  const int* new_slot; int R;
  int* gen; int gen_ld; int bos; int pad;
  int* drafts; const int* drafts_new; int nd;                         // N * D ints per row
  uint8_t* src_valid; const uint8_t* valid_new; int Ls_cap; int Ls_new;
  float* memkv; const float* memkv_new; int kv_row;                   // floats per source position (Ld * 2 * d)
  int64_t* out_rows; short* traj_rows; int* fin_rows; int max_len; int traj_ld; int first_row;   // caller arrays of these rows
};

// grid (R, 1 + Ls_new): block (i, 0) initialises row i's slot scalars and its caller-side rows, block (i, 1 + key)
// copies the cross K/V of one source position.
__global__ __launch_bounds__(256) void k_pool_fill(PoolFillArgs a) {
  const int i = blockIdx.x;
  const int b = a.new_slot[i];
  const int t = threadIdx.x;
  if (blockIdx.y == 0) {
    for (int c = t; c < a.gen_ld; c += blockDim.x) a.gen[(size_t)b * a.gen_ld + c] = (c == 0) ? a.bos : a.pad;
    for (int c = t; c < a.nd; c += blockDim.x) a.drafts[(size_t)b * a.nd + c] = a.drafts_new[(size_t)i * a.nd + c];
    for (int c = t; c < a.Ls_cap; c += blockDim.x)
      a.src_valid[(size_t)b * a.Ls_cap + c] = (c < a.Ls_new) ? a.valid_new[(size_t)i * a.Ls_new + c] : (uint8_t)0;
    const size_t row = (size_t)(a.first_row + i);
    for (int c = t; c < a.max_len; c += blockDim.x) a.out_rows[row * a.max_len + c] = a.pad;
    for (int c = t; c < a.traj_ld; c += blockDim.x) a.traj_rows[row * a.traj_ld + c] = (c == 0) ? 0 : -1;
    if (t == 0) a.fin_rows[row] = 0;
  } else {
    const int key = blockIdx.y - 1;
    const float4* src = reinterpret_cast<const float4*>(a.memkv_new + ((size_t)i * a.Ls_new + key) * a.kv_row);
    float4* dst = reinterpret_cast<float4*>(a.memkv + ((size_t)b * a.Ls_cap + key) * a.kv_row);
    for (int c = t; c < a.kv_row / 4; c += blockDim.x) dst[c] = src[c];
  }
}

// Strided int64 -> int32 token copy + validity bytes for a chunk of rows of a wider matrix.
__global__ void k_prepare_tokens_2d(const int64_t* in, int ld_in, int* out, uint8_t* valid, int rows, int cols, int pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * cols) {
    const int r = i / cols, c = i - r * cols;
    const int tk = (int)in[(size_t)r * ld_in + c];
    out[i] = tk;
    valid[i] = (tk != pad) ? 1 : 0;
  }
}

// Plain greedy decoding (standard_decoding.py:45-53) on the same step machinery with N = 1, D = 0: every row
// appends its argmax token each step; nothing retires; the loop ends when every row emitted EOS or PAD at
// the same step, or after max_len - 1 steps.
__global__ __launch_bounds__(256) void k_greedy_accept(LoopArgs a) {
  __shared__ int s_running;
  DecState* st = a.st;
  const int Bc = st->n_active;
  if (Bc == 0) return;
  if (threadIdx.x == 0) s_running = 0;
  __syncthreads();
  const int f = a.front[0];                        // all rows share the same front in greedy decoding
  for (int b = threadIdx.x; b < Bc; b += blockDim.x) {
    const int t = a.pred[b];
    a.gen[(size_t)b * a.gen_ld + f + 1] = t;
    a.front[b] = f + 1;
    a.rec[b] = CopyRec{b, 0, 0, f, 0};
    if (t != a.eos && t != a.pad) s_running = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    st->n_copy = Bc;
    st->steps += 1;
    st->produced += Bc;
    st->verified_positions += Bc;
    st->kv_prefix_positions += (long long)Bc * f;
    st->src_positions += (long long)Bc * a.Ls;
    st->width = f + 2;
    const int stop = (!s_running || f + 1 >= a.max_len - 1) ? 1 : 0;
    st->stop = stop;
    if (stop) { st->n_active = 0; st->r_rows = 0; st->m_rows = 0; }
    a.host->width = f + 2;
    a.host->steps_done = st->steps;
    a.host->stop = stop;
    __threadfence_system();
  }
}

__global__ void k_gen_to_out(const int* gen, int gen_ld, int64_t* out, int B, int max_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * max_len) out[i] = gen[(size_t)(i / max_len) * gen_ld + (i % max_len)];
}

// Copy the K/V rows of the accepted positions (chosen draft, j = 0..nacc) from the step's packed
// QKV buffer of every decoder layer into the KV cache at positions front_old + j.
struct KvCopyArgs {
  const DecState* st; const CopyRec* rec;
  const float* qkv; long long qkv_layer_stride;      // [Ld][Mmax][3d]
  float* kcache; float* vcache; long long cache_layer_stride; long long cache_seq_stride;
  int N, D, d;
};

__global__ __launch_bounds__(256) void k_kvcopy(KvCopyArgs a) {
  const int slot = blockIdx.x, l = blockIdx.y;
  if (slot >= a.st->n_copy) return;
  const CopyRec r = a.rec[slot];
  const int RPS = step_rps(a.N, a.D);
  const float* src = a.qkv + (size_t)l * a.qkv_layer_stride + ((size_t)slot * RPS) * 3 * a.d;   // the slot's step rows
  float* kc = a.kcache + (size_t)l * a.cache_layer_stride + (size_t)r.b * a.cache_seq_stride + (size_t)r.front_old * a.d;
  float* vc = a.vcache + (size_t)l * a.cache_layer_stride + (size_t)r.b * a.cache_seq_stride + (size_t)r.front_old * a.d;
  const int per_row = a.d / 4;                    // float4 per K (or V) row
  const int total = (r.nacc + 1) * per_row;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int j = e / per_row, c = (e % per_row) * 4;
    const int srow = (j == 0) ? 0 : 1 + r.best * a.D + (j - 1);   // position front_old + j of the chosen draft
    const float* p = src + (size_t)srow * 3 * a.d;
    *reinterpret_cast<float4*>(kc + (size_t)j * a.d + c) = *reinterpret_cast<const float4*>(p + a.d + c);
    *reinterpret_cast<float4*>(vc + (size_t)j * a.d + c) = *reinterpret_cast<const float4*>(p + 2 * a.d + c);
  }
}

// ------------------------------------------------------------------------------------------------
// Beam-speculative bookkeeping kernels.
//
// k_nucleus: mask_with_num_logits_according_nucleus (speculative_decoding.py:871-904) without the full sort: one wave
// per distribution finds the n_best largest logits in descending order by repeated wavefront arg-max (each lane
// keeps V/64 candidates in registers), accumulates their softmax mass in rank order in fp32 and keeps rank i while
// the mass ranked above it is < nucleus (rank 0 always).  Either writes the masked row (kept logits, `fill`
// elsewhere) or, fused with calculate_n_accepted_in_drafts (:847-869), only counts how many leading draft tokens
// fall inside their position's kept set.
constexpr int NUC_MAX_KEEP = 32;
constexpr int NUC_VPL = 16;                 // logits per lane held in registers: V <= 1024

struct NucleusArgs {
  const float* logits; int rows; int V;    // [rows, V]
  float nucleus; int n_best; float fill;
  float* masked;                           // [rows, V] or null
  // fused acceptance: rows are (r, j) pairs, j = 0..D (D+1 distributions per draft row); drafts [R, D]
  const int64_t* drafts; int D; int* n_ok; // n_ok [R] (null: not fused)
};

__device__ __forceinline__ void nucleus_select(const float* __restrict__ p, int V, float nucleus, int n_best, int lane,
                                               int (&kept_idx)[NUC_MAX_KEEP], float (&kept_val)[NUC_MAX_KEEP], int& n_kept) {
  float v[NUC_VPL];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NUC_VPL; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < V ? p[c] : -INFINITY;
    m = fmaxf(m, v[i]);
  }
  m = wave_max(m);
  float z = 0.f;
#pragma unroll
  for (int i = 0; i < NUC_VPL; ++i) z += (lane + 64 * i < V) ? expf(v[i] - m) : 0.f;
  z = wave_sum(z);
  float above = 0.f;                        // softmax mass of the ranks already taken
  n_kept = 0;
  for (int rank = 0; rank < n_best && rank < V; ++rank) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NUC_VPL; ++i)
      if (v[i] > best || (v[i] == best && lane + 64 * i < bi && v[i] != -INFINITY)) { best = v[i]; bi = lane + 64 * i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (bi == 0x7fffffff) break;            // nothing finite left
    const bool keep = (rank == 0) || (above < nucleus);
    if (!keep) break;                       // the mass above only grows: no later rank can be kept
    kept_idx[n_kept] = bi;
    kept_val[n_kept] = best;
    ++n_kept;
    above += expf(best - m) / z;
#pragma unroll
    for (int i = 0; i < NUC_VPL; ++i)
      if (lane + 64 * i == bi) v[i] = -INFINITY;
  }
}

// The same selection with the result spread over the lanes instead of a register array indexed at run time (which the
// compiler can only keep in scratch): lane r holds the rank-r entry.  Also hands back the softmax statistics (m, z)
// computed exactly as above.
// VPL = logits per lane held in registers (V <= 64 * VPL): the loops below run over VPL, so a small vocabulary does not pay
// for 1 024 columns; the arithmetic (and so every result) is the same for any VPL that covers V.
template <int VPL>
__device__ __forceinline__ void topk_to_lanes(const float* __restrict__ p, int V, float nucleus, int n_best, int lane,
                                              int& my_idx, float& my_val, int& n_kept, float& m_out, float& z_out) {
  float v[VPL];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < V ? p[c] : -INFINITY;
    m = fmaxf(m, v[i]);
  }
  m = wave_max(m);
  float z = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) z += (lane + 64 * i < V) ? expf(v[i] - m) : 0.f;
  z = wave_sum(z);
  m_out = m; z_out = z;
  n_kept = 0;
  my_idx = -1; my_val = 0.f;
  float above = 0.f;                        // softmax mass of the ranks already taken
  for (int rank = 0; rank < n_best && rank < V; ++rank) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if (v[i] > best || (v[i] == best && lane + 64 * i < bi && v[i] != -INFINITY)) { best = v[i]; bi = lane + 64 * i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (bi == 0x7fffffff) break;            // nothing finite left
    if (rank != 0 && !(above < nucleus)) break;   // the mass above only grows: no later rank can be kept
    if (lane == rank) { my_idx = bi; my_val = best; }
    ++n_kept;
    above += expf(best - m) / z;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if (lane + 64 * i == bi) v[i] = -INFINITY;
  }
}

__global__ __launch_bounds__(256) void k_nucleus(NucleusArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (a.n_ok) {
    // one wave per draft row: walk its D positions until the first draft token outside the kept set
    if (row >= a.rows) return;
    int ok = 0;
    for (int j = 0; j < a.D; ++j) {
      int ki[NUC_MAX_KEEP];
      float kv[NUC_MAX_KEEP];
      int nk;
      nucleus_select(a.logits + ((size_t)row * (a.D + 1) + j) * a.V, a.V, a.nucleus, a.n_best, lane, ki, kv, nk);
      const int tok = (int)a.drafts[(size_t)row * a.D + j];
      bool hit = false;
      for (int i = 0; i < nk; ++i) hit |= (ki[i] == tok);
      if (!hit) break;
      ++ok;
    }
    if (lane == 0) a.n_ok[row] = ok;
    return;
  }
  if (row >= a.rows) return;
  int ki[NUC_MAX_KEEP];
  float kv[NUC_MAX_KEEP];
  int nk;
  nucleus_select(a.logits + (size_t)row * a.V, a.V, a.nucleus, a.n_best, lane, ki, kv, nk);
  float* out = a.masked + (size_t)row * a.V;
  for (int c = lane; c < a.V; c += 64) {
    float val = a.fill;
    for (int i = 0; i < nk; ++i)
      if (ki[i] == c) val = kv[i];
    out[c] = val;
  }
}

// k_ragged_topk: topk_in_each_group (speculative_decoding.py:177-238): the k largest scores of every consecutive
// group, best first, with their flat indices.  One workgroup per group, k rounds of block-wide arg-max
// (ties: lower index first).
struct RaggedTopkArgs {
  const float* score; const int* offsets;  // offsets [G+1] (exclusive prefix sums of the group lengths)
  int k; float* top; int64_t* idx;         // [G, k]
};

__global__ __launch_bounds__(256) void k_ragged_topk(RaggedTopkArgs a) {
  extern __shared__ float vals[];          // the group's scores (taken ones become -inf)
  __shared__ float s_best[4];
  __shared__ int s_bi[4];
  const int g = blockIdx.x;
  const int lo = a.offsets[g], n = a.offsets[g + 1] - lo;
  for (int i = threadIdx.x; i < n; i += blockDim.x) vals[i] = a.score[lo + i];
  __syncthreads();
  for (int r = 0; r < a.k; ++r) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float v = vals[i];
      if (v > best || (v == best && i < bi && bi == 0x7fffffff)) { best = v; bi = i; }
    }
    // first maximum among equal values: per-thread scan ascends in i, so `bi` is already the lowest index it saw
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_best[threadIdx.x >> 6] = best; s_bi[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w)
        if (s_best[w] > best || (s_best[w] == best && s_bi[w] < bi)) { best = s_best[w]; bi = s_bi[w]; }
      if (bi == 0x7fffffff) bi = (r < n) ? r : 0;      // group exhausted (all -inf): any remaining slot, as padding
      a.top[(size_t)g * a.k + r] = best;
      a.idx[(size_t)g * a.k + r] = lo + bi;
      if (bi < n) vals[bi] = -INFINITY;
      s_bi[0] = bi;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Beam-speculative candidate expansion (SURVEY.md §2.3 K12 + K14; speculative_decoding.py:294-400 `sample`, :573-598).
//   leaf enumeration (beam_leaves_core, called by k_bs_leaves): one workgroup per candidate.  For every position p <= n_accepted of the candidate's chosen draft: the
//     n_best largest logits (nucleus >= 1 mode) minus the accepted draft token (positions < n_accepted), minus <BOS> at
//     the first rejected position, minus logits that are exactly 0 — each survivor is a leaf "keep p draft tokens, then
//     this token".  Leaf score = log-prob of the root + log-softmax of the kept tokens summed in position order (fp32,
//     sequential) + log-softmax of the leaf token.  Leaves are stored per (candidate, position) in ascending token id,
//     which is the order torch.nonzero enumerates them in.
//   k_beam_select: one workgroup per source.  The n_best best leaves of the source's candidates, best first (ties: earlier
//     in enumeration order), and the rows of the new candidates: root tokens, the kept draft tokens, the leaf token.
// Core of the leaf enumeration for one candidate `c` (the whole workgroup): `rowp(p)` = logits row of position p along
// the candidate's chosen draft, `chosen(p)` = its p-th draft token.
template <int VPL, class RowPtr, class Chosen>
__device__ __forceinline__ void beam_leaves_core(int c, int nacc, float root, int dl, int V, int K, int bos, RowPtr rowp, Chosen chosen,
                                                 float* leaf_score, int* leaf_tok, int* leaf_cnt, float* lp_kept) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = (int)(blockDim.x >> 6);
  const int dl1 = dl + 1;
  float* run = lp_kept + dl1;                 // [dl+1] sequential prefix sums
  // pass 1: per position softmax statistics, kept-token log-prob, and the surviving top-K in ascending token id
  for (int p = wave; p < dl1; p += n_waves) {
    if (p > nacc) {
      if (lane == 0) { lp_kept[p] = 0.f; leaf_cnt[(size_t)c * dl1 + p] = 0; }
      continue;
    }
    const float* row = rowp(p);
    int my_idx, nk;
    float my_val, m, z;
    topk_to_lanes<VPL>(row, V, 20.0f, K, lane, my_idx, my_val, nk, m, z);     // lane r: the rank-r logit (nucleus 20 keeps every rank)
    const int excl = (p < nacc) ? chosen(p) : ((p < dl) ? bos : -1);
    // survivors (not the excluded token, not an exact-zero logit) go out in ascending token id: a survivor's place is the
    // number of survivors with a smaller id
    const bool valid = lane < nk && my_idx != excl && my_val != 0.0f;
    const unsigned long long vm = __ballot(valid);
    int pos = 0;
    for (int j = 0; j < nk; ++j) {
      const int oj = __shfl(my_idx, j, 64);
      pos += (((vm >> j) & 1ull) && oj < my_idx) ? 1 : 0;
    }
    if (valid) {
      leaf_tok[((size_t)c * dl1 + p) * K + pos] = my_idx;
      leaf_score[((size_t)c * dl1 + p) * K + pos] = logf(expf(my_val - m) / z);     // log(softmax), as the reference writes it
    }
    if (lane == 0) {
      lp_kept[p] = (p < nacc) ? logf(expf(row[chosen(p)] - m) / z) : 0.f;
      leaf_cnt[(size_t)c * dl1 + p] = __popcll(vm);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {                      // run[p] = ((lp0 + lp1) + ...) + lp_{p-1}, summed in position order
    float acc = 0.f;
    for (int p = 0; p < dl1; ++p) { run[p] = acc; acc = (p == 0) ? lp_kept[0] : acc + lp_kept[p]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < dl1 * K; e += blockDim.x) {
    const int p = e / K, i = e % K;
    if (p <= nacc && i < leaf_cnt[(size_t)c * dl1 + p]) {
      float* ls = leaf_score + ((size_t)c * dl1 + p) * K + i;
      const float stepsum = (p == 0) ? *ls : run[p] + *ls;     // the torch path adds columns 0..p in order, then zeros
      *ls = root + stepsum;
    }
  }
}

template <typename TokT>
struct BeamSelectArgs {
  const float* leaf_score; const int* leaf_tok; const int* leaf_cnt;
  const TokT* cand; int width;                 // [n_cand, width] current rows (left-aligned, >= dl+1 PAD columns at the end)
  int ld_in, ld_out;                           // row strides of `cand` and `new_cand` (>= width)
  const int* len;                              // [n_cand] real tokens per row
  const int64_t* chosen; const int* chosen_slot;   // [n_cand, dl], [n_cand] draft slot of the chosen draft
  const uint8_t* finished;                     // [n_cand] row already holds EOS
  int B, beam, dl, K, pad, eos;
  int64_t* new_cand; float* new_logp; int* parent; int* parent_draft; int* mark;   // [B*K, width], [B*K] ...
  int* summary;                                // [4]: candidates with EOS, min PAD count, sum of marks >= 0, count of marks >= 0; [4] error
  int* new_len; uint8_t* new_finished;         // optional [B*K]: real tokens of every new row / whether it holds EOS
};

template <typename TokT>
__global__ __launch_bounds__(256) void k_beam_select(BeamSelectArgs<TokT> a) {
  extern __shared__ float sh[];                // scores [L] then codes [L] (as int)
  const int b = blockIdx.x;
  const int dl1 = a.dl + 1;
  const int L = a.beam * dl1 * a.K;            // strided capacity; entries beyond a (c,p) count hold -inf
  float* sc = sh;
  int* code = reinterpret_cast<int*>(sh + L);  // enumeration rank of each strided entry (for tie-breaking) or -1
  __shared__ int s_off[1024];                  // exclusive prefix of leaf counts over (candidate, position) of this source
  const int nseg = a.beam * dl1;
  for (int sidx = threadIdx.x; sidx < nseg; sidx += blockDim.x) s_off[sidx] = a.leaf_cnt[(size_t)b * nseg + sidx];
  __syncthreads();
  if (threadIdx.x == 0) {                      // counts -> exclusive prefix, in LDS
    int acc = 0;
    for (int sidx = 0; sidx < nseg; ++sidx) { const int n = s_off[sidx]; s_off[sidx] = acc; acc += n; }
    s_off[nseg] = acc;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < L; e += blockDim.x) {
    const int seg = e / a.K, i = e % a.K;
    const int cnt = a.leaf_cnt[(size_t)b * nseg + seg];
    sc[e] = (i < cnt) ? a.leaf_score[((size_t)b * nseg + seg) * a.K + i] : -INFINITY;
    code[e] = (i < cnt) ? s_off[seg] + i : 0x7fffffff;
  }
  __syncthreads();
  if (s_off[nseg] < a.K) { if (threadIdx.x == 0) a.summary[4] = 1; return; }   // the reference asserts len >= k
  __shared__ int s_win[NUC_MAX_KEEP];          // strided entry of the r-th best leaf
  __shared__ float s_wsc[NUC_MAX_KEEP];
  // Order: higher score first, equal scores by enumeration code (codes are unique).  Two levels, one barrier: every wave
  // takes the K best of ITS quarter of the entries in K rounds of wave-wide arg-max (no workgroup barrier inside the
  // rounds), then the <= 4K survivors are ranked against each other by counting — the K best overall are among them.
  __shared__ float s_csc[4 * NUC_MAX_KEEP];
  __shared__ int s_ccode[4 * NUC_MAX_KEEP];
  __shared__ int s_ce[4 * NUC_MAX_KEEP];
  {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per_wave = (L + 3) / 4, lo = wave * per_wave, hi = min(L, lo + per_wave);
    for (int r = 0; r < a.K; ++r) {
      float best = -INFINITY;
      int bc = 0x7fffffff, be = -1;
      for (int e = lo + lane; e < hi; e += 64) {
        const float v = sc[e];
        const int cd = code[e];
        if (cd != 0x7fffffff && (be < 0 || v > best || (v == best && cd < bc))) { best = v; bc = cd; be = e; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oc = __shfl_xor(bc, o, 64);
        const int oe = __shfl_xor(be, o, 64);
        if (oe >= 0 && (be < 0 || ov > best || (ov == best && oc < bc))) { best = ov; bc = oc; be = oe; }
      }
      if (lane == 0) {
        s_csc[wave * a.K + r] = best; s_ccode[wave * a.K + r] = bc; s_ce[wave * a.K + r] = be;
        if (be >= 0) code[be] = 0x7fffffff;      // taken (the score stays: it is read again below)
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 4 * a.K; t += blockDim.x) {
    const int e = s_ce[t];
    if (e < 0) continue;
    const float v = s_csc[t];
    const int cd = s_ccode[t];
    int rank = 0;
    for (int u = 0; u < 4 * a.K; ++u) {
      if (s_ce[u] < 0) continue;
      const float ov = s_csc[u];
      rank += (ov > v || (ov == v && s_ccode[u] < cd)) ? 1 : 0;
    }
    if (rank < a.K) { s_win[rank] = e; s_wsc[rank] = v; }
  }
  __syncthreads();
  // the K new rows, all at once: root tokens, the kept draft tokens, the leaf token
  for (int e = threadIdx.x; e < a.K * a.width; e += blockDim.x) {
    const int r = e / a.width, col = e - r * a.width;
    const int sel = s_win[r];
    const int seg = sel / a.K, i = sel % a.K;
    const int cl_local = seg / dl1, p = seg % dl1;
    const int c = b * a.beam + cl_local;
    const int tok = a.leaf_tok[((size_t)b * nseg + seg) * a.K + i];
    const int lc = a.len[c];
    int64_t t = (int64_t)a.cand[(size_t)c * a.ld_in + col];
    const int j = col - lc;
    if (j >= 0 && j <= a.dl) t = (j < p) ? a.chosen[(size_t)c * a.dl + j] : (j == p ? (int64_t)tok : (int64_t)a.pad);
    a.new_cand[(size_t)(b * a.K + r) * a.ld_out + col] = t;
  }
  for (int r = threadIdx.x; r < a.K; r += blockDim.x) {
    const int sel = s_win[r];
    const int seg = sel / a.K, i = sel % a.K;
    const int cl_local = seg / dl1, p = seg % dl1;
    const int c = b * a.beam + cl_local, out = b * a.K + r;
    const int tok = a.leaf_tok[((size_t)b * nseg + seg) * a.K + i];
    const int lc = a.len[c];
    a.new_logp[out] = s_wsc[r];
    a.parent[out] = c;
    a.parent_draft[out] = a.chosen_slot[c];
    const int fin_root = a.finished[c];
    a.mark[out] = fin_root ? -1 : p;
    const bool has_eos = fin_root || tok == a.eos;       // accepted draft tokens are never EOS (drafting.py:65)
    if (has_eos) atomicAdd(&a.summary[0], 1);
    const int real = (tok == a.pad) ? lc + p : lc + p + 1;   // PAD columns of the new row: everything after its last real token
    atomicMin(&a.summary[1], a.width - real);
    if (a.new_len) { a.new_len[out] = real; a.new_finished[out] = has_eos ? 1 : 0; }
    if (!fin_root) { atomicAdd(&a.summary[2], p); atomicAdd(&a.summary[3], 1); }
  }
}

// ------------------------------------------------------------------------------------------------
// Tree (beam) decoding with a per-candidate KV cache: SURVEY.md §2.3 K12-K14's decoder side.  A "candidate" is one
// hypothesis row (n_best per source); its cache is rebuilt every step from its parent's cache plus the parent's
// accepted step rows, then the same verify-step kernels run with candidate = running row.
// new_cache[c][0 .. len_c-2] = parent's cache [0 .. len_p-2] ++ parent's front row ++ parent's accepted draft rows.
struct TreeCacheArgs {
  const int* len; const int* parent; const int* parent_draft; const int* prev_len; const uint8_t* active;
  const float* k_old; const float* v_old; float* k_new; float* v_new;
  long long cache_layer_stride, cache_seq_stride;      // floats
  const float* qkv_prev; long long qkv_layer_stride;    // previous step's packed QKV rows, [Ld][M][3d]
  const int* prev_slot_of;                               // previous step: candidate -> slot in its compact active list (-1: inactive)
  int prev_N, prev_D, d;
};

__global__ __launch_bounds__(256) void k_tree_cache(TreeCacheArgs a) {
  const int c = blockIdx.x, l = blockIdx.y;
  if (!a.active[c]) return;
  const int p = a.parent[c];
  if (p < 0) return;                                     // fresh candidate (<BOS> only): nothing cached yet
  const int lc = a.len[c], lp = a.prev_len[p];
  const int per_row = a.d / 4;
  const float* ko = a.k_old + (size_t)l * a.cache_layer_stride + (size_t)p * a.cache_seq_stride;
  const float* vo = a.v_old + (size_t)l * a.cache_layer_stride + (size_t)p * a.cache_seq_stride;
  float* kn = a.k_new + (size_t)l * a.cache_layer_stride + (size_t)c * a.cache_seq_stride;
  float* vn = a.v_new + (size_t)l * a.cache_layer_stride + (size_t)c * a.cache_seq_stride;
  const int n_old = lp - 1;                              // positions the parent had cached
  for (int e = threadIdx.x; e < n_old * per_row; e += blockDim.x) {
    reinterpret_cast<float4*>(kn)[e] = reinterpret_cast<const float4*>(ko)[e];
    reinterpret_cast<float4*>(vn)[e] = reinterpret_cast<const float4*>(vo)[e];
  }
  const int n_new = (lc - 1) - n_old;                    // parent's front row + accepted draft rows
  const int slot = a.prev_slot_of[p];
  if (n_new <= 0 || slot < 0) return;
  const int RPS = step_rps(a.prev_N, a.prev_D);
  const float* src = a.qkv_prev + (size_t)l * a.qkv_layer_stride + ((size_t)slot * RPS) * 3 * a.d;
  const int dp = a.parent_draft[c];
  for (int e = threadIdx.x; e < n_new * per_row; e += blockDim.x) {
    const int j = e / per_row, col = (e % per_row) * 4;
    const int srow = (j == 0) ? 0 : 1 + dp * a.prev_D + (j - 1);
    const float* q = src + (size_t)srow * 3 * a.d;
    *reinterpret_cast<float4*>(kn + (size_t)(n_old + j) * a.d + col) = *reinterpret_cast<const float4*>(q + a.d + col);
    *reinterpret_cast<float4*>(vn + (size_t)(n_old + j) * a.d + col) = *reinterpret_cast<const float4*>(q + 2 * a.d + col);
  }
}

// ------------------------------------------------------------------------------------------------
// Native beam-speculative loop (ttx_beam_speculative_generate; speculative_decoding.py:428-598 all drafts, :600-845 smart
// drafts).  One iteration = k_bs_prep -> k_tree_cache -> k_bs_list -> the verify step (run_step) -> k_bs_hits ->
// k_bs_leaves -> k_beam_select<int> -> k_bs_publish.  The host knows every scalar of an iteration (candidate count, draft
// length, logical width) from what the previous one published, so they travel as kernel arguments; only the list of
// running candidates and the per-candidate choices live on the device.

// Inclusive scan of one int per thread over a 256-thread workgroup (Hillis-Steele in LDS).
__device__ __forceinline__ int block_scan_incl256(int v, int* s_scan) {
  s_scan[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int u = (threadIdx.x >= off) ? s_scan[threadIdx.x - off] : 0;
    __syncthreads();
    s_scan[threadIdx.x] += u;
    __syncthreads();
  }
  return s_scan[threadIdx.x];
}

struct BeamHost { int steps_done; int summary[5]; };     // pinned, device-mapped: written by k_bs_publish

struct BeamCounters {          // device-resident sums of one generate call
  long long model_calls, input_lines, running_rows;
  long long verified_positions;   // decoder positions the KV-cached algorithm needs: running candidates + their drafts' tokens
  long long executed_positions;   // rows of the step GEMMs (unused draft slots of smart mode included)
  long long kv_prefix_positions;  // cached prefix positions attended (sum over running candidates of len - 1)
  long long running_cands;        // sum over iterations of running candidates
  int max_group;               // smart drafts: largest number of drafts any candidate tries in the current iteration
  int pad_;
};

constexpr int BS_MAX_SLOTS = 64;     // draft slots per candidate (n_drafts) the bookkeeping kernels hold in LDS

struct BeamPrepArgs {
  const int64_t* cand_next; int ld;                 // rows the previous selection produced (or the <BOS> rows), [max_cand, ld]
  const int* len_next; const uint8_t* fin_next; const float* logp_next;
  int n_cand, beam, dl, N, pad;
  int smart, n_lib, lib_ld;                         // smart drafts: windows per source, tokens per window (first = key token)
  const int* drafts_all; int D0;                    // all drafts: [B, N, D0]
  const int* lib;                                   // smart drafts: [B, n_lib, lib_ld]
  int* gen; int* front; int* len; uint8_t* active; uint8_t* finished; float* logp; int* per_cand;
  int* drafts32;                                    // [max_cand, N, dl]: the step's draft slots
};

// One workgroup per candidate: row -> the step's loop state, and the candidate's draft slots.  All-drafts mode: the N
// drafts of its source (:484-500).  Smart mode (:690-738): the first `N` windows of the source's library whose first token
// equals the candidate's last token, in library order (window 0 if there is none); unused slots repeat the first draft.
__global__ __launch_bounds__(256) void k_bs_prep(BeamPrepArgs a) {
  __shared__ int s_scan[256];
  __shared__ int s_match[BS_MAX_SLOTS];
  const int c = blockIdx.x, t = threadIdx.x;
  if (c >= a.n_cand) {
    if (t == 0) { a.active[c] = 0; a.per_cand[c] = 0; }
    return;
  }
  const int64_t* row = a.cand_next + (size_t)c * a.ld;
  for (int col = t; col < a.ld; col += 256) a.gen[(size_t)c * a.ld + col] = (int)row[col];
  const int lc = a.len_next[c];
  const int fin = a.fin_next[c];
  if (t == 0) {
    a.len[c] = lc; a.front[c] = lc - 1; a.finished[c] = (uint8_t)fin; a.active[c] = fin ? 0 : 1; a.logp[c] = a.logp_next[c];
  }
  const int b = c / a.beam;
  int* dst = a.drafts32 + (size_t)c * a.N * a.dl;
  if (!a.smart) {
    const int* src = a.drafts_all + (size_t)b * a.N * a.D0;
    for (int e = t; e < a.N * a.dl; e += 256) dst[e] = src[(e / a.dl) * a.D0 + e % a.dl];
    if (t == 0) a.per_cand[c] = a.N;
    return;
  }
  const int last = (int)row[lc - 1];
  const int* lib = a.lib + (size_t)b * a.n_lib * a.lib_ld;
  int running = 0;
  for (int base = 0; base < a.n_lib && running < a.N; base += 256) {
    const int i = base + t;
    const int flag = (i < a.n_lib && lib[(size_t)i * a.lib_ld] == last) ? 1 : 0;
    const int incl = block_scan_incl256(flag, s_scan);
    const int pos = running + incl - 1;
    if (flag && pos < a.N) s_match[pos] = i;
    running += s_scan[255];
    __syncthreads();
  }
  int count = running < a.N ? running : a.N;
  if (count == 0) {                                   // "each line needs at least one draft" (:417)
    if (t == 0) s_match[0] = 0;
    count = 1;
  }
  __syncthreads();
  for (int e = t; e < a.N * a.dl; e += 256) {
    const int n = e / a.dl, j = e % a.dl;
    dst[e] = lib[(size_t)s_match[n < count ? n : 0] * a.lib_ld + 1 + j];
  }
  if (t == 0) a.per_cand[c] = count;
}

struct BeamListArgs {
  const uint8_t* active; const int* per_cand; const int* len;
  int n_cand, N, dl;
  int* act_idx; int* slot_of; int* prev_len; DecState* st; BeamCounters* cnt; int* summary;
};

// One workgroup: compact list of the running candidates (candidate order), the DecState the step kernels size their
// work from, the iteration's counters, and the reset of the selection summary.
__global__ __launch_bounds__(256) void k_bs_list(BeamListArgs a) {
  __shared__ int s_scan[256];
  __shared__ int s_lines, s_run, s_maxg, s_prefix;
  const int t = threadIdx.x;
  if (t == 0) { s_lines = 0; s_run = 0; s_maxg = 0; s_prefix = 0; }
  __syncthreads();
  int before = 0;
  for (int base = 0; base < a.n_cand; base += 256) {
    const int c = base + t;
    const int act = (c < a.n_cand && a.active[c]) ? 1 : 0;
    const int incl = block_scan_incl256(act, s_scan);
    if (c < a.n_cand) {
      const int pc = a.per_cand[c];
      a.slot_of[c] = act ? before + incl - 1 : -1;
      a.prev_len[c] = a.len[c];
      if (act) { a.act_idx[before + incl - 1] = c; atomicAdd(&s_run, pc); atomicAdd(&s_prefix, a.len[c] - 1); }
      atomicAdd(&s_lines, pc);
      atomicMax(&s_maxg, pc);
    }
    before += s_scan[255];
    __syncthreads();
  }
  if (t == 0) {
    DecState s;
    s.n_active = before; s.r_rows = before * a.N; s.m_rows = before * step_rps(a.N, a.dl);
    s.stop = 0; s.width = 0; s.steps = 0; s.error = 0; s.n_copy = 0;
    s.accepted = s.produced = s.verified_positions = s.kv_prefix_positions = s.src_positions = 0;
    *a.st = s;
    a.cnt->model_calls += 1;
    a.cnt->input_lines += s_lines;
    a.cnt->running_rows += s_run;
    a.cnt->verified_positions += before + (long long)s_run * a.dl;
    a.cnt->executed_positions += (long long)before * step_rps(a.N, a.dl);
    a.cnt->kv_prefix_positions += s_prefix;
    a.cnt->running_cands += before;
    a.cnt->max_group = s_maxg;
    a.summary[0] = 0; a.summary[1] = 0x7fffffff; a.summary[2] = 0; a.summary[3] = 0; a.summary[4] = 0;
  }
}

struct BeamHitsArgs {
  const float* logits; int V;                        // the step's logits, [n_active * RPS, V]
  const uint8_t* finished; const int* slot_of; const int* per_cand; const int* drafts32;
  int n_cand, N, dl, K;
  float nucleus;
  uint8_t* hit;                                      // [max_cand, N * dl]: draft token inside the kept set of its position?
};

// Acceptance test of every (draft, position) pair of every running candidate, one wave per pair, spread over the whole
// chip (grid = candidates x groups of four pairs): is the draft token among the <= K tokens inside the nucleus of its
// position (:539-548, :847-869)?  The positions are independent — only the count of LEADING hits matters, and k_bs_leaves
// takes it from these flags.  (One workgroup per candidate doing all its pairs kept 20-80 CUs busy for 40-60 us.)
constexpr int BS_HITS_WAVES = 4;
template <int VPL>
__global__ __launch_bounds__(BS_HITS_WAVES * 64) void k_bs_hits(BeamHitsArgs a) {
  const int c = blockIdx.x;
  if (c >= a.n_cand || a.finished[c]) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = blockIdx.y * BS_HITS_WAVES + wave;
  if (p >= a.per_cand[c] * a.dl) return;
  const int i = p / a.dl, j = p % a.dl;
  const int srow = (j == 0) ? 0 : 1 + i * a.dl + (j - 1);
  const float* row = a.logits + ((size_t)a.slot_of[c] * step_rps(a.N, a.dl) + srow) * a.V;
  const int tok = a.drafts32[((size_t)c * a.N + i) * a.dl + j];
  int my_idx, nk;
  float my_val, m, z;
  topk_to_lanes<VPL>(row, a.V, a.nucleus, a.K, lane, my_idx, my_val, nk, m, z);
  const bool h = __ballot(lane < nk && my_idx == tok) != 0ull;
  if (lane == 0) a.hit[(size_t)c * a.N * a.dl + p] = h ? 1 : 0;
}

struct BeamLeaves2Args {
  const float* logits; int V;
  const uint8_t* finished; const int* slot_of; const int* per_cand; const int* drafts32; const float* logp;
  const uint8_t* hit; const BeamCounters* cnt;
  int n_cand, N, dl, K, bos, pad, smart;
  int* best_n; int* best_slot; int64_t* chosen;      // [max_cand], [max_cand], [max_cand, dl]
  float* leaf_score; int* leaf_tok; int* leaf_cnt;
};

// One workgroup per candidate.  (1) Accepted length of each of its drafts = leading hits of k_bs_hits (finished candidates
// see the artificial "35 on PAD" logits, under which no draft token survives), then the best draft exactly as the
// reference's topk(1) picks it among equal counts (ttx_select.h): over the N drafts, or in smart mode over the table padded
// with -1 to the longest group.  (2) `sample` (:294-400) on the step's own logits rows along that draft.  A finished
// candidate has exactly one leaf: PAD at position 0 with log-softmax(35 on PAD, 0 elsewhere)[PAD] =
// log(1 / (1 + (V-1) e^-35)), which is 0 in fp32.
constexpr int BS_LEAVES_THREADS = 768;        // a wave per position of the chosen draft up to draft_len 11
template <int VPL>
__global__ __launch_bounds__(BS_LEAVES_THREADS) void k_bs_leaves(BeamLeaves2Args a) {
  extern __shared__ float lp_kept[];
  __shared__ int s_nok[BS_MAX_SLOTS];
  __shared__ long long s_v[BS_MAX_SLOTS];
  __shared__ int s_ix[BS_MAX_SLOTS];
  __shared__ int s_best;
  const int c = blockIdx.x;
  if (c >= a.n_cand) return;
  const int dl1 = a.dl + 1;
  const int pc = a.per_cand[c];
  const bool fin = a.finished[c] != 0;
  for (int i = threadIdx.x; i < pc; i += blockDim.x) {
    int ok = 0;
    if (!fin) {
      const uint8_t* h = a.hit + (size_t)c * a.N * a.dl + (size_t)i * a.dl;
      while (ok < a.dl && h[ok]) ++ok;
    }
    s_nok[i] = ok;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int W = a.smart ? a.cnt->max_group : a.N;
    for (int i = 0; i < W; ++i) s_v[i] = (i < pc) ? (long long)s_nok[i] : -1ll;
    const int best = ttxsel::topk1_index(s_v, s_ix, W);
    s_best = best;
    a.best_n[c] = s_nok[best];
    a.best_slot[c] = best;
  }
  __syncthreads();
  const int best = s_best;
  for (int j = threadIdx.x; j < a.dl; j += blockDim.x) a.chosen[(size_t)c * a.dl + j] = (int64_t)a.drafts32[((size_t)c * a.N + best) * a.dl + j];
  if (fin) {
    for (int p = threadIdx.x; p < dl1; p += blockDim.x) a.leaf_cnt[(size_t)c * dl1 + p] = (p == 0) ? 1 : 0;
    if (threadIdx.x == 0) {
      const float z = 1.0f + (float)(a.V - 1) * expf(-35.0f);
      a.leaf_tok[(size_t)c * dl1 * a.K] = a.pad;
      a.leaf_score[(size_t)c * dl1 * a.K] = a.logp[c] + logf(1.0f / z);
    }
    return;
  }
  const int RPS = step_rps(a.N, a.dl);
  const float* base = a.logits + (size_t)a.slot_of[c] * RPS * a.V;
  const int* dr = a.drafts32 + ((size_t)c * a.N + best) * a.dl;
  beam_leaves_core<VPL>(c, s_nok[best], a.logp[c], a.dl, a.V, a.K, a.bos,
                        [&](int p) { return base + (size_t)((p == 0) ? 0 : 1 + best * a.dl + (p - 1)) * a.V; },
                        [&](int p) { return dr[p]; },
                        a.leaf_score, a.leaf_tok, a.leaf_cnt, lp_kept);
}

// Last kernel of an iteration: the selection summary and the iteration count go to the pinned words the host polls (the
// count comes from the device-side counter so that the kernel's arguments are the same in every iteration: graph replay).
__global__ void k_bs_publish(const int* summary, BeamHost* host, const BeamCounters* cnt) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    for (int i = 0; i < 5; ++i) host->summary[i] = summary[i];
    __threadfence_system();
    host->steps_done = (int)cnt->model_calls;
    __threadfence_system();
  }
}

// First candidates of a call: one <BOS> row per source.
__global__ void k_bs_init(int64_t* cand_next, int ld, int* len_next, uint8_t* fin_next, float* logp_next, int* parent, int* parent_draft,
                          int max_cand, int B, int bos, int pad, BeamCounters* cnt) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (int i = tid; i < max_cand * ld; i += nth) cand_next[i] = (i % ld == 0 && i / ld < B) ? bos : pad;
  for (int i = tid; i < max_cand; i += nth) { len_next[i] = 1; fin_next[i] = 0; logp_next[i] = 0.f; parent[i] = -1; parent_draft[i] = 0; }
  if (tid == 0) {
    cnt->model_calls = 0; cnt->input_lines = 0; cnt->running_rows = 0; cnt->verified_positions = 0; cnt->executed_positions = 0;
    cnt->kv_prefix_positions = 0; cnt->running_cands = 0; cnt->max_group = 0; cnt->pad_ = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// Standard beam search (standard_decoding.py:131-171): one workgroup per source.  total[k][v] =
// score[k] + log(softmax(logits of candidate k))[v]  (finished candidates: the artificial "35 on PAD" row, :133-135), the
// beam best of the beam*V totals best first (ties: lower flat index), and the new rows: parent's tokens + the new token.
struct BeamStepArgs {
  const float* logits; int V;                   // the step's logits, one row per running candidate (compact order)
  const int* slot_of; const uint8_t* finished; const float* score;   // [n_cand]
  const int* gen; int ld; int width;            // current rows [n_cand, ld], `width` tokens each
  int B, beam, K, pad, eos;
  int64_t* new_cand; float* new_score; int* parent; int* new_len; uint8_t* new_finished; int* parent_draft;
  int* summary;                                 // [0] += new candidates holding EOS
};

__global__ __launch_bounds__(256) void k_beam_step(BeamStepArgs a) {
  extern __shared__ float tot[];                // [beam * V]
  __shared__ float s_best[4];
  __shared__ int s_bi[4];
  __shared__ int s_sel;
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = wave; k < a.beam; k += 4) {
    const int c = b * a.beam + k;
    const bool fin = a.finished[c] != 0;
    const float* row = fin ? nullptr : a.logits + (size_t)a.slot_of[c] * a.V;
    float m = -INFINITY;
    for (int v = lane; v < a.V; v += 64) m = fmaxf(m, fin ? (v == a.pad ? 35.0f : 0.0f) : row[v]);
    m = wave_max(m);
    float z = 0.f;
    for (int v = lane; v < a.V; v += 64) z += expf((fin ? (v == a.pad ? 35.0f : 0.0f) : row[v]) - m);
    z = wave_sum(z);
    const float sc = a.score[c];
    for (int v = lane; v < a.V; v += 64) {
      const float x = fin ? (v == a.pad ? 35.0f : 0.0f) : row[v];
      tot[k * a.V + v] = sc + logf(expf(x - m) / z);
    }
  }
  __syncthreads();
  const int n = a.beam * a.V;
  for (int r = 0; r < a.K; ++r) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float v = tot[i];
      if (v > best) { best = v; bi = i; }       // ascending i per thread: the first of equal values stays
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { s_best[wave] = best; s_bi[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      int sel = s_bi[0];
      float bv = s_best[0];
      for (int w = 1; w < 4; ++w)
        if (s_best[w] > bv || (s_best[w] == bv && s_bi[w] < sel)) { sel = s_bi[w]; bv = s_best[w]; }
      if (sel == 0x7fffffff) sel = 0;           // every total is -inf / NaN: keep the indexing in range
      s_sel = sel;
    }
    __syncthreads();
    const int sel = s_sel;
    const int k = sel / a.V, tok = sel % a.V;
    const int c = b * a.beam + k, out = b * a.K + r;
    const int* root = a.gen + (size_t)c * a.ld;
    int64_t* dst = a.new_cand + (size_t)out * a.ld;
    for (int col = threadIdx.x; col < a.ld; col += 256) dst[col] = col < a.width ? (int64_t)root[col] : (col == a.width ? (int64_t)tok : (int64_t)a.pad);
    if (threadIdx.x == 0) {
      a.new_score[out] = tot[sel];
      a.parent[out] = c;
      a.parent_draft[out] = 0;
      a.new_len[out] = a.width + 1;
      const bool has_eos = a.finished[c] || tok == a.eos;
      a.new_finished[out] = has_eos ? 1 : 0;
      if (has_eos) atomicAdd(&a.summary[0], 1);
      tot[sel] = -INFINITY;
    }
    __syncthreads();
  }
}

}  // namespace ttx
