// Attention translation unit of libttx_hip.so: k_attn3 (the verify step: one wave per (sequence,
// head, 32 step rows), registers only), k_attn2 (encoder and full-prefix decoder: K staged in LDS, MFMA), k_attn (streaming
// fallback for key counts beyond k_attn2's LDS images), and their launcher.
#include "ttx_internal.h"

#include <algorithm>

namespace ttx {

// ------------------------------------------------------------------------------------------------
// Attention.  One wave per (row, head, tile of <= MAXQ queries).  Keys come in two segments:
//   A: `nA` keys every query may see (KV cache prefix / encoder memory), individually maskable;
//   B: `nB` keys with the causal rule  key j visible to query i  <=>  j <= qpos0 + i.
// Scores live in LDS as S[key][MAXQ+1]; softmax by wavefront shuffles; P·V with lane = head dim.
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
      const size_t cb = a.cache_slot ? a.cache_slot[b] : b;      // the sequence's slot in the cache (batch pool: candidate -> slot map)
      const float* kc = a.kcache + cb * a.cache_seq_stride + hd;
      const float* vc = a.vcache + cb * a.cache_seq_stride + hd;
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
                                           QFlag qflag, float* __restrict__ out, int ldo, float scale, float* lds, int qcap) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int r = lane & 31, h = lane >> 5;
  const int nkp = a2_nkp(nk);
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
    for (int e = t; e < A2_MT * ATT_DH; e += 256) {
      const int ql = e >> 5, c = e & 31;
      const float o = Op[ql * 33 + c] + Op[A2_MT * 33 + ql * 33 + c] + Op[2 * A2_MT * 33 + ql * 33 + c] +
                      Op[3 * A2_MT * 33 + ql * 33 + c];
      if (q0 + ql < nq) out[(size_t)(q0 + ql) * ldo + c] = o * inv[ql];
    }
    if (mt + 1 < n_mt) __syncthreads();               // S, inv and Op are rewritten by the next row tile
  }
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
      const size_t cb = a.cache_slot ? a.cache_slot[b] : b;      // the sequence's slot in the cache (batch pool: candidate -> slot map)
      const float* kc = a.kcache + cb * a.cache_seq_stride + hd;
      const float* vc = a.vcache + cb * a.cache_seq_stride + hd;
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
                 a.out + (srow0 + r0) * a.d + hd, a.d, a.scale, lds, a2_qcap(RPS));
    } else {
      const size_t mrow0 = (size_t)(a.src_of ? a.src_of[b] : b) * a.Lk;
      const uint8_t* kv = a.key_pad + mrow0;
      const float* kb = a.k + mrow0 * a.ldkv + hd;
      const float* vb = a.v + mrow0 * a.ldkv + hd;
      const int ld = a.ldkv;
      attn2_core(a.q + (srow0 + r0) * a.ldq + hd, a.ldq, nq, a.src_len ? a.src_len[b] : a.Lk,   // see k_attn: slot pool
                 [=](int key, const float*& kp, const float*& vp) { kp = kb + (size_t)key * ld; vp = vb + (size_t)key * ld; },
                 [=](int key) { return kv[key] != 0 ? A2_ALL : A2_MASKED; }, q_any,
                 a.out + (srow0 + r0) * a.d + hd, a.d, a.scale, lds, a2_qcap(RPS));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Attention v3 for the verify step: ONE WAVE per (running sequence, head, 32 step rows), no barrier in the main loop.
//
// Everything stays in the layout the fp32 MFMA produces.  Scores are computed TRANSPOSED, S^T = K Q^T (A operand =
// 32 keys of the tile, B operand = the 32 queries), so a lane (r, h) ends up with 16 scores of ONE query r (keys
// (v&3) + 8(v>>2) + 4h of the tile): the online-softmax statistics of a query live in its own two lanes (one
// exchange with lane^32), and the probabilities are already the B operand of the second product
// O^T = V^T P^T in exactly the key pairing (own register t of the h = 0 lane with own register t of the h = 1 lane)
// the MFMA contracts; its A operand V^T is 16 coalesced 128-B row reads per tile.  O^T again keeps one query per
// lane, so rescaling by exp(m_old - m_new) and the final 1/l are per-lane scalars.  Keys are visited in tiles of
// 32 in a fixed order: a row's arithmetic does not depend on the batch it sits in.
//
// Arithmetic (the same whichever wave computes a tile): every 32-key tile i yields a partial (m_i, l_i, O_i) with its
// own maximum; the partials are folded IN TILE ORDER into (M, L, O) by  M' = max(M, m_i),  L' = L e^(M-M') + l_i e^(m_i-M'),
// O' likewise.  Two kernels evaluate it:
//   k_attn3s  many sequences (row groups, slot pools): one wave per unit (sequence, head, 32 step rows), folding as it goes,
//             the units of a launch STREAMED through a grid of the machine's size (see the kernel);
//   k_attn3   few sequences (a 32-row batch): the four waves of a workgroup take tiles w, w+4, ... of ONE unit, park the
//             partials in LDS and then fold them in tile order (wave w finishing dims 8w + 4h .. +3): bit-identical results,
//             four times the parallelism.  The host picks by launch size.
constexpr int A3_QT = 32;            // step rows per wave
#ifndef TTX_A3S_WGS_PER_CU
#define TTX_A3S_WGS_PER_CU 2         // grid of k_attn3s: workgroups per CU (two waves per SIMD are resident)
#endif
#ifndef TTX_A3_SPLIT_BELOW
#define TTX_A3_SPLIT_BELOW 1024      // launches with fewer units (slots x heads x 32-row tiles) share a unit's key tiles over four waves
#endif

// A3Tile = what one 32-key tile needs from memory, per lane.
struct A3Tile {
  float4 k0, k1, k2, k3;     // A operand of S^T: key key0 + r, dims 8g + 4h .. +3
  float v[16];               // A operand of O^T: V[key(t, h)][r], key(t, h) = key0 + (t&3) + 8(t>>2) + 4h
  int own;                   // validity word of key key0 + r (token / source-valid byte)
};

__device__ __forceinline__ void a3_fold(float& M, float& L, float mi, float li, float& a, float& b) {
  const float Mn = fmaxf(M, mi);
  a = (M == -INFINITY) ? 0.f : __expf(M - Mn);
  b = (mi == -INFINITY) ? 0.f : __expf(mi - Mn);
  L = __fmaf_rn(L, a, __fmul_rn(li, b));
  M = Mn;
}
constexpr int A3_PART = 16 * 64 + 64;          // floats of one parked tile partial: O_i [16][64], m_i [32], l_i [32]

// The scalars of one unit.  Keys of STEP_SELF: cached prefix [0, f) | step row 0 (position f) | the rows of every draft that has
// a query in this unit's 32 rows.  Keys below `lin_limit` sit at klin / vlin + key * lin_ld (the cache, or the encoder memory): a
// tile made of such keys takes its addresses from one base.  `n_plain` = number of leading keys that every query sees whenever
// they are real tokens (cached prefix and front token, or all encoder positions): tiles made of such keys skip the per-key
// visibility flags altogether.
struct A3Unit {
  const float* q; float* out;
  const float* klin; const float* vlin;
  const float* kb; const float* vb;            // STEP_SELF: this step's K / V rows of the sequence (packed QKV buffer)
  const int* tk;                               // STEP_SELF: the sequence's token row
  const uint8_t* kvalid;                       // STEP_CROSS: source-valid bytes
  int nq, nk, n_plain, lin_limit, ntiles;
  int f, kr0, n_lo, r0;
};

__device__ __forceinline__ int a3_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// x / D for 0 <= x < 2^32 / D without a division per lane: magic = floor(2^32 / D) + 1 (D >= 2)
__device__ __forceinline__ unsigned a3_magic(int D) { return D >= 2 ? (unsigned)(0x100000000ull / (unsigned)D) + 1u : 0u; }
__device__ __forceinline__ int a3_div(int x, int D, unsigned magic) { return D >= 2 ? (int)__umulhi((unsigned)x, magic) : x; }

template <int MODE>
__device__ __forceinline__ A3Unit a3_unit(const AttnArgs& a, int slot, int head, int qt, int RPS, unsigned magic) {
  A3Unit c;
  const int hd = head * ATT_DH;
  const int r0 = qt * A3_QT;
  const int b = a3_uniform(a.act_idx[slot]);
  const size_t srow0 = (size_t)slot * RPS;
  c.r0 = r0;
  c.nq = min(A3_QT, RPS - r0);
  c.q = a.q + (srow0 + r0) * a.ldq + hd;
  c.out = a.out + (srow0 + r0) * a.d + hd;
  if constexpr (MODE == ATT_STEP_SELF) {
    const int D = a.D;
    const int f = a3_uniform(a.front[b]);
    const int rlast = r0 + c.nq - 1;
    const int n_lo = (r0 == 0) ? 0 : a3_div(r0 - 1, D, magic);
    const int n_hi = (rlast == 0) ? -1 : a3_div(rlast - 1, D, magic);
    const int n_draft_keys = (n_hi >= n_lo && D > 0) ? (n_hi - n_lo + 1) * D : 0;
    c.f = f;
    c.n_lo = n_lo;
    c.kr0 = 1 + n_lo * D;
    c.tk = a.tok + (size_t)b * a.gen_ld;
    const size_t cb = a.cache_slot ? (size_t)a3_uniform(a.cache_slot[b]) : (size_t)b;
    c.klin = a.kcache + cb * a.cache_seq_stride + hd;
    c.vlin = a.vcache + cb * a.cache_seq_stride + hd;
    c.kb = a.k + srow0 * a.ldkv + hd;
    c.vb = a.v + srow0 * a.ldkv + hd;
    c.kvalid = nullptr;
    c.nk = f + 1 + n_draft_keys;
    c.n_plain = f + 1;
    c.lin_limit = f;
  } else {
    const size_t mrow0 = (size_t)(a.src_of ? a3_uniform(a.src_of[b]) : b) * a.Lk;
    const int nkeys = a.src_len ? a3_uniform(a.src_len[b]) : a.Lk;      // see k_attn: slot pool
    c.kvalid = a.key_pad + mrow0;
    c.klin = a.k + mrow0 * a.ldkv + hd;
    c.vlin = a.v + mrow0 * a.ldkv + hd;
    c.kb = c.klin;
    c.vb = c.vlin;
    c.tk = nullptr;
    c.f = 0; c.kr0 = 0; c.n_lo = 0;
    c.nk = nkeys;
    c.n_plain = (nkeys + 31) & ~31;
    c.lin_limit = nkeys;
  }
  c.ntiles = (c.nk + 31) >> 5;
  return c;
}

// K / V row of key `key` (0 <= key < nk) as an offset in floats from the unit's bases: 32-bit arithmetic (a sequence's cache, its
// step rows and a source's memory rows each span far less than 2^31 floats)
template <int MODE>
__device__ __forceinline__ void a3_keyrow(const AttnArgs& a, const A3Unit& c, int key, const float*& kp, const float*& vp) {
  if constexpr (MODE == ATT_STEP_SELF) {
    const bool cached = key < c.f;
    const int srow = (key == c.f) ? 0 : c.kr0 + (key - c.f - 1);
    const int off = cached ? key * a.d : srow * a.ldkv;
    kp = (cached ? c.klin : c.kb) + off;
    vp = (cached ? c.vlin : c.vb) + off;
  } else {
    const int off = key * a.ldkv;
    kp = c.klin + off;
    vp = c.vlin + off;
  }
}
template <int MODE>
__device__ __forceinline__ int a3_keyown(const AttnArgs& a, const A3Unit& c, int key) {
  if constexpr (MODE == ATT_STEP_SELF) return c.tk[min(key, c.f)] != a.pad ? 1 : 0;      // prefix / front token is a real token
  else return (int)c.kvalid[key];
}
// visibility flag of key `key` (a2_visible's convention); `real` = the key's validity word; keys past nk are masked
template <int MODE>
__device__ __forceinline__ int a3_keyflag(const AttnArgs& a, const A3Unit& c, int key, bool real, unsigned magic) {
  if (key >= c.nk) return A2_MASKED;
  if constexpr (MODE == ATT_STEP_SELF) {
    const int D = a.D;
    const int kr = c.kr0 + max(key - c.f - 1, 0);
    const int kn = a3_div(kr - 1, max(D, 1), magic);
    const int draft_flag = a2_flag(kn - c.n_lo, kr - 1 - kn * D);
    return key <= c.f ? (real ? A2_ALL : A2_MASKED) : draft_flag;
  } else {
    return real ? A2_ALL : A2_MASKED;
  }
}
template <int MODE>
__device__ __forceinline__ int a3_qflag(const AttnArgs& a, const A3Unit& c, int qi, unsigned magic) {
  if constexpr (MODE == ATT_STEP_SELF) {
    const int D = a.D;
    const int qr = c.r0 + qi;
    if (qr == 0 || D == 0) return a2_flag(0x3fff, 0);           // sees prefix + front token only
    const int qn = a3_div(qr - 1, D, magic);
    return a2_flag(qn - c.n_lo, qr - 1 - qn * D);
  } else {
    return 0;
  }
}

struct A3Query {
  float qx[16];              // B operand of S^T: query r, dims 8g + 4h .. +3, already times the scale
  int qf;                    // (rows past nq repeat the last query; they are never stored)
};

template <int MODE>
__device__ __forceinline__ A3Query a3_load_query(const AttnArgs& a, const A3Unit& c, int r, int h, unsigned magic) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  A3Query qq;
  const float* qp = c.q + (size_t)min(r, c.nq - 1) * a.ldq + 4 * h;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v = *reinterpret_cast<const f32x4*>(qp + 8 * g);
    v *= a.scale;
    qq.qx[4 * g] = v.x; qq.qx[4 * g + 1] = v.y; qq.qx[4 * g + 2] = v.z; qq.qx[4 * g + 3] = v.w;
  }
  qq.qf = a3_qflag<MODE>(a, c, min(r, c.nq - 1), magic);
  return qq;
}

// every load of a tile is unconditional (key indices clamped to nk - 1; such keys are masked): a conditional load
// costs a branch and a full vmcnt(0) round trip each
template <int MODE>
__device__ __forceinline__ A3Tile a3_load_tile(const AttnArgs& a, const A3Unit& c, int key0, int r, int h) {
  A3Tile tl;
  const int lin_ld = (MODE == ATT_STEP_SELF) ? a.d : a.ldkv;
  if (key0 + 32 <= c.lin_limit) {                   // uniform: all 32 keys exist and are laid out linearly
    const float* kp = c.klin + (key0 + r) * lin_ld + 4 * h;
    tl.k0 = *reinterpret_cast<const float4*>(kp);
    tl.k1 = *reinterpret_cast<const float4*>(kp + 8);
    tl.k2 = *reinterpret_cast<const float4*>(kp + 16);
    tl.k3 = *reinterpret_cast<const float4*>(kp + 24);
    tl.own = a3_keyown<MODE>(a, c, key0 + r);
    const float* vp = c.vlin + (key0 + 4 * h) * lin_ld + r;
#pragma unroll
    for (int t = 0; t < 16; ++t) tl.v[t] = vp[((t & 3) + 8 * (t >> 2)) * lin_ld];
    return tl;
  }
  const float *kp, *vp;
  const int kown = min(key0 + r, c.nk - 1);
  a3_keyrow<MODE>(a, c, kown, kp, vp);
  tl.k0 = *reinterpret_cast<const float4*>(kp + 4 * h);
  tl.k1 = *reinterpret_cast<const float4*>(kp + 4 * h + 8);
  tl.k2 = *reinterpret_cast<const float4*>(kp + 4 * h + 16);
  tl.k3 = *reinterpret_cast<const float4*>(kp + 4 * h + 24);
  tl.own = a3_keyown<MODE>(a, c, kown);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float *kq, *vq;
    a3_keyrow<MODE>(a, c, min(key0 + (t & 3) + 8 * (t >> 2) + 4 * h, c.nk - 1), kq, vq);
    tl.v[t] = vq[r];
  }
  return tl;
}

// One tile's partial (m_i = mx, l_i = rs, O_i = oi) from its loaded operands.
template <int MODE>
__device__ __forceinline__ void a3_tile_partial(const AttnArgs& a, const A3Unit& c, const A3Tile& cur, const A3Query& q, int key0,
                                                int r, int h, unsigned magic, float& mx, float& rs, f32x16& oi) {
  f32x16 sacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k0.x, q.qx[0], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k0.y, q.qx[1], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k0.z, q.qx[2], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k0.w, q.qx[3], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k1.x, q.qx[4], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k1.y, q.qx[5], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k1.z, q.qx[6], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k1.w, q.qx[7], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k2.x, q.qx[8], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k2.y, q.qx[9], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k2.z, q.qx[10], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k2.w, q.qx[11], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k3.x, q.qx[12], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k3.y, q.qx[13], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k3.z, q.qx[14], sacc, 0, 0, 0);
  sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.k3.w, q.qx[15], sacc, 0, 0, 0);
  mx = -INFINITY;
  if (key0 + 32 <= c.n_plain) {                       // uniform: a tile of plain keys (most tiles)
    // validity of the tile's 32 keys as a bit mask (lanes 0..31 hold keys key0 .. key0+31), keys past nk cleared
    unsigned valid = (unsigned)__ballot(cur.own != 0);
    if (c.nk - key0 < 32) valid &= (1u << (c.nk - key0)) - 1u;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int j = (t & 3) + 8 * (t >> 2) + 4 * h;
      sacc[t] = ((valid >> j) & 1u) ? sacc[t] : -INFINITY;
      mx = fmaxf(mx, sacc[t]);
    }
  } else {
    // the flag of key key0 + r is worked out ONCE, by the lane that loaded the key (one magic-number division per lane and
    // tile), and handed to the lanes that hold its scores: key (t&3) + 8(t>>2) + 4h sits in lane (t&3) + 8(t>>2) (+ 4)
    const int kflag = a3_keyflag<MODE>(a, c, key0 + r, cur.own != 0, magic);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int j0 = (t & 3) + 8 * (t >> 2);
      const int kf0 = __builtin_amdgcn_readlane(kflag, j0), kf1 = __builtin_amdgcn_readlane(kflag, j0 + 4);
      const bool vis = a2_visible(q.qf, h ? kf1 : kf0);
      sacc[t] = vis ? sacc[t] : -INFINITY;
      mx = fmaxf(mx, sacc[t]);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));                               // m_i
  const float base = (mx == -INFINITY) ? 0.f : mx;                  // nothing visible in this tile: every exp below is 0
  rs = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    sacc[t] = __expf(sacc[t] - base);
    rs += sacc[t];
  }
  rs += __shfl_xor(rs, 32);                                          // l_i
#pragma unroll
  for (int i = 0; i < 16; ++i) oi[i] = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) oi = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.v[t], sacc[t], oi, 0, 0, 0);
}

// Few sequences: one workgroup per unit, its key tiles shared out over the four waves.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_attn3(AttnArgs a) {
  static_assert(MODE == ATT_STEP_SELF || MODE == ATT_STEP_CROSS, "k_attn3 serves the verify step");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float a3_lds[];      // one A3_PART per key tile
  const int slot = blockIdx.x;
  if (slot >= a.st->n_active) return;
  const int RPS = step_rps(a.N, a.D);
  if ((int)blockIdx.z * A3_QT >= RPS) return;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const unsigned magic = a3_magic(a.D);
  const A3Unit c = a3_unit<MODE>(a, slot, blockIdx.y, blockIdx.z, RPS, magic);
  const A3Query q = a3_load_query<MODE>(a, c, r, h, magic);
  A3Tile cur = a3_load_tile<MODE>(a, c, min(wave, c.ntiles - 1) * 32, r, h);
  for (int it = wave; it < c.ntiles; it += 4) {
    // the next tile's loads go out before this tile's arithmetic (the empty asm keeps them above it); the copy at the
    // bottom of the loop is where they are waited for
    A3Tile nxt = a3_load_tile<MODE>(a, c, min(it + 4, c.ntiles - 1) * 32, r, h);
    asm volatile("" ::: "memory");
    float mx, rs;
    f32x16 oi;
    a3_tile_partial<MODE>(a, c, cur, q, it * 32, r, h, magic, mx, rs, oi);
    float* part = a3_lds + (size_t)it * A3_PART;
#pragma unroll
    for (int i = 0; i < 16; ++i) part[i * 64 + lane] = oi[i];
    if (h == 0) { part[16 * 64 + r] = mx; part[16 * 64 + 32 + r] = rs; }
    cur = nxt;
  }
  __syncthreads();
  float m = -INFINITY, l = 0.f;
  float o4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < c.ntiles; ++it) {            // the fold, in tile order; this wave owns values 4w .. 4w+3
    const float* part = a3_lds + (size_t)it * A3_PART;
    float fa, fb;
    a3_fold(m, l, part[16 * 64 + r], part[16 * 64 + 32 + r], fa, fb);
#pragma unroll
    for (int v = 0; v < 4; ++v) o4[v] = __fmaf_rn(o4[v], fa, __fmul_rn(part[(4 * wave + v) * 64 + lane], fb));
  }
  if (r < c.nq) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    f32x4 w = {o4[0] * inv, o4[1] * inv, o4[2] * inv, o4[3] * inv};
    *reinterpret_cast<f32x4*>(c.out + (size_t)r * a.d + 4 * h + 8 * wave) = w;
  }
}

// Many sequences: the units of the launch as a STREAM.  A pool launch reads ~8 KB of K/V per tile and head, and a wave that
// handled one unit and exited would spend its first tile's memory latency (and the dependent scalar loads in front of it:
// slot -> sequence -> front / source) idle, in phase with every other wave that started with it.  Here the grid has the
// machine's size (two waves per SIMD) and every wave walks units u = wave, wave + W, ...: while the last tile of a unit is in the
// matrix pipe, the first tile and the queries of the NEXT unit are already in flight, and that unit's scalars were fetched one
// unit earlier.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_attn3s(AttnArgs a, int H, int qtiles) {
  static_assert(MODE == ATT_STEP_SELF || MODE == ATT_STEP_CROSS, "k_attn3s serves the verify step");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int RPS = step_rps(a.N, a.D);
  const unsigned magic = a3_magic(a.D);
  const int W = (int)gridDim.x * 4;
  const int n_units = a.st->n_active * H * qtiles;
  int u = a3_uniform((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  if (u >= n_units) return;
  // unit u = ((slot * qtiles) + q tile) * H + head
  auto unit = [&](int uu) {
    const int head = uu % H, t = uu / H;
    return a3_unit<MODE>(a, t / qtiles, head, t % qtiles, RPS, magic);
  };
  A3Unit cu = unit(u);
  A3Query cq = a3_load_query<MODE>(a, cu, r, h, magic);
  A3Tile cur = a3_load_tile<MODE>(a, cu, 0, r, h);
  int un = u + W;
  A3Unit nu = unit(min(un, n_units - 1));               // the next unit's scalars, one unit ahead
  float m = -INFINITY, l = 0.f;
  f32x16 o;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.f;
  int it = 0;
  for (;;) {
    const bool last = it + 1 >= cu.ntiles;
    const bool more = un < n_units;
    // what the matrix pipe needs next goes out before this tile's arithmetic: the unit's next tile, or the first tile and the
    // queries of the next unit
    A3Tile nxt = cur;
    A3Query nq = cq;
    if (!last) {
      nxt = a3_load_tile<MODE>(a, cu, it * 32 + 32, r, h);
    } else if (more) {
      nxt = a3_load_tile<MODE>(a, nu, 0, r, h);
      nq = a3_load_query<MODE>(a, nu, r, h, magic);
    }
    asm volatile("" ::: "memory");
    float mx, rs, fa, fb;
    f32x16 oi;
    a3_tile_partial<MODE>(a, cu, cur, cq, it * 32, r, h, magic, mx, rs, oi);
    a3_fold(m, l, mx, rs, fa, fb);
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = __fmaf_rn(o[i], fa, __fmul_rn(oi[i], fb));
    cur = nxt;
    if (!last) {
      ++it;
      continue;
    }
    // o[v] = O[query r][dim (v&3) + 8(v>>2) + 4h]: four float4 per lane
    if (r < cu.nq) {
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      float* op = cu.out + (size_t)r * a.d + 4 * h;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        f32x4 w = {o[4 * v] * inv, o[4 * v + 1] * inv, o[4 * v + 2] * inv, o[4 * v + 3] * inv};
        *reinterpret_cast<f32x4*>(op + 8 * v) = w;
      }
    }
    if (!more) break;
    cu = nu;
    cq = nq;
    un += W;
    nu = unit(min(un, n_units - 1));
    it = 0;
    m = -INFINITY;
    l = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
static constexpr size_t kAttn2LdsLimit = 150 * 1024;

template <int MODE>
static int launch_attn_mode(ttx_session* s, hipStream_t st, const AttnArgs& a, int groups, int H, int q_per_group, int max_keys, int N, int D1) {
  if (groups <= 0 || q_per_group <= 0) return TTX_OK;
  constexpr bool step = (MODE == ATT_STEP_SELF || MODE == ATT_STEP_CROSS);
  // step modes: q_per_group = RPS = 1 + N*D rows per running sequence.  Self-attention keys of one workgroup:
  // prefix (< max_keys) + front row + the rows of every draft with a query among its 64 rows.
  const int D = D1 - 1;
  if constexpr (step) {
    // the verify step: one wave per (sequence, head, 32 step rows), registers only — no key-count limit
    if (H % 4 == 0 && !s->attn_fallback) {
      // few sequences (a 32-row batch): the key tiles of one (sequence, head) are shared out over the four waves of
      // a workgroup (k_attn3); many (row groups, slot pools): one wave per (sequence, head), streamed (k_attn3s).
      // Bit-identical either way.
      const int qtiles = cdiv(q_per_group, A3_QT);
      const int keys3 = (MODE == ATT_STEP_SELF) ? max_keys + 1 + N * std::max(D, 0) : max_keys;
      const size_t lds3 = sizeof(float) * (size_t)A3_PART * cdiv(keys3, 32);
      const bool split = s->attn_split != 0 && (s->attn_split > 0 || (long long)groups * H * qtiles < TTX_A3_SPLIT_BELOW) && lds3 <= 64 * 1024;
      if (split) {
        hipLaunchKernelGGL((k_attn3<MODE>), dim3(groups, H, qtiles), dim3(256), lds3, st, a);
      } else {
        // a grid of the machine's size (two workgroups = eight waves per CU), every wave walking its share of the units
        const long long units = (long long)groups * H * qtiles;
        const int wgs = (int)std::min<long long>((units + 3) / 4, (long long)TTX_A3S_WGS_PER_CU * s->m->n_cu);
        hipLaunchKernelGGL((k_attn3s<MODE>), dim3(wgs), dim3(256), 0, st, a, H, qtiles);
      }
      HIP_TRY(hipGetLastError());
      return TTX_OK;
    }
  }
  const int draft_keys = (D > 0) ? (std::min(N, (A2_QT + D - 2) / D + 1)) * D : 0;
  const int keys2 = (MODE == ATT_STEP_SELF) ? max_keys + 1 + draft_keys : max_keys;
  const size_t lds2 = attn2_lds_bytes(keys2, a2_qcap(q_per_group));
  if (lds2 <= kAttn2LdsLimit && attn2_fits(keys2) && !s->attn_fallback) {
    const int tiles = cdiv(q_per_group, A2_QT);
    if (lds2 > 64 * 1024 && !s->attr_attn2[MODE]) {             // per device: kept per session, set outside graph capture
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn2<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)kAttn2LdsLimit));
      s->attr_attn2[MODE] = true;
    }
    hipLaunchKernelGGL((k_attn2<MODE>), dim3(groups, H, tiles), dim3(256), lds2, st, a);
    HIP_TRY(hipGetLastError());
    return TTX_OK;
  }
  const int keys1 = (MODE == ATT_STEP_SELF) ? max_keys + q_per_group : max_keys;
  const size_t lds = attn_lds_bytes(keys1);
  if (lds > kAttn2LdsLimit) return fail(TTX_ERR_INVALID, "sequence too long for the attention kernels' LDS score buffer");
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_attn<MODE>), dim3(groups, H, cdiv(q_per_group, ATT_MAXQ)), dim3(64), lds, st, a);
  HIP_TRY(hipGetLastError());
  return TTX_OK;
}

int launch_attn(int mode, ttx_session* s, hipStream_t st, const AttnArgs& a, int groups, int H, int q_per_group, int max_keys,
                int N, int D1) {
  switch (mode) {
    case ATT_ENC: return launch_attn_mode<ATT_ENC>(s, st, a, groups, H, q_per_group, max_keys, N, D1);
    case ATT_FULL_SELF: return launch_attn_mode<ATT_FULL_SELF>(s, st, a, groups, H, q_per_group, max_keys, N, D1);
    case ATT_FULL_CROSS: return launch_attn_mode<ATT_FULL_CROSS>(s, st, a, groups, H, q_per_group, max_keys, N, D1);
    case ATT_STEP_SELF: return launch_attn_mode<ATT_STEP_SELF>(s, st, a, groups, H, q_per_group, max_keys, N, D1);
    case ATT_STEP_CROSS: return launch_attn_mode<ATT_STEP_CROSS>(s, st, a, groups, H, q_per_group, max_keys, N, D1);
  }
  return fail(TTX_ERR_INVALID, "unknown attention mode");
}

}  // namespace ttx
