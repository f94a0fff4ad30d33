// Types shared by the translation units of libttx_hip.so (GEMM, attention, loop kernels + C ABI): device-resident loop
// state, kernel argument blocks, wave reductions.  Hand-written HIP for gfx950 (CDNA4, wave64), fp32 throughout: the
// reference runs with `precision: null` (configs/cfg_standard_product_prediction.yaml:7) and the parity bar is token
// identity, so every dense contraction uses the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ttx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
  int slice_k;               // canonical slice length (see ttx_gemm.hip): the result is the ordered sum over the K range's
                             // slices of `slice_k` k's, each accumulated from zero; 0 = one chain over the whole range
  int big_min_tiles;         // k_gemm24: smallest 128x64-tile count (x slabs) at which that tiling is used instead of 64x64 (0: never)
};


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


// y = LN2?( LN( (resid + bias) + (slab[0] + slab[1] + ... in slab order) ) ) — one wave per row, VPL contiguous values per lane.
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

// ------------------------------------------------------------------------------------------------
constexpr int ATT_DH = 32;                  // head dimension every attention kernel is written for
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
  const int* cache_slot;           // STEP_SELF: running row -> its sequence's slot in the cache (null: the row index itself)
  int gen_ld; int N; int D;
};

}  // namespace ttx
