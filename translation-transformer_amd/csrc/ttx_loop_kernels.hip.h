// Loop and bookkeeping kernels of libttx_hip.so (included by ttx_api.hip only): embedding, argmax, draft making, the
// greedy-speculative accept / retire / KV-commit kernels and their slot pool, the nucleus / top-k helpers and the native
// beam-search and beam-speculative iterations.  Hand-written HIP for gfx950, wave64.
#pragma once
#include "ttx_common.hip.h"
#include "ttx_select.h"

namespace ttx {

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
    // ordered compaction: position = kept slots of the earlier waves + kept lanes below this one (ballots, two barriers)
    const unsigned long long mask = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_scan[wave] = __popcll(mask);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
      const int c = s_scan[w];
      before += (w < wave) ? c : 0;
      total += c;
    }
    if (keep) {
      a.act_idx[nn_before + before + __popcll(mask & ((1ull << lane) - 1ull))] = code;
      if (a.haspad[code]) s_suspect = 1;
      atomicMax(&s_maxf_new, a.front[code]);
    }
    nn_before += total;
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
  }                                                  // (the kernel's end releases it; the graph's next node is far away)
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
__device__ __forceinline__ void beam_leaves_core(int c, int nacc, float root, int dl, int dl_logic, int V, int K, int bos, RowPtr rowp, Chosen chosen,
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
    const int excl = (p < nacc) ? chosen(p) : ((p < dl_logic) ? bos : -1);     // dl_logic: the draft length in force (<= dl, the layout's)
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

// The K best leaves among the `nseg` (candidate, position) segments starting at segment `seg0` (each segment holds up to K
// leaves in leaf_score / leaf_tok, `leaf_cnt[seg]` of them real), best first: s_win[r] = index of the r-th best as
// local_segment * K + i, s_wsc[r] its score.  Order: higher score first, equal scores by enumeration order.  Returns false
// when there are fewer than K leaves (the reference asserts there, speculative_decoding.py:195).  Whole 256-thread workgroup;
// `sh` = 2 * nseg * K floats of dynamic LDS.
__device__ __forceinline__ bool select_best_leaves(const float* leaf_score, const int* leaf_cnt, size_t seg0, int nseg, int K, float* sh,
                                                   int* s_win, float* s_wsc) {
  const int L = nseg * K;                      // strided capacity; entries beyond a segment's count hold -inf
  float* sc = sh;
  int* code = reinterpret_cast<int*>(sh + L);  // enumeration rank of each strided entry (for tie-breaking) or taken / absent
  __shared__ int s_off[1024];                  // exclusive prefix of leaf counts over the segments
  for (int sidx = threadIdx.x; sidx < nseg; sidx += blockDim.x) s_off[sidx] = leaf_cnt[seg0 + sidx];
  __syncthreads();
  if (threadIdx.x == 0) {                      // counts -> exclusive prefix, in LDS
    int acc = 0;
    for (int sidx = 0; sidx < nseg; ++sidx) { const int n = s_off[sidx]; s_off[sidx] = acc; acc += n; }
    s_off[nseg] = acc;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < L; e += blockDim.x) {
    const int seg = e / K, i = e % K;
    const int cnt = leaf_cnt[seg0 + seg];
    sc[e] = (i < cnt) ? leaf_score[(seg0 + seg) * K + i] : -INFINITY;
    code[e] = (i < cnt) ? s_off[seg] + i : 0x7fffffff;
  }
  __syncthreads();
  if (s_off[nseg] < K) return false;
  // Two levels, one barrier: every wave takes the K best of ITS quarter of the entries in K rounds of wave-wide arg-max (no
  // workgroup barrier inside the rounds), then the <= 4K survivors are ranked against each other by counting — the K best
  // overall are among them.
  __shared__ float s_csc[4 * NUC_MAX_KEEP];
  __shared__ int s_ccode[4 * NUC_MAX_KEEP];
  __shared__ int s_ce[4 * NUC_MAX_KEEP];
  {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per_wave = (L + 3) / 4, lo = wave * per_wave, hi = min(L, lo + per_wave);
    for (int r = 0; r < K; ++r) {
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
        s_csc[wave * K + r] = best; s_ccode[wave * K + r] = bc; s_ce[wave * K + r] = be;
        if (be >= 0) code[be] = 0x7fffffff;      // taken (the score stays: it is read again below)
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 4 * K; t += blockDim.x) {
    const int e = s_ce[t];
    if (e < 0) continue;
    const float v = s_csc[t];
    const int cd = s_ccode[t];
    int rank = 0;
    for (int u = 0; u < 4 * K; ++u) {
      if (s_ce[u] < 0) continue;
      const float ov = s_csc[u];
      rank += (ov > v || (ov == v && s_ccode[u] < cd)) ? 1 : 0;
    }
    if (rank < K) { s_win[rank] = e; s_wsc[rank] = v; }
  }
  __syncthreads();
  return true;
}

template <typename TokT>
__global__ __launch_bounds__(256) void k_beam_select(BeamSelectArgs<TokT> a) {
  extern __shared__ float sh[];                // scores [L] then codes [L] (as int)
  const int b = blockIdx.x;
  const int dl1 = a.dl + 1;
  const int nseg = a.beam * dl1;
  __shared__ int s_win[NUC_MAX_KEEP];          // strided entry of the r-th best leaf
  __shared__ float s_wsc[NUC_MAX_KEEP];
  if (!select_best_leaves(a.leaf_score, a.leaf_cnt, (size_t)b * nseg, nseg, a.K, sh, s_win, s_wsc)) {   // the reference asserts len >= k
    if (threadIdx.x == 0) a.summary[4] = 1;
    return;
  }
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
  // batch pool: ONE cache buffer (k_old == k_new) and a candidate -> slot map.  A child that took over its parent's slot only
  // appends; the others copy the parent's rows into a slot no candidate of this iteration descends from (k_bsp_select's choice).
  const int* slot_parent; const int* slot_self;          // null: cache row = candidate index, two buffers
};

__global__ __launch_bounds__(256) void k_tree_cache(TreeCacheArgs a) {
  const int c = blockIdx.x, l = blockIdx.y;
  if (!a.active[c]) return;
  const int p = a.parent[c];
  if (p < 0) return;                                     // fresh candidate (<BOS> only): nothing cached yet
  const int lc = a.len[c], lp = a.prev_len[p];
  const int per_row = a.d / 4;
  const int sp = a.slot_parent ? a.slot_parent[c] : p, sc = a.slot_self ? a.slot_self[c] : c;
  const float* ko = a.k_old + (size_t)l * a.cache_layer_stride + (size_t)sp * a.cache_seq_stride;
  const float* vo = a.v_old + (size_t)l * a.cache_layer_stride + (size_t)sp * a.cache_seq_stride;
  float* kn = a.k_new + (size_t)l * a.cache_layer_stride + (size_t)sc * a.cache_seq_stride;
  float* vn = a.v_new + (size_t)l * a.cache_layer_stride + (size_t)sc * a.cache_seq_stride;
  const int n_old = lp - 1;                              // positions the parent had cached
  if (!(a.slot_self && sp == sc)) {                      // (in its parent's slot the rows are already there)
    for (int e = threadIdx.x; e < n_old * per_row; e += blockDim.x) {
      reinterpret_cast<float4*>(kn)[e] = reinterpret_cast<const float4*>(ko)[e];
      reinterpret_cast<float4*>(vn)[e] = reinterpret_cast<const float4*>(vo)[e];
    }
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
  __shared__ int s_cnt[4];
  __shared__ int s_lines, s_run, s_maxg, s_prefix;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) { s_lines = 0; s_run = 0; s_maxg = 0; s_prefix = 0; }
  __syncthreads();
  auto wsum = [](int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };
  auto wmax = [](int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64)); return v; };
  int before = 0;
  for (int base = 0; base < a.n_cand; base += 256) {
    const int c = base + t;
    const bool in = c < a.n_cand;
    const int act = (in && a.active[c]) ? 1 : 0;
    const int pc = in ? a.per_cand[c] : 0, ln = in ? a.len[c] : 1;
    // ordered compaction by ballots: running candidates of the earlier waves + running lanes below this one
    const unsigned long long mask = __ballot(act);
    if (lane == 0) s_cnt[wave] = __popcll(mask);
    __syncthreads();
    int below = 0, total = 0;
    for (int w = 0; w < 4; ++w) { below += (w < wave) ? s_cnt[w] : 0; total += s_cnt[w]; }
    const int pos = before + below + __popcll(mask & ((1ull << lane) - 1ull));
    if (in) {
      a.slot_of[c] = act ? pos : -1;
      a.prev_len[c] = ln;
      if (act) a.act_idx[pos] = c;
    }
    const int lines = wsum(pc), run = wsum(act ? pc : 0), prefix = wsum(act ? ln - 1 : 0), maxg = wmax(pc);
    if (lane == 0) { atomicAdd(&s_lines, lines); atomicAdd(&s_run, run); atomicAdd(&s_prefix, prefix); atomicMax(&s_maxg, maxg); }
    before += total;
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
  const int* dl_of;                                  // pool: [max_cand] draft length of the candidate's batch (<= dl, the row layout's); null: dl
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
  if (a.dl_of && j >= a.dl_of[c]) return;            // beyond the batch's (shrunk) draft: k_bs_leaves never looks there
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
  // source pool (ttx_beam_speculative_generate_pool); all null on the per-batch path
  const uint8_t* live;     // [max_cand] 0: no candidate in this slot (its segments get no leaves)
  const int* dl_of;        // [max_cand] draft length in force for the candidate's batch (<= dl, the row layout's)
  const int* grp_of;       // smart mode: [batch slots] longest draft group of each batch in this iteration (the width of the
  const int* cand_batch;   //             -1-padded table the best draft is picked from, :779-784) and [max_cand] -> batch slot
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
  if (a.live && !a.live[c]) {
    for (int p = threadIdx.x; p < dl1; p += blockDim.x) a.leaf_cnt[(size_t)c * dl1 + p] = 0;
    return;
  }
  const int pc = a.per_cand[c];
  const bool fin = a.finished[c] != 0;
  const int dl_c = a.dl_of ? a.dl_of[c] : a.dl;
  for (int i = threadIdx.x; i < pc; i += blockDim.x) {
    int ok = 0;
    if (!fin) {
      const uint8_t* h = a.hit + (size_t)c * a.N * a.dl + (size_t)i * a.dl;
      while (ok < dl_c && h[ok]) ++ok;
    }
    s_nok[i] = ok;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int W = !a.smart ? a.N : (a.grp_of ? a.grp_of[a.cand_batch[c]] : a.cnt->max_group);
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
  beam_leaves_core<VPL>(c, s_nok[best], a.logp[c], a.dl, dl_c, a.V, a.K, a.bos,
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

// ------------------------------------------------------------------------------------------------
// Calibration of bench.py's per-launch HIP event pairs: a kernel of KNOWN duration (it waits `ticks` of the 100 MHz realtime
// counter and reports what it saw elapse) between two events — pair time minus in-kernel time is what the bracketing costs.
__global__ void k_spin(unsigned long long* out, int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t1 = t0;
  while ((long long)(t1 - t0) < (long long)ticks) t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------
// Beam-speculative BATCH POOL (ttx_beam_speculative_generate_pool): continuous batching over many given batches.
// In the reference's loop (speculative_decoding.py:428-598, :600-845) the sources of a batch meet only in batch-wide scalars —
// the draft length min(max_len - longest row - 1, draft_len) (:476 / :671), the stop rule (every row holds EOS, :586; or no
// room left, :464), the tensor width, and in smart mode the width of the -1-padded table the best draft is picked from (the
// batch's longest draft group, :779-784 -> :225) — while candidates, leaves, scores and the per-source selection depend on the
// source alone, and a source all of whose n_best rows hold EOS is a fixed point of the iteration (one PAD leaf of log-prob + 0
// per row, same order).  The pool therefore owns C source slots (K = n_best candidate slots each) and NB batch slots: a given
// batch is admitted as a whole (its sources start together and stay in lock-step), ONE verify step per iteration serves every
// live candidate of every batch in the pool (M ~ 10^4 step rows instead of ~1.5 k), and the batch-wide scalars are kept per
// batch slot on the device, exactly as the reference computes them:
//   bat_dl      the draft length in force (non-increasing; the step's row layout keeps draft_len slots per draft, a batch
//               whose draft shrank simply stops looking beyond its own length)
//   bat_grp     smart mode: the longest draft group of the batch's candidates in this iteration (finished sources count 1)
//   stop        every source finished -> its slots are free; no room left -> its running sources are retired as they stand
// A source whose K rows all hold EOS retires at once (its slots are reused by the next batch) and only its longest row stays
// behind in the batch slot.  Per source the longest new row of every iteration is recorded, from which the host derives the
// result width, the model calls and the counters of each given batch (scheduling.replay_beam_batch).
//   iteration = k_bsp_prep -> k_tree_cache -> k_bs_list -> verify step -> k_bs_hits -> k_bs_leaves (pool arguments) ->
//               k_bsp_select -> k_bsp_batches -> k_bsp_retire -> k_bsp_publish      (admission: k_bsp_admit -> k_bsp_fill)
enum BeamPoolStatus { BP_RUNNING = 0, BP_DONE = 1, BP_ERR_LEAVES = 2, BP_STOPPED = 3, BP_MAX_STEPS = 4 };

struct BeamPoolHost { int steps_done; int n_live; int n_running; int error; };     // pinned, device-mapped: written by k_bsp_publish

// Caller-side arrays of one pool call (device pointers; constant for the call), in work-list order.
struct BeamPoolIo {
  int64_t* out;              // [R_total][K][max_len]: the hypotheses of every retired source, best first, PAD beyond
  short* trace_len;          // [R_total][T_cap]: longest hypothesis (tokens) of the source after each of its iterations
  int* summary;              // [R_total][8]: iterations, BeamPoolStatus, input lines, running rows, accepted sum, accepted count,
                             //               longest hypothesis at the end, decoded candidates summed over the iterations
  const int* len_all;        // [R_total] source length (position after the last non-PAD token)
  const int* batch_all;      // [R_total] index of the source's batch in the work list
  const int* given_all;      // [n_batches] padded width of each given batch (smart mode: library size)
  int T_cap;
  int pad_;
};

struct BeamPoolArgs {
  int C, K, N, D0, Ls_cap, max_len, ld;          // ld: row stride of the candidate rows (max_len + D0 + 2)
  int smart, lib_ld, pad, bos, eos, repl, max_steps;
  // source slots [C]
  int* row_of; int* slot_batch; int* src_acc; int* src_state;    // src_acc [C][8]; src_state: BeamPoolStatus decided this iteration
  const int* tok;                                // [C][Ls_cap] source tokens per slot (PAD beyond the source)
  const int* drafts_all;                         // all drafts: [C][N][D0]
  // batch slots [C] (a batch has at least one source)
  int* bat_id; int* bat_iter; int* bat_dl; int* bat_grp; int* bat_live; int* bat_nfin; int* bat_longest_fin; int* bat_longest_cur;
  int* bat_state; int* bat_given;
  // candidates ([C*K])
  int64_t* cand_next; int* len_next; uint8_t* fin_next; float* logp_next; int* parent; int* parent_draft;
  int* gen; int* front; int* len; uint8_t* active; uint8_t* finished; uint8_t* live; float* logp; int* per_cand; int* drafts32;
  int* cand_dl; int* cand_batch;
  int* cache_slot; int* cache_slot_parent;        // candidate -> slot of its KV cache; slot its parent's rows sit in (k_tree_cache)
  const int* chosen_slot; const int64_t* chosen;
  const float* leaf_score; const int* leaf_tok; const int* leaf_cnt;
  BeamCounters* cnt; const BeamPoolIo* io; BeamPoolHost* host; int* dev_summary;   // dev_summary [4]: live sources, running candidates, error, -
};

__global__ void k_bsp_init(BeamPoolArgs a, int* src_of, int* cand_src_len) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  const int MC = a.C * a.K;
  for (int i = tid; i < a.C; i += nth) {
    a.row_of[i] = -1; a.slot_batch[i] = 0; a.src_state[i] = BP_RUNNING;
    a.bat_id[i] = -1; a.bat_iter[i] = 0; a.bat_dl[i] = a.D0; a.bat_grp[i] = 0; a.bat_live[i] = 0; a.bat_nfin[i] = 0;
    a.bat_longest_fin[i] = 0; a.bat_longest_cur[i] = 0; a.bat_state[i] = 0; a.bat_given[i] = 0;
  }
  for (int i = tid; i < a.C * 8; i += nth) a.src_acc[i] = 0;
  for (int i = tid; i < MC; i += nth) {
    a.len_next[i] = 1; a.fin_next[i] = 1; a.logp_next[i] = 0.f; a.parent[i] = -1; a.parent_draft[i] = 0;
    a.active[i] = 0; a.finished[i] = 1; a.live[i] = 0; a.per_cand[i] = 0; a.cand_dl[i] = a.D0; a.cand_batch[i] = 0;
    a.cache_slot[i] = i; a.cache_slot_parent[i] = i;
    src_of[i] = i / a.K; cand_src_len[i] = 1;
  }
  if (tid == 0) {
    a.cnt->model_calls = 0; a.cnt->input_lines = 0; a.cnt->running_rows = 0; a.cnt->verified_positions = 0; a.cnt->executed_positions = 0;
    a.cnt->kv_prefix_positions = 0; a.cnt->running_cands = 0; a.cnt->max_group = 0; a.cnt->pad_ = 0;
    a.dev_summary[0] = 0; a.dev_summary[1] = 0; a.dev_summary[2] = 0; a.dev_summary[3] = 0;
    a.host->steps_done = 0; a.host->n_live = 0; a.host->n_running = 0; a.host->error = 0;
    __threadfence_system();
  }
}

// One thread.  The R new sources (rows first_row .. first_row + R - 1 of the work list: whole batches, in order) take the
// lowest free source slots; every new batch takes the lowest free batch slot.
__global__ void k_bsp_admit(BeamPoolArgs a, int* new_slot, int* cand_src_len, int R, int first_row) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const BeamPoolIo io = *a.io;
  int got = 0, s = 0, b = 0, cur_batch = -1, cur_slot = -1;
  for (; got < R; ++got) {
    const int row = first_row + got;
    while (s < a.C && a.row_of[s] >= 0) ++s;
    if (s >= a.C) break;
    const int batch = io.batch_all[row];
    if (batch != cur_batch) {
      while (b < a.C && a.bat_id[b] >= 0) ++b;
      if (b >= a.C) break;
      cur_batch = batch; cur_slot = b;
      a.bat_id[b] = batch; a.bat_iter[b] = 0; a.bat_dl[b] = min(a.D0, a.max_len - 2); a.bat_grp[b] = 0; a.bat_live[b] = 0; a.bat_nfin[b] = 0;   // :457, :476
      a.bat_longest_fin[b] = 0; a.bat_longest_cur[b] = 0; a.bat_state[b] = 0; a.bat_given[b] = io.given_all[batch];
    }
    new_slot[got] = s;
    a.row_of[s] = row;
    a.slot_batch[s] = cur_slot;
    a.src_state[s] = BP_RUNNING;
    a.bat_live[cur_slot] += 1;
    for (int q = 0; q < 8; ++q) a.src_acc[s * 8 + q] = 0;
    for (int k = 0; k < a.K; ++k) {
      cand_src_len[s * a.K + k] = io.len_all[row]; a.cand_batch[s * a.K + k] = cur_slot;
      a.cache_slot[s * a.K + k] = s * a.K + k; a.cache_slot_parent[s * a.K + k] = s * a.K + k;      // a new source: its K slots are all free
    }
    ++s;
  }
  // the host only admits as many sources as it knows to be free, so got == R
  if (got != R) { a.dev_summary[2] = 1; a.host->error = 1; __threadfence_system(); }
}

struct BeamPoolFillArgs {
  const int* new_slot; int R; int first_row;
  int C, K, N, D0, ld, Ls_cap, Ls_new, max_len, pad, bos;
  int64_t* cand_next; int* len_next; uint8_t* fin_next; float* logp_next; int* parent; int* parent_draft;
  int* tok; const int* tok_new;                                       // [C][Ls_cap] <- [R][Ls_new]
  uint8_t* src_valid; const uint8_t* valid_new;
  int* drafts_all; const int* drafts_new;                             // all drafts: [C][N*D0] <- [R][N*D0] (null in smart mode)
  float* memkv; const float* memkv_new; int kv_row;                   // floats per source position (Ld * 2 * d)
  const BeamPoolIo* io;
};

// grid (R, 1 + Ls_new): block (i, 0) initialises source i's slot (candidate rows, tokens, drafts) and its caller-side rows,
// block (i, 1 + key) copies the cross K/V of one source position.
__global__ __launch_bounds__(256) void k_bsp_fill(BeamPoolFillArgs a) {
  const int i = blockIdx.x;
  const int s = a.new_slot[i];
  const int t = threadIdx.x;
  if (blockIdx.y == 0) {
    for (int e = t; e < a.K * a.ld; e += blockDim.x) a.cand_next[(size_t)s * a.K * a.ld + e] = (e == 0) ? a.bos : a.pad;
    for (int k = t; k < a.K; k += blockDim.x) {
      const int c = s * a.K + k;
      a.len_next[c] = 1; a.fin_next[c] = 0; a.logp_next[c] = 0.f; a.parent[c] = -1; a.parent_draft[c] = 0;
    }
    for (int c = t; c < a.Ls_cap; c += blockDim.x) {
      a.tok[(size_t)s * a.Ls_cap + c] = (c < a.Ls_new) ? a.tok_new[(size_t)i * a.Ls_new + c] : a.pad;
      a.src_valid[(size_t)s * a.Ls_cap + c] = (c < a.Ls_new) ? a.valid_new[(size_t)i * a.Ls_new + c] : (uint8_t)0;
    }
    if (a.drafts_new)
      for (int c = t; c < a.N * a.D0; c += blockDim.x) a.drafts_all[(size_t)s * a.N * a.D0 + c] = a.drafts_new[(size_t)i * a.N * a.D0 + c];
    const size_t row = (size_t)(a.first_row + i);
    const BeamPoolIo io = *a.io;
    for (int c = t; c < a.K * a.max_len; c += blockDim.x) io.out[row * a.K * a.max_len + c] = a.pad;
    for (int c = t; c < io.T_cap; c += blockDim.x) io.trace_len[row * io.T_cap + c] = -1;
    for (int c = t; c < 8; c += blockDim.x) io.summary[row * 8 + c] = 0;
  } else {
    const int key = blockIdx.y - 1;
    const float4* src = reinterpret_cast<const float4*>(a.memkv_new + ((size_t)i * a.Ls_new + key) * a.kv_row);
    float4* dst = reinterpret_cast<float4*>(a.memkv + ((size_t)s * a.Ls_cap + key) * a.kv_row);
    for (int c = t; c < a.kv_row / 4; c += blockDim.x) dst[c] = src[c];
  }
}

// One workgroup per candidate slot c = s * K + k: k_bs_prep for the pool.  A slot without a source, or a candidate index the
// source does not have yet (in its batch's first iteration a source has ONE <BOS> candidate, :447-459), is dead: not decoded,
// no leaves.  Smart mode reads the library straight from the source tokens: make_drafts(src, draft_len + 1, Ls - 5, ...)
// (:603-615) returns ALL Ls - 5 stride-1 windows of the row padded to the GIVEN batch's width Ls, in order (it asks for as
// many drafts as there are windows), with EOS / PAD replaced by the replace token — window i is tokens i .. i + lib_ld - 1.
__global__ __launch_bounds__(256) void k_bsp_prep(BeamPoolArgs a) {
  __shared__ int s_scan[256];
  __shared__ int s_match[BS_MAX_SLOTS];
  const int c = blockIdx.x, t = threadIdx.x;
  const int s = c / a.K, k = c - s * a.K;
  const int row_id = a.row_of[s];
  const int b = a.slot_batch[s];
  const int beam = (row_id >= 0) ? (a.bat_iter[b] == 0 ? 1 : a.K) : 0;
  if (k >= beam) {
    if (t == 0) { a.active[c] = 0; a.finished[c] = 1; a.live[c] = 0; a.per_cand[c] = 0; a.len[c] = 1; a.front[c] = 0; }
    return;
  }
  const int64_t* row = a.cand_next + (size_t)c * a.ld;
  for (int col = t; col < a.ld; col += 256) a.gen[(size_t)c * a.ld + col] = (int)row[col];
  const int lc = a.len_next[c];
  const int fin = a.fin_next[c];
  if (t == 0) {
    a.len[c] = lc; a.front[c] = lc - 1; a.finished[c] = (uint8_t)fin; a.active[c] = fin ? 0 : 1; a.live[c] = 1; a.logp[c] = a.logp_next[c];
    a.cand_dl[c] = a.bat_dl[b];
  }
  const int dl = a.D0;                                // the step's row layout; the batch may look at fewer tokens (cand_dl)
  int* dst = a.drafts32 + (size_t)c * a.N * dl;
  if (!a.smart) {
    const int* src = a.drafts_all + (size_t)s * a.N * a.D0;
    for (int e = t; e < a.N * dl; e += 256) dst[e] = src[e];
    if (t == 0) a.per_cand[c] = a.N;
    return;
  }
  const int last = (int)row[lc - 1];
  const int* tk = a.tok + (size_t)s * a.Ls_cap;
  const int n_lib = a.bat_given[b] - 5;
  auto lib_tok = [&](int pos) {                      // token `pos` of the padded, service-token-free source row
    const int v = (pos < a.Ls_cap) ? tk[pos] : a.pad;
    return (v == a.eos || v == a.pad) ? a.repl : v;
  };
  int running = 0;
  for (int base = 0; base < n_lib && running < a.N; base += 256) {
    const int i = base + t;
    const int flag = (i < n_lib && lib_tok(i) == last) ? 1 : 0;
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
  for (int e = t; e < a.N * dl; e += 256) {
    const int n = e / dl, j = e % dl;
    dst[e] = lib_tok(s_match[n < count ? n : 0] + 1 + j);
  }
  if (t == 0) {
    a.per_cand[c] = count;
    atomicMax(&a.bat_grp[b], count);                  // the batch's longest draft group of this iteration (:779-784)
  }
}

// One workgroup per source slot: k_beam_select for the pool (the n_best best leaves of the source's candidates and the new
// rows) and the source's iteration record; whether the source retires is decided per batch afterwards (k_bsp_batches).
__global__ __launch_bounds__(256) void k_bsp_select(BeamPoolArgs a) {
  extern __shared__ float sh[];
  __shared__ int s_win[NUC_MAX_KEEP];
  __shared__ float s_wsc[NUC_MAX_KEEP];
  __shared__ int s_eos, s_maxreal, s_accsum, s_acccnt, s_running_next;
  const int s = blockIdx.x, t = threadIdx.x;
  const int row_id = a.row_of[s];
  if (row_id < 0) return;
  const int b = a.slot_batch[s];
  const int it0 = a.bat_iter[b];
  const int beam = it0 == 0 ? 1 : a.K;
  const int dl = a.D0, dl1 = dl + 1, K = a.K;
  const int c0 = s * K;
  const size_t seg0 = (size_t)c0 * dl1;
  if (t == 0) { s_eos = 0; s_maxreal = 0; s_accsum = 0; s_acccnt = 0; s_running_next = 0; }
  const bool enough = select_best_leaves(a.leaf_score, a.leaf_cnt, seg0, beam * dl1, K, sh, s_win, s_wsc);   // ends with a barrier
  const BeamPoolIo io = *a.io;
  if (enough) {
    // the K new rows: root tokens, the kept draft tokens, the leaf token (the roots are read from `gen`, which k_bsp_prep copied)
    for (int e = t; e < K * a.ld; e += blockDim.x) {
      const int r = e / a.ld, col = e - r * a.ld;
      const int sel = s_win[r];
      const int seg = sel / K, i = sel % K;
      const int cl = seg / dl1, p = seg % dl1;
      const int c = c0 + cl;
      const int tok = a.leaf_tok[(seg0 + seg) * K + i];
      const int lc = a.len[c];
      int64_t v = (int64_t)a.gen[(size_t)c * a.ld + col];
      const int j = col - lc;
      if (j >= 0 && j <= dl) v = (j < p) ? a.chosen[(size_t)c * dl + j] : (j == p ? (int64_t)tok : (int64_t)a.pad);
      a.cand_next[(size_t)(c0 + r) * a.ld + col] = v;
    }
    for (int r = t; r < K; r += blockDim.x) {
      const int sel = s_win[r];
      const int seg = sel / K, i = sel % K;
      const int cl = seg / dl1, p = seg % dl1;
      const int c = c0 + cl, out = c0 + r;
      const int tok = a.leaf_tok[(seg0 + seg) * K + i];
      const int lc = a.len[c];
      const int fin_root = a.finished[c];
      const bool has_eos = fin_root || tok == a.eos;       // accepted draft tokens are never EOS (drafting.py:65)
      const int real = (tok == a.pad) ? lc + p : lc + p + 1;
      a.logp_next[out] = s_wsc[r];
      a.parent[out] = c;
      a.parent_draft[out] = a.chosen_slot[c];
      a.len_next[out] = real;
      a.fin_next[out] = has_eos ? 1 : 0;
      if (has_eos) atomicAdd(&s_eos, 1); else atomicAdd(&s_running_next, 1);
      atomicMax(&s_maxreal, real);
      if (!fin_root) { atomicAdd(&s_accsum, p); atomicAdd(&s_acccnt, 1); }
    }
  }
  __syncthreads();
  if (t == 0) {
    const int it = it0 + 1;
    int lines = 0, running = 0, run_cands = 0;
    for (int k = 0; k < beam; ++k) {
      const int pc = a.per_cand[c0 + k];
      lines += pc;
      if (!a.finished[c0 + k]) { running += pc; ++run_cands; }
    }
    int* acc = a.src_acc + s * 8;
    acc[0] += lines; acc[1] += running; acc[2] += s_accsum; acc[3] += s_acccnt; acc[4] += run_cands; acc[5] = s_maxreal; acc[6] = s_running_next;
    if (it - 1 < io.T_cap) io.trace_len[(size_t)row_id * io.T_cap + it - 1] = (short)s_maxreal;
    int st = BP_RUNNING;
    if (!enough) { st = BP_ERR_LEAVES; atomicMax(&a.bat_state[b], 2); }       // the reference asserts for the whole batch (:195)
    else if (s_eos == K) st = BP_DONE;
    a.src_state[s] = st;
    if (enough) atomicMax(&a.bat_longest_cur[b], s_maxreal);
    if (enough) {
      // Cache slots of the K new candidates (k_tree_cache): a parent's slot goes to ONE of its children — a running one if there
      // is one — which then only appends its accepted rows; every further child copies the parent's rows into the slot of a
      // candidate without children (nobody reads that slot again).  K children, K slots: it always works out.  The candidates
      // themselves stay in the order the selection gave them; only the attention's cache pointer goes through the map.
      int old_slot[NUC_MAX_KEEP], par[NUC_MAX_KEEP], new_slot[NUC_MAX_KEEP];
      for (int q = 0; q < K; ++q) { old_slot[q] = a.cache_slot[c0 + q]; new_slot[q] = -1; }
      for (int r = 0; r < K; ++r) par[r] = (s_win[r] / K) / dl1;
      unsigned taken = 0;
      for (int pass = 0; pass < 2; ++pass)
        for (int r = 0; r < K; ++r) {
          if (new_slot[r] >= 0 || (a.fin_next[c0 + r] ? 1 : 0) != pass) continue;
          if (!((taken >> par[r]) & 1u)) { new_slot[r] = old_slot[par[r]]; taken |= 1u << par[r]; }
        }
      int q = 0;
      for (int r = 0; r < K; ++r) {
        if (new_slot[r] >= 0) continue;
        while ((taken >> q) & 1u) ++q;
        new_slot[r] = old_slot[q];
        taken |= 1u << q;
      }
      for (int r = 0; r < K; ++r) { a.cache_slot_parent[c0 + r] = old_slot[par[r]]; a.cache_slot[c0 + r] = new_slot[r]; }
    }
  }
}

// One thread per batch slot: the reference's loop scalars of every batch in the pool after this iteration (:578-598):
// longest row (finished sources included), room, the next draft length, and whether the loop goes on.
__global__ void k_bsp_batches(BeamPoolArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.C || a.bat_id[b] < 0) return;
  const int it = a.bat_iter[b] + 1;
  a.bat_iter[b] = it;
  if (a.bat_state[b] == 2) return;                               // error: its sources retire with BP_ERR_LEAVES
  const int longest = max(a.bat_longest_fin[b], a.bat_longest_cur[b]);
  const int room = a.max_len - longest - 1;                      // possible_draft_len (:596)
  if (room < 1) a.bat_state[b] = 1;                              // `while possible_draft_len >= 1` (:464) ends the loop
  else if (a.max_steps > 0 && it >= a.max_steps) a.bat_state[b] = 3;   // guard (not in the reference) — unless everything finished
  a.bat_dl[b] = min(room < 1 ? 1 : room, a.bat_dl[b]);           // :476 for the next iteration
  a.bat_longest_cur[b] = 0;
}

// One workgroup per source slot: sources that finished (every row holds EOS) or whose batch stopped hand their hypotheses and
// their summary to the caller's arrays and free the slot; the last source of a batch frees the batch slot.
__global__ __launch_bounds__(256) void k_bsp_retire(BeamPoolArgs a) {
  const int s = blockIdx.x, t = threadIdx.x;
  const int row_id = a.row_of[s];
  if (row_id < 0) return;
  const int b = a.slot_batch[s];
  const int bst = a.bat_state[b];
  int st = a.src_state[s];
  // a finished source stays finished whatever happens to its batch later; a running one follows its batch
  if (st == BP_RUNNING) st = (bst == 1) ? BP_STOPPED : (bst == 2) ? BP_ERR_LEAVES : (bst == 3) ? BP_MAX_STEPS : BP_RUNNING;
  else if (st == BP_DONE && bst == 2) st = BP_ERR_LEAVES;         // the batch raised in this very iteration
  const int* acc = a.src_acc + s * 8;
  if (st == BP_RUNNING) {
    if (t == 0) { atomicAdd(&a.dev_summary[0], 1); atomicAdd(&a.dev_summary[1], acc[6]); }
    return;
  }
  const BeamPoolIo io = *a.io;
  const int K = a.K;
  if (st == BP_DONE || st == BP_STOPPED)
    for (int e = t; e < K * a.max_len; e += blockDim.x) {
      const int r = e / a.max_len, col = e - r * a.max_len;
      io.out[((size_t)row_id * K + r) * a.max_len + col] = a.cand_next[(size_t)(s * K + r) * a.ld + col];
    }
  if (t == 0) {
    int* sm = io.summary + (size_t)row_id * 8;
    sm[0] = a.bat_iter[b]; sm[1] = st; sm[2] = acc[0]; sm[3] = acc[1]; sm[4] = acc[2]; sm[5] = acc[3]; sm[6] = acc[5]; sm[7] = acc[4];
    a.row_of[s] = -1;
    atomicMax(&a.bat_longest_fin[b], acc[5]);                     // a finished source's rows stay in the batch's tensor
    atomicAdd(&a.bat_nfin[b], 1);
    if (atomicSub(&a.bat_live[b], 1) == 1) a.bat_id[b] = -1;      // last source of the batch: the batch slot is free
  }
}

// Last kernel of an iteration: live sources / running candidates of the NEXT iteration and the iteration count go to the
// pinned words the host polls; per-iteration tallies are reset (smart mode: a batch with finished sources starts its longest
// draft group at 1 — a finished row has exactly one draft, :417).
__global__ void k_bsp_publish(BeamPoolArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < a.C) a.bat_grp[b] = (a.bat_id[b] >= 0 && a.bat_nfin[b] > 0) ? 1 : 0;
  if (b == 0) {
    a.host->n_live = a.dev_summary[0];
    a.host->n_running = a.dev_summary[1];
    if (a.dev_summary[2]) a.host->error = 1;
    a.dev_summary[0] = 0; a.dev_summary[1] = 0;
    __threadfence_system();
    a.host->steps_done = (int)a.cnt->model_calls;
    __threadfence_system();
  }
}

}  // namespace ttx
