// Host-side types shared by the translation units of libttx_hip.so (ttx_api.hip: C ABI, runtime, loop kernels;
// ttx_gemm.hip: GEMM family; ttx_attn.hip: attention family).  Kernels are launched from the unit that defines them;
// the other units reach them through the launchers declared at the bottom.
#pragma once
#include "ttx.h"
#include "ttx_common.hip.h"

#include <map>
#include <set>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace ttx {
struct BeamHost;
struct BeamPoolHost;
}

struct LayerW {
  // offsets (in floats) into the blob
  size_t sa_in_w, sa_in_b, sa_out_w, sa_out_b;
  size_t ca_in_w, ca_in_b, ca_out_w, ca_out_b;  // decoder only
  size_t l1_w, l1_b, l2_w, l2_b;
  size_t n1_w, n1_b, n2_w, n2_b, n3_w, n3_b;
};

struct ttx_model {
  ttx_config cfg;
  int device;
  int n_cu = 256;                  // compute units of the device (grids sized to the machine: k_attn3s)
  float* blob = nullptr;
  size_t blob_floats = 0;
  std::map<std::string, std::pair<size_t, size_t>> index;  // name -> (offset, numel)
  std::vector<LayerW> enc, dec;
  size_t src_emb, tgt_emb, enc_norm_w, enc_norm_b, dec_norm_w, dec_norm_b, cls_w, cls_b, pe;
  size_t cross_kv_w, cross_kv_b;  // packed [Ld*2d, d] / [Ld*2d]: cross-attention K,V rows of every decoder layer
  const float* p(size_t off) const { return blob + off; }
};

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  uint64_t* owner_gen = nullptr;   // the owning session's alloc_generation: bumped whenever this buffer moves
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Which kernels carry a verify step's GEMMs.  Every contraction is DEFINED as the ordered sum of fixed K slices, each
// accumulated from zero in k order (ttx_gemm.hip), and both variants evaluate exactly that sum — so the choice is free
// to follow the live row count of a step (it is made per step on the host) without touching a single bit of the result.
//   GV_BIG    one workgroup walks all slices of its output tile (128x64 / 64x64 tiles by live row count)
//   GV_SMALL  short dependent chains for steps of a few hundred rows: one wave per slice (32x32 tiles) for the K = 256
//             GEMMs up to 768 columns, one workgroup per slice (split-K slabs, summed in order by k_finish_ln) for FFN2
//   GV_BIG_FFN2_SLABS  GV_BIG for every GEMM of the step except FFN2, which runs as in GV_SMALL (between small_rows and
//             ffn2_slab_rows live rows one workgroup per 256-k slice beats the 128x64 / 64x64 tiles walking all 2 048 k's)
//   GV_MID    between qkv_small_rows and small_rows: the short-chain kernels for the d-wide K = d GEMMs, the classifier and FFN2,
//             the tiles for QKV and FFN1 (a 32x32 launch over 768 columns loses to 64x64 tiles from about 800 rows on)
enum GemmVariant { GV_BIG = 0, GV_SMALL = 1, GV_BIG_FFN2_SLABS = 2, GV_MID = 3 };

struct GraphKey {
  int B, Ls, N, D, max_len, mode, kcap, variant;   // mode: 0 speculative, 1 plain greedy, 2 per-row rule, 3 slot pool
  bool operator<(const GraphKey& o) const {
    return std::tie(B, Ls, N, D, max_len, mode, kcap, variant) < std::tie(o.B, o.Ls, o.N, o.D, o.max_len, o.mode, o.kcap, o.variant);
  }
};

#ifndef TTX_BIG_MIN_TILES
#define TTX_BIG_MIN_TILES 400
#endif
struct ttx_session {
  ttx_model* m;
  std::vector<Buf*> all;
  // activations (shared by encoder / full decoder / step)
  Buf x, x1, x2, xf, ao, q2, hbuf, slab, qkv, logits, ckv;
  // sources
  Buf tok_src, src_valid, memory, memkv;
  // full decoder
  Buf tok_tgt, mem_pad_tmp;
  // loop
  Buf drafts, gen, front, act_idx, rec, pred, state, kcache, vcache, src32, outbuf, haspad, traj, fin_step;
  // slot pool (continuous batching)
  Buf rstep, row_of, src_len, new_slot, pool_io, memkv_new, valid_new, drafts_new;
  // snapshot of one verify step for the logits parity test (ttx_gen_params.want_logits)
  Buf snap_logits, snap_act, snap_front, snap_gen, snap_state;
  int snap_B = 0, snap_rps = 0, snap_gen_ld = 0, snap_step = 0;
  Buf leaf_score, leaf_tok, leaf_cnt, beam_summary;
  // native beam-speculative loop
  Buf bs_cand_next, bs_len_next, bs_fin_next, bs_logp_next, bs_len, bs_fin, bs_active, bs_logp, bs_per_cand, bs_best_n, bs_best_slot,
      bs_chosen, bs_parent, bs_parent_draft, bs_mark, bs_drafts_src, bs_cnt, bs_hit;
  ttx::BeamHost* beam_host = nullptr;   // pinned + device-mapped, written by k_bs_publish
  // tree (beam) decoding
  Buf tk[2], tv[2], t_prev_len, t_slot_of, t_src_of;
  // beam-speculative source pool (continuous batching over sources of many batches)
  Buf bp_row_of, bp_batch, bp_cand, bp_cand_len, bp_tok, bp_new_slot, bp_io, bp_grp, bp_src_acc, bp_enc_qkv;
  ttx::BeamPoolHost* bp_host = nullptr; // pinned + device-mapped, written by k_bsp_publish
  ttx::HostInfo* host_info = nullptr;   // pinned + device-mapped, written by the accept kernels
  hipStream_t own_stream = nullptr;     // the loops run on a session-owned stream (the caller's may be the null stream)
  // Captured graphs hold raw pointers into the workspaces.  EVERY growth of a buffer of this session (whichever entry
  // point caused it) bumps alloc_generation through Buf::owner_gen; the graph cache remembers the generation it was
  // captured under and is dropped as soon as the two differ (graphs_current(), called before any replay or capture).
  uint64_t alloc_generation = 0;
  uint64_t graphs_generation = 0;
  bool dead = false;               // a verify step never published its result (watchdog): the stream may still be stuck
  bool use_graphs = true;
  std::map<GraphKey, hipGraphExec_t> graphs;
  std::set<GraphKey> warmed;
  hipEvent_t ev_done = nullptr;
  std::map<std::vector<int>, hipGraphExec_t> beam_graphs;   // one iteration of the beam-speculative loop per shape
  std::set<std::vector<int>> beam_warmed;
  void drop_graphs() {
    for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
    graphs.clear(); warmed.clear();
    for (auto& kv : beam_graphs) (void)hipGraphExecDestroy(kv.second);
    beam_graphs.clear(); beam_warmed.clear();
  }
  void graphs_current() { if (graphs_generation != alloc_generation) { drop_graphs(); graphs_generation = alloc_generation; } }
  ttx::DecState* host_state = nullptr;  // pinned copy target
  // function attributes (dynamic LDS limits) are per device: set once per session, outside graph capture
  bool attr_attn2[8] = {false, false, false, false, false, false, false, false};
  bool attr_select = false, attr_step = false, attr_topk = false, attr_pool_select = false;
  // GEMM policy (all choices are between bit-identical evaluations, see GemmVariant)
  int qkv_small_rows = 800;        // a verify step with fewer live rows than this runs under GV_SMALL (TTX_QKV_SMALL_ROWS)
  int small_rows = 2000;           // ... with fewer than this under GV_MID (TTX_SMALL_ROWS)
  int ffn2_slab_rows = 5600;       // ... and with fewer than this (and at least small_rows) under GV_BIG_FFN2_SLABS (TTX_FFN2_SLAB_ROWS; 0: never)
  // k_gemm24 picks the tiling per launch from the live row count: 128x64 tiles once there are big_min_tiles of them,
  // else 64x64 (TTX_BIG_MIN_TILES)
  int big_min_tiles = TTX_BIG_MIN_TILES;
  int attn_split = -1;             // -1 by launch size, 0 never, 1 always (key tiles of a head over 4 waves)
  bool attn_fallback = false;      // TTX_ATTN_FALLBACK=1 (test hook): every attention launch on the streaming kernel k_attn
  // profiling of the GEMM launches (bench.py roofline): a HIP event pair around every GEMM launch
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double prof_ms = 0;
  double prof_empty_pair_ms = -1;
  long long prof_launches = 0;
  bool host_timing = false;
  double host_launch_us = 0;
  long long host_captures = 0;      // TTX_HOST_TIMING: iteration graphs captured on this session and the host time they took
  double host_capture_us = 0;
  long long host_launches = 0;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
  ttx_session() { for (Buf* b : {&x, &x1, &x2, &xf, &ao, &q2, &hbuf, &slab, &qkv, &logits, &ckv, &tok_src, &src_valid, &memory,
                                 &memkv, &tok_tgt, &mem_pad_tmp, &drafts, &gen, &front, &act_idx, &rec, &pred, &state,
                                 &kcache, &vcache, &src32, &outbuf, &haspad, &traj, &fin_step, &rstep, &row_of, &src_len, &new_slot,
                                 &pool_io, &memkv_new, &valid_new, &drafts_new, &tk[0], &tk[1], &tv[0], &tv[1],
                                 &t_prev_len, &t_slot_of, &t_src_of,
                                 &snap_logits, &snap_act, &snap_front, &snap_gen, &snap_state, &leaf_score, &leaf_tok, &leaf_cnt,
                                 &beam_summary, &bs_cand_next, &bs_len_next, &bs_fin_next, &bs_logp_next, &bs_len, &bs_fin, &bs_active,
                                 &bs_logp, &bs_per_cand, &bs_best_n, &bs_best_slot, &bs_chosen, &bs_parent, &bs_parent_draft, &bs_mark,
                                 &bs_drafts_src, &bs_cnt, &bs_hit, &bp_row_of, &bp_batch, &bp_cand, &bp_cand_len, &bp_tok, &bp_new_slot,
                                 &bp_io, &bp_grp, &bp_src_acc, &bp_enc_qkv}) { b->owner_gen = &alloc_generation; all.push_back(b); } }
};

namespace ttx {

// error text of the calling thread (ttx_last_error); returns `code`
int fail(int code, const std::string& msg);

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      return ttx::fail(TTX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                                        std::to_string(__LINE__) + ")");                               \
  } while (0)

#define TTX_TRY(expr)        \
  do {                       \
    int _r = (expr);         \
    if (_r != TTX_OK) return _r; \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- ttx_gemm.hip ------------------------------------------------------------------------------------------------
// Canonical slice length of a contraction over K (0: one chain; K is then not a multiple of 64).
int gemm_slice_k(int K);
// Split-K slabs of a d-wide step GEMM under `variant` (1 for everything but FFN2 under GV_SMALL).
int gemm_splits(int N, int K, bool step, int variant);
// Y = act(X W^T + b) (splits == 0) or `splits` raw slabs [splits][Mmax][ldy] (k_finish_ln sums them in slab order).
// m_ptr != null marks a verify-step launch (live row count on the device, capacity Mmax); `variant` is a GemmVariant.
int launch_gemm(ttx_session* s, hipStream_t st, const float* X, int ldx, const float* W, int ldw, const float* bias,
                float* Y, int ldy, const int* m_ptr, int Mmax, int N, int K, bool relu, int splits, long long slab_stride,
                int variant);
int launch_finish(ttx_session* s, hipStream_t st, const float* slabs, int n_slabs, long long slab_stride, const float* bias,
                  const float* resid, const float* g1, const float* b1, const float* g2, const float* b2,
                  const uint8_t* row_valid, float* Y, const int* m_ptr, int Mmax);
int gemm_bench(ttx_session* s, int M, int N, int K, int splits, int variant, int reps, double* us_per_launch, double* max_abs_diff);

// ---- ttx_attn.hip ------------------------------------------------------------------------------------------------
// `groups` = sources / decoder rows / running-sequence slots; `q_per_group` = query rows of one group (Ls, Lt, or the
// 1 + N*D step rows of a sequence); for the step modes `D1`/`N` shape the draft tiles.
int launch_attn(int mode, ttx_session* s, hipStream_t st, const AttnArgs& a, int groups, int H, int q_per_group, int max_keys,
                int N = 1, int D1 = 1);

}  // namespace ttx
