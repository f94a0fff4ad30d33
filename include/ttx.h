/*
 * ttx.h — C ABI of libttx_hip.so: MI355X-native (gfx950) encoder–decoder forward and speculative
 * decoding for the Molecular Transformer hot path of Academich/translation-transformer.
 *
 * The reference has no FFI: its hot path is Python calling stock torch ops.  Each entry point below
 * names the reference function it replaces (paths relative to the reference root).  Conventions:
 *   - every pointer named d_* is a DEVICE pointer owned by the caller (a torch tensor's data_ptr());
 *     the library borrows it for the duration of the call and never frees it;
 *   - token tensors are int64 row-major exactly as the reference's LongTensors; floats are fp32;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued on it; calls that return host-side results synchronise that stream before returning;
 *   - every function returns 0 on success or a negative ttx_status; ttx_last_error() gives the text
 *     (thread-local).  Nothing throws across the boundary;
 *   - one host thread drives one ttx_session at a time (the Lightning predict loop is sequential:
 *     src/model/lightning_model.py:209-212); distinct sessions on distinct streams may run concurrently.
 */
#ifndef TTX_H
#define TTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTX_ABI_VERSION 4

typedef enum ttx_status {
  TTX_OK = 0,
  TTX_ERR_INVALID = -1,      /* bad argument / shape                                             */
  TTX_ERR_HIP = -2,          /* a HIP runtime call failed                                        */
  TTX_ERR_NO_DEVICE = -3,    /* no gfx950 device visible: the library has NO CPU fallback         */
  TTX_ERR_REFERENCE = -4,    /* input on which the reference itself raises (see ttx_last_error)  */
  TTX_ERR_NOMEM = -5,
  TTX_ERR_ROW_REPLAY = -6,   /* ttx_greedy_speculative_generate_rows only: decode the batches as given instead */
  TTX_ERR_MAX_STEPS = -7     /* beam-speculative: the ttx_beam_params.max_steps guard tripped (non-terminating input) */
} ttx_status;

/* Model hyper-parameters: the init_args of VanillaTransformer (src/model/modules.py:11-38). */
typedef struct ttx_config {
  int32_t vocab_size;           /* tgt (= src when shared) vocabulary                          */
  int32_t src_vocab_size;
  int32_t embedding_dim;        /* d: multiple of 64, <= 1024                                  */
  int32_t num_heads;            /* d / num_heads must be 32                                    */
  int32_t feedforward_dim;      /* multiple of 64                                              */
  int32_t num_encoder_layers;
  int32_t num_decoder_layers;
  int32_t pad_token;            /* src == tgt pad id (0 in the reference: tokenizer_base.py:27) */
  int32_t max_positions;        /* rows of the sinusoid table minus one (embeddings.py:31: 5000) */
  float   layer_norm_eps;       /* modules.py:52: 1e-5                                         */
} ttx_config;

typedef struct ttx_model ttx_model;      /* weights resident in HBM                          */
typedef struct ttx_session ttx_session;  /* workspaces, KV caches, captured graphs, one stream at a time */

/* One named fp32 host tensor of the reference state dict ("model." prefix stripped). */
typedef struct ttx_tensor {
  const char*  name;
  const float* data;      /* HOST pointer, row-major */
  int64_t      numel;
} ttx_tensor;

/* Library / device ----------------------------------------------------------------------------- */
int         ttx_abi_version(void);
const char* ttx_last_error(void);
/* Number of visible gfx950 devices (0 when there is none; never raises). */
int         ttx_device_count(void);

/* Weights in: replaces VanillaTransformer.__init__ + load_state_dict (modules.py:11-84; checkpoint key
 * layout of lightning_model.py / tests/test_batching.py:48-49).  Tensors are looked up by name; a
 * missing or mis-sized tensor is TTX_ERR_INVALID.  The sinusoid table (embeddings.py:38-45, not in the
 * state dict) is rebuilt on the host with the same fp32 formula. */
int  ttx_model_create(const ttx_config* cfg, const ttx_tensor* tensors, int n_tensors, int device,
                      ttx_model** out);
void ttx_model_destroy(ttx_model* m);
/* Packed weight blob (all weights, one contiguous device allocation) for the RCCL broadcast of
 * rank 0 creates the model from host tensors, every other rank creates it with
 * ttx_model_create_empty and receives the blob with one ncclBroadcast on [ptr, ptr+bytes). */
int  ttx_model_create_empty(const ttx_config* cfg, int device, ttx_model** out);
int  ttx_model_blob(ttx_model* m, void** d_ptr, int64_t* bytes);

int  ttx_session_create(ttx_model* m, ttx_session** out);
void ttx_session_destroy(ttx_session* s);

/* Model protocol ---------------------------------------------------------- */

/* VanillaTransformer.encode_src (modules.py:110-116).  d_src: int64 [B,Ls]; PAD keys masked
 * (src == pad_token); d_memory: fp32 [B,Ls,d] out, rows at PAD positions are written as zeros. */
int ttx_encode_src(ttx_session* s, const int64_t* d_src, int B, int Ls, float* d_memory, void* stream);

/* VanillaTransformer.decode_tgt (modules.py:118-138): full-prefix decoder forward + classifier.
 * d_tgt int64 [R,Lt]; d_memory fp32 [Rm,Ls,d]; d_mem_pad uint8 [Rm,Ls] (1 = PAD key);
 * d_mem_row int32 [R] maps decoder row -> memory row (NULL: identity, Rm == R, i.e. the reference's
 * inflated memory); d_logits fp32 [R,Lt,V] out. */
int ttx_decode_tgt(ttx_session* s, const int64_t* d_tgt, int R, int Lt, const float* d_memory,
                   const uint8_t* d_mem_pad, const int32_t* d_mem_row, int Rm, int Ls, float* d_logits,
                   void* stream);

/* VanillaTransformer.forward (modules.py:86-108; step 0 of standard beam search). d_logits [B,Lt,V]. */
int ttx_forward(ttx_session* s, const int64_t* d_src, int B, int Ls, const int64_t* d_tgt, int Lt,
                float* d_logits, void* stream);

/* Draft maker: make_drafts (src/utils/drafting.py:5-67) on the device.  d_src int64 [B,L];
 * d_drafts int64 [B,n_drafts,D] out with D = clamp(draft_len, min_draft_len, max_draft_len). */
int ttx_make_drafts(ttx_session* s, const int64_t* d_src, int B, int L, int draft_len, int n_drafts,
                    int min_draft_len, int max_draft_len, int eos_token, int pad_token, int replace_token,
                    int64_t* d_drafts, void* stream);

/* Generators -------------------------------------------------------------- */

typedef struct ttx_gen_params {
  int32_t max_len;
  int32_t draft_len;       /* greedy-speculative: D (clamped to [1,max_len] as speculative_decoding.py:64-73) */
  int32_t n_drafts;        /* N */
  int32_t pad_token, bos_token, eos_token, replace_token;
  int32_t want_logits;     /* parity tests: k > 0 records verify step k (1-based) for ttx_debug_step_snapshot */
} ttx_gen_params;

typedef struct ttx_gen_stats {
  int64_t model_calls;         /* decoder invocations == verify steps (generator.model_calls_num)      */
  int64_t accepted_tokens;     /* draft tokens accepted over all rows and steps                        */
  int64_t produced_tokens;     /* tokens written (accepted + one bonus token per row per step)         */
  int64_t verified_positions;  /* sum over steps of Bc*N*(D+1): rows of the step's GEMMs               */
  int64_t kv_prefix_positions; /* sum over steps and running rows of the cached prefix length          */
  int64_t src_positions;       /* sum over steps of Bc*Ls (cross-attention keys read)                  */
  double  encode_ms, decode_ms;/* device time (HIP events on `stream`) of the two phases               */
  int64_t src_tokens_padded;   /* encoder positions computed: rows x padded length, summed over encoder passes         */
  int64_t status;              /* this batch's own status (TTX_OK, TTX_ERR_REFERENCE, ...): the *_many calls return the
                                  first failure but decode every batch                                  */
} ttx_gen_stats;

/* TranslationInferenceGreedySpeculative.generate (src/decoding/speculative_decoding.py:39-174) with a
 * KV cache: encoder once, cross-attention K/V projected once per source, D+1 new positions per draft per
 * step.  d_src int64 [B,Ls]; d_out int64 [B,1,max_len] (PAD-filled; rows that never reach EOS stay
 * all-PAD exactly as the reference leaves them).  Returns TTX_ERR_REFERENCE where the reference raises
 * (a row finishing at a width beyond max_len, :158). */
int ttx_greedy_speculative_generate(ttx_session* s, const int64_t* d_src, int B, int Ls,
                                    const ttx_gen_params* p, int64_t* d_out, ttx_gen_stats* stats,
                                    void* stream);

/* TranslationInferenceGreedy.generate (src/decoding/standard_decoding.py:30-55) with a KV cache.
 * d_out int64 [B,1,max_len]. */
int ttx_greedy_generate(ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_gen_params* p,
                        int64_t* d_out, ttx_gen_stats* stats, void* stream);

/* Beam-speculative bookkeeping kernels.
 * ttx_nucleus_mask: mask_with_num_logits_according_nucleus (src/decoding/speculative_decoding.py:871-904): per row of
 *   d_logits [rows,V] keep the best logit and further ones, best first, while the softmax mass ranked above is
 *   < nucleus, never more than n_best (<= 32); everything else becomes `fill`.  d_out [rows,V].  V <= 1024.
 * ttx_accepted_lengths: the nucleus mask (0.9975-style) fused with calculate_n_accepted_in_drafts (:847-869):
 *   d_logits [R,D+1,V], d_drafts int64 [R,D] -> d_n_ok int32 [R] = leading draft tokens inside their position's kept set.
 * ttx_ragged_topk: topk_in_each_group (:177-238): d_score [sum of group lengths], d_offsets int32 [G+1] exclusive prefix
 *   sums, every group >= k entries; d_top fp32 [G,k] and d_idx int64 [G,k] (flat indices) best first; equal scores: lower
 *   index first (torch leaves ties unspecified). */
int ttx_nucleus_mask(ttx_session* s, const float* d_logits, int rows, int V, float nucleus, int n_best, float fill,
                     float* d_out, void* stream);
int ttx_accepted_lengths(ttx_session* s, const float* d_logits, const int64_t* d_drafts, int R, int D, int V, float nucleus,
                         int n_best, int32_t* d_n_ok, void* stream);
int ttx_ragged_topk(ttx_session* s, const float* d_score, const int32_t* d_offsets, int G, int max_group, int k,
                    float* d_top, int64_t* d_idx, void* stream);
/* TranslationInferenceBeamSearchSpeculative.generate (src/decoding/speculative_decoding.py:241-869) — the whole loop on the
 * device: generate_trying_all_the_drafts (:428-598) or generate_with_smart_drafts (:600-845), `sample` (:294-400),
 * calculate_n_accepted_in_drafts (:847-869), mask_with_num_logits_according_nucleus (:871-904) and topk_in_each_group
 * (:177-238), on a per-candidate KV cache (encoder and cross K/V once per source).  Ties between drafts with the same
 * accepted length are resolved as torch's CPU topk(1) resolves them (csrc/ttx_select.h), so the candidates are those of the
 * reference run on the CPU.  draft_len is clamped to [5, 200] as the reference's constructor does (:278-284).
 *   d_src int64 [B, Ls];  d_out int64 [B, n_best, max_len] (row stride max_len): the reference's result tensor
 *   [B, n_best, W] occupies the first W = stats->out_width columns of every row (W <= max_len), hypotheses best first.
 * Errors: TTX_ERR_REFERENCE where the reference asserts/raises (fewer candidate leaves than n_best for a source, :195;
 * no drafts, drafting.py:39-43; max_len < 3); TTX_ERR_MAX_STEPS when the guard below trips.
 * Limits of the native loop (the reference has none; TTX_ERR_INVALID names the one exceeded): n_best <= 32, n_drafts <= 64,
 * vocabulary <= 1024, n_best * (draft_len + 1) <= 1023 and 2 * n_best^2 * (draft_len + 1) * 4 bytes <= 150 KB of LDS. */
typedef struct ttx_beam_params {
  int32_t max_len;
  int32_t n_best;            /* <= 32                                                                  */
  int32_t draft_len;
  int32_t n_drafts;          /* <= 64: drafts per candidate (all drafts) / most drafts tried per candidate (smart) */
  int32_t smart_drafts_mode;
  int32_t pad_token, bos_token, eos_token, replace_token;
  int32_t max_steps;         /* 0: none (reference behaviour: its loop does not end when a candidate keeps emitting PAD
                                before any EOS); > 0: fail once more than this many iterations would be needed */
} ttx_beam_params;

typedef struct ttx_beam_stats {
  int64_t model_calls;             /* iterations == decoder invocations (generator.model_calls_num)         */
  int64_t input_lines;             /* (candidate, draft) rows built over all iterations (model_input_lines_num) */
  int64_t running_rows;            /* of those, rows of unfinished candidates (the reference's b_sz)        */
  int64_t accepted_tokens;         /* generator.accepted_tokens_num                                         */
  int64_t produced_non_pad_tokens; /* generator.produced_non_pad_tokens                                     */
  int64_t verified_positions;      /* decoder positions the KV-cached algorithm needs (running candidates + draft tokens) */
  int64_t executed_positions;      /* rows of the step GEMMs actually computed (unused smart-mode draft slots included) */
  int64_t kv_prefix_positions;     /* cached prefix positions attended                                      */
  int64_t running_candidates;      /* sum over iterations of unfinished candidates                          */
  int64_t src_tokens_padded;       /* encoder positions computed: B x Ls                                    */
  double  encode_ms, decode_ms;    /* device time (HIP events) of encoder + cross K/V + drafts / of the loop */
  int32_t out_width;               /* W: columns of the result tensor                                       */
  int32_t status;                  /* this batch's own status (the *_many call returns the first failure)   */
} ttx_beam_stats;

int ttx_beam_speculative_generate(ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_beam_params* p,
                                  int64_t* d_out, ttx_beam_stats* stats, void* stream);
/* Several batches in flight (batch i on sessions[i % n_sessions], each on its own stream; one host thread drives all of
 * them): per-batch outputs and stats are those of n_batches calls of ttx_beam_speculative_generate. */
int ttx_beam_speculative_generate_many(ttx_session** sessions, int n_sessions, int n_batches, const int64_t* const* d_src,
                                       const int* B, const int* Ls, const ttx_beam_params* p, int64_t* const* d_out,
                                       ttx_beam_stats* stats, void* stream);

/* The same generator with continuous batching over many given batches (SURVEY.md §8(f) #1 for the beam path).  In the
 * reference's loop the sources of a batch meet only in batch-wide scalars: the draft length min(max_len - longest row - 1,
 * draft_len) (:476 / :671), the stop rule (every row holds EOS, :586 / :826, or no room left, :464 / :652), the tensor width
 * and, in smart mode, the width of the -1-padded table the best draft is picked from (the batch's longest draft group,
 * :779-784 -> :225); a source all of whose n_best rows hold EOS is a fixed point of the iteration.  A session here owns a pool
 * of `capacity` source slots: given batches are admitted whole (their sources run in lock-step), ONE verify step per iteration
 * serves every live candidate of every batch in the pool, the batch-wide scalars are kept per batch on the device exactly as
 * the reference computes them, a source that finished frees its slots at once, and the pool is refilled from the work list.
 * Per source it returns
 *   d_out        int64 [R_total][n_best][max_len]  hypotheses best first, PAD beyond
 *   d_trace_len  int16 [R_total][trace_cap]        longest hypothesis after each of the source's iterations (-1 past the last)
 *   d_summary    int32 [R_total][8]                iterations of its batch when it retired, status (1 every row holds EOS,
 *                                                  3 its batch ran out of room: rows as they stood, 2 fewer leaves than
 *                                                  n_best: the reference asserts for the batch, 4 max_steps), input lines,
 *                                                  running rows, accepted-token sum, accepted count, longest hypothesis,
 *                                                  decoded candidates summed over the iterations
 * from which translation-transformer_amd/scheduling.py:replay_beam_batch derives each given batch's result width, model calls
 * and counters.  d_src int64 [R_total][Ls_all]: all sources right-padded, batch after batch in admission order; HOST arrays:
 * h_len int32 [R_total] a source's length (position after its last non-PAD token), h_batch_of int32 [R_total] its batch
 * (0, 0, .., 1, 1, ...: non-decreasing, every batch non-empty and at most `capacity` sources), h_given_ls int32 [n_batches] the
 * padded width each batch was given in (smart mode builds its window library over that width, :603-615).  `stats` receives the
 * sums of what the device executed (model_calls = iterations of all pools). */
int ttx_beam_speculative_generate_pool(ttx_session** sessions, int n_sessions, const int64_t* d_src, int R_total, int Ls_all,
                                       const int32_t* h_len, const int32_t* h_batch_of, int n_batches, const int32_t* h_given_ls,
                                       int capacity, const ttx_beam_params* p, int64_t* d_out, int16_t* d_trace_len,
                                       int32_t* d_summary, int trace_cap, ttx_beam_stats* stats, void* stream);

/* TranslationInferenceBeamSearch.generate (src/decoding/standard_decoding.py:89-174) — the whole loop on the device with a
 * per-hypothesis KV cache: the <BOS> step (:102), then up to max_len - 2 iterations of {decoder on the unfinished hypotheses,
 * artificial "35 on PAD" logits for the finished ones (:133-135), log(softmax) + running score, topk(beam) over beam x V per
 * source (:151-153), row assembly (:154-161)}, ending early once every hypothesis holds EOS (:166).
 *   d_src int64 [B, Ls];  d_out int64 [B, beam_size, max_len] (row stride max_len): the reference's result [B, beam, W]
 *   occupies the first W = stats->out_width columns, hypotheses best first. */
typedef struct ttx_beam_search_params {
  int32_t max_len, beam_size;
  int32_t pad_token, bos_token, eos_token;
} ttx_beam_search_params;
typedef struct ttx_beam_search_stats {
  int64_t model_calls;     /* generator.model_calls_num                                              */
  int64_t running_rows;    /* generator.b_sz: decoder rows over all calls (unfinished hypotheses)    */
  int32_t out_width;
  int32_t pad_;
} ttx_beam_search_stats;
int ttx_beam_generate(ttx_session* s, const int64_t* d_src, int B, int Ls, const ttx_beam_search_params* p, int64_t* d_out,
                      ttx_beam_search_stats* stats, void* stream);

/* Several batches in flight on one GPU (the scheduling SURVEY.md §8(f) #1 names; the reference's predict loop
 * is strictly one batch at a time, src/model/lightning_model.py:209-212).  Batch i is decoded on
 * sessions[i % n_sessions]; each session runs on its own internal stream that first waits for `stream`
 * (the stream the inputs were produced on); the call returns when every output is complete.  Per-batch
 * outputs and stats are identical to n_batches calls of ttx_greedy_speculative_generate. */
int ttx_greedy_speculative_generate_many(ttx_session** sessions, int n_sessions, int n_batches,
                                         const int64_t* const* d_src, const int* B, const int* Ls,
                                         const ttx_gen_params* p, int64_t* const* d_out, ttx_gen_stats* stats,
                                         void* stream);

/* Row-scheduled decoding (second half: batches regrouped by length).  The reference's loop
 * couples the rows of a batch only through the shared width of `generated_tokens` (speculative_decoding.py:93,
 * :97-102, :145, :158): the loop ends once max(front) + draft_len + 2 >= max_len, and a row finishing at a width
 * beyond max_len raises.  Tokens, drafts and accepted lengths of a row do not depend on its neighbours.  This
 * entry point therefore decodes every row under the rule it would see ALONE in a batch (continue while
 * front + draft_len + 2 < max_len) and returns, besides d_out[i] (int64 [B_i][max_len]; rows that never produced
 * EOS stay PAD), each row's front after every verify step: d_traj[i] int16 [B_i][max_len + 1] (column 0 = 0, -1
 * past the row's last step) and d_fin_step[i] int32 [B_i] (the step that produced EOS, 0 = none).  From these a
 * caller that regrouped rows (e.g. sorted by source length) replays the reference's width rule over the ORIGINAL
 * batches and obtains exactly their outputs, errors and model-call counts; translation-transformer_amd/decoding.py
 * `generate_many(..., reorder=True)` does that.  Returns TTX_ERR_ROW_REPLAY when a row emitted PAD inside its
 * sequence (reference quirk: the outcome then depends on the neighbours; decode those batches as given). */
int ttx_greedy_speculative_generate_rows(ttx_session** sessions, int n_sessions, int n_batches,
                                         const int64_t* const* d_src, const int* B, const int* Ls,
                                         const ttx_gen_params* p, int64_t* const* d_out, int16_t* const* d_traj,
                                         int32_t* const* d_fin_step, ttx_gen_stats* stats, void* stream);

/* The same contract as ttx_greedy_speculative_generate_rows with continuous batching: rows are not cut into fixed
 * groups; every session keeps a pool of up to `capacity` slots and admits the next rows of the work list (encoder,
 * cross K/V, drafts, slot state) whenever at least a quarter of its slots are free, so the verify step keeps
 * close to capacity * (1 + n_drafts * draft_len) rows until the list is exhausted.  d_src int64 [R_total][Ls_all] holds
 * ALL rows right-padded, in the order they are to be admitted (sorted by length pads least: a chunk of rows is encoded at the
 * width of its longest row); h_len (HOST, int32 [R_total]) their lengths (position after the last non-PAD token).  d_out int64 [R_total][max_len], d_traj int16 [R_total][max_len + 1], d_fin_step int32
 * [R_total] as in ttx_greedy_speculative_generate_rows, in the order of d_src.  `stats` (zero it first) receives the
 * sums over all sessions; stats->model_calls counts the verify steps the device executed.  Returns
 * TTX_ERR_ROW_REPLAY like the rows call. */
int ttx_greedy_speculative_generate_pool(ttx_session** sessions, int n_sessions, const int64_t* d_src, int R_total, int Ls_all,
                                         const int32_t* h_len, int capacity, const ttx_gen_params* p, int64_t* d_out,
                                         int16_t* d_traj, int32_t* d_fin_step, ttx_gen_stats* stats, void* stream);

/* Host-side string work either side of the hot path (no GPU) ------------------------------------
 * ChemSMILESTokenizer (src/data_handling/tokenizer_smiles.py:8-39), the pad_sequence collate
 * (src/data_handling/seq2seq_wrappers.py:121-127) and GenericTokenizer.decode (tokenizer_base.py:80-91).
 * ttx_tokenizer_create takes the vocabulary as parallel arrays (token string, id), i.e. the reference's vocab.json
 * (decoder_dict); service ids are the reference's fixed PAD=0, BOS=1, EOS=2, UNK=3.
 * encode: number of ids the line needs (BOS/EOS included), writing min(that, cap) of them;
 * encode_batch: int64 [B, cap_cols] padded with PAD, returns the padded width (or -needed if cap_cols is too small);
 * decode: skips service tokens, stops at the first EOS, returns the string length, writes a NUL-terminated string. */
typedef struct ttx_tokenizer ttx_tokenizer;
int  ttx_tokenizer_create(const char* const* tokens, const int32_t* ids, int n, ttx_tokenizer** out);
void ttx_tokenizer_destroy(ttx_tokenizer* t);
int  ttx_tokenizer_encode(const ttx_tokenizer* t, const char* line, int32_t* out, int cap);
int  ttx_tokenizer_encode_batch(const ttx_tokenizer* t, const char* const* lines, int B, int64_t* out, int cap_cols);
int  ttx_tokenizer_decode(const ttx_tokenizer* t, const int64_t* ids, int n, char* out, int cap);

/* Parity instrumentation for the KV-cached verify step: the step selected by ttx_gen_params.want_logits (1-based step
 * number) of the most recent generate call on `s` — its pre-argmax logits and the loop state they were computed from.
 * HOST destinations: h_logits [n_active*rps, V], h_act int32 [n_active] (running rows in slot order), h_front int32 [B],
 * h_gen int32 [B, gen_ld]; info[0..5] = n_active, rps (= 1 + N*D rows per sequence), B, gen_ld, V, step.  Any of the
 * four array pointers may be null (e.g. a first call to learn the sizes). */
int ttx_debug_step_snapshot(ttx_session* s, int32_t* info, float* h_logits, int32_t* h_act, int32_t* h_front,
                            int32_t* h_gen);

/* Timing of the dominant kernel for bench.py's roofline: summed HIP-event time (events recorded on the launch stream around
 * every GEMM launch) and launch count of the generate calls on this session since the previous read (reading resets the sums);
 * `empty_pair_ms` is what the bracketing adds to a launch's figure, calibrated on the same stream with pairs around a kernel of
 * known duration (pair time minus the realtime ticks the kernel saw go by; median of 32).  Only collected when the session was
 * created with TTX_PROFILE_GEMM=1 in the environment (that session launches eagerly, without graphs).  The *_pool and *_many
 * entry points run the sessions of a call ONE AFTER ANOTHER when sessions[0] is a profiling session (each pool decodes the share of
 * the work list it takes when all start together), so that an event pair times its own launch and not the other sessions' kernels;
 * every session of the call must then be a profiling one (NativeTransformer built under TTX_PROFILE_GEMM=1 sees to that). */
int ttx_last_kernel_profile(ttx_session* s, double* gemm_ms, int64_t* gemm_launches, double* empty_pair_ms);

/* Development aid (tools/bench_gemm.py): times one GEMM shape (K = 64, 128 or a multiple of 256) in isolation on random
 * operands.  variant 2 / 46 = 64x64, 128x64 tiles; 24 = the production kernel's own choice from the row count;
 * 3 = one wave per canonical slice (32x32 tiles, K = 256); 8 = one workgroup per slice with `splits` raw slabs (FFN2).  Returns
 * microseconds per launch over `reps` back-to-back launches and the largest absolute difference to the 64x64 tiling's result
 * — 0.0 for every variant: all of them evaluate the same ordered sum of K slices (csrc/ttx_gemm.hip). */
int ttx_debug_gemm_bench(ttx_session* s, int M, int N, int K, int splits, int variant, int reps, double* us_per_launch,
                         double* max_abs_diff);

#ifdef __cplusplus
}
#endif
#endif /* TTX_H */
